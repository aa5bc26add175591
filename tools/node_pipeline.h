// node_pipeline.h -- one rank of north_star's whole-node run, as host-only control flow: the rank's planes are
// transformed chunk by chunk on a compute stream, every chunk's coefficients are all-gathered on a second stream that
// waits for the chunk's kernel through an event, so chunk k's gather runs under chunk k+1's kernel.  Three ways to run
// the same chunks -- compute only, gather only, pipelined -- give SURVEY.md 8(e)'s three figures.
//
// Nothing of HIP or RCCL in here: the back end (streams, events, what "compute chunk k" and "gather chunk k" launch) is a
// template parameter.  tools/simd_dct_cli.cpp instantiates it over hipStream_t / hipEvent_t, mdct_batch_run and
// mdct_allgather_rows; tests/node_pipeline_driver.cpp over worker-thread streams on host buffers, with the real
// mdct_allgather_rows of libmdct_hip.so bound to tests/fake_rccl.c (world 2 and 8, every gathered byte compared, TSan).
// The reference has no communication at all; its only parallelism hook is the caller-side row range simd_dct.cpp:2245-2255.
//
// Dev must provide:
//   typename Stream, typename Event
//   Stream compute_stream(); Stream comm_stream(); Event event(int chunk);
//   int launch_compute(int chunk, Stream);   0 = ok   (asynchronous on the stream)
//   int launch_gather(int chunk, Stream);    0 = ok   (asynchronous on the stream; collective: every rank issues the same sequence)
//   int record(Event, Stream); int wait(Stream, Event); int sync(Stream);
#ifndef MDCT_NODE_PIPELINE_H
#define MDCT_NODE_PIPELINE_H

namespace mdct_node
{

template <class Dev>
class Pipeline
{
public:
  Pipeline(Dev &dev, int n_chunks) : d(dev), n(n_chunks) {}

  // every chunk's kernel back to back, one wait at the end
  int compute_only()
  {
    int rc = 0;
    for (int k = 0; k < n && rc == 0; k++)
      rc = d.launch_compute(k, d.compute_stream());
    const int s = d.sync(d.compute_stream());
    return rc ? rc : s;
  }

  // every chunk's all-gather back to back (the buffers hold whatever the last transform left: the bytes moved are the same)
  int gather_only()
  {
    int rc = 0;
    for (int k = 0; k < n && rc == 0; k++)
      rc = d.launch_gather(k, d.comm_stream());
    const int s = d.sync(d.comm_stream());
    return rc ? rc : s;
  }

  // kernel k on the compute stream, an event behind it; the communication stream waits for that event and gathers chunk k
  // while the compute stream is already running kernel k + 1.  A failed launch stops the issue; both streams are drained
  // before returning either way, so that the caller may free or reuse the buffers.
  // NOTE for callers with several ranks: a rank that returns early here has issued fewer collectives than its peers --
  // the caller must make the failure known to them (simd_dct_cli: the shared abort flag and the parent's deadline).
  int pipelined()
  {
    int rc = 0;
    for (int k = 0; k < n && rc == 0; k++)
    {
      if ((rc = d.launch_compute(k, d.compute_stream())))
        break;
      if ((rc = d.record(d.event(k), d.compute_stream())))
        break;
      if ((rc = d.wait(d.comm_stream(), d.event(k))))
        break;
      rc = d.launch_gather(k, d.comm_stream());
    }
    const int s1 = d.sync(d.compute_stream());
    const int s2 = d.sync(d.comm_stream());
    return rc ? rc : (s1 ? s1 : s2);
  }

private:
  Dev &d;
  int n;
};

// Layout of the gather buffer all three runs share: [chunk][owner rank][plane of the chunk], every slot `chunk_planes` planes.
// A chunk is then world * chunk_planes equal "rows" of plane_bytes for mdct_allgather_rows, rank r owning rows
// [r * chunk_planes, (r + 1) * chunk_planes) -- equal shards: ONE in-place ncclAllGather per chunk.
struct BatchShape
{
  int world, rank;
  int planes;       // whole batch
  int per_rank;     // planes / world
  int chunk_planes; // planes of one rank per chunk
  int chunks;       // per_rank / chunk_planes
  // global index of plane i of chunk c as owned by rank r
  int plane_id(int r, int c, int i) const { return r * per_rank + c * chunk_planes + i; }
  // slot of (chunk c, owner r, plane i) in the gather buffer, in planes
  long slot(int c, int r, int i) const { return ((long)c * world + r) * chunk_planes + i; }
};

// planes % world == 0 is required (equal shards); the chunk is shrunk to the largest divisor of per_rank not above `want`
inline bool make_shape(int planes, int world, int rank, int want_chunk, BatchShape &s)
{
  if (planes <= 0 || world <= 0 || rank < 0 || rank >= world || planes % world != 0)
    return false;
  s.world = world;
  s.rank = rank;
  s.planes = planes;
  s.per_rank = planes / world;
  int c = want_chunk < 1 ? 1 : (want_chunk > s.per_rank ? s.per_rank : want_chunk);
  while (s.per_rank % c != 0)
    c--;
  s.chunk_planes = c;
  s.chunks = s.per_rank / c;
  return true;
}

} // namespace mdct_node
#endif
