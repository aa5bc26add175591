// simd_dct_cli.cpp -- benchmark / parity CLI on top of the drop-in API.
//
// Counterpart of the reference's harness (src/main.cpp) for the MI355X engine: same
// positional arguments and options (main.cpp:84-102, parser :207-444), same statistics
// (min, mean +- sigma over --runs, print_perf_info main.cpp:34-80) and the same calls
// (simdDCT_*Buffer(in, out, table, X, Y, 0, Y), main.cpp:514/543/572) -- only the three
// functions now come from libmdct_hip.so.  Host code stays C++; HIP is used here only to
// place the buffers in HBM for the --resident mode.
//
//   simd_dct_cli <raw_grayscale_image_file | synthetic:noise | synthetic:photo> <X> <Y>
//        [--to <file>] [--quality <n>] [--runs <n>] [--mode enc-quant|enc-quant32|enc-quant-stereo]...
//        [--max-simd avx512bw|avx512f|avx2|avx|sse4.2|sse4.1|ssse3|sse3|sse2|none] [--cpu-core <n>] [--resident [--async]] [--pin] [--cold] [--device <n>] [--gpus <n>]
//   simd_dct_cli synthetic:photo 0 0 --gpus <n> --batch <planes>x<X>x<Y> [--chunk <planes>] [--runs <n>]
//        north_star's whole-node run at BASELINE.json configs[3]'s shape: int16 planes, forward only, planes sharded over the ranks,
//        every chunk all-gathered under the next chunk's kernel (tools/node_pipeline.h); compute-only / gather-only / pipelined seconds
//
// Build: hipcc -O2 -std=c++17 -Iinclude tools/simd_dct_cli.cpp -Lsimd_dct_amd -lmdct_hip -Wl,-rpath,'$ORIGIN/../simd_dct_amd' -o tools/simd_dct_cli
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>
#include <x86intrin.h>

#include <atomic>

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "mdct.h"
#include "node_pipeline.h"
#include "simd_dct_shim.h"

namespace
{

double now_ns()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e9 + ts.tv_nsec;
}

// deterministic planes, same generator as simd_dct_amd/synth.py
uint32_t mix32(uint32_t x)
{
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  return x;
}

void synth(std::vector<uint8_t> &img, size_t W, size_t H, bool photo)
{
  const uint32_t seed = 20261003u;
  for (size_t i = 0; i < W * H; i++)
  {
    const uint32_t h = mix32((uint32_t)i ^ seed);
    if (!photo)
    {
      img[i] = (uint8_t)(h >> 24);
      continue;
    }
    const uint64_t x = i % W, y = i / W;
    const int b = (int)(((x * 3 + y * 5) >> 2) & 0xFF);
    const int tri = b < 128 ? b : 255 - b;
    int px = 48 + tri + (int)((h >> 24) % 49) - 24;
    img[i] = (uint8_t)(px < 0 ? 0 : (px > 255 ? 255 : px));
  }
}

struct Stats
{
  double min_ns, mean_ns, sd_ns;
};

// min / mean / sigma of one series (nanoseconds or TSC clocks alike)
Stats stats(const std::vector<double> &ns)
{
  Stats s{1e300, 0, 0};
  for (double v : ns)
  {
    s.min_ns = v < s.min_ns ? v : s.min_ns;
    s.mean_ns += v;
  }
  s.mean_ns /= ns.size();
  for (double v : ns)
    s.sd_ns += (v - s.mean_ns) * (v - s.mean_ns);
  s.sd_ns = ns.size() > 1 ? std::sqrt(s.sd_ns / (ns.size() - 1)) : 0;
  return s;
}

// ---- --gpus N: one process per GPU (forked BEFORE any HIP call), block rows sharded with the
// reference's own startY/endY hook (simd_dct.cpp:2245-2255), coefficients all-gathered over RCCL
// through the C-ABI (mdct_comm_*, mdct_allgather_*).  Every rank ends up with the whole output.
struct Shared
{
  unsigned char id[MDCT_UNIQUE_ID_BYTES];
  std::atomic<int> id_ready;
  std::atomic<int> abort; // a rank failed: nobody enters another collective
  double compute_ns[64], gather_ns[64], pipe_ns[64];
  int rc[64];
  std::atomic<int> arrived[8]; // barriers between the phases of the batch run (one counter per barrier)
};


// the device of a rank: its own GPU.  MDCT_CLI_SHARE_DEVICES=1 (rehearsals on a box with fewer GPUs than ranks, tests/test_two_rank_gpu.py):
// rank modulo the number of devices -- RCCL itself may still refuse two ranks on one device, which the run then reports like any other failure
static int rank_device(int rank)
{
  const char *e = getenv("MDCT_CLI_SHARE_DEVICES");
  if (!(e && e[0] == '1'))
    return rank;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
    return rank;
  return rank % n;
}

int run_rank_body(Shared *sh, int rank, int world, const std::vector<uint8_t> &in, size_t X, size_t Y, const float *table, bool stereo, size_t runs, const char *out_file);

// whatever goes wrong in a rank, the others learn of it before their next collective (and the parent ends those that
// are already inside one, run_multi_gpu)
int run_rank(Shared *sh, int rank, int world, const std::vector<uint8_t> &in, size_t X, size_t Y, const float *table, bool stereo, size_t runs, const char *out_file)
{
  const int rc = run_rank_body(sh, rank, world, in, X, Y, table, stereo, runs, out_file);
  if (rc != 0)
    sh->abort.store(1);
  return rc;
}

int run_rank_body(Shared *sh, int rank, int world, const std::vector<uint8_t> &in, size_t X, size_t Y, const float *table, bool stereo, size_t runs, const char *out_file)
{
  if (mdct_init(rank_device(rank)) != MDCT_SUCCESS)
  {
    printf("rank %d: mdct_init(%d) failed: %s\n", rank, rank_device(rank), mdct_last_error());
    return 2;
  }
  if (rank == 0)
  {
    if (mdct_comm_get_unique_id(sh->id) != MDCT_SUCCESS)
    {
      printf("rank 0: %s\n", mdct_last_error());
      sh->id_ready.store(-1);
      return 2;
    }
    sh->id_ready.store(1);
  }
  while (sh->id_ready.load() == 0 && !sh->abort.load())
    usleep(1000);
  if (sh->id_ready.load() <= 0 || sh->abort.load())
    return 2;
  mdct_comm *comm = nullptr;
  if (mdct_comm_init(&comm, rank, world, sh->id) != MDCT_SUCCESS)
  {
    printf("rank %d: %s\n", rank, mdct_last_error());
    return 2;
  }
  const size_t bytes = X * Y;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  hipStream_t stream;
  if (hipMalloc((void **)&d_in, bytes) != hipSuccess || hipMalloc((void **)&d_out, bytes) != hipSuccess || hipStreamCreate(&stream) != hipSuccess ||
      hipMemcpy(d_in, in.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_out, 0, bytes) != hipSuccess)
  {
    printf("rank %d: device allocation failed\n", rank);
    return 2;
  }
  mdct_shim_set_stream(stream);
  mdct_shim_set_async(1);
  // whole plane: q32 through the sizeY = 2H call form, stereo as is; shard = block rows [b0, b1)
  size_t b0, b1;
  mdct_shard_rows(stereo ? Y / 16 : Y / 8, world, rank, &b0, &b1);
  double best_c = 1e300, best_g = 1e300;
  int rc = 0;
  for (size_t i = 0; i < runs + 1 && rc == 0; i++) // first run warms up
  {
    const double t0 = now_ns();
    simdDctResult r = sdr_Success;
    if (b1 > b0)
      r = stereo ? simdDCT_EncodeQuantizeReorderStereoBuffer(d_in, d_out, table, X, Y, 16 * b0, 16 * (b1 - 1))
                 : simdDCT_EncodeQuantize32ReorderBuffer(d_in, d_out, table, X, 2 * Y, 16 * b0, 16 * (b1 - 1));
    if (r != sdr_Success || hipStreamSynchronize(stream) != hipSuccess)
    {
      printf("rank %d: %s\n", rank, mdct_last_error());
      rc = 3;
      break;
    }
    const double t1 = now_ns();
    if (sh->abort.load())
    {
      printf("rank %d: another rank failed, leaving before the all-gather\n", rank);
      rc = 3;
      break;
    }
    const int g = stereo ? mdct_allgather_stereo(comm, d_out, X, Y, stream) : mdct_allgather_rows(comm, d_out, 8 * X, Y / 8, stream);
    if (g != MDCT_SUCCESS || hipStreamSynchronize(stream) != hipSuccess)
    {
      printf("rank %d: %s\n", rank, mdct_last_error());
      rc = 3;
      break;
    }
    const double t2 = now_ns();
    if (i > 0)
    {
      best_c = t1 - t0 < best_c ? t1 - t0 : best_c;
      best_g = t2 - t1 < best_g ? t2 - t1 : best_g;
    }
  }
  sh->compute_ns[rank] = best_c;
  sh->gather_ns[rank] = best_g;
  if (rc == 0 && out_file && rank == 0)
  {
    std::vector<uint8_t> out(bytes);
    FILE *f = fopen(out_file, "wb");
    if (hipMemcpy(out.data(), d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess || !f || fwrite(out.data(), 1, bytes, f) != bytes)
      rc = 1;
    if (f)
      fclose(f);
  }
  mdct_comm_destroy(comm);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

// ---- --gpus N --batch PxWxH: BASELINE.json configs[3] through the C-ABI, host code in C++ ---------------------------
// Every rank owns P / N whole planes (the same bytes per rank as 1/N of every plane's block rows), transforms them chunk
// by chunk with ONE launch per chunk (mdct_batch_run) straight into its slot of the gather buffer, and all-gathers each
// chunk in place (mdct_allgather_rows: equal shards, one ncclAllGather) on a second stream ordered by events, so that
// chunk k's gather runs under chunk k + 1's kernel.  The control flow is tools/node_pipeline.h -- the code the CPU test
// runs with world 2 and 8 over tests/fake_rccl.c.
struct BatchJob
{
  int planes;
  size_t W, H;
  int chunk;
};

struct HipNode
{
  typedef hipStream_t Stream;
  typedef hipEvent_t Event;
  mdct_node::BatchShape shape;
  size_t plane_bytes = 0;
  mdct_comm *comm = nullptr;
  char *gbuf = nullptr;
  hipStream_t s_compute = nullptr, s_comm = nullptr;
  std::vector<hipEvent_t> events;
  std::vector<mdct_batch *> batches;
  Shared *sh = nullptr;

  Stream compute_stream() { return s_compute; }
  Stream comm_stream() { return s_comm; }
  Event event(int k) { return events[k]; }
  int launch_compute(int k, Stream s) { return mdct_batch_run(batches[k], s); }
  int launch_gather(int k, Stream s)
  {
    if (sh->abort.load())
      return 3; // a peer has failed: it will not show up for this collective
    return mdct_allgather_rows(comm, gbuf + (size_t)shape.slot(k, 0, 0) * plane_bytes, plane_bytes, (size_t)shape.world * shape.chunk_planes, s);
  }
  int record(Event e, Stream s) { return hipEventRecord(e, s) == hipSuccess ? 0 : 3; }
  int wait(Stream s, Event e) { return hipStreamWaitEvent(s, e, 0) == hipSuccess ? 0 : 3; }
  int sync(Stream s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : 3; }
};

// all ranks meet here (shared-memory counter; gives up when a rank has failed)
bool rank_barrier(Shared *sh, int which, int world)
{
  sh->arrived[which].fetch_add(1);
  while (sh->arrived[which].load() < world)
  {
    if (sh->abort.load())
      return false;
    usleep(50);
  }
  return !sh->abort.load();
}

void synth_i16(std::vector<int16_t> &pl, size_t W, size_t H, uint32_t seed)
{ // simd_dct_amd/synth.py plane_i16 "photo": the 8-bit picture minus 128
  for (size_t i = 0; i < W * H; i++)
  {
    const uint32_t h = mix32((uint32_t)i ^ seed);
    const uint64_t x = i % W, y = i / W;
    const int b = (int)(((x * 3 + y * 5) >> 2) & 0xFF);
    const int tri = b < 128 ? b : 255 - b;
    int px = 48 + tri + (int)((h >> 24) % 49) - 24;
    pl[i] = (int16_t)((px < 0 ? 0 : (px > 255 ? 255 : px)) - 128);
  }
}

int comm_setup(Shared *sh, int rank, int world, mdct_comm **comm)
{
  if (mdct_init(rank_device(rank)) != MDCT_SUCCESS)
  {
    printf("rank %d: mdct_init(%d) failed: %s\n", rank, rank_device(rank), mdct_last_error());
    return 2;
  }
  if (rank == 0)
  {
    if (mdct_comm_get_unique_id(sh->id) != MDCT_SUCCESS)
    {
      printf("rank 0: %s\n", mdct_last_error());
      sh->id_ready.store(-1);
      return 2;
    }
    sh->id_ready.store(1);
  }
  while (sh->id_ready.load() == 0 && !sh->abort.load())
    usleep(1000);
  if (sh->id_ready.load() <= 0 || sh->abort.load())
    return 2;
  if (mdct_comm_init(comm, rank, world, sh->id) != MDCT_SUCCESS)
  {
    printf("rank %d: %s\n", rank, mdct_last_error());
    return 2;
  }
  return 0;
}

int run_rank_batch(Shared *sh, int rank, int world, const BatchJob &job, size_t runs)
{
  HipNode node;
  node.sh = sh;
  int rc = comm_setup(sh, rank, world, &node.comm);
  if (rc)
    return rc;
  if (!mdct_node::make_shape(job.planes, world, rank, job.chunk, node.shape))
  {
    printf("rank %d: %d planes do not split evenly over %d ranks\n", rank, job.planes, world);
    return 1;
  }
  const mdct_node::BatchShape &sp = node.shape;
  const size_t W = job.W, H = job.H, elems = W * H;
  node.plane_bytes = elems * sizeof(int16_t);
  char *src = nullptr, *base = nullptr, *scratch = nullptr;
  constexpr int kBase = 4; // distinct pictures; plane p of the batch shows picture p % 4
  if (hipMalloc((void **)&src, (size_t)sp.per_rank * node.plane_bytes) != hipSuccess || hipMalloc((void **)&node.gbuf, (size_t)sp.planes * node.plane_bytes) != hipSuccess ||
      hipMalloc((void **)&base, kBase * node.plane_bytes) != hipSuccess || hipMalloc((void **)&scratch, node.plane_bytes) != hipSuccess ||
      hipStreamCreate(&node.s_compute) != hipSuccess || hipStreamCreate(&node.s_comm) != hipSuccess)
  {
    printf("rank %d: device allocation failed (%zu MiB for the batch's coefficients)\n", rank, ((size_t)sp.planes * node.plane_bytes) >> 20);
    return 2;
  }
  {
    std::vector<int16_t> pic(elems);
    for (int b = 0; b < kBase; b++)
    {
      synth_i16(pic, W, H, 20261003u + 100u + (uint32_t)b);
      if (hipMemcpy(base + (size_t)b * node.plane_bytes, pic.data(), node.plane_bytes, hipMemcpyHostToDevice) != hipSuccess)
        return 2;
    }
    for (int p = 0; p < sp.per_rank; p++)
      if (hipMemcpy(src + (size_t)p * node.plane_bytes, base + (size_t)((rank * sp.per_rank + p) % kBase) * node.plane_bytes, node.plane_bytes, hipMemcpyDeviceToDevice) != hipSuccess)
        return 2;
    if (hipMemset(node.gbuf, 0xEE, (size_t)sp.planes * node.plane_bytes) != hipSuccess)
      return 2;
  }
  node.events.resize(sp.chunks);
  node.batches.resize(sp.chunks, nullptr);
  for (int c = 0; c < sp.chunks; c++)
  {
    std::vector<mdct_plane_i16> pl(sp.chunk_planes);
    for (int i = 0; i < sp.chunk_planes; i++)
    {
      pl[i].from = (const int16_t *)(src + (size_t)(c * sp.chunk_planes + i) * node.plane_bytes);
      pl[i].to = (int16_t *)(node.gbuf + (size_t)sp.slot(c, rank, i) * node.plane_bytes);
      pl[i].pitch_in = pl[i].pitch_out = pl[i].sizeX = W;
      pl[i].sizeY = H;
      pl[i].lut = nullptr;
    }
    if (hipEventCreateWithFlags(&node.events[c], hipEventDisableTiming) != hipSuccess || mdct_batch_create(&node.batches[c], MDCT_MODE_FWD, pl.data(), sp.chunk_planes) != MDCT_SUCCESS)
    {
      printf("rank %d: %s\n", rank, mdct_last_error());
      return 2;
    }
  }
  mdct_node::Pipeline<HipNode> pipe(node, sp.chunks);
  double best[3] = {1e300, 1e300, 1e300};
  int bar = 0;
  for (int phase = 0; phase < 3 && rc == 0; phase++)
  { // 0 compute only, 1 gather only, 2 pipelined; every phase entered by all ranks together, first pass of each untimed
    if (!rank_barrier(sh, bar++, world))
      return 3;
    for (size_t i = 0; i < runs + 1 && rc == 0; i++)
    {
      const double t0 = now_ns();
      rc = phase == 0 ? pipe.compute_only() : (phase == 1 ? pipe.gather_only() : pipe.pipelined());
      const double dt = now_ns() - t0;
      if (i > 0 && dt < best[phase])
        best[phase] = dt;
    }
    if (rc)
      printf("rank %d: phase %d failed (%d): %s\n", rank, phase, rc, mdct_last_error());
  }
  // what arrived: plane 0 of every owner's slot in the first and the last chunk against this rank's own transform of that picture
  if (rc == 0)
  {
    std::vector<int16_t> got(elems), want(elems);
    const int cs[2] = {0, sp.chunks - 1};
    for (int ci = 0; ci < (sp.chunks > 1 ? 2 : 1) && rc == 0; ci++)
      for (int r = 0; r < world && rc == 0; r++)
      {
        const int id = sp.plane_id(r, cs[ci], 0);
        if (mdct_fwd_i16((const int16_t *)(base + (size_t)(id % kBase) * node.plane_bytes), (int16_t *)scratch, W, W, nullptr, W, H, 0, H / 8, nullptr) != MDCT_SUCCESS ||
            hipDeviceSynchronize() != hipSuccess || hipMemcpy(want.data(), scratch, node.plane_bytes, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(got.data(), node.gbuf + (size_t)sp.slot(cs[ci], r, 0) * node.plane_bytes, node.plane_bytes, hipMemcpyDeviceToHost) != hipSuccess)
          rc = 3;
        else if (memcmp(got.data(), want.data(), node.plane_bytes) != 0)
        {
          printf("rank %d: gathered plane %d (owner %d, chunk %d) differs from the transform of its picture\n", rank, id, r, cs[ci]);
          rc = 6;
        }
      }
  }
  sh->compute_ns[rank] = best[0];
  sh->gather_ns[rank] = best[1];
  sh->pipe_ns[rank] = best[2];
  for (mdct_batch *b : node.batches)
    mdct_batch_destroy(b);
  for (hipEvent_t e : node.events)
    (void)hipEventDestroy(e);
  mdct_comm_destroy(node.comm);
  (void)hipFree(src);
  (void)hipFree(node.gbuf);
  (void)hipFree(base);
  (void)hipFree(scratch);
  return rc;
}

int run_multi_gpu(int world, const std::vector<uint8_t> &in, size_t X, size_t Y, const float *table, bool stereo, size_t runs, const char *out_file, const BatchJob *job = nullptr)
{
  if (world > 64)
  {
    puts("Invalid Parameter.");
    return 1;
  }
  Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (sh == MAP_FAILED)
    return 2;
  new (sh) Shared();
  sh->id_ready.store(0);
  sh->abort.store(0);
  for (auto &a : sh->arrived)
    a.store(0);
  std::vector<pid_t> kids;
  for (int r = 0; r < world; r++)
  {
    fflush(stdout); // (nothing buffered may be inherited twice)
    const pid_t p = fork(); // nothing has touched HIP yet in this process
    if (p == 0)
    {
      int code;
      if (job)
      {
        code = run_rank_batch(sh, r, world, *job, runs);
        if (code != 0)
          sh->abort.store(1);
      }
      else
        code = run_rank(sh, r, world, in, X, Y, table, stereo, runs, out_file);
      fflush(stdout); // (_exit does not flush: a rank's messages would be lost whenever stdout is a pipe or a file)
      _exit(code);
    }
    kids.push_back(p);
  }
  // Poll, with a deadline: a rank that fails while the others are already inside a collective (ncclCommInitRank, the
  // all-gather) would leave them blocked for good.  Once one rank has failed the rest get a grace period, then SIGKILL;
  // the same when the whole job exceeds MDCT_CLI_GPUS_TIMEOUT seconds (default 600).
  const char *te = getenv("MDCT_CLI_GPUS_TIMEOUT");
  const double deadline_ns = now_ns() + (te ? atof(te) : 600.0) * 1e9;
  int rc = 0;
  size_t alive = kids.size();
  std::vector<bool> done(kids.size(), false);
  double kill_at_ns = 0;
  bool killed = false;
  while (alive)
  {
    for (size_t i = 0; i < kids.size(); i++)
    {
      if (done[i])
        continue;
      int st = 0;
      const pid_t w = waitpid(kids[i], &st, killed ? 0 : WNOHANG); // after the kill: one blocking wait per remaining child
      if (w == 0)
        continue;
      done[i] = true;
      alive--;
      const int code = (w > 0 && WIFEXITED(st)) ? WEXITSTATUS(st) : 4;
      rc = code > rc ? code : rc;
      if (code != 0 && kill_at_ns == 0)
      {
        sh->abort.store(1);
        kill_at_ns = now_ns() + 5e9;
      }
    }
    const double t = now_ns();
    if (alive && !killed && ((kill_at_ns != 0 && t > kill_at_ns) || t > deadline_ns))
    {
      printf("--gpus: %s; ending %zu remaining rank(s)\n", t > deadline_ns ? "timed out" : "a rank failed", alive);
      for (size_t i = 0; i < kids.size(); i++)
        if (!done[i])
          kill(kids[i], SIGKILL);
      rc = rc ? rc : 5;
      killed = true; // said and done once
    }
    if (alive && !killed)
      usleep(2000);
  }
  if (rc == 0 && job)
  {
    double c = 0, g = 0, pl = 0;
    for (int r = 0; r < world; r++)
    {
      c = sh->compute_ns[r] > c ? sh->compute_ns[r] : c;
      g = sh->gather_ns[r] > g ? sh->gather_ns[r] : g;
      pl = sh->pipe_ns[r] > pl ? sh->pipe_ns[r] : pl;
    }
    mdct_node::BatchShape sp;
    mdct_node::make_shape(job->planes, world, 0, job->chunk, sp);
    const double px = (double)job->planes * job->W * job->H, out_bytes = px * 2;
    const double bus = world > 1 ? out_bytes * (world - 1) / world / g : 0.0; // bytes per ns == GB/s each rank receives
    printf("fwd-i16 batch %dx%zux%zu over %d GPU(s): %d planes per rank in %d chunk(s) of %d, one launch + one in-place all-gather per chunk | result sdr_Success (sampled slots verified)\n",
           job->planes, job->W, job->H, world, sp.per_rank, sp.chunks, sp.chunk_planes);
    printf("  slowest rank, best of %zu: compute only %.6f s | gather only %.6f s | pipelined (gather k under kernel k+1) %.6f s\n", runs, c * 1e-9, g * 1e-9, pl * 1e-9);
    printf("  whole batch: %.0f Mpx/s pipelined, %.0f Mpx/s compute only (%.1f GB/s algorithmic per GPU = %.1f %% of 8 TB/s) | gather bus %.1f GB/s per GPU = %.1f %% of 7 x 153 GB/s xGMI\n",
           px / (pl * 1e-9) / 1e6, px / (c * 1e-9) / 1e6, 4.0 * px / world / c, 100.0 * 4.0 * px / world / c / 8000.0, bus, 100.0 * bus / (7 * 153.0));
  }
  else if (rc == 0)
  {
    double c = 0, g = 0;
    for (int r = 0; r < world; r++)
    {
      c = sh->compute_ns[r] > c ? sh->compute_ns[r] : c;
      g = sh->gather_ns[r] > g ? sh->gather_ns[r] : g;
    }
    const double px = (double)X * Y;
    printf("%s over %d GPU(s): block rows sharded, RCCL all-gather via the C-ABI | result sdr_Success | slowest rank: transform %.1f us, all-gather %.1f us | %.1f Mpx/s incl. gather, %.1f Mpx/s transform only | gather bus %.1f GB/s\n",
           stereo ? "enc-quant-stereo" : "enc-quant32", world, c * 1e-3, g * 1e-3, px / ((c + g) * 1e-9) / 1e6, px / (c * 1e-9) / 1e6, world > 1 ? px * (world - 1) / world / g : 0.0);
  }
  munmap(sh, sizeof(Shared));
  return rc;
}

const char *result_name(int r) { return r == 0 ? "sdr_Success" : (r == 1 ? "sdr_InvalidParameter" : "sdr_NotSupported"); }

} // namespace

int main(int argc, char **argv)
{
  if (argc < 4)
  {
    puts("Invalid Parameter.\n\nUsage: simd_dct_cli <raw_grayscale_image_file | synthetic:noise | synthetic:photo> <resolutionX> <resolutionY>");
    puts("\t--to <file_name>\t\tStore the last output in the specified file.");
    puts("\t--quality <n>\t\t\tMultiplies the base quantization table (integer, as the reference parses it).");
    puts("\t--runs <uint>\t\t\tRun the benchmark for a specified amount of times.");
    puts("\t--max-simd <avx512bw / avx512f / avx2 / avx / sse4.2 / sse4.1 / ssse3 / sse3 / sse2 / none>\tHighest reference tier to reproduce.");
    puts("\t--cpu-core <uint>\t\tPin the host thread to a CPU core.");
    puts("\t--mode <enc-quant / enc-quant32 / enc-quant-stereo>\tOnly execute a specified mode (repeatable).");
    puts("\t--resident\t\t\tKeep input and output in HBM (device pointers through the same API).");
    puts("\t--pin\t\t\t\tPage-lock the host buffers once (mdct_shim_pin): host-pointer calls then DMA in place.");
    puts("\t--cold\t\t\t\tDo not call mdct_shim_warmup() first: the first run then contains the one-time initialisation.");
    puts("\t--device <n>\t\t\tHIP device ordinal.");
    puts("\t--gpus <n>\t\t\tOne process per GPU: block rows sharded, coefficients all-gathered over RCCL (enc-quant32 or enc-quant-stereo, device-resident).");
    puts("\t--batch <P>x<X>x<Y> [--chunk <n>]\tWith --gpus: P int16 planes, forward only, sharded by planes; chunks all-gathered under the next chunk's kernel.");
    puts("\t--async\t\t\t\tWith --resident: also issue the runs back to back on one stream with a single wait at the end.");
    return 1;
  }
  const std::string filename = argv[1];
  const size_t X = strtoull(argv[2], nullptr, 10), Y = strtoull(argv[3], nullptr, 10);
  bool batch_mode = false;
  for (int i = 4; i < argc; i++)
    batch_mode = batch_mode || std::string(argv[i]) == "--batch";
  if ((X == 0 || Y == 0) && !batch_mode)
  {
    puts("Invalid Resolution Specified. Aborting.");
    return 1;
  }
  const char *out_file = nullptr;
  size_t runs = 128; // main.cpp:21
  float quality = 1.0f;
  bool resident = false, pin = false, cold = false, async_calls = false;
  int device = 0, max_simd = MDCT_SIMD_AVX2, gpus = 0;
  BatchJob job{0, 0, 0, 8};
  bool m_encq = false, m_q32 = false, m_stereo = false;
  for (int i = 4; i < argc; i++)
  {
    const std::string a = argv[i];
    auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--to") out_file = next();
    else if (a == "--quality") quality = (float)strtoull(next(), nullptr, 10); // main.cpp:214
    else if (a == "--runs") runs = strtoull(next(), nullptr, 10);
    else if (a == "--resident") resident = true;
    else if (a == "--pin") pin = true;
    else if (a == "--cold") cold = true;
    else if (a == "--device") device = atoi(next());
    else if (a == "--gpus") gpus = atoi(next());
    else if (a == "--async") async_calls = true;
    else if (a == "--chunk") job.chunk = atoi(next());
    else if (a == "--batch")
    {
      unsigned long long bw = 0, bh = 0;
      if (sscanf(next(), "%dx%llux%llu", &job.planes, &bw, &bh) != 3 || job.planes <= 0 || bw == 0 || bh == 0 || bw % 8 || bh % 8)
      {
        puts("Invalid Parameter: --batch <planes>x<X>x<Y> with X, Y multiples of 8.");
        return 1;
      }
      job.W = bw;
      job.H = bh;
    }
    else if (a == "--mode")
    {
      const std::string m = next();
      if (m == "enc-quant") m_encq = true;
      else if (m == "enc-quant32") m_q32 = true;
      else if (m == "enc-quant-stereo") m_stereo = true;
      else { printf("Invalid Parameter '%s'. Aborting.", m.c_str()); return 1; }
    }
    else if (a == "--max-simd")
    {
      // the reference's spellings (main.cpp:87-97); each caps the tier like its flag clearing does (:283-438)
      const std::string m = next();
      if (m == "avx512bw" || m == "avx512f" || m == "avx2") max_simd = MDCT_SIMD_AVX2;
      else if (m == "avx" || m == "sse4.2" || m == "sse4.1") max_simd = MDCT_SIMD_SSE41;
      else if (m == "ssse3") max_simd = MDCT_SIMD_SSSE3;
      else if (m == "sse3" || m == "sse2") max_simd = MDCT_SIMD_SSE2;
      else if (m == "none") max_simd = MDCT_SIMD_NONE;
      else { printf("Invalid SIMD Variant '%s' specified.", m.c_str()); return 1; }
    }
    else if (a == "--cpu-core")
    { // main.cpp:239-259: pin the calling (host) thread, for steadier host-pointer timings
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET((int)strtoull(next(), nullptr, 10), &set);
      pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    else { printf("Invalid Parameter '%s'. Aborting.", a.c_str()); return 1; }
  }
  if (!m_encq && !m_q32 && !m_stereo)
    m_encq = m_q32 = m_stereo = true;
  if (runs == 0 || runs > 1024)
  {
    puts("Invalid Parameter.");
    return 1;
  }
  if (job.planes > 0)
  { // north_star's whole-node run (int16 planes, forward only); forks before the first HIP call of this process
    if (gpus <= 0)
    {
      puts("--batch needs --gpus <n> (1 is fine).");
      return 1;
    }
    return run_multi_gpu(gpus, std::vector<uint8_t>(), 0, 0, nullptr, false, runs > 16 ? 5 : runs, nullptr, &job);
  }

  const size_t fileSize = X * Y;
  std::vector<uint8_t> in(fileSize), out(fileSize, 0);
  if (filename.rfind("synthetic:", 0) == 0)
    synth(in, X, Y, filename == "synthetic:photo");
  else
  {
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f || fread(in.data(), 1, fileSize, f) != fileSize)
    {
      puts("Failed to read file.");
      return 1;
    }
    fclose(f);
  }

  // main.cpp:179-189
  float table[64] = {.17f, .11f, .10f, .16f, .24f, .40f, .51f, .61f, .12f, .12f, .14f, .19f, .26f, .58f, .60f, .55f, .14f, .13f, .16f, .24f, .40f, .57f, .69f, .56f,
                     .14f, .17f, .22f, .29f, .51f, .87f, .80f, .62f, .18f, .22f, .37f, .56f, .68f, 1.09f, 1.03f, .77f, .24f, .35f, .55f, .64f, .81f, 1.04f, 1.13f, .92f,
                     .49f, .64f, .78f, .87f, 1.03f, 1.21f, 1.20f, 1.01f, .72f, .92f, .95f, .98f, 1.12f, 1.00f, 1.03f, .99f};
  for (float &t : table)
    t *= quality;

  if (gpus > 0)
  { // must fork before the first HIP call of this process
    if (m_stereo == m_q32 || m_encq)
    {
      puts("--gpus needs exactly one of --mode enc-quant32 / enc-quant-stereo.");
      return 1;
    }
    mdct_shim_set_max_simd(max_simd);
    return run_multi_gpu(gpus, in, X, Y, table, m_stereo, runs, out_file);
  }
  if (mdct_init(device) != MDCT_SUCCESS)
  {
    printf("mdct_init failed: %s\n", mdct_last_error());
    return 2;
  }
  mdct_device_info di;
  mdct_get_device_info(&di);
  mdct_shim_set_max_simd(max_simd);
  printf("File: '%s' (%" PRIu64 " Bytes)\nDevice: '%s' (%d CUs, wave%d, %.0f GB HBM) via %s pointers, reference tier <= %s\n", filename.c_str(), (uint64_t)fileSize, di.name, di.compute_units,
         di.wavefront_size, di.hbm_bytes / 1e9, resident ? "device" : "host", max_simd == MDCT_SIMD_AVX2 ? "AVX2" : (max_simd == MDCT_SIMD_SSE41 ? "SSE4.1" : (max_simd == MDCT_SIMD_SSSE3 ? "SSSE3" : (max_simd == MDCT_SIMD_SSE2 ? "SSE2" : "scalar"))));

  if (pin && !resident && (mdct_shim_pin(in.data(), fileSize) != 0 || mdct_shim_pin(out.data(), fileSize) != 0))
  {
    puts("mdct_shim_pin failed.");
    return 2;
  }
  uint8_t *d_in = nullptr, *d_out = nullptr;
  const uint8_t *p_in = in.data();
  uint8_t *p_out = out.data();
  if (resident)
  {
    if (hipMalloc((void **)&d_in, fileSize) != hipSuccess || hipMalloc((void **)&d_out, fileSize) != hipSuccess || hipMemcpy(d_in, in.data(), fileSize, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d_out, 0, fileSize) != hipSuccess)
    {
      puts("Memory allocation failure.");
      return 2;
    }
    p_in = d_in;
    p_out = d_out;
  }

  if (!cold && mdct_shim_warmup(fileSize) != MDCT_SUCCESS)
    printf("mdct_shim_warmup failed: %s\n", mdct_last_error());
  // --resident: this box's measured copy rate for the same footprint (read N + write N bytes), the "measured roofline"
  double copy_GBps = 0;
  if (resident)
  {
    for (int i = 0; i < 20; i++)
      (void)mdct_stream_copy(d_in, d_out, fileSize, nullptr);
    (void)hipDeviceSynchronize();
    const int reps = 50;
    const double t0 = now_ns();
    for (int i = 0; i < reps; i++)
      (void)mdct_stream_copy(d_in, d_out, fileSize, nullptr);
    (void)hipDeviceSynchronize();
    copy_GBps = 2.0 * fileSize * reps / (now_ns() - t0);
    (void)hipMemset(d_out, 0, fileSize);
  }

  // host pointers: what the link of this box does with pinned memory when both directions run at once (tools/pcie_bench.cpp measures more
  // patterns) -- the ceiling of any host-pointer call, which has to move `bytes` in and `bytes` out; seconds for `bytes` each way
  auto link_seconds = [&](size_t bytes) -> double {
    uint8_t *da = nullptr, *db = nullptr, *pa = nullptr, *pb = nullptr;
    hipStream_t s0 = nullptr, s1 = nullptr;
    double best = 0;
    if (hipMalloc((void **)&da, bytes) == hipSuccess && hipMalloc((void **)&db, bytes) == hipSuccess && hipHostMalloc((void **)&pa, bytes, hipHostMallocDefault) == hipSuccess &&
        hipHostMalloc((void **)&pb, bytes, hipHostMallocDefault) == hipSuccess && hipStreamCreateWithFlags(&s0, hipStreamNonBlocking) == hipSuccess &&
        hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess)
    {
      memset(pa, 1, bytes);
      memset(pb, 2, bytes);
      for (int i = 0; i < 8; i++)
      {
        const double t0 = now_ns();
        (void)hipMemcpyAsync(da, pa, bytes, hipMemcpyHostToDevice, s0);
        (void)hipMemcpyAsync(pb, db, bytes, hipMemcpyDeviceToHost, s1);
        (void)hipStreamSynchronize(s0);
        (void)hipStreamSynchronize(s1);
        const double dt = (now_ns() - t0) * 1e-9;
        if (i >= 2 && (best == 0 || dt < best))
          best = dt;
      }
    }
    (void)hipGetLastError();
    if (s0) (void)hipStreamDestroy(s0);
    if (s1) (void)hipStreamDestroy(s1);
    if (pa) (void)hipHostFree(pa);
    if (pb) (void)hipHostFree(pb);
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    return best;
  };
  double link_half_s = 0, link_full_s = 0; // half the file each way (enc-quant, enc-quant32: the reference's top-half loop), the whole file (stereo)
  if (!resident)
  {
    link_half_s = link_seconds(fileSize / 2);
    link_full_s = link_seconds(fileSize);
  }

  struct Mode
  {
    const char *name;
    bool on;
    simdDctResult (*fn)(const uint8_t *, uint8_t *, const float *, size_t, size_t, size_t, size_t);
    double covered; // fraction of the plane the reference semantics really transform
  };
  const Mode modes[] = {{"enc-quant", m_encq, simdDCT_EncodeQuantizeBuffer, 0.5}, {"enc-quant32", m_q32, simdDCT_EncodeQuantize32ReorderBuffer, 0.5}, {"enc-quant-stereo", m_stereo, simdDCT_EncodeQuantizeReorderStereoBuffer, 1.0}};
  // the reference's columns (print_perf_info, main.cpp:72-73: min and mean clk/byte, min and mean MiB/s, nominal = whole
  // file / time) next to ns/byte, the pixels really transformed, algorithmic GB/s (2 B/px) and its share of the HBM
  // spec (8 TB/s) and of the copy rate measured above (--resident only)
  puts("mode             | result               |  min clk/byte | mean clk/byte (sigma) |   min ns/byte |  mean ns/byte (sigma) |  min MiB/s (nominal) | mean MiB/s | actual Mpx/s (min) | alg. GB/s | % of 8 TB/s | % of measured copy | % of measured link");
  if (resident)
    printf("(measured copy of %.1f MiB in + out: %.1f GB/s)\n", fileSize / 1048576.0, copy_GBps);
  else if (link_half_s > 0 && link_full_s > 0)
    printf("(measured link, pinned memory, both directions at once: %.1f MiB each way in %.3f ms = %.1f GB/s each way; %.1f MiB each way in %.3f ms = %.1f GB/s: what a host-pointer call of this size could at best take)\n",
           fileSize / 2 / 1048576.0, link_half_s * 1e3, fileSize / 2 / link_half_s / 1e9, fileSize / 1048576.0, link_full_s * 1e3, fileSize / link_full_s / 1e9);
  int rc_all = 0;
  for (const Mode &m : modes)
  {
    if (!m.on)
      continue;
    std::vector<double> ns(runs), clk(runs);
    simdDctResult r = sdr_Success;
    for (size_t i = 0; i < runs && r == sdr_Success; i++)
    {
      const double t0 = now_ns();
      const uint64_t c0 = __rdtsc();            // main.cpp:512
      r = m.fn(p_in, p_out, table, X, Y, 0, Y); // main.cpp:514
      const uint64_t c1 = __rdtsc();
      ns[i] = now_ns() - t0;
      clk[i] = (double)(c1 - c0);
    }
    if (r != sdr_Success)
    {
      printf("%-16s | %-20s | %s\n", m.name, result_name(r), mdct_last_error());
      rc_all = 3;
      continue;
    }
    const Stats s = stats(ns), c = stats(clk);
    const double mib = fileSize / (1024.0 * 1024.0);
    const double px = fileSize * m.covered;
    const double alg = 2.0 * px / s.min_ns; // bytes per ns == GB/s
    char of_copy[32] = "-";
    if (copy_GBps > 0)
      snprintf(of_copy, sizeof of_copy, "%.1f", 100.0 * alg / copy_GBps);
    char of_link[32] = "-";
    const double link_s = m.covered < 1.0 ? link_half_s : link_full_s; // the bytes a host-pointer call of this mode moves each way
    if (!resident && link_s > 0)
      snprintf(of_link, sizeof of_link, "%.1f", 100.0 * link_s / (s.min_ns * 1e-9));
    printf("%-16s | %-20s | %13.5f | %10.5f (%8.5f) | %13.5f | %10.5f (%8.5f) | %20.2f | %10.2f | %18.1f | %9.1f | %11.1f | %18s | %18s\n", m.name, result_name(r), c.min_ns / fileSize, c.mean_ns / fileSize,
           c.sd_ns / fileSize, s.min_ns / fileSize, s.mean_ns / fileSize, s.sd_ns / fileSize, mib / (s.min_ns * 1e-9), mib / (s.mean_ns * 1e-9), px / (s.min_ns * 1e-9) / 1e6, alg, 100.0 * alg / 8000.0, of_copy, of_link);
  }

  if (resident && async_calls)
  { // The rows above are what a caller of the reference API gets who, like main.cpp:514, waits for every call.  The same calls issued
    // back to back (mdct_shim_set_async: a device-pointer call returns once its kernel is queued) with one wait at the end:
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess)
      return 2;
    mdct_shim_set_stream(st);
    mdct_shim_set_async(1);
    puts("mode (async)     | result               | calls back to back, one final wait: us per call | actual Mpx/s | alg. GB/s | % of 8 TB/s | % of measured copy");
    for (const Mode &m : modes)
    {
      if (!m.on)
        continue;
      simdDctResult r = sdr_Success;
      for (int warm = 0; warm < 2 && r == sdr_Success; warm++)
      {
        const double t0 = now_ns();
        for (size_t i = 0; i < runs && r == sdr_Success; i++)
          r = m.fn(p_in, p_out, table, X, Y, 0, Y);
        if (hipStreamSynchronize(st) != hipSuccess)
          r = sdr_NotSupported;
        const double per = (now_ns() - t0) / runs;
        if (warm == 1 && r == sdr_Success)
        {
          const double px = fileSize * m.covered, alg = 2.0 * px / per;
          printf("%-16s | %-20s | %47.2f | %12.1f | %9.1f | %11.1f | %18.1f\n", m.name, result_name(r), per * 1e-3, px / (per * 1e-9) / 1e6, alg, 100.0 * alg / 8000.0, copy_GBps > 0 ? 100.0 * alg / copy_GBps : 0.0);
        }
      }
      if (r != sdr_Success)
      {
        printf("%-16s | %-20s | %s\n", m.name, result_name(r), mdct_last_error());
        rc_all = 3;
      }
    }
    mdct_shim_set_async(0);
    mdct_shim_set_stream(nullptr);
    (void)hipStreamDestroy(st);
  }

  if (out_file)
  {
    if (resident && hipMemcpy(out.data(), d_out, fileSize, hipMemcpyDeviceToHost) != hipSuccess)
    {
      puts("Failed to copy the output back.");
      return 2;
    }
    FILE *f = fopen(out_file, "wb");
    if (!f || fwrite(out.data(), 1, fileSize, f) != fileSize) // main.cpp:594-606
    {
      puts("Failed to write file.");
      return 1;
    }
    fclose(f);
  }
  if (resident)
  {
    (void)hipFree(d_in);
    (void)hipFree(d_out);
  }
  if (pin && !resident)
  {
    mdct_shim_unpin(in.data());
    mdct_shim_unpin(out.data());
  }
  mdct_shim_release();
  return rc_all;
}
