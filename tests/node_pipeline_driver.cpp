// node_pipeline_driver.cpp -- TEST INFRASTRUCTURE: one rank of the whole-node run's control flow (tools/node_pipeline.h,
// the code tools/simd_dct_cli --gpus N --batch ... runs) on HOST buffers.  Streams are worker threads, events are
// generation counters, "compute chunk k" writes a closed-form transform of the rank's synthetic planes into the rank's
// slot of the gather buffer, and "gather chunk k" is the product's own mdct_allgather_rows (libmdct_hip.so) bound to
// tests/fake_rccl.c through MDCT_RCCL_LIB.  One process per rank (tests/test_node_pipeline.py starts world 2 and 8).
// After each of compute-only / gather-only / pipelined the whole gather buffer is compared byte for byte with what every
// owner must have produced; a missing event wait between a chunk's kernel and its gather shows up as canary bytes (and as
// a ThreadSanitizer report in the sanitizer build).
//
//   node_pipeline_driver <rank> <world> <idfile> <planes> <plane_elems> <chunk_planes>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "mdct.h"
#include "node_pipeline.h"

#define CHECK(c)                                                            \
  do                                                                        \
  {                                                                         \
    if (!(c))                                                               \
    {                                                                       \
      fprintf(stderr, "%s:%d: CHECK failed: %s (%s)\n", __FILE__, __LINE__, #c, mdct_last_error()); \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

namespace
{

struct WorkerStream
{
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool stop = false, busy = false;
  std::thread th;
  WorkerStream() : th([this] { loop(); }) {}
  ~WorkerStream()
  {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void loop()
  {
    for (;;)
    {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty())
          return;
        f = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      f();
      {
        std::lock_guard<std::mutex> lk(m);
        busy = false;
      }
      idle.notify_all();
    }
  }
  void push(std::function<void()> f)
  {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(std::move(f));
    }
    cv.notify_all();
  }
  void drain()
  {
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [&] { return q.empty() && !busy; });
  }
};

struct GenEvent
{
  std::mutex m;
  std::condition_variable cv;
  long issued = 0, reached = 0; // records enqueued / records the stream has got to
};

uint32_t mix32(uint32_t x)
{
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  return x;
}
int16_t input_of(int plane_id, size_t j) { return (int16_t)(mix32((uint32_t)j ^ (0x51ED270Bu * (uint32_t)(plane_id + 1))) >> 16); }
int16_t transform_of(int16_t v, int plane_id) { return (int16_t)(v * 3 + plane_id); } // stands in for the kernel: any closed form will do

struct HostNode
{
  typedef WorkerStream *Stream;
  typedef GenEvent *Event;
  mdct_node::BatchShape shape;
  size_t plane_elems;
  mdct_comm *comm;
  std::vector<int16_t> gbuf; // [chunk][owner][plane of chunk][plane_elems]
  WorkerStream s_compute, s_comm;
  std::vector<GenEvent> events;
  int fail_compute_at = -1;      // chunk whose launch fails (error-path test)
  bool skip_event_wait = false;  // negative control: the bug the events exist to prevent
  std::mutex err_m;
  int async_rc = 0;

  HostNode(const mdct_node::BatchShape &sh, size_t elems, mdct_comm *c) : shape(sh), plane_elems(elems), comm(c), gbuf((size_t)sh.planes * elems), events(sh.chunks) {}

  Stream compute_stream() { return &s_compute; }
  Stream comm_stream() { return &s_comm; }
  Event event(int k) { return &events[k]; }
  int16_t *slot(int c, int r, int i) { return gbuf.data() + (size_t)shape.slot(c, r, i) * plane_elems; }

  int launch_compute(int k, Stream s)
  {
    if (k == fail_compute_at)
      return 42;
    s->push([this, k] {
      std::this_thread::sleep_for(std::chrono::milliseconds(2)); // a kernel takes its time: a gather that did not wait for it moves canaries
      for (int i = 0; i < shape.chunk_planes; i++)
      {
        const int id = shape.plane_id(shape.rank, k, i);
        int16_t *out = slot(k, shape.rank, i);
        for (size_t j = 0; j < plane_elems; j++)
          out[j] = transform_of(input_of(id, j), id);
      }
    });
    return 0;
  }
  int launch_gather(int k, Stream s)
  {
    s->push([this, k] {
      const int rc = mdct_allgather_rows(comm, slot(k, 0, 0), plane_elems * sizeof(int16_t), (size_t)shape.world * shape.chunk_planes, nullptr);
      if (rc)
      {
        std::lock_guard<std::mutex> lk(err_m);
        async_rc = rc;
      }
    });
    return 0;
  }
  int record(Event e, Stream s)
  {
    {
      std::lock_guard<std::mutex> lk(e->m);
      e->issued++;
    }
    s->push([e] {
      {
        std::lock_guard<std::mutex> lk(e->m);
        e->reached++;
      }
      e->cv.notify_all();
    });
    return 0;
  }
  int wait(Stream s, Event e)
  {
    if (skip_event_wait)
      return 0;
    long target;
    {
      std::lock_guard<std::mutex> lk(e->m);
      target = e->issued;
    }
    s->push([e, target] {
      std::unique_lock<std::mutex> lk(e->m);
      e->cv.wait(lk, [&] { return e->reached >= target; });
    });
    return 0;
  }
  int sync(Stream s)
  {
    s->drain();
    std::lock_guard<std::mutex> lk(err_m);
    return async_rc;
  }

  void fill(int16_t v) { std::fill(gbuf.begin(), gbuf.end(), v); }
  // bytes of the gather buffer that differ from what their owners must have produced; own_only: only this rank's slots are expected
  size_t mismatches(bool own_only, int16_t canary)
  {
    size_t bad = 0;
    for (int c = 0; c < shape.chunks; c++)
      for (int r = 0; r < shape.world; r++)
        for (int i = 0; i < shape.chunk_planes; i++)
        {
          const int id = shape.plane_id(r, c, i);
          const int16_t *got = slot(c, r, i);
          const bool expect_data = !own_only || r == shape.rank;
          for (size_t j = 0; j < plane_elems; j++)
            bad += got[j] != (expect_data ? transform_of(input_of(id, j), id) : canary);
        }
    return bad;
  }
};

} // namespace

int main(int argc, char **argv)
{
  if (argc < 7)
  {
    fprintf(stderr, "usage: %s rank world idfile planes plane_elems chunk_planes\n", argv[0]);
    return 2;
  }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  const char *idfile = argv[3];
  const int planes = atoi(argv[4]);
  const size_t elems = strtoull(argv[5], nullptr, 10);
  const int chunk = atoi(argv[6]);

  unsigned char id[MDCT_UNIQUE_ID_BYTES];
  if (rank == 0)
  {
    CHECK(mdct_comm_get_unique_id(id) == MDCT_SUCCESS);
    std::string tmp = std::string(idfile) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    CHECK(f && fwrite(id, 1, sizeof id, f) == sizeof id);
    fclose(f);
    CHECK(rename(tmp.c_str(), idfile) == 0);
  }
  else
  {
    FILE *f = nullptr;
    for (int t = 0; t < 12000 && !(f = fopen(idfile, "rb")); t++)
      usleep(10000);
    CHECK(f && fread(id, 1, sizeof id, f) == sizeof id);
    fclose(f);
  }
  mdct_comm *comm = nullptr;
  CHECK(mdct_comm_init(&comm, rank, world, id) == MDCT_SUCCESS);

  mdct_node::BatchShape shape;
  CHECK(mdct_node::make_shape(planes, world, rank, chunk, shape));
  CHECK(shape.per_rank * world == planes && shape.chunks * shape.chunk_planes == shape.per_rank && shape.chunk_planes <= (chunk < 1 ? 1 : chunk));
  {
    mdct_node::BatchShape bad;
    CHECK(!mdct_node::make_shape(planes + 1, world == 1 ? 2 : world, 0, chunk, bad) || (planes + 1) % (world == 1 ? 2 : world) == 0);
    CHECK(!mdct_node::make_shape(planes, world, world, chunk, bad));
  }
  const int16_t CANARY = (int16_t)0xEEEE;
  HostNode node(shape, elems, comm);
  node.skip_event_wait = getenv("NODE_PIPELINE_SKIP_EVENT_WAIT") != nullptr; // negative control of the test itself
  mdct_node::Pipeline<HostNode> pipe(node, shape.chunks);

  node.fill(CANARY);
  CHECK(pipe.compute_only() == 0);
  CHECK(node.mismatches(/*own_only=*/true, CANARY) == 0); // this rank's slots only; nobody else's bytes touched
  CHECK(pipe.gather_only() == 0);
  CHECK(node.mismatches(false, CANARY) == 0); // every byte of every rank's planes
  for (int rep = 0; rep < 3; rep++)
  {
    node.fill(CANARY);
    CHECK(pipe.pipelined() == 0);
    CHECK(node.mismatches(false, CANARY) == 0);
  }
  if (world == 1)
  { // error path and negative control (no peers to leave behind in a collective)
    node.fill(CANARY);
    node.fail_compute_at = shape.chunks > 1 ? 1 : 0;
    CHECK(pipe.pipelined() == 42); // stops issuing, drains both streams, reports the launch's code
    node.fail_compute_at = -1;
    CHECK(pipe.pipelined() == 0 && node.mismatches(false, CANARY) == 0); // and the pipeline is usable afterwards
  }
  CHECK(mdct_comm_destroy(comm) == MDCT_SUCCESS);
  printf("node pipeline ok rank %d of %d: %d planes, %d chunks of %d\n", rank, world, planes, shape.chunks, shape.chunk_planes);
  return 0;
}
