// mdct_api.hip -- the extern "C" boundary of libmdct_hip.so (declared in include/mdct.h).
//
// Thin by design: validate like the reference's dispatchers (simd_dct.cpp:71-133), build the
// 64-entry multiplier table on the host with the reference's exact float expression,
// fill a by-value argument block and launch.  No allocation, no synchronisation and no
// host<->device copy on any launch path (hipGraph-capturable).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "mdct.h"
#include "mdct_kernels.h"

namespace
{

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

} // namespace

// shared with comm.hip; not part of the public ABI
extern "C" __attribute__((visibility("hidden"))) int mdct_set_error(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace
{

int hip_fail(hipError_t e, const char *what)
{
  return fail(MDCT_NOT_SUPPORTED, "%s: %s", what, hipGetErrorString(e));
}

constexpr int kMaxDevices = 64;
std::mutex g_mu;
mdct_device_info g_info[kMaxDevices];
bool g_have[kMaxDevices];

// Replaces the x86 CPUID probe (simd_platform.c:68-178) with a HIP device probe.
int probe(int device, const mdct_device_info **out)
{
  if (device < 0 || device >= kMaxDevices)
    return fail(MDCT_INVALID_PARAMETER, "device ordinal %d out of range", device);
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_have[device])
  {
    hipDeviceProp_t p;
    const hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess)
      return hip_fail(e, "hipGetDeviceProperties");
    mdct_device_info &i = g_info[device];
    memset(&i, 0, sizeof(i));
    i.device = device;
    i.compute_units = p.multiProcessorCount;
    i.wavefront_size = p.warpSize;
    i.lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    i.hbm_bytes = p.totalGlobalMem;
    snprintf(i.name, sizeof(i.name), "%s", p.gcnArchName);
    i.is_gfx950 = strncmp(p.gcnArchName, "gfx950", 6) == 0;
    g_have[device] = true;
  }
  if (out)
    *out = &g_info[device];
  return MDCT_SUCCESS;
}

int current(const mdct_device_info **out)
{
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return hip_fail(e, "hipGetDevice (is a HIP device visible?)");
  const int r = probe(dev, out);
  if (r != MDCT_SUCCESS)
    return r;
  if (!(*out)->is_gfx950 || (*out)->wavefront_size != 64)
    return fail(MDCT_NOT_SUPPORTED, "device %d is %s; this library contains gfx950 (wave64) code only", dev, (*out)->name);
  return MDCT_SUCCESS;
}

// blocks in [by0,by1) x bpr must fit the kernels' 32-bit linear block index
int count_blocks(size_t bpr, size_t rows, uint32_t *n)
{
  const size_t total = bpr * rows;
  if (bpr > 0xFFFFFFFFull || (rows != 0 && total / rows != bpr) || total > 0x7FFFFFFFull)
    return fail(MDCT_NOT_SUPPORTED, "launch of %zu x %zu blocks exceeds the 2^31 block limit; split the row range", bpr, rows);
  *n = (uint32_t)total;
  return MDCT_SUCCESS;
}

// The fast quantiser forms are exact only while |coefficient * q| < 2^31 (see "Quantisers" in
// mdct_kernels.hip).  |coefficient| <= 2040 for 8-bit input in every reference tier, so any
// |q| <= 2^17 is safe; beyond that, or for inf/NaN, the exact x86-convert emulation is used.
bool table_needs_safe(const float *q)
{
  for (int i = 0; i < 64; i++)
    if (!std::isfinite(q[i]) || std::fabs(q[i]) > 131072.0f)
      return true;
  return false;
}

// the reference tiers' constants as the packed-fp32 kernels take them (mdct_kernels.hip: PkConsts)
static mdct::PkConstsArg pk_consts(bool q32, bool scalar)
{
  const mdct::DctConsts c;
  // scalar tiers (mdct_kernels.hip: encode_block_pk): nm.y = the bias 127/255 of :245 / :362, bias = (255, 0.5 - 2^-25)
  return mdct::PkConstsArg{{c.a, c.f}, {c.c, c.d}, {c.b, c.e}, {c.n, scalar ? 127.0f / 255.0f : (q32 ? c.magic23 + 128.0f : c.magic23)},
                           {c.d, c.a}, {c.f, c.d}, {c.f, c.c}, {c.c, c.a},
                           {scalar ? 255.0f : 1.f / (float)0xFF, scalar ? nextafterf(0.5f, 0.0f) : 127.0f},
                           {1.f / 255.f, (float)(1.0 / 255.0 - (double)(1.f / 255.f))}}; // (c, c2): 0x3b808081, 0xaf7efeff
}

// the q32 product's multipliers 255 / (lut * 0.95) (simd_dct.cpp:2239) in the packed kernels' pair order: (v * 4 + j) * 2 + {0, 1} = coefficient
// (v, kPairA[j]) / (v, kPairB[j]); the fast quantiser works on -v (complement trick), the safe one on v
static void q32_pair_table(const float *lut, bool negate, float (&out)[64])
{
  static const int pa[4] = {0, 2, 1, 5}, pb[4] = {4, 6, 3, 7}; // == mdct::kPairA / kPairB
  constexpr float vr = .95f;
  for (int m = 0; m < 8; m++)
    for (int j = 0; j < 4; j++)
    {
      const float qa = 255.0f / (lut[m * 8 + pa[j]] * vr), qb = 255.0f / (lut[m * 8 + pb[j]] * vr);
      out[(m * 4 + j) * 2] = negate ? -qa : qa;
      out[(m * 4 + j) * 2 + 1] = negate ? -qb : qb;
    }
}

int own_plane_args(const void *from, const void *to, size_t esz, size_t pitch_in, size_t pitch_out, size_t sizeX, size_t sizeY, size_t by0, size_t by1)
{
  if (from == nullptr || to == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null plane pointer");
  if (sizeX % 8 != 0 || sizeY % 8 != 0)
    return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu is not a multiple of 8x8", sizeX, sizeY);
  if (pitch_in < sizeX || pitch_out < sizeX || by0 > by1 || by1 > sizeY / 8)
    return fail(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows", by0, by1, sizeY / 8);
  if (((uintptr_t)from | (uintptr_t)to | (pitch_in * esz) | (pitch_out * esz)) & 15)
    return fail(MDCT_INVALID_PARAMETER, "plane rows must be 16-byte aligned");
  return MDCT_SUCCESS;
}

// AAN scale factors a_0 = 1, a_k = sqrt(2) cos(k pi / 16); the 2-D tables are products of
// doubles rounded once to float (the CPU checker uses the identical expression).
const double kAanScale[8] = {1.0, 1.387039845322148, 1.306562964876377, 1.175875602419359, 1.0, 0.785694958387102, 0.541196100146197, 0.275899379282943};

void aan_tables_compute(float *fwd, float *inv)
{
  for (int v = 0; v < 8; v++)
    for (int u = 0; u < 8; u++)
    {
      const double a = kAanScale[v] * kAanScale[u];
      fwd[v * 8 + u] = (float)(1.0 / (8.0 * a));
      inv[v * 8 + u] = (float)(a / 8.0);
    }
}

void aan_tables(float *fwd, float *inv)
{
  static float s_fwd[64], s_inv[64];
  static std::once_flag once;
  std::call_once(once, [] { aan_tables_compute(s_fwd, s_inv); });
  memcpy(fwd, s_fwd, sizeof(s_fwd));
  memcpy(inv, s_inv, sizeof(s_inv));
}

// forward multiplier = (1/lut) * scale, inverse multiplier = lut * scale, each one float op
// pair_order: the fused round trip runs on packed fp32 and wants both tables in the register-pair order of its
// column pass, j-major: (j*8 + v)*2 + {0,1} = (v, kAanPairA[j]) / (v, kAanPairB[j])  (mdct_kernels.hip: i16_roundtrip_rows)
int make_own_tables(const float *lut, mdct::OwnTables &tb, bool pair_order = false)
{
  float ft[64], it[64];
  aan_tables(ft, it);
  for (int i = 0; i < 64; i++)
  {
    if (lut && !(std::isfinite(lut[i]) && lut[i] != 0.0f))
      return fail(MDCT_INVALID_PARAMETER, "quantisation table entry %d is %g; the int16 paths need finite non-zero entries", i, (double)lut[i]);
    tb.qf[i] = lut ? (1.0f / lut[i]) * ft[i] : ft[i];
    tb.dq[i] = lut ? lut[i] * it[i] : it[i];
  }
  if (pair_order)
  {
    static const int pa[4] = {0, 2, 5, 1}, pb[4] = {4, 6, 3, 7}; // the pairs aan_fwd_h produces (mdct_kernels.hip)
    mdct::OwnTables t = tb;
    for (int v = 0; v < 8; v++)
      for (int j = 0; j < 4; j++)
      {
        tb.qf[(j * 8 + v) * 2] = t.qf[v * 8 + pa[j]];
        tb.qf[(j * 8 + v) * 2 + 1] = t.qf[v * 8 + pb[j]];
        tb.dq[(j * 8 + v) * 2] = t.dq[v * 8 + pa[j]];
        tb.dq[(j * 8 + v) * 2 + 1] = t.dq[v * 8 + pb[j]];
      }
  }
  return MDCT_SUCCESS;
}

// every entry >= 8.01 in magnitude: |orthonormal 8x8 DCT coefficient of int16 samples| <= 8 * 32768, so coefficient / entry stays inside
// int16 with ~40 units to spare for the rounding of the float pipeline (mdct_kernels.hip: i16_roundtrip_rows<.., SAT = false>)
static bool lut_bounded(const float *lut)
{
  if (!lut)
    return false;
  for (int i = 0; i < 64; i++)
    if (!(fabsf(lut[i]) >= 8.01f))
      return false;
  return true;
}

// ---- table cache: quantiser tables parked in device memory -------------------------------------------------------
// A kernel's argument segment is written afresh by the host for every launch, so tables passed by value are cold in every
// cache each time; the same 512 bytes in device memory stay hot in L2 across launches (8192^2 round trip with a table: 46.4 ->
// 44.8 us, the 8K 4:2:0 frame 36.9-38.0 -> 35.7).  Round 5: everything about the cache is asynchronous and stream-ordered --
//   first sight   a one-wave upload kernel carries the 512 bytes in ITS argument segment and writes them into a slot, enqueued on the
//                 caller's stream right before the launch that reads the slot: no host block, no staging buffer whose lifetime
//                 anybody has to track (round 4 did a blocking hipMemcpy here)
//   another stream meets a slot whose upload has not finished: hipStreamWaitEvent on the slot's `ready` event
//   capacity      kTableSlots per device, least recently used evicted.  A hit costs no HIP call at all: a slot only remembers WHICH streams
//                 have read it since its upload (table_release).  The evicting upload fences them when it happens: an event recorded
//                 on each of those streams there and then -- behind everything they have queued, the readers included -- which the
//                 uploading stream waits for on the device.  (Round 5's first cut left an event behind every launch instead: the
//                 marker between back-to-back kernels cost a 3840x2160 plane 8 -> 17 us.)  A stream that cannot be fenced any more
//                 (destroyed with work in flight) keeps its slots from being victims; so does a call between acquire and release.
//   capture       a capturing stream never touches the cache, not even for a table that is parked: a replayed graph would read the
//                 slot at any later time, after any number of evictions.  Its tables travel in the kernel arguments (same results).
// When nothing can be parked (capture, allocation failure, every slot held) the caller works from its arguments.
constexpr int kTableSlots = 256;
constexpr int kSlotStreams = 4; // reader streams remembered per slot; a slot read by more is fenced against every stream the cache has seen
constexpr int kKnownStreams = 64;
struct TableSlot
{
  mdct::OwnTables host;        // what the slot holds
  uint64_t hash = 0, tick = 0; // tick: last acquire (LRU)
  int pins = 0;                // acquired, not yet released
  bool used = false, ready_done = false;
  hipStream_t ready_stream = nullptr;
  hipEvent_t ready = nullptr;  // recorded behind the upload kernel
  hipStream_t readers[kSlotStreams] = {};
  int n_readers = 0;           // kSlotStreams + 1: more than that
};
struct TableCache
{
  std::mutex mu;
  mdct::OwnTables *dev = nullptr;
  bool failed = false;
  uint64_t tick = 0;
  std::vector<TableSlot> slots;
  std::unordered_multimap<uint64_t, int> index; // content hash -> slot
  hipStream_t known[kKnownStreams] = {};        // every stream that has read a parked table (for slots with many readers)
  int n_known = 0;
  bool known_overflow = false;
  hipEvent_t fence = nullptr;                   // re-recorded for every fence: hipStreamWaitEvent takes the event's state at the call
  uint64_t stat[MDCT_TABLE_STAT_COUNT] = {};
};

uint64_t table_hash(const mdct::OwnTables &tb)
{ // FNV-1a over the 128 multipliers' bit patterns
  uint64_t h = 1469598103934665603ull;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(&tb);
  for (size_t i = 0; i < sizeof(tb) / 4; i++)
    h = (h ^ w[i]) * 1099511628211ull;
  return h;
}
TableCache g_tables[kMaxDevices];

bool stream_is_capturing(hipStream_t s)
{
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipError_t e = hipStreamIsCapturing(s, &st);
  if (e != hipSuccess)
  {
    (void)hipGetLastError(); // the legacy stream while another stream captures in global mode: treat as "not now"
    return true;
  }
  return st != hipStreamCaptureStatusNone;
}

// allocation and event creation must not disturb a capture another thread has open in global mode
struct RelaxedCaptureMode
{
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  RelaxedCaptureMode() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
  ~RelaxedCaptureMode() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
};

bool make_event(hipEvent_t *ev)
{
  if (*ev)
    return true;
  RelaxedCaptureMode relaxed;
  if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess)
  {
    (void)hipGetLastError();
    *ev = nullptr;
    return false;
  }
  return true;
}

struct TableRef
{ // a parked table held by a call between table_acquire and table_release; dev == nullptr: not parked, read the kernel arguments
  int device = -1, slot = -1;
  const mdct::OwnTables *dev = nullptr;
};

// `stream` must wait (on the device) for everything that still reads or writes the slot: its upload, and the launches of other
// streams that read it -- fenced by an event recorded on each such stream now.  false: some reader cannot be fenced, not a victim.
bool slot_quiesce_for(TableCache &c, TableSlot &sl, hipStream_t stream)
{
  if (!sl.used)
    return true;
  if (!sl.ready_done && sl.ready_stream != stream && hipStreamWaitEvent(stream, sl.ready, 0) != hipSuccess)
    return false;
  const bool many = sl.n_readers > kSlotStreams;
  if (many && c.known_overflow)
    return false;
  const hipStream_t *set = many ? c.known : sl.readers;
  const int n = many ? c.n_known : sl.n_readers;
  for (int i = 0; i < n; i++)
  {
    if (set[i] == stream)
      continue; // in order on the stream itself
    if (!make_event(&c.fence) || hipEventRecord(c.fence, set[i]) != hipSuccess || hipStreamWaitEvent(stream, c.fence, 0) != hipSuccess)
      return false; // (a stream destroyed with work in flight cannot be fenced)
  }
  return true;
}

TableRef table_acquire(int device, const mdct::OwnTables &tb, hipStream_t stream)
{
  TableCache &c = g_tables[device];
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.failed)
    return TableRef();
  if (stream_is_capturing(stream))
  {
    c.stat[MDCT_TABLE_STAT_FROM_ARGUMENTS]++;
    return TableRef();
  }
  const uint64_t h = table_hash(tb);
  const auto range = c.index.equal_range(h);
  for (auto it = range.first; it != range.second; ++it)
  {
    TableSlot &sl = c.slots[it->second];
    if (memcmp(&sl.host, &tb, sizeof(tb)) != 0)
      continue;
    if (!sl.ready_done && sl.ready_stream != stream)
    { // uploaded on another stream: done by now, or this stream waits for it (on the device)
      if (hipEventQuery(sl.ready) == hipSuccess)
        sl.ready_done = true;
      else
      {
        (void)hipGetLastError();
        if (hipStreamWaitEvent(stream, sl.ready, 0) != hipSuccess)
        {
          (void)hipGetLastError();
          c.stat[MDCT_TABLE_STAT_FROM_ARGUMENTS]++;
          return TableRef();
        }
        c.stat[MDCT_TABLE_STAT_STREAM_WAITS]++;
      }
    }
    sl.tick = ++c.tick;
    sl.pins++;
    c.stat[MDCT_TABLE_STAT_HITS]++;
    return TableRef{device, it->second, c.dev + it->second};
  }
  // first sight: a free slot, or the least recently used one nobody holds
  if (!c.dev)
  {
    RelaxedCaptureMode relaxed;
    if (hipMalloc(reinterpret_cast<void **>(&c.dev), kTableSlots * sizeof(mdct::OwnTables)) != hipSuccess)
    {
      (void)hipGetLastError();
      c.dev = nullptr;
      c.failed = true; // keep working from the arguments
      return TableRef();
    }
    c.slots.reserve(kTableSlots);
  }
  int k = -1;
  if (c.slots.size() < (size_t)kTableSlots)
  {
    c.slots.emplace_back();
    k = (int)c.slots.size() - 1;
    if (!make_event(&c.slots[k].ready))
      k = -2;
  }
  else
  { // the least recently used slot that nobody holds and whose readers can be fenced (a few tries: fencing enqueues waits on `stream`)
    uint64_t after = 0;
    for (int tries = 0; tries < 4 && k < 0; tries++)
    {
      int v = -1;
      for (int i = 0; i < kTableSlots; i++)
        if (c.slots[i].pins == 0 && c.slots[i].tick > after && (v < 0 || c.slots[i].tick < c.slots[v].tick))
          v = i;
      if (v < 0)
        break;
      after = c.slots[v].tick;
      if (slot_quiesce_for(c, c.slots[v], stream))
        k = v;
      else
      { // a reader that cannot be fenced any more (its stream was destroyed, or more streams than the cache tracks): the slot stays
        (void)hipGetLastError();
        c.stat[MDCT_TABLE_STAT_UNFENCEABLE]++;
      }
    }
  }
  if (k < 0 || mdct::launch_park_table(tb, c.dev + k, stream) != hipSuccess)
  { // nothing was overwritten: every slot keeps what it held
    (void)hipGetLastError();
    c.stat[MDCT_TABLE_STAT_FROM_ARGUMENTS]++;
    if ((k >= 0 && !c.slots[k].used) || k == -2)
    { // the slot appended above never held a table: give its event back with it
      if (c.slots.back().ready)
        (void)hipEventDestroy(c.slots.back().ready);
      c.slots.pop_back();
    }
    return TableRef();
  }
  TableSlot &sl = c.slots[k];
  bool upload_done = false;
  if (hipEventRecord(sl.ready, stream) != hipSuccess)
  { // the upload is enqueued but other streams could not be told when it ends: the one error path that waits for it
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream);
    upload_done = true;
  }
  if (sl.used)
  {
    const auto old = c.index.equal_range(sl.hash);
    for (auto it = old.first; it != old.second; ++it)
      if (it->second == k)
      {
        c.index.erase(it);
        break;
      }
    c.stat[MDCT_TABLE_STAT_EVICTIONS]++;
  }
  sl.n_readers = 0; // the upload is ordered behind all of them
  sl.host = tb;
  sl.hash = h;
  sl.used = true;
  sl.ready_done = upload_done;
  sl.ready_stream = stream;
  sl.tick = ++c.tick;
  sl.pins = 1;
  c.index.emplace(h, k);
  c.stat[MDCT_TABLE_STAT_UPLOADS]++;
  return TableRef{device, k, c.dev + k};
}

// after the launch(es) that read the slot were enqueued on `stream` (launched = false: nothing was enqueued, just let go).
// No HIP call: the slot and the cache only remember the stream.
void table_release(const TableRef &r, hipStream_t stream, bool launched = true)
{
  if (!r.dev)
    return;
  TableCache &c = g_tables[r.device];
  std::lock_guard<std::mutex> lk(c.mu);
  TableSlot &sl = c.slots[r.slot];
  sl.pins--;
  if (!launched)
    return;
  // The cache-wide list FIRST, for every launched read: a slot with more than kSlotStreams readers is fenced against this list, so a
  // stream whose reads only ever met such slots must be on it too (round 5 returned before this for many-reader slots: a 6th stream
  // could stay unknown to the cache and an evicting upload could overwrite a table its queued kernels were still reading).
  bool seen = false;
  for (int i = 0; i < c.n_known && !seen; i++)
    seen = c.known[i] == stream;
  if (!seen)
  {
    if (c.n_known < kKnownStreams)
      c.known[c.n_known++] = stream;
    else
      c.known_overflow = true; // slots with many readers stop being victims: 64 streams sharing tables is not this cache's case
  }
  if (sl.n_readers > kSlotStreams)
    return;
  for (int i = 0; i < sl.n_readers; i++)
    if (sl.readers[i] == stream)
      return;
  if (sl.n_readers < kSlotStreams)
    sl.readers[sl.n_readers] = stream;
  sl.n_readers++;
}

int run_i16(int mode, const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  // arguments first (like the reference's dispatchers), device second
  int r = own_plane_args(from, to, sizeof(int16_t), pitch_in, pitch_out, sizeX, sizeY, by0, by1);
  if (r)
    return r;
  const mdct_device_info *di;
  if ((r = current(&di)))
    return r;
  mdct::I16Args a;
  a.from = from;
  a.to = to;
  a.pitch_in = pitch_in;
  a.pitch_out = pitch_out;
  a.bpr = (uint32_t)(sizeX / 8);
  a.by0 = (uint32_t)by0;
  if ((r = count_blocks(sizeX / 8, by1 - by0, &a.nblocks)))
    return r;
  if ((r = make_own_tables(lut, a.tb, mode == mdct::MODE_ROUNDTRIP)))
    return r;
  const TableRef parked = (lut || mode != mdct::MODE_ROUNDTRIP) ? table_acquire(di->device, a.tb, (hipStream_t)stream) : TableRef(); // (no table, round trip: no multipliers read)
  a.tb_dev = parked.dev;
  const hipError_t e = mdct::launch_i16(a, mode, lut != nullptr, (hipStream_t)stream, lut_bounded(lut));
  table_release(parked, (hipStream_t)stream, e == hipSuccess);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "i16 kernel launch");
}

// one plane of 8-bit pixels and one of int16 coefficients (either direction), block rows [by0, by1)
int u8_i16_plane_args(const void *px, const void *coef, size_t pitch_px, size_t pitch_coef, size_t sizeX, size_t sizeY, size_t by0, size_t by1)
{
  if (px == nullptr || coef == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null plane pointer");
  if (sizeX % 8 != 0 || sizeY % 8 != 0)
    return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu is not a multiple of 8x8", sizeX, sizeY);
  if (pitch_px < sizeX || pitch_coef < sizeX || by0 > by1 || by1 > sizeY / 8)
    return fail(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows", by0, by1, sizeY / 8);
  if (((uintptr_t)coef | (pitch_coef * sizeof(int16_t))) & 15)
    return fail(MDCT_INVALID_PARAMETER, "coefficient rows must be 16-byte aligned");
  return MDCT_SUCCESS;
}

// pixels -> quantised coefficients, one plane (k_u8_i16; the inverse of one plane runs as a batch of one, mdct_inv_i16_u8)
int run_fwd_u8_i16(const uint8_t *px, int16_t *coef, size_t pitch_px, size_t pitch_coef, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  int r = u8_i16_plane_args(px, coef, pitch_px, pitch_coef, sizeX, sizeY, by0, by1);
  if (r)
    return r;
  const mdct_device_info *di;
  if ((r = current(&di)))
    return r;
  mdct::U8I16Args a;
  a.px = px;
  a.coef = coef;
  a.pitch_px = pitch_px;
  a.pitch_coef = pitch_coef;
  a.bpr = (uint32_t)(sizeX / 8);
  a.by0 = (uint32_t)by0;
  if ((r = count_blocks(sizeX / 8, by1 - by0, &a.nblocks)))
    return r;
  if ((r = make_own_tables(lut, a.tb)))
    return r;
  a.dc_shift = level_shift ? 64.0f * 128.0f : 0.0f;
  const hipError_t e = mdct::launch_u8_i16_fwd(a, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "u8 -> i16 kernel launch");
}

int run_u8_records(const void *px_, bool i16_in, size_t pitch_px, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts,
                   void *stream)
{
  const uint8_t *px = static_cast<const uint8_t *>(px_);
  if (px == nullptr || levels == nullptr || runs == nullptr || counts == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null pointer");
  if (sizeX % 8 != 0 || sizeY % 8 != 0)
    return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu is not a multiple of 8x8", sizeX, sizeY);
  if (pitch_px < sizeX || by0 > by1 || by1 > sizeY / 8)
    return fail(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows", by0, by1, sizeY / 8);
  if (((uintptr_t)levels | (uintptr_t)runs) & 15)
    return fail(MDCT_INVALID_PARAMETER, "levels and runs must be 16-byte aligned");
  if (i16_in && (((uintptr_t)px | (pitch_px * sizeof(int16_t))) & 15))
    return fail(MDCT_INVALID_PARAMETER, "rows of the int16 plane must be 16-byte aligned");
  const mdct_device_info *di;
  int r = current(&di);
  if (r)
    return r;
  mdct::U8RecArgs a;
  a.px = px;
  a.levels = levels;
  a.runs = runs;
  a.counts = counts;
  a.pitch_px = pitch_px;
  a.bpr = (uint32_t)(sizeX / 8);
  a.by0 = (uint32_t)by0;
  if ((r = count_blocks(sizeX / 8, by1 - by0, &a.nblocks)))
    return r;
  if ((r = make_own_tables(lut, a.tb, /*pair_order=*/true)))
    return r;
  a.dc_shift = level_shift ? 64.0f * 128.0f : 0.0f;
  // 8-bit pixels: |coefficient| <= 8 * 255 = 2040, so with every table entry >= 1/16 in magnitude (or no table) the quantised value stays
  // inside int16 (<= 32640) and the kernel's saturations cannot fire (mdct_kernels.hip: k_u8_records<.., CLAMP>)
  bool clamp = i16_in;
  for (int i = 0; lut && !clamp && i < 64; i++)
    clamp = !(fabsf(lut[i]) >= 0.0625f);
  const hipError_t e = mdct::launch_u8_records(a, i16_in, (hipStream_t)stream, clamp);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "u8 -> records kernel launch");
}

int run_f32(int mode, const float *from, float *to, size_t pitch_in, size_t pitch_out, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  int r = own_plane_args(from, to, sizeof(float), pitch_in, pitch_out, sizeX, sizeY, by0, by1);
  if (r)
    return r;
  const mdct_device_info *di;
  if ((r = current(&di)))
    return r;
  mdct::F32Args a;
  a.from = from;
  a.to = to;
  a.pitch_in = pitch_in;
  a.pitch_out = pitch_out;
  a.bpr = (uint32_t)(sizeX / 8);
  a.by0 = (uint32_t)by0;
  if ((r = count_blocks(sizeX / 8, by1 - by0, &a.nblocks)))
    return r;
  {
    float ft[64], it[64];
    aan_tables(ft, it);
    memcpy(a.scale, mode == mdct::MODE_FWD ? ft : it, sizeof(a.scale));
  }
  const hipError_t e = mdct::launch_f32(a, mode, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "f32 kernel launch");
}


// ---- plane batches (mdct_*_i16_batch, mdct_batch_*): validation, table de-duplication, argument blocks ---------------
struct BatchInput
{
  std::vector<mdct::OwnTables> tables; // distinct tables of the whole list
  std::vector<int> table_id;           // per plane: index into `tables`, -1 = takes no table slot
  std::vector<unsigned char> has_lut;  // per plane: a quantisation table was given
  std::vector<unsigned char> bounded;  // per plane: lut_bounded() (int16 planes) / u8_table_is_tame() (8-bit planes)
};

// internal batch modes beside the public MDCT_MODE_*: the 8-bit tile kernel (k_u8_batch) as fused round trip, pixels -> coefficients, coefficients -> pixels
constexpr int kModeRoundtripU8 = 3, kModeFwdU8 = 4, kModeInvU8 = 5;
inline bool is_u8_mode(int mode) { return mode >= kModeRoundtripU8 && mode <= kModeInvU8; }
constexpr int kModeQ32 = 6; // 8-bit planes -> the reference's q32 product (k_q32_batch)
// every batch kernel (k_i16_batch, k_u8_batch, k_q32_batch) tiles planes whose rows end in half a tile over PAIRS of block rows (batch_plan.h: kDescPaired)
// (MDCT_PAIRED_ROWS=0 in the environment: one tile grid per block row as before round 6 -- the A/B knob of tools/experiments/exp_paired_rows.py)
inline bool pairs_rows(int)
{
  const char *e = getenv("MDCT_PAIRED_ROWS");
  return !(e && e[0] == '0');
}

// what the layout code sees of a plane of the mixed (8-bit pixels <-> int16 coefficients) batches
struct GenPlane
{
  const void *from;
  void *to;
  size_t pitch_in, pitch_out; // bytes on the pixel side, elements on the coefficient side
  size_t sizeX, sizeY;
  const float *lut;
};

// 8-bit planes: pitches in bytes, no alignment requirement (the reference's loads are unaligned too, simd_dct.cpp:2109)
int u8_plane_args(const void *from, const void *to, size_t pitch_in, size_t pitch_out, size_t sizeX, size_t sizeY, size_t by0, size_t by1)
{
  if (from == nullptr || to == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null plane pointer");
  if (sizeX % 8 != 0 || sizeY % 8 != 0)
    return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu is not a multiple of 8x8", sizeX, sizeY);
  if (pitch_in < sizeX || pitch_out < sizeX || by0 > by1 || by1 > sizeY / 8)
    return fail(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows", by0, by1, sizeY / 8);
  return MDCT_SUCCESS;
}

// The fast build of k_u8_batch (mdct_kernels.hip) leaves out the quantiser's saturations and packs the output with
// v_sat_pk_u8_i16, which sees rne(x) + shift as an int16.  Both are safe when
//   every |lut[i]| >= 1/16   |coefficient| <= 8 * 255 for 8-bit pixels, so |coefficient / lut| <= 32640 < 32767.5
//   |lut|_2 <= 40000         Parseval on the orthonormal transform: |x|_inf <= |z|_2 <= |y|_2 + |z - y|_2 <= 2040 + |lut|_2 / 2 = 22040
//                            (z = dequantised coefficients, each within lut[i] / 2 of y), far inside int16 with the shift added
// No table: all ones.  Any other finite non-zero table takes the general build.
bool u8_table_is_tame(const float *lut)
{
  if (!lut)
    return true;
  double ss = 0.0;
  for (int i = 0; i < 64; i++)
  {
    if (!(fabsf(lut[i]) >= 0.0625f))
      return false;
    ss += (double)lut[i] * lut[i];
  }
  return ss <= 40000.0 * 40000.0;
}

// every |entry| >= 1/16 (or no table): |coefficient / lut| <= 16 * 2040 for 8-bit pixels, inside int16 -- the forward 8-bit batch may leave its saturations out
bool u8_table_is_bounded(const float *lut)
{
  for (int i = 0; lut && i < 64; i++)
    if (!(fabsf(lut[i]) >= 0.0625f))
      return false;
  return true;
}

template <class Plane>
int batch_input(int mode, const Plane *planes, int n, BatchInput &in)
{
  constexpr bool U8 = std::is_same<Plane, mdct_plane_u8>::value, GEN = std::is_same<Plane, GenPlane>::value;
  const bool mode_ok = U8 ? (mode == kModeRoundtripU8 || mode == kModeQ32) : (GEN ? (mode == kModeFwdU8 || mode == kModeInvU8) : (mode == mdct::MODE_FWD || mode == mdct::MODE_INV || mode == mdct::MODE_ROUNDTRIP));
  if (!mode_ok)
    return fail(MDCT_INVALID_PARAMETER, "batch mode %d (MDCT_MODE_FWD / _INV / _ROUNDTRIP)", mode);
  if (n < 0 || (planes == nullptr && n > 0))
    return fail(MDCT_INVALID_PARAMETER, "null plane list");
  const bool rt = mode == mdct::MODE_ROUNDTRIP, q32 = mode == kModeQ32;
  in.table_id.assign(n, -1);
  in.has_lut.assign(n, 0);
  in.bounded.assign(n, 0);
  std::vector<const float *> src; // what each distinct table was made from (nullptr = no quantisation)
  bool q32_safe = false;          // q32: one form of the quantiser (and of its tables) for the whole call
  if constexpr (U8)
    for (int i = 0; q32 && i < n; i++)
    { // argument checks in the order of the reference's dispatcher: null -> 1, shape -> 2 (simd_dct.cpp:117-118); the table is null-checked too
      const Plane &p = planes[i];
      if (p.from == nullptr || p.to == nullptr || p.lut == nullptr)
        return fail(MDCT_INVALID_PARAMETER, "null pointer (plane %d)", i);
      if (p.sizeX % 64 != 0 || p.sizeY % 8 != 0)
        return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu: width must be a multiple of 64 and height of 8 for the q32 layout", p.sizeX, p.sizeY);
      if (p.pitch_out < 8 * p.sizeX || p.pitch_out % 16 != 0)
        return fail(MDCT_INVALID_PARAMETER, "output strip pitch %zu must be >= 8*sizeX = %zu and a multiple of 16", p.pitch_out, 8 * p.sizeX);
      // NOT in place (the reference's product is not either: a group's 512 output bytes land over pixels other blocks have not read yet,
      // simd_dct.cpp:2227-2230 against :2103-2110 -- SURVEY.md 8b): an output strip range that meets the plane's own input is refused
      const uintptr_t in0 = (uintptr_t)p.from, in1 = in0 + (p.sizeY ? (p.sizeY - 1) * p.pitch_in + p.sizeX : 0);
      const uintptr_t out0 = (uintptr_t)p.to, out1 = out0 + (p.sizeY >= 8 ? (p.sizeY / 8 - 1) * p.pitch_out + 8 * p.sizeX : 0);
      if (in0 < out1 && out0 < in1)
        return fail(MDCT_INVALID_PARAMETER, "plane %d: the q32 product cannot be written over its own input (to overlaps from)", i);
      float q[64];
      q32_pair_table(p.lut, false, q);
      q32_safe = q32_safe || table_needs_safe(q);
    }
  for (int i = 0; i < n; i++)
  { // validate everything before launching anything
    const Plane &p = planes[i];
    int r;
    if constexpr (U8 || GEN)
      r = u8_plane_args(p.from, p.to, p.pitch_in, p.pitch_out, p.sizeX, p.sizeY, 0, p.sizeY / 8);
    else
      r = own_plane_args(p.from, p.to, sizeof(int16_t), p.pitch_in, p.pitch_out, p.sizeX, p.sizeY, 0, p.sizeY / 8);
    if (r)
      return r;
    if constexpr (GEN)
    { // the coefficient side is read / written as 16-byte rows per lane
      const bool fwd = mode == kModeFwdU8;
      if ((((uintptr_t)(fwd ? p.to : p.from)) | ((fwd ? p.pitch_out : p.pitch_in) * sizeof(int16_t))) & 15)
        return fail(MDCT_INVALID_PARAMETER, "coefficient rows must be 16-byte aligned");
    }
    in.has_lut[i] = p.lut != nullptr;
    in.bounded[i] = q32 ? !q32_safe : (U8 ? u8_table_is_tame(p.lut) : (GEN ? (mode == kModeFwdU8 && u8_table_is_bounded(p.lut)) : lut_bounded(p.lut)));
    if (rt && !p.lut)
      continue; // the fused int16 round trip without a table needs no multipliers at all (1/64 rides in the rounding)
    int id = -1;
    for (size_t k = 0; k < src.size() && id < 0; k++)
      if (src[k] == p.lut || (src[k] && p.lut && memcmp(src[k], p.lut, 64 * sizeof(float)) == 0))
        id = (int)k;
    if (id < 0)
    {
      mdct::OwnTables tb;
      if (q32)
      {
        memset(&tb, 0, sizeof(tb));
        q32_pair_table(p.lut, !q32_safe, tb.qf);
      }
      else if ((r = make_own_tables(p.lut, tb, /*pair_order=*/rt || U8 || GEN)))
        return r;
      id = (int)src.size();
      src.push_back(p.lut);
      in.tables.push_back(tb);
    }
    in.table_id[i] = id;
  }
  return MDCT_SUCCESS;
}

struct BatchLaunch
{
  mdct::BatchArgs args;
  uint32_t total;
  int lutmode;
  bool sat;
};

// header of the argument block from a layout; descriptors / tables are attached by the caller (embedded or device memory)
void batch_header(const mdct::BatchLayout &lay, const BatchInput &in, BatchLaunch &l)
{
  mdct::BatchArgs &a = l.args;
  a.consts = mdct::DctConsts();
  memset(a.px, 0, sizeof(a.px));
  a.pk = pk_consts(/*q32=*/true, /*scalar=*/false);
  memset(&a.head, 0, sizeof(a.head));
  a.head.n = (uint32_t)lay.descs.size();
  a.head.uniform = lay.uniform;
  a.head.pp_m = lay.pp.m;
  a.head.pp_s = lay.pp.s;
  a.head.table_bytes = (uint32_t)(lay.tables.size() * sizeof(mdct::OwnTables));
  memcpy(a.head.first8, lay.first8, sizeof(a.head.first8));
  a.descs = nullptr;
  a.tables = nullptr;
  l.total = lay.total;
  l.lutmode = lay.with_lut == 0 ? mdct::BATCH_NO_LUT : (lay.with_lut == (int)lay.descs.size() ? mdct::BATCH_ALL_LUT : mdct::BATCH_MIXED);
  bool all_bounded = true;
  for (int i : lay.plane)
    all_bounded = all_bounded && in.bounded[i];
  l.sat = !all_bounded;
}

void u8_px_consts(int level_shift, float (&px)[4])
{
  const float shift = level_shift ? 128.0f : 0.0f;
  px[0] = 64.0f * shift; // forward: the level shift is "raw DC - 64 * shift"
  px[1] = shift;         // inverse: it comes back as "z00 + shift" before the transform (a constant plane is the DC term)
  px[2] = px[3] = 0.0f;
}

hipError_t launch_batch(const BatchLaunch &l, int mode, hipStream_t s)
{
  if (mode == kModeQ32)
    return mdct::launch_q32_batch(l.args, l.total, l.sat, s);
  return is_u8_mode(mode) ? mdct::launch_u8_batch(l.args, l.total, mode - kModeRoundtripU8, l.sat, s) : mdct::launch_i16_batch(l.args, l.total, mode, l.lutmode, l.sat, s);
}

// no allocation on the device, no copy: descriptors and tables travel in the kernel arguments, as many planes per launch as fit
template <class Plane>
int run_batch(int mode, const Plane *planes, int n, int level_shift, void *stream)
{
  BatchInput in;
  int r = batch_input(mode, planes, n, in);
  if (r)
    return r;
  const mdct_device_info *di;
  if ((r = current(&di)))
    return r;
  for (int i0 = 0; i0 < n;)
  {
    mdct::BatchLayout lay;
    mdct::batch_layout(planes, in.table_id.data(), in.has_lut.data(), i0, n, mdct::kBatchBlob, sizeof(mdct::OwnTables), lay, pairs_rows(mode));
    if (lay.consumed == 0)
      return fail(MDCT_NOT_SUPPORTED, "plane %d (%zux%zu) exceeds the limit of 2^26 - 1 tiles per launch; split it", i0, planes[i0].sizeX, planes[i0].sizeY);
    i0 += lay.consumed;
    BatchLaunch l;
    batch_header(lay, in, l);
    u8_px_consts(level_shift, l.args.px);
    for (size_t k = 0; k < lay.tables.size(); k++)
      memcpy(l.args.blob + k * sizeof(mdct::OwnTables), &in.tables[lay.tables[k]], sizeof(mdct::OwnTables));
    // the chunk's tables from the device's table cache when all of them are parked there (descriptors: always in the arguments)
    std::vector<TableRef> parked(lay.tables.size());
    bool all_parked = !lay.tables.empty();
    for (size_t k = 0; k < lay.tables.size() && all_parked; k++)
      all_parked = (parked[k] = table_acquire(di->device, in.tables[lay.tables[k]], (hipStream_t)stream)).dev != nullptr;
    if (all_parked)
    {
      const mdct::OwnTables *base = parked[0].dev;
      for (size_t k = 1; k < parked.size(); k++)
        base = parked[k].dev < base ? parked[k].dev : base;
      l.args.tables = base;
      for (mdct::BatchDesc &d : lay.descs)
        d.table = (uint32_t)((parked[d.table / sizeof(mdct::OwnTables)].dev - base) * sizeof(mdct::OwnTables));
    }
    memcpy(l.args.blob + l.args.head.table_bytes, lay.descs.data(), lay.descs.size() * sizeof(mdct::BatchDesc));
    const hipError_t e = launch_batch(l, mode, (hipStream_t)stream);
    for (const TableRef &r : parked)
      table_release(r, (hipStream_t)stream, all_parked && e == hipSuccess); // (an upload that was enqueued for nothing is harmless)
    if (e != hipSuccess)
      return hip_fail(e, "plane batch launch");
  }
  return MDCT_SUCCESS;
}

int run_i16_batch(int mode, const mdct_plane_i16 *planes, int n, void *stream) { return run_batch(mode, planes, n, 0, stream); }

} // namespace

extern "C" {

int mdct_init(int device)
{
  const hipError_t e = hipSetDevice(device);
  if (e != hipSuccess)
    return hip_fail(e, "hipSetDevice");
  const mdct_device_info *di;
  const int r = current(&di);
  if (r != MDCT_SUCCESS)
    return r;
  // one-time costs belong here: load both code objects onto this device now (otherwise the caller's first transform pays ~1.5 ms for it).
  // Failure is not fatal -- the first launch would load them anyway.
  (void)mdct::preload_kernels();
  (void)mdct::preload_stage_kernels();
  return MDCT_SUCCESS;
}

int mdct_get_device_info(mdct_device_info *info)
{
  if (!info)
    return fail(MDCT_INVALID_PARAMETER, "null info");
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return hip_fail(e, "hipGetDevice");
  const mdct_device_info *di;
  const int r = probe(dev, &di);
  if (r)
    return r;
  *info = *di;
  return MDCT_SUCCESS;
}

const char *mdct_last_error(void) { return g_err; }

// pitch_out: 0 = the reference's addressing (strips of 8*sizeX bytes, tight)
static int fwd_quant_u8(const uint8_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int layout, int profile, void *stream)
{
  // argument checks in the order of the reference dispatchers: null -> 1, shape -> 2
  // (simd_dct.cpp:75-76, :97-98, :117-118); the table is not null-checked there, here it is.
  if (from == nullptr || to == nullptr || lut == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null pointer");
  const bool combo = (layout == MDCT_LAYOUT_Q32 && profile == MDCT_PROFILE_REF_AVX) || (layout == MDCT_LAYOUT_STEREO && (profile == MDCT_PROFILE_REF_SSE || profile == MDCT_PROFILE_REF_SCALAR)) ||
                     (layout == MDCT_LAYOUT_BLOCK && profile == MDCT_PROFILE_REF_SCALAR) || (layout == MDCT_LAYOUT_BLOCK_SSE && profile == MDCT_PROFILE_REF_SSE);
  if (!combo)
    return fail(MDCT_NOT_SUPPORTED, "layout %d with profile %d is not a reference tier", layout, profile);
  const size_t xmul = layout == MDCT_LAYOUT_Q32 ? 64 : ((profile == MDCT_PROFILE_REF_SSE) ? 16 : 8);
  const size_t ymul = layout == MDCT_LAYOUT_STEREO ? 16 : 8;
  if (sizeX == 0 || sizeX % xmul != 0 || sizeY % ymul != 0)
    return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu: width must be a multiple of %zu and height of %zu for this layout", sizeX, sizeY, xmul, ymul);
  if (pitch_in < sizeX || by0 > by1 || by1 > sizeY / ymul)
    return fail(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows", by0, by1, sizeY / ymul);
  if (pitch_out != 0 && layout != MDCT_LAYOUT_Q32 && layout != MDCT_LAYOUT_BLOCK)
    return fail(MDCT_NOT_SUPPORTED, "an output strip pitch exists for the Q32 and BLOCK layouts only");
  if (pitch_out != 0 && (pitch_out < 8 * sizeX || pitch_out % 16 != 0))
    return fail(MDCT_INVALID_PARAMETER, "output strip pitch %zu must be >= 8*sizeX = %zu and a multiple of 16", pitch_out, 8 * sizeX);

  const mdct_device_info *di;
  int r = current(&di);
  if (r)
    return r;

  mdct::U8Args a;
  memset(&a, 0, sizeof(a));
  a.consts = mdct::DctConsts();
  a.from = from;
  a.to = to;
  constexpr float vr = .95f;
  for (int i = 0; i < 64; i++) // simd_dct.cpp:2239, :910 (x255 tiers) / :192 (scalar tiers)
    a.qt.q[i] = profile == MDCT_PROFILE_REF_SCALAR ? 1.f / (lut[i] * vr) : 255.0f / (lut[i] * vr);
  const bool safe = profile != MDCT_PROFILE_REF_SCALAR && table_needs_safe(a.qt.q);
  { // the packed-fp32 kernels want the multipliers in their register-pair order (mdct_kernels.hip: encode_block_*_pk):
    // pair (m, j) = coefficients (kPairA[j], kPairB[j]) of the FIRST pass at index m of the SECOND pass.
    // Q32 stores v*8+u with rows first (first-pass index = u); the encq layouts store u*8+v with rows first;
    // the stereo layout stores v*8+u with columns first (first-pass index = v).
    static const int pa[4] = {0, 2, 1, 5}, pb[4] = {4, 6, 3, 7}; // == mdct::kPairA / kPairB
    const bool q32 = layout == MDCT_LAYOUT_Q32;
    float t[64];
    for (int m = 0; m < 8; m++)
      for (int j = 0; j < 4; j++)
      {
        t[(m * 4 + j) * 2] = a.qt.q[q32 ? m * 8 + pa[j] : pa[j] * 8 + m];
        t[(m * 4 + j) * 2 + 1] = a.qt.q[q32 ? m * 8 + pb[j] : pb[j] * 8 + m];
      }
    const bool negate = q32 && !safe; // the q32 fast quantiser works on -v (complement trick)
    for (int i = 0; i < 64; i++)
      a.qt.q[i] = negate ? -t[i] : t[i];
    a.pk = pk_consts(q32, profile == MDCT_PROFILE_REF_SCALAR);
  }
  a.pitch = pitch_in;
  a.sizeX = sizeX;
  a.out_strip = pitch_out ? pitch_out : 8 * sizeX;
  a.out_tight = a.out_strip == 8 * sizeX;
  a.eye_offset = pitch_in * (sizeY / 2);
  a.plane_stride = (sizeX * sizeY) / 64;
  a.bpr = (uint32_t)(sizeX / 8);
  a.by0 = (uint32_t)by0;
  a.by_last = by1 > by0 ? (uint32_t)(by1 - 1) : 0;
  const size_t rows = (by1 - by0) * (layout == MDCT_LAYOUT_STEREO ? 2 : 1);
  if ((r = count_blocks(sizeX / 8, rows, &a.nblocks)))
    return r;
  a.spill_ok = by1 * 8 * sizeX + 64 <= sizeX * sizeY;
  const hipError_t e = mdct::launch_fwd_quant_u8(a, layout, profile, safe, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "u8 kernel launch");
}

int mdct_fwd_quant_u8(const uint8_t *from, uint8_t *to, size_t pitch_in, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int layout, int profile, void *stream)
{
  return fwd_quant_u8(from, to, pitch_in, 0, lut, sizeX, sizeY, by0, by1, layout, profile, stream);
}

int mdct_fwd_quant_u8_pitched(const uint8_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int layout, int profile, void *stream)
{
  if (pitch_out == 0)
    return fail(MDCT_INVALID_PARAMETER, "output strip pitch is 0");
  return fwd_quant_u8(from, to, pitch_in, pitch_out, lut, sizeX, sizeY, by0, by1, layout, profile, stream);
}

int mdct_fwd_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  return run_i16(mdct::MODE_FWD, from, to, pitch_in, pitch_out, lut, sizeX, sizeY, by0, by1, stream);
}

int mdct_inv_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  return run_i16(mdct::MODE_INV, from, to, pitch_in, pitch_out, lut, sizeX, sizeY, by0, by1, stream);
}

int mdct_roundtrip_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  return run_i16(mdct::MODE_ROUNDTRIP, from, to, pitch_in, pitch_out, lut, sizeX, sizeY, by0, by1, stream);
}

int mdct_fwd_u8_records(const uint8_t *px, size_t pitch, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream)
{
  return run_u8_records(px, false, pitch, lut, level_shift, sizeX, sizeY, by0, by1, levels, runs, counts, stream);
}

int mdct_fwd_i16_records(const int16_t *from, size_t pitch, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream)
{
  return run_u8_records(from, true, pitch, lut, 0, sizeX, sizeY, by0, by1, levels, runs, counts, stream);
}

extern "C" __attribute__((visibility("hidden"))) void mdct_huff_build(int which, uint32_t *tab, int ntab); // stages.hip

struct PxScanOut
{ // the one-launch form: the rows go on into a contiguous scan (mdct_fwd_*_jpeg_scan)
  uint64_t *row_work;
  int first_rst;
  uint8_t *scan;
  size_t capacity;
  uint64_t *row_offsets;
};

constexpr size_t kChainMaxRows = 16384; // beyond: two launches (see run_px_huffman)
static int run_px_huffman(const void *px_, bool i16_in, size_t pitch_px, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride,
                          uint32_t *seg_bytes, uint32_t *ff_counts, const PxScanOut *pack, void *stream)
{
  const uint8_t *px = static_cast<const uint8_t *>(px_);
  if (px == nullptr || out == nullptr || (seg_bytes == nullptr && !pack))
    return fail(MDCT_INVALID_PARAMETER, "null pointer");
  if (pack && (pack->row_work == nullptr || pack->scan == nullptr || pack->row_offsets == nullptr))
    return fail(MDCT_INVALID_PARAMETER, "null pointer");
  if (pack && ((((uintptr_t)pack->row_work | (uintptr_t)pack->row_offsets) & 7) || pack->first_rst < 0 || pack->first_rst > 7))
    return fail(MDCT_INVALID_PARAMETER, "row_work and row_offsets 8-byte aligned; first_rst in 0..7");
  if (sizeX == 0 || sizeX % 8 != 0 || sizeY % 8 != 0)
    return fail(MDCT_NOT_SUPPORTED, "plane %zux%zu is not a multiple of 8x8", sizeX, sizeY);
  const size_t bpr = sizeX / 8;
  if (pitch_px < sizeX || by0 > by1 || by1 > sizeY / 8 || bpr > 0xFFFF)
    return fail(MDCT_INVALID_PARAMETER, "bad pitch or block-row range [%zu,%zu) for %zu rows, or more than 65535 blocks per row (a restart interval)", by0, by1, sizeY / 8);
  if (seg_stride < bpr * 208 + 8 || seg_stride % 4 != 0 || ((uintptr_t)out & 3))
    return fail(MDCT_INVALID_PARAMETER, "seg_stride must be a multiple of 4 and >= 208 * blocks per row + 8 = %zu (worst case of F.1.2); out 4-byte aligned", bpr * 208 + 8);
  if (i16_in && (((uintptr_t)px | (pitch_px * sizeof(int16_t))) & 15))
    return fail(MDCT_INVALID_PARAMETER, "rows of the int16 plane must be 16-byte aligned");
  if (by1 - by0 > 0x7FFFFFFFull)
    return fail(MDCT_NOT_SUPPORTED, "too many block rows for one call");
  if (pack && by0 == by1)
    return fail(MDCT_INVALID_PARAMETER, "an empty block-row range has no scan");
  const mdct_device_info *di;
  int r = current(&di);
  if (r)
    return r;
  mdct::PxHuffArgs a;
  memset(&a, 0, sizeof(a));
  a.consts = mdct::DctConsts();
  a.px = px;
  a.out = out;
  a.seg_bytes = seg_bytes;
  a.ff_counts = ff_counts;
  a.seg_stride = seg_stride;
  a.pitch_px = pitch_px;
  a.bpr = (uint32_t)bpr;
  a.by0 = (uint32_t)by0;
  if (pack)
  {
    a.scan = pack->scan;
    a.capacity = pack->capacity;
    a.row_off = reinterpret_cast<unsigned long long *>(pack->row_offsets);
    a.work = reinterpret_cast<unsigned long long *>(pack->row_work);
    a.n_rows = (uint32_t)(by1 - by0);
    a.first_rst = (uint32_t)pack->first_rst;
  }
  if ((r = make_own_tables(lut, a.tb, /*pair_order=*/true)))
    return r;
  a.dc_shift = level_shift ? 64.0f * 128.0f : 0.0f;
  mdct_huff_build(chroma ? 2 : 0, a.dc, 12);
  mdct_huff_build(chroma ? 3 : 1, a.ac, 256);
  // 8-bit pixels through a table with every entry >= 1.01: an AC coefficient of 8-bit pixels is at most 1020 in magnitude (an exact bound -- half of
  // 8 * 255, reached by the sign patterns of basis functions (0,4), (4,0), (4,4) -- not the looser 8 * 128), so |AC level| <= 1020 / 1.01 = 1009.9 < 1023
  // (13 levels of margin for the float pipeline's rounding; tests/test_entropy.py drives all 64 patterns through 1.01) and the DC fits int16 by far: the
  // kernel's saturations can never fire (mdct_kernels.hip: CLAMP)
  bool clamp = true;
  if (!i16_in && lut)
  {
    clamp = false;
    for (int i = 0; i < 64; i++)
      clamp = clamp || !(fabsf(lut[i]) >= 1.01f);
  }
  if (pack && by1 - by0 > kChainMaxRows)
  { // Every row of the one-launch form adds up the lengths of all rows before it: quadratic, and past ~16 k rows slower than the packing
    // kernel's scan (tools/time_many_rows.py).  Taller planes therefore take the fused kernel and mdct_jpeg_pack_rows_counted, with the
    // per-row byte and 0xFF counts parked in the caller's row_work (8 bytes per row are there), which is zeroed again behind them --
    // row_work[0..1] (epoch, failure word) are not touched, so the array stays valid for one-launch calls.
    const size_t n = by1 - by0;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(pack->row_work + 2);
    a.scan = nullptr;
    a.seg_bytes = cnt - by0; // (the kernel indexes both by the plane's row number)
    a.ff_counts = cnt + n - by0;
    hipError_t e = mdct::launch_px_huffman(a, i16_in, false, clamp, (uint32_t)n, (hipStream_t)stream);
    if (e != hipSuccess)
      return hip_fail(e, "pixels -> Huffman rows kernel launch");
    r = mdct_jpeg_pack_rows_counted(out + by0 * seg_stride, cnt, cnt + n, seg_stride, n, pack->first_rst, pack->scan, pack->capacity, pack->row_offsets, stream);
    if (r)
      return r;
    e = hipMemsetAsync(pack->row_work + 2, 0, n * sizeof(uint64_t), (hipStream_t)stream);
    return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "hipMemsetAsync(row_work)");
  }
  const hipError_t e = mdct::launch_px_huffman(a, i16_in, pack != nullptr, clamp, (uint32_t)(by1 - by0), (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "pixels -> Huffman rows kernel launch");
}

int mdct_fwd_u8_huffman_rows(const uint8_t *px, size_t pitch, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride,
                             uint32_t *seg_bytes, uint32_t *ff_counts, void *stream)
{
  return run_px_huffman(px, false, pitch, lut, level_shift, sizeX, sizeY, by0, by1, chroma, out, seg_stride, seg_bytes, ff_counts, nullptr, stream);
}

int mdct_fwd_i16_huffman_rows(const int16_t *from, size_t pitch, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride, uint32_t *seg_bytes,
                              uint32_t *ff_counts, void *stream)
{
  return run_px_huffman(from, true, pitch, lut, 0, sizeX, sizeY, by0, by1, chroma, out, seg_stride, seg_bytes, ff_counts, nullptr, stream);
}

int mdct_fwd_u8_jpeg_scan(const uint8_t *px, size_t pitch, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma, uint8_t *seg_work, size_t seg_stride,
                          uint64_t *row_work, int first_rst, uint8_t *out, size_t out_capacity, uint64_t *row_offsets, void *stream)
{
  const PxScanOut pack = {row_work, first_rst, out, out_capacity, row_offsets};
  return run_px_huffman(px, false, pitch, lut, level_shift, sizeX, sizeY, by0, by1, chroma, seg_work, seg_stride, nullptr, nullptr, &pack, stream);
}

int mdct_fwd_i16_jpeg_scan(const int16_t *from, size_t pitch, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma, uint8_t *seg_work, size_t seg_stride, uint64_t *row_work,
                           int first_rst, uint8_t *out, size_t out_capacity, uint64_t *row_offsets, void *stream)
{
  const PxScanOut pack = {row_work, first_rst, out, out_capacity, row_offsets};
  return run_px_huffman(from, true, pitch, lut, 0, sizeX, sizeY, by0, by1, chroma, seg_work, seg_stride, nullptr, nullptr, &pack, stream);
}

int mdct_fwd_u8_i16(const uint8_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  return run_fwd_u8_i16(from, to, pitch_in, pitch_out, lut, level_shift, sizeX, sizeY, by0, by1, stream);
}

int mdct_inv_i16_u8(const int16_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  // the strip as a batch of one through k_u8_batch<U8_INV>: 3-8 % faster than a kernel of its own at every size measured
  // (profiles/r05_exp_u8_i16_single_vs_batch.log; the forward direction keeps k_u8_i16, which wins on large planes)
  const int r = u8_i16_plane_args(to, from, pitch_out, pitch_in, sizeX, sizeY, by0, by1);
  if (r)
    return r;
  const GenPlane strip = {from + by0 * 8 * pitch_in, to + by0 * 8 * pitch_out, pitch_in, pitch_out, sizeX, (by1 - by0) * 8, lut};
  return run_batch(kModeInvU8, &strip, 1, level_shift, stream);
}

int mdct_fwd_f32(const float *from, float *to, size_t pitch_in, size_t pitch_out, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  return run_f32(mdct::MODE_FWD, from, to, pitch_in, pitch_out, sizeX, sizeY, by0, by1, stream);
}

int mdct_inv_f32(const float *from, float *to, size_t pitch_in, size_t pitch_out, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  return run_f32(mdct::MODE_INV, from, to, pitch_in, pitch_out, sizeX, sizeY, by0, by1, stream);
}

// the call BASELINE.json configs[2] is quoted on (Y + Cb + Cr, own tables): the fused round trip of a plane batch
int mdct_roundtrip_i16_planes(const mdct_plane_i16 *planes, int n_planes, void *stream) { return run_i16_batch(mdct::MODE_ROUNDTRIP, planes, n_planes, stream); }

int mdct_fwd_i16_batch(const mdct_plane_i16 *planes, int n_planes, void *stream) { return run_i16_batch(mdct::MODE_FWD, planes, n_planes, stream); }
int mdct_inv_i16_batch(const mdct_plane_i16 *planes, int n_planes, void *stream) { return run_i16_batch(mdct::MODE_INV, planes, n_planes, stream); }
int mdct_roundtrip_i16_batch(const mdct_plane_i16 *planes, int n_planes, void *stream) { return run_i16_batch(mdct::MODE_ROUNDTRIP, planes, n_planes, stream); }

// A batch whose descriptors and tables live in device memory: created once (allocates, copies, synchronous), run any
// number of times as ONE launch each (grid limit permitting), capture-safe.
struct mdct_batch
{
  int mode = 0, device = 0, n_planes = 0;
  void *dev = nullptr;
  std::vector<BatchLaunch> launches;
};

} // extern "C" (a template cannot have C linkage)

template <class Plane>
static int batch_create(mdct_batch **out, int mode, const Plane *planes, int n_planes, int level_shift)
{
  if (out == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null batch handle");
  *out = nullptr;
  BatchInput in;
  int r = batch_input(mode, planes, n_planes, in);
  if (r)
    return r;
  const mdct_device_info *di;
  if ((r = current(&di)))
    return r;
  mdct_batch *b = new mdct_batch;
  b->mode = mode;
  b->device = di->device;
  b->n_planes = n_planes;
  std::vector<unsigned char> image; // [tables of launch 0][descriptors of launch 0][tables of launch 1] ...
  std::vector<size_t> at;           // where each launch's tables start in `image`
  for (int i0 = 0; i0 < n_planes;)
  {
    mdct::BatchLayout lay;
    mdct::batch_layout(planes, in.table_id.data(), in.has_lut.data(), i0, n_planes, 0, sizeof(mdct::OwnTables), lay, pairs_rows(mode));
    if (lay.consumed == 0)
    {
      delete b;
      return fail(MDCT_NOT_SUPPORTED, "plane %d (%zux%zu) exceeds the limit of 2^26 - 1 tiles per launch; split it", i0, planes[i0].sizeX, planes[i0].sizeY);
    }
    i0 += lay.consumed;
    if (lay.descs.empty())
      continue;
    BatchLaunch l;
    batch_header(lay, in, l);
    u8_px_consts(level_shift, l.args.px);
    memset(l.args.blob, 0, sizeof(l.args.blob));
    at.push_back(image.size());
    for (int id : lay.tables)
    {
      const unsigned char *t = reinterpret_cast<const unsigned char *>(&in.tables[id]);
      image.insert(image.end(), t, t + sizeof(mdct::OwnTables));
    }
    const unsigned char *d = reinterpret_cast<const unsigned char *>(lay.descs.data());
    image.insert(image.end(), d, d + lay.descs.size() * sizeof(mdct::BatchDesc));
    b->launches.push_back(l);
  }
  if (!image.empty())
  {
    hipError_t e = hipMalloc(&b->dev, image.size());
    if (e == hipSuccess)
      e = hipMemcpy(b->dev, image.data(), image.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
      if (b->dev)
        (void)hipFree(b->dev);
      delete b;
      return hip_fail(e, "plane batch: descriptor table");
    }
    for (size_t k = 0; k < b->launches.size(); k++)
    {
      mdct::BatchArgs &a = b->launches[k].args;
      unsigned char *base = static_cast<unsigned char *>(b->dev) + at[k];
      a.tables = reinterpret_cast<const mdct::OwnTables *>(base);
      a.descs = reinterpret_cast<const mdct::BatchDesc *>(base + a.head.table_bytes);
    }
  }
  *out = b;
  return MDCT_SUCCESS;
}

extern "C" {

int mdct_batch_create(mdct_batch **out, int mode, const mdct_plane_i16 *planes, int n_planes) { return batch_create(out, mode, planes, n_planes, 0); }
int mdct_batch_create_u8(mdct_batch **out, const mdct_plane_u8 *planes, int n_planes, int level_shift) { return batch_create(out, kModeRoundtripU8, planes, n_planes, level_shift); }

int mdct_roundtrip_u8_batch(const mdct_plane_u8 *planes, int n_planes, int level_shift, void *stream) { return run_batch(kModeRoundtripU8, planes, n_planes, level_shift, stream); }

// the reference's primary product (q32 layout, AVX2-tier arithmetic: mdct_fwd_quant_u8 with MDCT_LAYOUT_Q32 / MDCT_PROFILE_REF_AVX over every
// block row) on a plane list, one launch; pitch_out = bytes between the block rows' 8 * sizeX-byte output strips
int mdct_fwd_quant32_u8_batch(const mdct_plane_u8 *planes, int n_planes, void *stream) { return run_batch(kModeQ32, planes, n_planes, 0, stream); }
int mdct_batch_create_q32(mdct_batch **out, const mdct_plane_u8 *planes, int n_planes) { return batch_create(out, kModeQ32, planes, n_planes, 0); }

} // extern "C"

// pixels <-> coefficients batches: the public struct names the two sides, the layout code wants source and destination
static int gen_planes(const mdct_plane_u8_i16 *planes, int n, bool fwd, std::vector<GenPlane> &out)
{
  if (n < 0 || (planes == nullptr && n > 0))
    return fail(MDCT_INVALID_PARAMETER, "null plane list");
  out.resize(n > 0 ? n : 0);
  for (int i = 0; i < n; i++)
  {
    const mdct_plane_u8_i16 &p = planes[i];
    out[i] = fwd ? GenPlane{p.px, p.coef, p.pitch_px, p.pitch_coef, p.sizeX, p.sizeY, p.lut} : GenPlane{p.coef, p.px, p.pitch_coef, p.pitch_px, p.sizeX, p.sizeY, p.lut};
  }
  return MDCT_SUCCESS;
}

extern "C" {

int mdct_fwd_u8_i16_batch(const mdct_plane_u8_i16 *planes, int n_planes, int level_shift, void *stream)
{
  std::vector<GenPlane> g;
  const int r = gen_planes(planes, n_planes, true, g);
  return r ? r : run_batch(kModeFwdU8, g.data(), n_planes, level_shift, stream);
}

int mdct_inv_i16_u8_batch(const mdct_plane_u8_i16 *planes, int n_planes, int level_shift, void *stream)
{
  std::vector<GenPlane> g;
  const int r = gen_planes(planes, n_planes, false, g);
  return r ? r : run_batch(kModeInvU8, g.data(), n_planes, level_shift, stream);
}

int mdct_batch_create_u8_i16(mdct_batch **out, int mode, const mdct_plane_u8_i16 *planes, int n_planes, int level_shift)
{
  if (mode != MDCT_MODE_FWD && mode != MDCT_MODE_INV)
    return fail(MDCT_INVALID_PARAMETER, "batch mode %d (MDCT_MODE_FWD: pixels -> coefficients, MDCT_MODE_INV: coefficients -> pixels)", mode);
  std::vector<GenPlane> g;
  const int r = gen_planes(planes, n_planes, mode == MDCT_MODE_FWD, g);
  return r ? r : batch_create(out, mode == MDCT_MODE_FWD ? kModeFwdU8 : kModeInvU8, g.data(), n_planes, level_shift);
}

// one plane, block rows [by0, by1): the strip as a batch of one (the same kernel, descriptors and table in the arguments)
int mdct_roundtrip_u8(const uint8_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream)
{
  const int r = u8_plane_args(from, to, pitch_in, pitch_out, sizeX, sizeY, by0, by1);
  if (r)
    return r;
  const mdct_plane_u8 strip = {from + by0 * 8 * pitch_in, to + by0 * 8 * pitch_out, pitch_in, pitch_out, sizeX, (by1 - by0) * 8, lut};
  return run_batch(kModeRoundtripU8, &strip, 1, level_shift, stream);
}

int mdct_batch_run(const mdct_batch *b, void *stream)
{
  if (b == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null batch");
  int dev = -1;
  if (!b->launches.empty() && (hipGetDevice(&dev) != hipSuccess || dev != b->device))
    return fail(MDCT_INVALID_PARAMETER, "the batch was created on device %d, the calling thread's current device is %d", b->device, dev);
  for (const BatchLaunch &l : b->launches)
  {
    const hipError_t e = launch_batch(l, b->mode, (hipStream_t)stream);
    if (e != hipSuccess)
      return hip_fail(e, "plane batch launch");
  }
  return MDCT_SUCCESS;
}

int mdct_batch_launches(const mdct_batch *b) { return b ? (int)b->launches.size() : 0; }

int mdct_batch_destroy(mdct_batch *b)
{
  if (b == nullptr)
    return MDCT_SUCCESS;
  hipError_t e = hipSuccess;
  if (b->dev)
    e = hipFree(b->dev);
  delete b;
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "hipFree");
}

int mdct_table_cache_stats(uint64_t *stats, int n)
{
  if (stats == nullptr || n < 0)
    return fail(MDCT_INVALID_PARAMETER, "null stats");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return fail(MDCT_NOT_SUPPORTED, "no current HIP device");
  TableCache &c = g_tables[dev];
  std::lock_guard<std::mutex> lk(c.mu);
  for (int i = 0; i < n; i++)
    stats[i] = i < MDCT_TABLE_STAT_COUNT ? c.stat[i] : 0;
  return MDCT_SUCCESS;
}

int mdct_clock_probe(uint64_t *out, uint32_t ticks_100MHz, uint32_t waves, void *stream)
{
  if (out == nullptr || waves == 0 || waves > 65536 || ticks_100MHz == 0 || ticks_100MHz > 10000000u)
    return fail(MDCT_INVALID_PARAMETER, "clock probe: null output, 0 or more than 65536 waves, or more than 10^7 ticks");
  const mdct_device_info *di;
  const int r = current(&di);
  if (r)
    return r;
  const hipError_t e = mdct::launch_clock_probe(reinterpret_cast<unsigned long long *>(out), ticks_100MHz, waves, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "clock probe launch");
}

int mdct_stream_copy(const void *from, void *to, size_t bytes, void *stream)
{
  if (from == nullptr || to == nullptr)
    return fail(MDCT_INVALID_PARAMETER, "null pointer");
  if ((((uintptr_t)from | (uintptr_t)to | bytes) & 15) != 0)
    return fail(MDCT_INVALID_PARAMETER, "stream copy needs 16-byte aligned pointers and size");
  const mdct_device_info *di;
  const int r = current(&di);
  if (r)
    return r;
  const hipError_t e = mdct::launch_stream_copy(from, to, bytes, di->compute_units, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "stream copy launch");
}

struct mdct_timer
{
  hipEvent_t t0, t1;
};

mdct_timer *mdct_timer_create(void)
{
  mdct_timer *t = new mdct_timer;
  if (hipEventCreate(&t->t0) != hipSuccess || hipEventCreate(&t->t1) != hipSuccess)
  {
    fail(MDCT_NOT_SUPPORTED, "hipEventCreate failed");
    delete t;
    return nullptr;
  }
  return t;
}

void mdct_timer_destroy(mdct_timer *t)
{
  if (!t)
    return;
  (void)hipEventDestroy(t->t0);
  (void)hipEventDestroy(t->t1);
  delete t;
}

int mdct_timer_start(mdct_timer *t, void *stream)
{
  if (!t)
    return fail(MDCT_INVALID_PARAMETER, "null timer");
  const hipError_t e = hipEventRecord(t->t0, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "hipEventRecord");
}

int mdct_timer_stop(mdct_timer *t, void *stream)
{
  if (!t)
    return fail(MDCT_INVALID_PARAMETER, "null timer");
  const hipError_t e = hipEventRecord(t->t1, (hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "hipEventRecord");
}

double mdct_timer_elapsed_ms(mdct_timer *t)
{
  if (!t)
    return -1.0;
  hipError_t e = hipEventSynchronize(t->t1);
  if (e != hipSuccess)
  {
    hip_fail(e, "hipEventSynchronize");
    return -1.0;
  }
  float ms = 0.f;
  e = hipEventElapsedTime(&ms, t->t0, t->t1);
  if (e != hipSuccess)
  {
    hip_fail(e, "hipEventElapsedTime");
    return -1.0;
  }
  return (double)ms;
}

int mdct_timer_wait_spin(mdct_timer *t)
{
  if (!t)
    return fail(MDCT_INVALID_PARAMETER, "null timer");
  for (;;)
  {
    const hipError_t e = hipEventQuery(t->t1);
    if (e == hipSuccess)
      return MDCT_SUCCESS;
    if (e != hipErrorNotReady)
      return hip_fail(e, "hipEventQuery");
  }
}

int mdct_stream_synchronize(void *stream)
{
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  return e == hipSuccess ? MDCT_SUCCESS : hip_fail(e, "hipStreamSynchronize");
}

} // extern "C"
