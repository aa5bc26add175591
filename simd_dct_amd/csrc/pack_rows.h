// pack_rows.h -- device code shared by the scan packing kernels (stages.hip: k_pack_*) and the one-launch encoder
// (mdct_kernels.hip: k_px_huffman_rows<.., PACK>): a row segment of Huffman code -> its place in the contiguous scan,
// byte-stuffed (ITU-T T.81 B.1.1.5) and followed by its restart marker (E.1.4).  256 threads per workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include "wg_sync.h"
#include <stdint.h>

#include "huffman_rows.h" // ff_bytes

namespace mdct
{

constexpr uint32_t kPackBytesPerThread = 64;                      // source bytes per thread and window (rows up to 4 KiB: 16)
constexpr uint32_t kPackWindow = 256 * kPackBytesPerThread;       // 16 KiB of a row per iteration: most rows of an 8192-wide plane are one
constexpr uint32_t kPackStageWords = (2 * kPackWindow + 16) / 4;  // LDS staging: every byte may be 0xFF, + alignment slack

// 4 bytes at any address (gfx950 has unaligned LDS and global access: one ds_write_b32 / global_store_dword)
struct __attribute__((packed)) PackU32
{
  uint32_t v;
};

// sum over the 256 threads of the workgroup (returned to all) and the exclusive prefix of this thread
__device__ __forceinline__ uint32_t wg_scan256(uint32_t v, uint32_t *wave_tot /* LDS [4] */, uint32_t &total)
{
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t u = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d)
      incl += u;
  }
  wg_sync(); // the previous use of wave_tot is over
  if (lane == 63)
    wave_tot[wave] = incl;
  wg_sync();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4; w++)
  {
    const uint32_t t = wave_tot[w];
    before += w < wave ? t : 0;
    total += t;
  }
  return before + incl - v;
}

// sum of s over the 256 threads, returned to all
__device__ __forceinline__ unsigned long long wg_sum256(unsigned long long s, unsigned long long *wave_sum /* LDS [4] */)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    s += __shfl_xor(s, d, 64);
  wg_sync(); // the previous use of wave_sum is over
  if ((threadIdx.x & 63) == 0)
    wave_sum[threadIdx.x >> 6] = s;
  wg_sync();
  return wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// The copy of one row by one workgroup, in windows of 256 * BPT source bytes.  A thread takes BPT consecutive bytes; one
// workgroup scan per window gives it its place; the stuffed bytes are assembled in LDS at the position that has the alignment
// of their global address; the workgroup copies 16 bytes per thread and step out (only the two ends of a window are byte
// stores -- different rows and windows may share a dword of the scan, never a byte).
// In LDS, phase A writes every source dword where it would go if it held no 0xFF (an unaligned ds_write_b32; its
// range [p, p + 4 + #0xFF) belongs to it alone), phase B rewrites the few dwords that do, byte by byte with the zero after each
// 0xFF: 0.4 % of the bytes are 0xFF, so a wave goes through phase B two or three times for its 64 x BPT bytes.
// (The first version -- 8 bytes per thread and iteration, byte stores to global memory -- took 12 us for the 9.3 MB of an
// 8192 x 8192 scan; windows of 2 KiB through LDS 10 us)
// COHERENT: the segment was written by this workgroup earlier in the same kernel; the loads bypass the vector L1.
// 256 * BPT source bytes of a row in registers: BPT consecutive bytes per thread
template <bool COHERENT, uint32_t BPT>
struct PackWindow
{
  static constexpr uint32_t NW = BPT / 4;
  uint32_t w[NW];
  uint32_t nvalid;

  __device__ __forceinline__ void load(const uint8_t *seg, bool pair_ok, uint32_t nb, uint32_t c0)
  {
    const uint32_t i = c0 + threadIdx.x * BPT;
    nvalid = i < nb ? min(BPT, nb - i) : 0u;
#pragma unroll
    for (uint32_t j = 0; j < NW; j += 2)
    {
      w[j] = w[j + 1] = 0;
      if (4 * j < nvalid)
      {
        if (pair_ok)
        {
          unsigned long long v;
          if (COHERENT)
            v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(seg + i) + j / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            v = (reinterpret_cast<const unsigned long long *>(seg + i))[j / 2];
          w[j] = (uint32_t)v;
          w[j + 1] = (uint32_t)(v >> 32);
        }
        else
        {
          const uint32_t *q = reinterpret_cast<const uint32_t *>(seg + i) + j;
          w[j] = COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[0];
          if (4 * j + 4 < nvalid)
            w[j + 1] = COHERENT ? __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[1];
        }
      }
    }
  }
};

template <bool COHERENT, uint32_t BPT>
__device__ __forceinline__ void pack_write_windows(PackWindow<COHERENT, BPT> &win /* window 0, loaded */, const uint8_t *seg, bool pair_ok, uint32_t nb, uint8_t *out,
                                                   uint32_t *stage /* LDS, 16-byte aligned, 2 * 256 * BPT + 16 bytes */, uint32_t *wave_tot /* LDS [4] */, uint32_t &done)
{
  constexpr uint32_t NW = BPT / 4;
  uint8_t *stage_b = reinterpret_cast<uint8_t *>(stage);
  for (uint32_t c0 = 0; c0 < nb; c0 += 256 * BPT)
  {
    const uint32_t nvalid = win.nvalid;
    // 0xFF bytes per dword (only among the valid bytes), and which dwords phase B has to redo
    uint32_t cnt[NW], fix = 0, mine = nvalid;
#pragma unroll
    for (uint32_t j = 0; j < NW; j++)
    {
      uint32_t m = ff_bytes(win.w[j]);
      const uint32_t vb = nvalid > 4 * j ? min(4u, nvalid - 4 * j) : 0u;
      if (vb < 4)
        m = vb ? m & ((1u << (8 * vb)) - 1u) : 0u;
      cnt[j] = (uint32_t)__builtin_popcount(m);
      mine += cnt[j];
      fix |= (m || (vb && vb < 4)) ? 1u << j : 0u;
    }
    uint32_t total;
    const uint32_t at = wg_scan256(mine, wave_tot, total); // its barriers also end the previous window's copy out of LDS
    uint8_t *g = out + done;
    const uint32_t o = (uint32_t)((uintptr_t)g & 15);
    const uint32_t pos = o + at;
    { // phase A
      uint32_t c = 0;
#pragma unroll
      for (uint32_t j = 0; j < NW; j++)
      {
        if (4 * j < nvalid)
          reinterpret_cast<PackU32 *>(stage_b + pos + 4 * j + c)->v = win.w[j];
        c += cnt[j];
      }
    }
    if (c0 + 256 * BPT < nb)
      win.load(seg, pair_ok, nb, c0 + 256 * BPT); // the next window, in flight during phase B and the copy out
    { // phase B: in increasing j, so that c = the 0xFF bytes of this thread before dword j
      uint32_t c = 0;
      while (fix)
      {
        const uint32_t j = (uint32_t)__builtin_ctz(fix);
        fix &= fix - 1;
        uint32_t q = pos + 4 * j + c;
        const uint32_t v = reinterpret_cast<const PackU32 *>(stage_b + q)->v; // what phase A put there: w[j]
        const uint32_t vb = min(4u, nvalid - 4 * j);
        for (uint32_t k = 0; k < vb; k++)
        {
          const uint32_t b = (v >> (8 * k)) & 0xFFu;
          stage_b[q++] = (uint8_t)b;
          if (b == 0xFFu)
          {
            stage_b[q++] = 0;
            c++;
          }
        }
      }
    }
    wg_sync();
    // LDS byte x <-> global byte g - o + x, and g - o is 16-byte aligned
    uint8_t *gb = g - o;
    const uint32_t lim = o + total;
    for (uint32_t q = threadIdx.x; q * 16 < lim; q += 256)
    {
      const uint32_t lo = q * 16;
      const uint4 v = reinterpret_cast<const uint4 *>(stage)[q];
      if (lo >= o && lo + 16 <= lim)
        *reinterpret_cast<uint4 *>(gb + lo) = v;
      else
      {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t k = 0; k < 16; k++)
          if (lo + k >= o && lo + k < lim)
            gb[lo + k] = (uint8_t)(vv[k >> 2] >> (8 * (k & 3)));
      }
    }
    done += total;
  }
}

// The row in two steps, so that the caller can find out where the row goes (sum the lengths of the rows before it, or wait for
// them) while the row's first window is on its way from L2:  PackRow p; p.begin(..); base = ...; p.finish(out + base, ..);
template <bool COHERENT>
struct PackRow
{
  PackWindow<COHERENT, 16> small; // rows up to 4 KiB: more threads with something to do
  PackWindow<COHERENT, kPackBytesPerThread> big;
  const uint8_t *seg;
  uint32_t nb;
  bool pair_ok, is_small;

  __device__ __forceinline__ void begin(const uint8_t *seg_, size_t seg_stride, uint32_t nb_)
  {
    seg = seg_;
    nb = nb_;
    pair_ok = ((((uintptr_t)seg) | seg_stride) & 7) == 0; // 8-byte loads that stay inside the row's stride
    is_small = nb <= 256 * 16;
    if (is_small)
      small.load(seg, pair_ok, nb, 0);
    else
      big.load(seg, pair_ok, nb, 0);
  }

  __device__ __forceinline__ void finish(uint8_t *out, bool marker, uint32_t rst_m, uint32_t *stage /* LDS [kPackStageWords], 16-byte aligned */, uint32_t *wave_tot /* LDS [4] */)
  {
    uint32_t done = 0; // bytes of this row already written
    if (is_small)
      pack_write_windows<COHERENT, 16>(small, seg, pair_ok, nb, out, stage, wave_tot, done);
    else
      pack_write_windows<COHERENT, kPackBytesPerThread>(big, seg, pair_ok, nb, out, stage, wave_tot, done);
    if (threadIdx.x == 0 && marker)
    {
      out[done] = 0xFF;
      out[done + 1] = (uint8_t)(0xD0 + rst_m);
    }
  }
};

// ---- the one-launch encoder's chain between the rows (k_px_huffman_rows<.., PACK>)
// work[0]: launch epoch, work[1]: sticky failure word, work[2 + r]: epoch tag << 32 | stuffed length of row r.  The caller zeroes
// `work` once; every launch leaves it ready for the next one (also for a replay of the same captured launch, which is why the epoch
// lives on the device and not in the kernel arguments).  A row reads the epoch when it starts and publishes when it has coded its
// segment; the LAST row, once it has seen every other row's tag, opens the next epoch -- at that point every row of this launch
// has read the epoch long ago, and rows still waiting compare the tags with their own copy of it.
// Failure: a row whose predecessors do not all publish within kChainDeadlineTicks (ONE deadline for the whole wait, on the constant
// 100 MHz clock) sets bit 0 of work[1] and writes no scan.  The last row publishes row_off[n_rows] = UINT64_MAX when that bit is set -- it waits
// until every other row has DECIDED (chain_row_decided below), not merely published -- and does NOT open the next epoch; every later launch on the same work
// array sees work[1] at its start, writes UINT64_MAX and does nothing else, until the caller zeroes the array again.
// row_off[n_rows] == UINT64_MAX is thus the one failure indicator of a launch (include/mdct.h).
// Forward progress rests on in-order dispatch: a row only ever waits for rows with smaller workgroup indices, which the hardware has
// started before it (one workgroup queue per launch, dealt round-robin to the XCDs in index order), and every row publishes BEFORE it
// waits -- so the lowest row that has not published yet is running, not queued behind its waiters.
constexpr unsigned long long kChainDeadlineTicks = 100000000ull; // ~1 s of wall_clock64()

__device__ __forceinline__ uint32_t chain_epoch_tag(const unsigned long long *work)
{
  const uint32_t tag = (uint32_t)__hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  return tag ? tag : 1u; // 0 is what a slot of the zeroed work array holds
}
__device__ __forceinline__ bool chain_failed(const unsigned long long *work) { return (__hip_atomic_load(work + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) != 0; } // (bits 1..: rows of the running launch that have decided)

__device__ __forceinline__ void chain_publish(unsigned long long *work, uint32_t r, uint32_t tag, uint32_t len)
{
  __hip_atomic_store(work + 2 + r, ((unsigned long long)tag << 32) | len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum of the lengths of rows 0..r-1 once they are all published; ok = false if the deadline passed first (the caller then sets work[1])
__device__ __forceinline__ unsigned long long chain_base(const unsigned long long *work, uint32_t r, uint32_t tag, unsigned long long *wave_sum, bool &ok)
{
  unsigned long long s = 0;
  uint32_t bad = 0;
  const unsigned long long t0 = wall_clock64();
  for (uint32_t i0 = threadIdx.x; i0 < r; i0 += 4 * 256)
  { // four tags per thread in flight; then wait for those that were not there yet
    unsigned long long v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      v[k] = i0 + 256 * k < r ? __hip_atomic_load(work + 2 + i0 + 256 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
    {
      for (uint32_t spins = 0; (uint32_t)(v[k] >> 32) != tag; spins++)
      {
        if (bad || ((spins & 63) == 63 && wall_clock64() - t0 > kChainDeadlineTicks))
        {
          bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        v[k] = __hip_atomic_load(work + 2 + i0 + 256 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s += (uint32_t)v[k];
    }
  }
  const unsigned long long base = wg_sum256(s, wave_sum);
  ok = wg_sum256(bad, wave_sum) == 0;
  return base;
}

// work[1] = failure bit (bit 0, sticky) + the number of rows of the running launch that have decided, times two.  A row adds 2 once its
// own verdict is in work[1] (after its fetch_or, if it gave up); the last row waits for n_rows - 1 of them, reads the bit, and -- when the
// launch is good -- leaves the word zero for the next launch.  A failed launch leaves it non-zero: chain_failed() for every later one.
__device__ __forceinline__ void chain_row_decided(unsigned long long *work) { __hip_atomic_fetch_add(work + 1, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the last row, one thread: true = every other row decided and none gave up (work[1] is zero again); false = failure bit set / deadline
__device__ __forceinline__ bool chain_all_decided(unsigned long long *work, uint32_t others)
{
  const unsigned long long t0 = wall_clock64();
  for (uint32_t spins = 0;; spins++)
  {
    const unsigned long long v = __hip_atomic_load(work + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v & 1)
      return false;
    if ((v >> 1) >= others)
    {
      __hip_atomic_store(work + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if ((spins & 63) == 63 && wall_clock64() - t0 > kChainDeadlineTicks)
    {
      __hip_atomic_fetch_or(work + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// the last row, one thread, after chain_base
__device__ __forceinline__ void chain_next_epoch(unsigned long long *work, uint32_t tag)
{
  __hip_atomic_store(work, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace mdct
