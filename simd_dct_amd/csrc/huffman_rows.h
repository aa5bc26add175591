// huffman_rows.h -- baseline Huffman coding (ITU-T T.81 Annex C, F.1.2) of one chunk of a block row's run/level records,
// shared by k_huffman_rows (stages.hip: records from HBM) and the fused pixels -> Huffman rows kernel
// (mdct_kernels.hip: records only ever exist in LDS).  Device code only; no reference counterpart.
#pragma once
#include <hip/hip_runtime.h>
#include "wg_sync.h"
#include <stdint.h>

namespace mdct
{

struct HuffArgs
{
  const int16_t *levels;
  const uint8_t *runs, *counts;
  uint8_t *out;
  uint32_t *seg_bytes;
  size_t seg_stride;
  uint32_t bpr, by0;
  uint32_t dc[12];  // size << 16 | code per DC category
  uint32_t ac[256]; // size << 16 | code per RRRRSSSS
};

constexpr uint32_t kHuffRing = 2048; // words of bit stream held in LDS (power of two)

// 0x80 in every byte of w that equals 0xFF (exact: no carries between bytes)
__device__ __forceinline__ uint32_t ff_bytes(uint32_t w)
{
  const uint32_t t = ~w; // a zero byte of t is an 0xFF byte of w
  return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}

// A parked pair is  run << 16 | (uint16_t)level  (top 10 bits zero).  The counting walk replaces it by
// its token  length << 27 | Huffman code and amplitude bits  (length >= 2: top 5 bits non-zero); pairs
// that need ZRL codes first (rare) stay as they are and are coded again by the emitting walk.
struct HuffTok
{
  uint32_t bits, len;
};

__device__ __forceinline__ HuffTok huff_dc_token(int diff, const uint32_t *dc)
{
  diff = diff > 2047 ? 2047 : (diff < -2047 ? -2047 : diff);                        // 8-bit baseline: categories 0..11 (F.1.2.1.1)
  const int s = diff ? 32 - __builtin_clz((uint32_t)(diff < 0 ? -diff : diff)) : 0; // SSSS: bits of |DIFF|
  const uint32_t e = dc[s];
  return {((e & 0xFFFFu) << s) | ((uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << s) - 1u)), (e >> 16) + (uint32_t)s};
}

__device__ __forceinline__ HuffTok huff_ac_token(int r, int l, const uint32_t *ac)
{                                                 // r: 0..15 zeros before the coefficient
  l = l > 1023 ? 1023 : (l < -1023 ? -1023 : l); // categories 1..10 (F.1.2.2.1)
  const int amp = l + (l >> 31);                 // F.1.2.2.1: a negative value is coded as value - 1, low SSSS bits
  int lead; // leading bits equal to the sign bit
  asm("v_ffbh_i32 %0, %1" : "=v"(lead) : "v"(amp));
  const int s = 32 - lead; // SSSS = bits of |l| = significant bits of amp (l != 0)
  const uint32_t e = ac[(r << 4) | s];
  return {((e & 0xFFFFu) << s) | ((uint32_t)amp & ((1u << s) - 1u)), (e >> 16) + (uint32_t)s}; // <= 16 + 10 bits
}

// The state of one block row's bit stream across its chunks, and the coder of one chunk.
//   WAVES     waves of the workgroup; a chunk is 64 * WAVES consecutive blocks, lane = block
//   STAGE     pairs of a block that sit in its LDS row `rec` (one dword each)
//   TAIL      a block's pairs beyond STAGE are read from HBM (lv_g / rn_g); otherwise STAGE == 64 holds them all
//   LATE_DC   the DC predictor of a wave's first block is not known up front (the fused kernel: it is the previous wave's
//             last block, transformed in this very chunk): every wave publishes its first and last DC beside its bit count,
//             and all of them derive the four missing token lengths after the barrier
template <int WAVES, int STAGE, bool TAIL, bool LATE_DC>
struct HuffRowCoder
{
  const uint32_t *ac, *dc; // LDS copies of the tables
  uint32_t *ring;          // [kHuffRing], zeroed
  uint32_t (*tot)[WAVES];  // [2][WAVES]
  int (*dcx)[2][WAVES];    // [2][first / last][WAVES]  (LATE_DC)
  uint32_t *out_w;         // the row's segment
  uint32_t zrl, eob;       // size << 16 | code
  uint32_t bpr;
  uint32_t base_bits = 0;  // bits of the row produced by earlier chunks
  uint32_t par = 0;
  int carry_dc = 0;        // LATE_DC: DC of the previous chunk's last block

  // blocks c0 + 64 * wave + lane of the row; the lane's record: n pairs, the first min(n, STAGE) in rec[]
  __device__ __forceinline__ void chunk(uint32_t c0, uint32_t *rec, int n, bool live, int prev_dc, const int16_t *lv_g, const uint8_t *rn_g)
  {
    constexpr uint32_t kChunk = 64 * WAVES;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = n < STAGE ? n : STAGE; // pairs of this block that sit in LDS
    // 1. DC of this block and of its predecessor
    const uint32_t e0 = rec[0];
    const bool has_dc = n > 0 && (e0 >> 16) == 0; // the first pair sits at scan position 0: it is the DC coefficient
    const int my_dc = has_dc ? (int)(int16_t)e0 : 0;
    const int up = __shfl_up(my_dc, 1, 64);
    HuffTok dct = huff_dc_token(my_dc - (lane == 0 ? prev_dc : up), dc);
    const bool dc_later = LATE_DC && lane == 0;
    if constexpr (LATE_DC)
    {
      if (lane == 0)
        dcx[par][0][wave] = my_dc;
      if (lane == 63)
        dcx[par][1][wave] = my_dc;
    }
    // 2. bits of this block; the parked pairs become tokens
    uint32_t bits = 0;
    bool need_eob = false;
    const int first_ac = has_dc ? 1 : 0;
    int ac_end = n; // pairs [first_ac, ac_end) are coded
    if (live)
    {
      bits = dc_later ? 0u : dct.len;
      int pos = has_dc ? 0 : -1; // scan position of the last coded coefficient
      uint32_t lmin = 0xFFFFu;   // smallest AC level seen, as a 16-bit pattern: 0 only for a zero level
      auto count = [&](uint32_t e, int run, int l) -> uint32_t { // the pair's token, or e itself if ZRL codes precede it
        const int r = run + (pos >> 31); // zeros before it among the AC positions (position 0 is the DC's)
        pos += run + 1;
        lmin = min(lmin, (uint32_t)l & 0xFFFFu);
        const HuffTok t = huff_ac_token(r & 15, l, ac);
        bits += t.len + (uint32_t)(r >> 4) * (zrl >> 16);
        return r > 15 ? e : t.len << 27 | t.bits;
      };
      int i = first_ac;
      for (; i + 1 < nl; i += 2)
      { // two pairs per trip: their LDS reads and table lookups overlap
        const uint32_t ea = rec[i], eb = rec[i + 1];
        const uint32_t ta = count(ea, (int)(ea >> 16), (int)(int16_t)ea), tb = count(eb, (int)(eb >> 16), (int)(int16_t)eb);
        rec[i] = ta;
        rec[i + 1] = tb;
      }
      if (i < nl)
      {
        const uint32_t e = rec[i];
        rec[i] = count(e, (int)(e >> 16), (int)(int16_t)e);
        i++;
      }
      if constexpr (TAIL)
        for (; i < n; i++)
          count(0u, (int)rn_g[i], (int)lv_g[i]);
      if (pos > 63 || lmin == 0)
      { // not a block (positions past 63, or a zero level): coded as its DC coefficient alone -- keeps the worst case of F.1.2
        ac_end = first_ac;
        bits = dc_later ? 0u : dct.len;
        pos = 0;
      }
      need_eob = pos < 63;
      if (need_eob)
        bits += eob >> 16;
    }
    // 3. exclusive scan: inside the wave, then over the waves
    uint32_t incl = bits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t v = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d)
        incl += v;
    }
    if (lane == 63)
      tot[par][wave] = incl;
    wg_sync(); // also: every wave has flushed (and cleared) the previous chunk's words
    uint32_t wave_start = 0, chunk_bits = 0, lane0_len = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WAVES; w++)
    {
      uint32_t t = tot[par][w];
      if constexpr (LATE_DC)
      { // the DC token of wave w's first block: predictor = the last block of wave w - 1 (of the previous chunk for w == 0)
        const int pred = w == 0 ? (c0 == 0 ? 0 : carry_dc) : dcx[par][1][w - 1];
        const HuffTok t0 = huff_dc_token(dcx[par][0][w] - pred, dc);
        const uint32_t l0 = c0 + 64 * w < bpr ? t0.len : 0u;
        t += l0;
        if (w == wave)
        {
          lane0_len = l0;
          if (lane == 0)
            dct = t0;
        }
      }
      wave_start += w < wave ? t : 0;
      chunk_bits += t;
    }
    if constexpr (LATE_DC)
      carry_dc = dcx[par][1][WAVES - 1];
    const uint32_t end_bits = base_bits + chunk_bits;
    const uint32_t w_first = base_bits >> 5, w_end = end_bits >> 5; // complete words of the row after this chunk: [.., w_end)
    // 4. + 5. emit and flush, one window of the ring at a time (one window unless the chunk exceeds 4 bit/px)
    for (uint32_t win = w_first; win <= w_end; win += kHuffRing)
    {
      if (win != w_first)
      { // the previous window's slots are cleared -- and the clearing ds_writes have LANDED: the compiler emits this barrier without
        // the s_waitcnt lgkmcnt(0) every other barrier of the kernel gets (ISA, ROCm 7.2), and a clear that is still queued when
        // another SIMD's wave passes the barrier overtakes that wave's ds_or: one row in ~10^5 of those that need several windows
        // lost bits of a word (tools/soak_jpeg_scan.py)
        wg_sync();
      }
      if (live)
      {
        const uint32_t cur = base_bits + wave_start + (incl - bits) + (lane ? lane0_len : 0u); // bit position in the row
        uint32_t widx = cur >> 5;
        uint32_t acc = 0;         // the bits not yet written: acc < 2^nacc
        uint32_t nacc = cur & 31; // pretend that many zero bits precede: OR leaves the neighbour's bits alone
        auto put = [&](uint32_t tok, uint32_t len) { // 1 <= len <= 27, tok < 2^len
          const uint32_t total = nacc + len;
          if (total < 32)
          {
            acc = (acc << len) | tok;
            nacc = total;
          }
          else
          { // the word is complete: the pending bits and the head of the token
            const uint32_t over = total - 32;
            if (widx - win < kHuffRing)
              atomicOr(&ring[widx & (kHuffRing - 1)], (acc << ((32 - nacc) & 31)) | (tok >> over));
            widx++;
            acc = tok & ((1u << over) - 1u);
            nacc = over;
          }
        };
        auto put_pair = [&](int r, int l) {
          for (; r > 15; r -= 16)
            put(zrl & 0xFFFFu, zrl >> 16);
          const HuffTok t = huff_ac_token(r, l, ac);
          put(t.bits, t.len);
        };
        put(dct.bits, dct.len);
        int i = first_ac;
        const int nl2 = ac_end < nl ? ac_end : nl;
        uint32_t e_next = rec[i]; // one token ahead (the row's skew dword makes rec[nl] readable)
        for (; i < nl2; i++)
        {
          const uint32_t e = e_next;
          e_next = rec[i + 1];
          if (e >> 27)
            put(e & 0x7FFFFFFu, e >> 27);
          else
            put_pair((int)(e >> 16) - (i == 0 ? 1 : 0), (int)(int16_t)e);
        }
        if constexpr (TAIL)
          for (; i < ac_end; i++)
            put_pair((int)rn_g[i] - (i == 0 ? 1 : 0), (int)lv_g[i]);
        if (need_eob)
          put(eob & 0xFFFFu, eob >> 16);
        if (nacc && widx - win < kHuffRing)
          atomicOr(&ring[widx & (kHuffRing - 1)], acc << (32 - nacc));
      }
      wg_sync();
      const uint32_t stop = min(w_end, win + kHuffRing);
      for (uint32_t w = win + tid; w < stop; w += kChunk)
      {
        out_w[w] = __builtin_bswap32(ring[w & (kHuffRing - 1)]);
        ring[w & (kHuffRing - 1)] = 0;
      }
    }
    base_bits = end_bits;
    par ^= 1;
  }

  // after the last chunk (behind a workgroup barrier): F.1.2.3 pads the last byte with 1-bits; returns the segment's length in bytes
  __device__ __forceinline__ uint32_t finish()
  {
    const uint32_t rem = base_bits & 31;
    if (rem)
    {
      const uint32_t pad = (8 - (rem & 7)) & 7;
      uint32_t w = ring[(base_bits >> 5) & (kHuffRing - 1)];
      if (pad)
        w |= ((1u << pad) - 1u) << (32 - rem - pad);
      out_w[base_bits >> 5] = __builtin_bswap32(w);
    }
    return (base_bits + 7) / 8;
  }
};

// ---------------------------------------------------------------------------------------
// The same coder on COMPACT records (the fused pixels -> Huffman rows kernel): the DC value travels in a register and
// the AC coefficients are 16-bit entries  run << 12 | (level & 0xFFF)  -- run 0..15, level within +-1023 (the range of
// baseline AC categories 1..10, F.1.2.2.1; the quantiser saturates there, as huff_ac_token does) -- with 0xF000 for
// "16 zeros" (ZRL, RRRRSSSS = 0xF0: an entry like any other).  A block's worst case is 63 entries = 126 bytes instead of
// 64 dwords, which is what lets four workgroups share a CU's LDS.  Entries cannot hold their 26-bit tokens, so the
// counting walk only sums lengths and the emitting walk looks the codes up again.
// ---------------------------------------------------------------------------------------
constexpr int kRec16Row = 66; // halfwords per block: 63 entries + the slot that takes the writes of zero coefficients, padded to an odd dword count

__device__ __forceinline__ HuffTok huff_ac_token12(uint32_t e, const uint32_t *ac)
{
  int l;
  asm("v_bfe_i32 %0, %1, 0, 12" : "=v"(l) : "v"(e));
  const int amp = l + (l >> 31); // F.1.2.2.1: a negative value is coded as value - 1, low SSSS bits
  int lead;
  asm("v_ffbh_i32 %0, %1" : "=v"(lead) : "v"(amp));
  const int s = l ? 32 - lead : 0; // SSSS; 0 for the ZRL entry
  const uint32_t c = ac[((e >> 12) << 4) | (uint32_t)s];
  return {((c & 0xFFFFu) << s) | ((uint32_t)amp & ((1u << s) - 1u)), (c >> 16) + (uint32_t)s};
}

template <int WAVES, uint32_t RING>
struct HuffRowCoder16
{
  const uint32_t *ac, *dc; // LDS copies of the tables
  uint32_t *ring;          // [RING], zeroed
  uint32_t (*tot)[WAVES];  // [2][WAVES]
  int (*dcx)[2][WAVES];    // [2][first / last][WAVES]
  uint32_t *out_w;         // the row's segment
  uint32_t eob;            // size << 16 | code
  uint32_t bpr;
  uint32_t base_bits = 0;
  uint32_t par = 0;
  int carry_dc = 0; // DC of the previous chunk's last block
  uint32_t ff = 0;  // 0xFF bytes among the words this thread has flushed (the bytes mdct_jpeg_pack_rows will stuff)

  // blocks c0 + 64 * wave + lane of the row; the lane's block: DC value, n AC entries in rec[], EOB needed unless position 63 is coded
  __device__ __forceinline__ void chunk(uint32_t c0, const uint16_t *rec, int n, bool live, int my_dc, bool need_eob)
  {
    constexpr uint32_t kChunk = 64 * WAVES;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *rec2 = reinterpret_cast<const uint32_t *>(rec); // two entries per LDS read
    if (!live)
    {
      my_dc = 0;
      n = 0;
    }
    const int up = __shfl_up(my_dc, 1, 64);
    HuffTok dct = huff_dc_token(my_dc - up, dc); // lane 0: replaced after the barrier
    if (lane == 0)
      dcx[par][0][wave] = my_dc;
    if (lane == 63)
      dcx[par][1][wave] = my_dc;
    // bits of this block
    uint32_t bits = 0;
    if (live)
    {
      bits = lane ? dct.len : 0u;
      for (int i = 0; i < n; i += 2)
      {
        const uint32_t w = rec2[i >> 1];
        bits += huff_ac_token12(w & 0xFFFFu, ac).len;
        if (i + 1 < n)
          bits += huff_ac_token12(w >> 16, ac).len;
      }
      if (need_eob)
        bits += eob >> 16;
    }
    uint32_t incl = bits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t v = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d)
        incl += v;
    }
    if (lane == 63)
      tot[par][wave] = incl;
    wg_sync(); // also: every wave has flushed (and cleared) the previous chunk's words
    uint32_t wave_start = 0, chunk_bits = 0, lane0_len = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WAVES; w++)
    { // the DC token of wave w's first block: predictor = the last block of wave w - 1 (of the previous chunk for w == 0)
      const int pred = w == 0 ? (c0 == 0 ? 0 : carry_dc) : dcx[par][1][w - 1];
      const HuffTok t0 = huff_dc_token(dcx[par][0][w] - pred, dc);
      const uint32_t l0 = c0 + 64 * w < bpr ? t0.len : 0u;
      const uint32_t t = tot[par][w] + l0;
      if (w == wave)
      {
        lane0_len = l0;
        if (lane == 0)
          dct = t0;
      }
      wave_start += w < wave ? t : 0;
      chunk_bits += t;
    }
    carry_dc = dcx[par][1][WAVES - 1];
    const uint32_t end_bits = base_bits + chunk_bits;
    const uint32_t w_first = base_bits >> 5, w_end = end_bits >> 5;
    for (uint32_t win = w_first; win <= w_end; win += RING)
    {
      if (win != w_first)
      { // the previous window's slots are cleared -- and the clearing ds_writes have LANDED: the compiler emits this barrier without
        // the s_waitcnt lgkmcnt(0) every other barrier of the kernel gets (ISA, ROCm 7.2), and a clear that is still queued when
        // another SIMD's wave passes the barrier overtakes that wave's ds_or: one row in ~10^5 of those that need several windows
        // lost bits of a word (tools/soak_jpeg_scan.py)
        wg_sync();
      }
      if (live)
      {
        const uint32_t cur = base_bits + wave_start + (incl - bits) + (lane ? lane0_len : 0u);
        uint32_t widx = cur >> 5;
        uint32_t acc = 0;
        uint32_t nacc = cur & 31;
        auto put = [&](uint32_t tok, uint32_t len) { // 1 <= len <= 27, tok < 2^len
          const uint32_t total = nacc + len;
          if (total < 32)
          {
            acc = (acc << len) | tok;
            nacc = total;
          }
          else
          {
            const uint32_t over = total - 32;
            if (widx - win < RING)
              atomicOr(&ring[widx & (RING - 1)], (acc << ((32 - nacc) & 31)) | (tok >> over));
            widx++;
            acc = tok & ((1u << over) - 1u);
            nacc = over;
          }
        };
        put(dct.bits, dct.len);
        for (int i = 0; i < n; i += 2)
        {
          const uint32_t w = rec2[i >> 1];
          const HuffTok ta = huff_ac_token12(w & 0xFFFFu, ac);
          put(ta.bits, ta.len);
          if (i + 1 < n)
          {
            const HuffTok tb = huff_ac_token12(w >> 16, ac);
            put(tb.bits, tb.len);
          }
        }
        if (need_eob)
          put(eob & 0xFFFFu, eob >> 16);
        if (nacc && widx - win < RING)
          atomicOr(&ring[widx & (RING - 1)], acc << (32 - nacc));
      }
      wg_sync();
      const uint32_t stop = min(w_end, win + RING);
      for (uint32_t w = win + tid; w < stop; w += kChunk)
      {
        const uint32_t v = ring[w & (RING - 1)];
        ff += (uint32_t)__builtin_popcount(ff_bytes(v));
        out_w[w] = __builtin_bswap32(v);
        ring[w & (RING - 1)] = 0;
      }
    }
    base_bits = end_bits;
    par ^= 1;
  }

  // one thread, after the last chunk (behind a workgroup barrier); *ff_last = 0xFF bytes in the padded last word
  __device__ __forceinline__ uint32_t finish(uint32_t *ff_last)
  {
    const uint32_t rem = base_bits & 31;
    *ff_last = 0;
    if (rem)
    {
      const uint32_t pad = (8 - (rem & 7)) & 7;
      uint32_t w = ring[(base_bits >> 5) & (RING - 1)];
      if (pad)
        w |= ((1u << pad) - 1u) << (32 - rem - pad);
      out_w[base_bits >> 5] = __builtin_bswap32(w);
      const uint32_t nbytes = (rem + 7) / 8; // the stream is MSB first: its bytes are the word's top ones
      *ff_last = (uint32_t)__builtin_popcount(ff_bytes(w) & (0xFFFFFFFFu << (8 * (4 - nbytes))));
    }
    return (base_bits + 7) / 8;
  }
};

} // namespace mdct
