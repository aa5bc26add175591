// mdct_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 8x8 block-DCT engine.
//
// Mapping (all kernels): ONE 8x8 BLOCK PER LANE, a wave64 carries 64 consecutive blocks.
// A lane owns its whole block in VGPRs, so the row pass, the column pass and the
// row<->column "transpose" are pure register renaming; cross-lane traffic exists only in
// the OUTPUT REORDER of the interleaved layouts and goes through wave-private LDS.
// There is no dense contraction anywhere: no MFMA.
//
// Bit-exactness contract (u8 path): IEEE binary32, one rounding per written operation,
// the exact association of the reference tier being reproduced, NO FMA.  This file is
// compiled with -ffp-contract=off and additionally pins contraction off below.
// (One fused multiply-add is written out by hand, in the scalar tiers' quantiser: it replaces  (uint8_t)roundf(c * 255.f)  by a form
//  proven equal for every float c in [0, 1] -- tools/check_roundf_forms.py -- not an operation of the reference's transform.)
//
// Reference lines restated (rainerzufalldererste/simd_dct, src/simd_dct.cpp):
//   1-D kernels   K_AVX :2158-2184   K_SSE :434-654   K_TRUE :138-172
//   block drivers B1 :2064-2262  B2 :896-1103  B3 :1540-1704  B4 :177-298  B5 :300-395
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "wg_sync.h"
#include "huffman_rows.h"
#include "pack_rows.h"
#include "mdct_kernels.h"
#include "scan_records.h"

#pragma clang fp contract(off)

namespace mdct
{

// The seven butterfly constants (simd_dct.cpp:140-146) travel in the kernarg segment
// (DctConsts, mdct_kernels.h) and therefore live in SGPRs: a VALU op with an SGPR operand is
// a 4-byte encoding, the same op with a 32-bit literal is 8 bytes, and these kernels are
// ~2000 straight-line VALU instructions per wave.

enum { K_AVX = 0, K_SSE = 1, K_TRUE = 2 }; // the reference's three 1-D kernels (the engine-own one is aan_fwd8 below)

// ---------------------------------------------------------------------------------------
// Engine-own 1-D kernels (int16 / float32 / 8-bit round-trip paths; no reference counterpart, so the
// reference tiers' no-FMA rule does not bind them): the scaled Arai-Agui-Nakajima butterfly, 5 mul + 29 add
// per 8 points instead of 28 + 28, with 4 of the 5 multiplies FUSED into the additions they feed (round 6):
// 30 operations per pass.  On MI355X a v_pk_fma_f32 issues in the time of a v_pk_mul_f32 (~4.2 cycles per
// wave64, profiles/valu_issue_costs.json), so every fused pair saves a packed issue slot in the packed forms
// below (the VALU-issue-bound k_u8_batch); a scalar v_fma_f32 costs about a v_mul + v_add, so the scalar forms
// here gain nothing and lose nothing -- they follow because all forms must produce the same bits.
// Explicit __builtin_fmaf only: contraction stays off, the compiler fuses nothing on its own.
// The scale factors live in 64-entry tables applied where a multiply exists anyway (quantise / dequantise),
// and for the fused round trip they cancel to exactly 1/64, which the final rounding step
// absorbs (see rne_i16_bits).  Same operations on the same operands as the CPU checker (orc_aan_*).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void aan_fwd8(const DctConsts &C, float &p0, float &p1, float &p2, float &p3, float &p4, float &p5, float &p6, float &p7)
{
  const float t0 = p0 + p7, t7 = p0 - p7, t1 = p1 + p6, t6 = p1 - p6;
  const float t2 = p2 + p5, t5 = p2 - p5, t3 = p3 + p4, t4 = p3 - p4;
  const float e10 = t0 + t3, e13 = t0 - t3, e11 = t1 + t2, e12 = t1 - t2;
  const float s1 = e12 + e13;
  const float o10 = t4 + t5, o11 = t5 + t6, o12 = t6 + t7;
  const float z5 = (o10 - o12) * C.c382;
  const float z2 = __builtin_fmaf(C.c541, o10, z5);
  const float z4 = __builtin_fmaf(C.c1306, o12, z5);
  const float z11 = __builtin_fmaf(o11, C.c707, t7), z13 = __builtin_fmaf(-o11, C.c707, t7);
  p0 = e10 + e11;
  p4 = e10 - e11;
  p2 = __builtin_fmaf(s1, C.c707, e13);
  p6 = __builtin_fmaf(-s1, C.c707, e13);
  p5 = z13 + z2;
  p3 = z13 - z2;
  p1 = z11 + z4;
  p7 = z11 - z4;
}

__device__ __forceinline__ void aan_inv8(const DctConsts &C, float &p0, float &p1, float &p2, float &p3, float &p4, float &p5, float &p6, float &p7)
{
  const float e10 = p0 + p4, e11 = p0 - p4;
  const float e13 = p2 + p6;
  const float e12 = __builtin_fmaf(p2 - p6, C.c1414, -e13);
  const float t0 = e10 + e13, t3 = e10 - e13, t1 = e11 + e12, t2 = e11 - e12;
  const float z13 = p5 + p3, z10 = p5 - p3, z11 = p1 + p7, z12 = p1 - p7;
  const float t7 = z11 + z13;
  const float z5 = (z10 + z12) * C.c1847;
  const float o10 = __builtin_fmaf(C.c1082, z12, -z5);
  const float o12 = __builtin_fmaf(-C.c2613, z10, z5);
  const float t6 = o12 - t7;
  const float t5 = __builtin_fmaf(z11 - z13, C.c1414, -t6);
  const float t4 = o10 + t5;
  p0 = t0 + t7;
  p7 = t0 - t7;
  p1 = t1 + t6;
  p6 = t1 - t6;
  p2 = t2 + t5;
  p5 = t2 - t5;
  p4 = t3 + t4;
  p3 = t3 - t4;
}

__device__ __forceinline__ void raw_fwd(const DctConsts &C, float (&b)[8][8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
    aan_fwd8(C, b[r][0], b[r][1], b[r][2], b[r][3], b[r][4], b[r][5], b[r][6], b[r][7]);
#pragma unroll
  for (int c = 0; c < 8; c++)
    aan_fwd8(C, b[0][c], b[1][c], b[2][c], b[3][c], b[4][c], b[5][c], b[6][c], b[7][c]);
}

__device__ __forceinline__ void raw_inv(const DctConsts &C, float (&b)[8][8])
{
#pragma unroll
  for (int c = 0; c < 8; c++)
    aan_inv8(C, b[0][c], b[1][c], b[2][c], b[3][c], b[4][c], b[5][c], b[6][c], b[7][c]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    aan_inv8(C, b[r][0], b[r][1], b[r][2], b[r][3], b[r][4], b[r][5], b[r][6], b[r][7]);
}

// ---------------------------------------------------------------------------------------
// K_AVX on packed fp32 (v_pk_add_f32 / v_pk_mul_f32): 28 instead of 56 instructions per 8-point
// transform, identical bits.  Each half of a packed op is an individually rounded IEEE op, and
// a + (-b) == a - b, so only the instruction count changes; on gfx950 a packed op issues in
// ~5 cycles against ~2.7 for each of the two scalar ops it replaces, and at fewer than 8 waves
// per SIMD the gap is wider (profiles/r01_valubench2.log, profiles/r02_q32_variants.md).
//
// Row pass ("horizontal"): the 8 values of one block row sit in 4 adjacent register pairs
// (p0,p1)(p2,p3)(p4,p5)(p6,p7).  op_sel picks which half of each source feeds each half of the
// result and neg_lo/neg_hi negate per half, so every butterfly stage is ONE packed op producing
// two DIFFERENT quantities of simd_dct.cpp:2160-2183:
//   (x07p,x16p) (x25p,x34p) (x07m,x61m) (x25m,x43m) (pp,qp) (pm,qm) (o0,o4) (o2,o6)
//   (t1,t3) (t5,t7) (u1,u3) (u5,u7) (o1,o3) (o5,o7)
// The outputs come out paired (0,4)(2,6)(1,3)(5,7) along u, identically for every row, so the
// column pass ("vertical", simd_dct.cpp:2189-2215) is the plain butterfly on 4 column pairs.
// No register moves anywhere.  Written as inline asm: the compiler folds only trivial
// shuffles into op_sel and materialises the rest as v_mov / extra packed ops.
// ---------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MDCT_PKA(d, a, b, mods) asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b))
#define MDCT_PKM(d, a, k, mods) asm("v_pk_mul_f32 %0, %1, %2 " mods : "=v"(d) : "v"(a), "s"(k))
#define MDCT_PKF(d, a, k, c, mods) asm("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(d) : "v"(a), "s"(k), "v"(c)) // d = a * k + c, one rounding
// halves of the constant operand (src1) used for (lo, hi) of the result
#define MDCT_K_LL "op_sel:[0,0] op_sel_hi:[1,0]"
#define MDCT_K_HH "op_sel:[0,1] op_sel_hi:[1,1]"
#define MDCT_K_LH "op_sel:[0,0] op_sel_hi:[1,1]"
#define MDCT_K_HL "op_sel:[0,1] op_sel_hi:[1,0]"
#define MDCT_X "op_sel:[0,1] op_sel_hi:[1,0]" // lo = a.lo (+) b.hi, hi = a.hi (+) b.lo
#define MDCT_NEG_B "neg_lo:[0,1] neg_hi:[0,1]"  // a - b in both halves

struct PkConsts
{
  f32x2 af, cd, be, nm; // (Ca,Cf) (Cc,Cd) (Cb,Ce) (Cn, rounding constant)
  f32x2 da, fd, fc, ca; // K_TRUE only: (Cd,Ca) (Cf,Cd) (Cf,Cc) (Cc,Ca)
  f32x2 bias;           // SSE tiers: (1/255, 127.0f);  scalar tiers: (255.0f, pred(0.5)) and nm = (Cn, 127/255)
  f32x2 div;            // scalar tiers: (c, c2) = (rn(1/255), rn(1/255 - c)), see encode_block_pk
};
static_assert(sizeof(PkConsts) == sizeof(PkConstsArg), "PkConstsArg (mdct_kernels.h) is the kernel-argument image of PkConsts");

// Phase priorities (template flag PRIO of the block functions): a wave raises its issue priority as it advances through its
// phases, so that of the waves sharing a SIMD the one closest to its stores goes first (shortest remaining work first)
// instead of all of them finishing late together.  Measured on the fused int16 round trip at 3 waves/SIMD: 46.1 -> 44.7 us
// (profiles/r03_exp_i16_tile_and_priorities.log).
#define MDCT_PHASE_PRIO(n)                   \
  do                                         \
  {                                          \
    if constexpr (PRIO)                      \
    {                                        \
      __builtin_amdgcn_sched_barrier(0);     \
      __builtin_amdgcn_s_setprio(n);         \
      __builtin_amdgcn_sched_barrier(0);     \
    }                                        \
  } while (0)
// one block row (or column) held in 4 pairs -> (o0,o4) (o2,o6) (o1,o3) (o5,o7), each already times Cn.
// K selects the reference 1-D kernel being reproduced; they differ in the odd part only:
//   K_AVX  :2176-2183  o1 = t1 + (Cd x25m - Cf x43m)       o3 = t3 - (Ca x25m + Cd x43m)   (k=3 quirk)
//   K_SSE  :547-577    o1 = t1 + (Cd x25m + Cf x43m) (k=1 quirk)  o3 = t3 + (Cd x43m - Ca x25m)
//   K_TRUE :163-171    left to right: o1 = (t1 + Cd x25m) - Cf x43m, o3 = (t3 - Ca x25m) + Cd x43m, ...
// Statement order matters for speed, not for bits: every MDCT_PK* is a separate inline-asm statement, and for an asm
// statement that reads a VGPR written by the asm statement IMMEDIATELY before it the compiler inserts a wait state
// (it cannot see inside and assumes the gfx940 dst_sel forwarding hazard): 64-76 s_nop per wave in the order the
// formulas are usually written in.  The operations are therefore emitted so that none consumes its predecessor's
// result (the textbook order and the statement-per-operation forms of the other passes: tools/experiments/pk_forms_textbook.h, for A/B runs).
template <int K1D>
__device__ __forceinline__ void dct8_h(const PkConsts &K, f32x2 a01, f32x2 a23, f32x2 a45, f32x2 a67, f32x2 &o04, f32x2 &o26, f32x2 &o13, f32x2 &o57)
{
  if constexpr (K1D != K_TRUE)
  { // The whole line as ONE asm statement, registers allocated by hand: the four input pairs are reused as scratch
    // (A = a01, B = a23, C = a45, D = a67) plus three temporaries, 28 instructions, no compiler-inserted wait states
    // inside; results leave in T0 = (o0,o4), T1 = (o2,o6), C = (o1,o3), D = (o5,o7).  Same operations, same operands,
    // same modifiers as the statement-per-operation form below (which K_TRUE still runs and which stays the readable spec).
    f32x2 T0, T1, T2;
#define MDCT_XS " op_sel:[0,1] op_sel_hi:[1,0]"
#define MDCT_DCT8_H_ASM(U13MOD, O13MOD) \
    asm("v_pk_add_f32 %4, %0, %3" MDCT_XS "\n\t"                                  /* T0 = s1 = (p0+p7, p1+p6) */ \
        "v_pk_add_f32 %5, %1, %2" MDCT_XS "\n\t"                                  /* T1 = s2 = (p2+p5, p3+p4) */ \
        "v_pk_add_f32 %0, %0, %3" MDCT_XS " neg_lo:[0,1] neg_hi:[1,0]\n\t"        /* A = d = (p0-p7, p6-p1) */ \
        "v_pk_add_f32 %1, %1, %2" MDCT_XS " neg_lo:[0,1] neg_hi:[1,0]\n\t"        /* B = e = (p2-p5, p4-p3) */ \
        "v_pk_add_f32 %2, %4, %5" MDCT_XS "\n\t"                                  /* C = (pp, qp) */ \
        "v_pk_add_f32 %3, %4, %5" MDCT_XS " neg_lo:[0,1] neg_hi:[0,1]\n\t"        /* D = (pm, qm) */ \
        "v_pk_add_f32 %4, %2, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"   /* T0 = (pp+qp, pp-qp) */ \
        "v_pk_mul_f32 %5, %3, %9 op_sel:[0,0] op_sel_hi:[1,0]\n\t"                /* T1 = r = (Cb pm, Cb qm) */ \
        "v_pk_mul_f32 %2, %0, %7 op_sel:[0,0] op_sel_hi:[1,1]\n\t"                /* C = m1 = (Ca x07m, Cf x61m) */ \
        "v_pk_mul_f32 %3, %3, %9 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                /* D = t = (Ce pm, Ce qm) */ \
        "v_pk_mul_f32 %6, %0, %8 op_sel:[0,0] op_sel_hi:[1,0]\n\t"                /* T2 = m2 = (Cc x07m, Cc x61m) */ \
        "v_pk_add_f32 %5, %5, %3" MDCT_XS " neg_hi:[1,0]\n\t"                     /* T1 = (Cb pm + Ce qm, Ce pm - Cb qm) */ \
        "v_pk_add_f32 %2, %2, %6" MDCT_XS " neg_lo:[0,1]\n\t"                     /* C = t13 */ \
        "v_pk_mul_f32 %3, %0, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                /* D = m3 = (Cd x07m, Cd x61m) */ \
        "v_pk_mul_f32 %6, %0, %7 op_sel:[0,1] op_sel_hi:[1,0]\n\t"                /* T2 = m4 = (Cf x07m, Ca x61m) */ \
        "v_pk_mul_f32 %0, %1, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                /* A = n1 = (Cd x25m, Cd x43m) */ \
        "v_pk_add_f32 %3, %3, %6" MDCT_XS "\n\t"                                  /* D = t57 */ \
        "v_pk_mul_f32 %6, %1, %7 op_sel:[0,0] op_sel_hi:[1,1]\n\t"                /* T2 = n2 = (Ca x25m, Cf x43m) */ \
        "v_pk_mul_f32 %4, %4, %10 op_sel:[0,0] op_sel_hi:[1,0]\n\t"               /* T0 *= Cn */ \
        "v_pk_add_f32 %0, %0, %6" MDCT_XS " " U13MOD "\n\t"                              /* A = u13: K_AVX neg_lo:[0,1] (Cd x25m - Cf x43m, ..), K_SSE neg_hi:[0,1] (k=1 quirk, :550) */ \
        "v_pk_mul_f32 %6, %1, %7 op_sel:[0,1] op_sel_hi:[1,0]\n\t"                /* T2 = n3 = (Cf x25m, Ca x43m) */ \
        "v_pk_mul_f32 %1, %1, %8 op_sel:[0,0] op_sel_hi:[1,0]\n\t"                /* B = n4 = (Cc x25m, Cc x43m) */ \
        "v_pk_mul_f32 %5, %5, %10 op_sel:[0,0] op_sel_hi:[1,0]\n\t"               /* T1 *= Cn */ \
        "v_pk_add_f32 %6, %6, %1" MDCT_XS " neg_lo:[0,1]\n\t"                     /* T2 = u57 */ \
        "v_pk_add_f32 %2, %2, %0 " O13MOD "\n\t"                                        /* C = o13: K_AVX neg_hi:[0,1] (the k=3 quirk of :2181), K_SSE plain */ \
        "v_pk_add_f32 %3, %3, %6\n\t"                                            /* D = o57 */ \
        "v_pk_mul_f32 %2, %2, %10 op_sel:[0,0] op_sel_hi:[1,0]\n\t" \
        "v_pk_mul_f32 %3, %3, %10 op_sel:[0,0] op_sel_hi:[1,0]" \
        : "+v"(a01), "+v"(a23), "+v"(a45), "+v"(a67), "=&v"(T0), "=&v"(T1), "=&v"(T2) \
        : "s"(K.af), "s"(K.cd), "s"(K.be), "s"(K.nm))
    if constexpr (K1D == K_AVX)
      MDCT_DCT8_H_ASM("neg_lo:[0,1]", "neg_hi:[0,1]");
    else
      MDCT_DCT8_H_ASM("neg_hi:[0,1]", "");
#undef MDCT_DCT8_H_ASM
#undef MDCT_XS
    o04 = T0; o26 = T1; o13 = a45; o57 = a67;
    return;
  }
  f32x2 s1, s2, d, e, pqp, pqm, r, t, m1, m2, m3, m4, t13, t57;
  MDCT_PKA(s1, a01, a67, MDCT_X);                                  // (p0+p7, p1+p6)
  MDCT_PKA(s2, a23, a45, MDCT_X);                                  // (p2+p5, p3+p4)
  MDCT_PKA(d, a01, a67, MDCT_X " neg_lo:[0,1] neg_hi:[1,0]");      // (p0-p7, p6-p1)
  MDCT_PKA(e, a23, a45, MDCT_X " neg_lo:[0,1] neg_hi:[1,0]");      // (p2-p5, p4-p3)
  MDCT_PKA(pqp, s1, s2, MDCT_X);                                   // (x07p+x34p, x16p+x25p)
  MDCT_PKA(pqm, s1, s2, MDCT_X " " MDCT_NEG_B);                    // (x07p-x34p, x16p-x25p)
  MDCT_PKM(m1, d, K.af, MDCT_K_LH);                                // (Ca x07m, Cf x61m)
  MDCT_PKA(o04, pqp, pqp, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]"); // (pp+qp, pp-qp)
  MDCT_PKM(r, pqm, K.be, MDCT_K_LL);                               // (Cb pm, Cb qm)
  MDCT_PKM(t, pqm, K.be, MDCT_K_HH);                               // (Ce pm, Ce qm)
  MDCT_PKM(m2, d, K.cd, MDCT_K_LL);                                // (Cc x07m, Cc x61m)
  MDCT_PKA(o26, r, t, MDCT_X " neg_hi:[1,0]");                     // (Cb pm + Ce qm, Ce pm - Cb qm)
  MDCT_PKM(m3, d, K.cd, MDCT_K_HH);                                // (Cd x07m, Cd x61m)
  MDCT_PKM(m4, d, K.af, MDCT_K_HL);                                // (Cf x07m, Ca x61m)
  MDCT_PKA(t13, m1, m2, MDCT_X " neg_lo:[0,1]");                   // (Ca x07m - Cc x61m, Cf x61m + Cc x07m)
  if constexpr (K1D == K_TRUE)
  { // sequential association: two more terms added one after the other
    f32x2 g1, g2, g3, g4, h13, h57;
    MDCT_PKM(g1, e, K.da, "op_sel:[0,0] op_sel_hi:[0,1]");         // (Cd x25m, Ca x25m)
    MDCT_PKA(t57, m3, m4, MDCT_X);                                 // (Cd x07m + Ca x61m, Cd x61m + Cf x07m)
    MDCT_PKM(g3, e, K.fc, "op_sel:[0,0] op_sel_hi:[0,1]");         // (Cf x25m, Cc x25m)
    MDCT_PKM(g2, e, K.fd, "op_sel:[1,0] op_sel_hi:[1,1]");         // (Cf x43m, Cd x43m)
    MDCT_PKA(h13, t13, g1, "neg_hi:[0,1]");                        // (t1 + Cd x25m, t3 - Ca x25m)
    MDCT_PKM(g4, e, K.ca, "op_sel:[1,0] op_sel_hi:[1,1]");         // (Cc x43m, Ca x43m)
    MDCT_PKA(h57, t57, g3, "");                                    // (t5 + Cf x25m, t7 + Cc x25m)
    MDCT_PKM(o04, o04, K.nm, MDCT_K_LL);
    MDCT_PKA(o13, h13, g2, "neg_lo:[0,1]");                        // (.. - Cf x43m, .. + Cd x43m)
    MDCT_PKM(o26, o26, K.nm, MDCT_K_LL);
    MDCT_PKA(o57, h57, g4, "neg_lo:[0,1]");                        // (.. - Cc x43m, .. + Ca x43m)
    MDCT_PKM(o13, o13, K.nm, MDCT_K_LL);
    MDCT_PKM(o57, o57, K.nm, MDCT_K_LL);
  }
  else
  {
    f32x2 n1, n2, n3, n4, u13, u57;
    MDCT_PKM(n1, e, K.cd, MDCT_K_HH);                              // (Cd x25m, Cd x43m)
    MDCT_PKA(t57, m3, m4, MDCT_X);                                 // (Cd x07m + Ca x61m, Cd x61m + Cf x07m)
    MDCT_PKM(n2, e, K.af, MDCT_K_LH);                              // (Ca x25m, Cf x43m)
    MDCT_PKM(n3, e, K.af, MDCT_K_HL);                              // (Cf x25m, Ca x43m)
    MDCT_PKM(n4, e, K.cd, MDCT_K_LL);                              // (Cc x25m, Cc x43m)
    if constexpr (K1D == K_AVX)
      MDCT_PKA(u13, n1, n2, MDCT_X " neg_lo:[0,1]");               // (Cd x25m - Cf x43m, Cd x43m + Ca x25m)
    else
    {
      static_assert(K1D == K_SSE, "unknown 1-D kernel");
      MDCT_PKA(u13, n1, n2, MDCT_X " neg_hi:[0,1]");               // (Cd x25m + Cf x43m [k=1 quirk, :550], Cd x43m - Ca x25m)
    }
    MDCT_PKM(o04, o04, K.nm, MDCT_K_LL);
    MDCT_PKA(u57, n3, n4, MDCT_X " neg_lo:[0,1]");                 // (Cf x25m - Cc x43m, Ca x43m + Cc x25m)
    MDCT_PKM(o26, o26, K.nm, MDCT_K_LL);
    if constexpr (K1D == K_AVX)
      MDCT_PKA(o13, t13, u13, "neg_hi:[0,1]");                     // (t1 + u1, t3 - u3): the k=3 quirk of :2181
    else
      MDCT_PKA(o13, t13, u13, "");
    MDCT_PKA(o57, t57, u57, "");                                   // (t5 + u5, t7 + u7)
    MDCT_PKM(o13, o13, K.nm, MDCT_K_LL);
    MDCT_PKM(o57, o57, K.nm, MDCT_K_LL);
  }
}

// the same transform down a PAIR of independent columns (or rows), p[r] = (B[r][u1], B[r][u2]), in place
template <int K1D>
__device__ __forceinline__ void dct8_v(const PkConsts &K, f32x2 (&p)[8])
{
  // Plain vector code: this pass needs no cross-half operand selection -- only whole-pair adds / subtracts and products
  // with one broadcast constant, which the compiler turns into v_pk_add_f32 / v_pk_mul_f32 itself (the broadcast and the
  // subtraction's sign fold into op_sel / neg modifiers: (-x)*c == -(x*c) and a + (-b) == a - b exactly; contraction is
  // off for the whole file).  Compiler-visible operations are scheduled with their latencies and draw no conservative
  // wait states (the asm statements of the horizontal pass do, see above).
  const f32x2 Ca = K.af.xx, Cf = K.af.yy, Cc = K.cd.xx, Cd = K.cd.yy, Cb = K.be.xx, Ce = K.be.yy, Cn = K.nm.xx;
  const f32x2 x07p = p[0] + p[7], x16p = p[1] + p[6], x25p = p[2] + p[5], x34p = p[3] + p[4];
  const f32x2 x07m = p[0] - p[7], x61m = p[6] - p[1], x25m = p[2] - p[5], x43m = p[4] - p[3];
  const f32x2 pp = x07p + x34p, pm = x07p - x34p, qp = x16p + x25p, qm = x16p - x25p;
  const f32x2 o0 = pp + qp, o4 = pp - qp;
  const f32x2 o2 = (Cb * pm) + (Ce * qm), o6 = (Ce * pm) - (Cb * qm);
  const f32x2 t1 = (Ca * x07m) - (Cc * x61m), t3 = (Cc * x07m) + (Cf * x61m);
  const f32x2 t5 = (Cd * x07m) + (Ca * x61m), t7 = (Cf * x07m) + (Cd * x61m);
  f32x2 o1, o3, o5, o7;
  if constexpr (K1D == K_TRUE)
  { // ((t + c1 x25m) +- c2 x43m), :166-171
    o1 = (t1 + (Cd * x25m)) - (Cf * x43m);
    o3 = (t3 - (Ca * x25m)) + (Cd * x43m);
    o5 = (t5 + (Cf * x25m)) - (Cc * x43m);
    o7 = (t7 + (Cc * x25m)) + (Ca * x43m);
  }
  else
  {
    const f32x2 u5 = (Cf * x25m) - (Cc * x43m), u7 = (Cc * x25m) + (Ca * x43m);
    if constexpr (K1D == K_AVX)
    {
      const f32x2 u1 = (Cd * x25m) - (Cf * x43m), u3 = (Ca * x25m) + (Cd * x43m);
      o1 = t1 + u1;
      o3 = t3 - u3; // the k=3 quirk of :2181
    }
    else
    {
      const f32x2 u1 = (Cd * x25m) + (Cf * x43m), u3 = (Cd * x43m) - (Ca * x25m); // k=1 quirk, :550
      o1 = t1 + u1;
      o3 = t3 + u3;
    }
    o5 = t5 + u5;
    o7 = t7 + u7;
  }
  p[0] = Cn * o0; p[1] = Cn * o1; p[2] = Cn * o2; p[3] = Cn * o3;
  p[4] = Cn * o4; p[5] = Cn * o5; p[6] = Cn * o6; p[7] = Cn * o7;
}

// ---------------------------------------------------------------------------------------
// Quantisers.  They return a word whose LOW BYTE is the quantised coefficient (the upper
// bits are not cleaned: every consumer stores or packs the low byte only).
//
// x86 cvtps_epi32 = RNE, "integer indefinite" 0x80000000 when out of range or NaN
// (simd_dct.cpp:2224, :1020).  For |v| < 2^31 the reference's clamp(127 + rne(v), 0, 255)
// equals 127 + rne(clamp(v, -127, 128)) (rne is monotone and the bounds are integers), and
// rne of a value in that range is the low byte of the float  v + 1.5*2^23  (the add rounds
// to nearest-even at unit granularity; 1.5*2^23 is even, so ties go the same way).  That is
// the fast form: v_mul, v_med3_f32, v_add_f32, v_add_u32 -- no convert instructions.
// The host selects SAFE (explicit emulation of the indefinite value) whenever a multiplier
// is non-finite or larger than 2^17: |coefficient| <= 2040 in every tier, so below that
// bound |v| < 2^31 always holds (mdct_api.hip).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t cvt_rne(float v) { return (int32_t)__builtin_rintf(v); } // v_rndne_f32 + v_cvt_i32_f32

__device__ __forceinline__ int32_t cvtps_epi32_exact(float v)
{
  if (!(__builtin_fabsf(v) < 2147483648.0f))
    return INT32_MIN;
  return cvt_rne(v);
}

__device__ __forceinline__ int32_t clamp255(int32_t v) { return min(max(v, 0), 255); } // v_med3_i32

// ---------------------------------------------------------------------------------------
// u8 forward + quantise + reorder.
// ---------------------------------------------------------------------------------------
// 8 pixels of one block row.  The reference takes any alignment (unaligned loads,
// simd_dct.cpp:2109); so does this: gfx950 global loads are alignment-free in hardware, the
// type below only stops the compiler from assuming 8-byte alignment.  Streamed once -> nt.
typedef unsigned int u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
__device__ __forceinline__ uint2 load8(const uint8_t *p)
{
  const u32x2_unaligned v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_unaligned *>(p));
  return make_uint2(v.x, v.y);
}

// byte N of a dword -> float in ONE instruction (v_cvt_f32_ubyteN).  Written as (pure,
// schedulable) inline asm, not as (float)((w >> 8N) & 0xFF): from the latter LLVM rewrites the
// first butterfly stage as integer SDWA adds followed by v_cvt_f32_i32 (exact, but ~1.6x the
// issue cycles on gfx950, where SDWA forms and converts are half rate).
template <int N>
__device__ __forceinline__ float ubyte_to_float(uint32_t w)
{
  float f;
  if constexpr (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
  else if constexpr (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
  else if constexpr (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
  else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
  return f;
}

// the lane's block as eight 8-byte rows, all eight loads in flight together
__device__ __forceinline__ void load_block_rows(const uint8_t *src, size_t pitch, uint2 (&rows)[8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
    rows[r] = load8(src + (size_t)r * pitch);
}

// B1 on packed fp32: raw bytes -> both passes -> quantise; out[v*8+u] = word whose low byte is the
// stored byte (SAFE) or its complement (fast form, see below).  `qt` holds the multipliers in PAIR
// ORDER, (v*4+j)*2 + {0,1} = coefficient (v, kPairA[j]) / (v, kPairB[j]), already negated for the
// fast form (mdct_api.hip).
//
// Fast form of clamp(127 + rne(v), 0, 255), |v| < 2^31 (simd_dct.cpp:2224): rne is odd-symmetric, so
// 127 + rne(v) = 255 - (rne(-v) + 128); -v = f * (-q) exactly; clamp -v to [-128, 127] and add
// 1.5*2^23 + 128 (even, so ties round as before): the low byte of that float is rne(-v) + 128, and
// the stored byte is its complement.  The complement is applied to whole dwords after the LDS
// reorder (4 v_not per lane instead of 64 integer adds).
constexpr int kPairA[4] = {0, 2, 1, 5}, kPairB[4] = {4, 6, 3, 7};
struct NoHook
{
  __device__ __forceinline__ void operator()() const {}
};
// after_rows: called between the row pass and the column pass, when the 16 registers of `rows` are dead (a caller that works on
// several tiles issues the next tile's loads there)
// qp[v * 4 + j]: the multiplier pair, from the kernel arguments (QuantTable) or through a wave-uniform pointer (plane batches)
template <bool SAFE, bool PRIO = false, class AfterRows = NoHook, class Pairs>
__device__ __forceinline__ void encode_block_avx_pk_t(const PkConsts &K, const uint2 (&rows)[8], const Pairs qp_of, uint32_t (&out)[64], AfterRows after_rows = AfterRows())
{
  MDCT_PHASE_PRIO(1);
  f32x2 col[4][8]; // col[j][r] = (B[r][kPairA[j]], B[r][kPairB[j]]) after the row pass
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const f32x2 a01 = {ubyte_to_float<0>(rows[r].x), ubyte_to_float<1>(rows[r].x)}; // :2143, raw 0..255
    const f32x2 a23 = {ubyte_to_float<2>(rows[r].x), ubyte_to_float<3>(rows[r].x)};
    const f32x2 a45 = {ubyte_to_float<0>(rows[r].y), ubyte_to_float<1>(rows[r].y)};
    const f32x2 a67 = {ubyte_to_float<2>(rows[r].y), ubyte_to_float<3>(rows[r].y)};
    dct8_h<K_AVX>(K, a01, a23, a45, a67, col[0][r], col[1][r], col[2][r], col[3][r]);
  }
  after_rows();
  MDCT_PHASE_PRIO(2);
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    dct8_v<K_AVX>(K, col[j]);
#pragma unroll
    for (int v = 0; v < 8; v++)
    {
      const f32x2 qp = qp_of[v * 4 + j];
      f32x2 m;
      m = col[j][v] * qp;
      if constexpr (SAFE)
      {
        out[v * 8 + kPairA[j]] = (uint32_t)clamp255((int32_t)((uint32_t)cvtps_epi32_exact(m.x) + 127u));
        out[v * 8 + kPairB[j]] = (uint32_t)clamp255((int32_t)((uint32_t)cvtps_epi32_exact(m.y) + 127u));
      }
      else
      {
        f32x2 t;
        m.x = __builtin_amdgcn_fmed3f(m.x, -128.0f, 127.0f);
        m.y = __builtin_amdgcn_fmed3f(m.y, -128.0f, 127.0f);
        t = m + K.nm.yy; // + (magic, magic)
        out[v * 8 + kPairA[j]] = __float_as_uint(t.x);
        out[v * 8 + kPairB[j]] = __float_as_uint(t.y);
      }
    }
  }
}

template <bool SAFE, bool PRIO = false, class AfterRows = NoHook>
__device__ __forceinline__ void encode_block_avx_pk(const PkConsts &K, const uint2 (&rows)[8], const QuantTable &qt, uint32_t (&out)[64], AfterRows after_rows = AfterRows())
{
  encode_block_avx_pk_t<SAFE, PRIO, AfterRows>(K, rows, reinterpret_cast<const f32x2 *>(qt.q), out, after_rows);
}

#ifndef MDCT_BLOCK_STAGE
#define MDCT_BLOCK_STAGE 1
#endif
// Scalar tiers: the reference divides every pixel by 255 (`px / 255.f`, simd_dct.cpp:222, :343).  For a byte x the correctly rounded
// quotient is, bit for bit,  fma(x, c, rn(x * c2))  with c = rn(1/255) and c2 = rn(1/255 - c): checked for all 256 values in exact
// rational arithmetic (tools/check_div255_forms.py, tests/test_div255_forms.py; plain x * c is wrong for 126 of them).  Two packed
// operations per pixel pair.  Rounds 2-4 kept the 256 quotients in an LDS table per workgroup: 64 random ds_read_b32 per lane, ~3
// bank-conflict cycles per LDS cycle (profiles/r04_pmc_raw.txt), a barrier, and LDS that capped the occupancy.
// B2..B5 on packed fp32.  The first pass runs "horizontally" over the 8 lines of the block (rows for the
// encq tiers, :347-358 / :1608-1636; columns for the stereo tiers, which transpose first, :961-1004 /
// :225-241), leaving the pairs (0,4)(2,6)(1,3)(5,7) of first-pass coefficients side by side; the second
// pass runs "vertically" on those four pairs.  P[j][m] then holds coefficients (kPairA[j], m) and
// (kPairB[j], m) in (first-pass index, second-pass index) terms, and both layouts store index
// first*8 + second: u*8+v for the encq tiers (:362, :1651), v*8+u for the stereo tiers.  `qt` holds the
// multipliers in that pair order, (m*4+j)*2 + {0,1} (mdct_api.hip).  Same bits as encode_block.
template <int PROFILE, int LAYOUT, bool SAFE>
__device__ __forceinline__ void encode_block_pk(const PkConsts &K, const uint2 (&rows)[8], const QuantTable &qt, uint32_t (&out)[64])
{
  constexpr int K1D = PROFILE == MDCT_PROFILE_REF_SSE ? K_SSE : K_TRUE;
  constexpr bool COLS_FIRST = LAYOUT == MDCT_LAYOUT_STEREO;
  // pixel (r, c) as the tier's float: byte c of row r
  auto px = [&](auto r_, auto c_) {
    constexpr int r = decltype(r_)::value, c = decltype(c_)::value;
    const uint32_t w = c < 4 ? rows[r].x : rows[r].y;
    return ubyte_to_float<(c & 3)>(w);
  };
  f32x2 P[4][8];
  auto line = [&](auto i_) {
    constexpr int i = decltype(i_)::value;
    auto at = [&](auto k_) { // k-th value of line i
      if constexpr (COLS_FIRST)
        return px(k_, i_);
      else
        return px(i_, k_);
    };
    using std::integral_constant;
    f32x2 a01 = {at(integral_constant<int, 0>{}), at(integral_constant<int, 1>{})};
    f32x2 a23 = {at(integral_constant<int, 2>{}), at(integral_constant<int, 3>{})};
    f32x2 a45 = {at(integral_constant<int, 4>{}), at(integral_constant<int, 5>{})};
    f32x2 a67 = {at(integral_constant<int, 6>{}), at(integral_constant<int, 7>{})};
    if constexpr (PROFILE == MDCT_PROFILE_REF_SSE)
    { // px * (1.0f / 255.0f), :949
      a01 = a01 * K.bias.xx; a23 = a23 * K.bias.xx; a45 = a45 * K.bias.xx; a67 = a67 * K.bias.xx;
    }
    else
    { // px / 255.f, :222 / :343: q = fma(x, c, rn(x * c2)) on both halves (see above)
      auto div255 = [&](f32x2 x) {
        f32x2 q;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
            : "=&v"(q)
            : "v"(x), "s"(K.div));
        return q;
      };
      a01 = div255(a01); a23 = div255(a23); a45 = div255(a45); a67 = div255(a67);
    }
    dct8_h<K1D>(K, a01, a23, a45, a67, P[0][i], P[1][i], P[2][i], P[3][i]);
  };
  line(std::integral_constant<int, 0>{}); line(std::integral_constant<int, 1>{}); line(std::integral_constant<int, 2>{}); line(std::integral_constant<int, 3>{});
  line(std::integral_constant<int, 4>{}); line(std::integral_constant<int, 5>{}); line(std::integral_constant<int, 6>{}); line(std::integral_constant<int, 7>{});
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    dct8_v<K1D>(K, P[j]);
#pragma unroll
    for (int m = 0; m < 8; m++)
    {
      const f32x2 qp = reinterpret_cast<const f32x2 *>(qt.q)[m * 4 + j];
      const int sa = kPairA[j] * 8 + m, sb = kPairB[j] * 8 + m;
      f32x2 v;
      v = P[j][m] * qp;
      if constexpr (PROFILE == MDCT_PROFILE_REF_SSE)
      { // B2/B3 :1020  clamp(rne(f*q + 127.0f), 0, 255)
        v = v + K.bias.yy;
        if constexpr (SAFE)
        {
          out[sa] = (uint32_t)clamp255(cvtps_epi32_exact(v.x));
          out[sb] = (uint32_t)clamp255(cvtps_epi32_exact(v.y));
        }
        else
        {
          // (v_cvt_pk_u8_f32 -- round to nearest even, saturate to [0, 255], NaN -> 0: probed on gfx950, tools/probe_cvt_pk_u8.hip -- would do
          // clamp + round in one instruction per value: 739 -> 706 vector instructions per wave for stereo/SSE, and no faster: 32.1-32.2 us
          // against 31.8 in the bench, encq/SSE 23.9-24.1 against 24.1.  The convert is not a full-rate instruction; the magic add stays.)
          f32x2 t;
          v.x = __builtin_amdgcn_fmed3f(v.x, 0.0f, 255.0f);
          v.y = __builtin_amdgcn_fmed3f(v.y, 0.0f, 255.0f);
          t = v + K.nm.yy;
          out[sa] = __float_as_uint(t.x);
          out[sb] = __float_as_uint(t.y);
        }
      }
      else
      { // B4/B5 :245, :362  (uint8_t)roundf(_clamp(f*qs + 127/255, 0, 1) * 255) in three operations per coefficient pair:
        //   c = clamp(v + 127/255, 0, 1)   one v_pk_add_f32 with the clamp modifier; NaN -> 0 (DX10_CLAMP), which is what
        //                                  _clamp (:50-54, NaN passes) followed by x86's (uint8_t) cast of NaN yields
        //   s = fma(c, 255, pred(0.5))     one rounding; pred(0.5) = 0.5 - 2^-25
        //   byte = trunc(s)                v_cvt_u32_f32
        // roundf(rn(c * 255)) == trunc(rn(c * 255 + pred(0.5))) for EVERY float c in [0, 1]: checked exhaustively over all 2^30 + 1
        // of them (tools/check_roundf_forms.py, profiles/r04_roundf_forms_exhaustive.log; with 0.5 instead of pred(0.5), fused or
        // not, exactly one c fails: 0x3B008080, whose c * 255 rounds to pred(0.5)).  The fused multiply-add is not the reference's
        // arithmetic, it is a shorter way to the same byte -- 5 vector instructions per pair where magic-number rounding plus the
        // tie fix-up took 11 (944 -> 752 per wave).
        f32x2 s2;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] clamp\n\t"
            "v_pk_fma_f32 %0, %0, %3, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
            : "=&v"(s2)
            : "v"(v), "s"(K.nm), "s"(K.bias));
        out[sa] = (uint32_t)s2.x;
        out[sb] = (uint32_t)s2.y;
      }
    }
  }
}

// four low bytes -> one dword, upper bits of the inputs ignored (3 x v_perm_b32)
__device__ __forceinline__ uint32_t pack4_lo8(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u);
  const uint32_t cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
  return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}

constexpr int kWG = 256;           // 4 waves
#ifndef MDCT_XCD_SWIZZLE
#define MDCT_XCD_SWIZZLE 0
#endif
// Workgroup index of the launch.  Hardware hands consecutive workgroups to the 8 XCDs round-robin, so with the plain
// linear order the 8 XCDs stream through neighbouring tiles (the same DRAM pages) at any moment; MDCT_XCD_SWIZZLE=1
// gives every XCD one contiguous eighth of the plane instead.  Measured (profiles/r02_xcd_mapping.log): the linear order
// is faster for every single plane (8192^2: copy 42.9 vs 51.5 us, round trip 46.8 vs 50.7, q32 31.0 vs 31.7); only the
// 17 GB batch of config 4 gains 2 % from the swizzle.  These kernels have no reuse for an XCD's L2 to keep: linear it is.
__device__ __forceinline__ uint32_t wg_index()
{
#if MDCT_XCD_SWIZZLE
  const uint32_t w = blockIdx.x, per = gridDim.x >> 3;
  return w < per * 8 ? (w & 7) * per + (w >> 3) : w;
#else
  return blockIdx.x;
#endif
}
// ---------------------------------------------------------------------------------------
// Wave-uniform addressing.  The u8 kernels are bound by VALU issue at the clock the chip holds under their load
// (profiles/r03_q32_timeline.md: every SIMD retires one wave per ~1.5 us, two waves in their compute phase saturate
// it, loads are hidden), so every VALU instruction that is not the reference's arithmetic costs time.  Per-lane
// 64-bit address arithmetic (an integer division, v_mad_u64_u32 per row) was ~100 of ~840 instructions per wave.
// Where a wave's 64 blocks lie in one block row (sizeX % 512 == 0) the launch is a 2-D grid -- blockIdx.y = block
// row, blockIdx.x = position in the row -- every base address is computed once per wave on the scalar unit, and a
// lane contributes only a 32-bit offset: global_load / global_store with an SGPR base.  sgpr_ptr() pins a
// (wave-uniform) pointer in SGPRs and keeps it in the global address space.
// ---------------------------------------------------------------------------------------
typedef const uint8_t __attribute__((address_space(1))) *gcptr_t;
typedef uint8_t __attribute__((address_space(1))) *gptr_t;
__device__ __forceinline__ gptr_t sgpr_ptr(const void *p)
{
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gptr_t)(((uint64_t)hi << 32) | lo); // (uint32_t halves: a sign-extended low half would corrupt the high one)
}
typedef unsigned int u32x2_unaligned_g __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned int u32x4_g __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_unaligned_g __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint2 load8_g(gcptr_t p)
{
  const u32x2_unaligned_g v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_unaligned_g __attribute__((address_space(1))) *>(p));
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void store16_g(gptr_t p, u32x4_g v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_g __attribute__((address_space(1))) *>(p)); }
__device__ __forceinline__ void store16_unaligned_g(gptr_t p, u32x4_unaligned_g v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_unaligned_g __attribute__((address_space(1))) *>(p)); }
// the lane's block as eight 8-byte rows: `base` = row 0 of the wave's first block (wave-uniform), lane offset in bytes
__device__ __forceinline__ void load_block_rows_g(const uint8_t *base, size_t pitch, uint32_t lane_off, uint2 (&rows)[8])
{
#pragma unroll
  for (int r = 0; r < 8; r++)
    rows[r] = load8_g(sgpr_ptr(base + (size_t)r * pitch) + lane_off);
}

constexpr int kQ32RowStride = 72;  // 64 lanes + 8 pad: keeps rows 8-byte aligned for ds_read_b64
constexpr int kStereoRowStride = kWG + 16; // 256 blocks + pad, rows stay 16-byte aligned for ds_read_b128

// ---------------------------------------------------------------------------------------
// B1 (simd_dct.cpp:2064-2262): the reference's primary product, its own kernel.
// Output of 8 consecutive blocks (one reference "group") is 512 contiguous bytes
// [coef*8 + blk] (:2227-2230) and group G of the plane sits at G*512, so a wave's 64 blocks
// cover 4096 contiguous output bytes.  They are staged through wave-private LDS as rows
// [coef][lane] so that every lane then stores 16 contiguous bytes.  No workgroup barrier:
// only the wave that wrote a row reads it.  Everything that is live across the transform is
// wave-uniform (SGPRs), which is what lets the fast form fit 6 waves/SIMD (80 VGPRs).
// ---------------------------------------------------------------------------------------
#ifndef MDCT_Q32_MINW
#define MDCT_Q32_MINW 6
#endif
// GENERAL = false: the launch is a whole number of waves and the output strips are tight (the host
// checks), so nothing is guarded; GENERAL = true adds the partial last wave and pitched strips.
template <bool SAFE, bool GENERAL>
__global__ __launch_bounds__(kWG, (SAFE || GENERAL) ? 1 : MDCT_Q32_MINW) void k_q32_avx(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWG / 64][64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave_t0 = wg_index() * kWG + wave * 64; // first block of this wave within the launch
  if (wave_t0 >= a.nblocks) // whole waves past the end of the launch (the grid is in workgroups of 4 waves)
    return;
  uint32_t wave_blocks = 64, t = wave_t0 + lane;
  if constexpr (GENERAL)
  {
    wave_blocks = min(64u, a.nblocks - wave_t0);
    t = wave_t0 + min(lane, wave_blocks - 1); // lanes past the end redo the last block; their bytes are never read back
  }
  uint32_t q[64];
  {
    const uint32_t row = t / a.bpr;
    const uint32_t bx = t - row * a.bpr;
    uint2 rows[8];
    load_block_rows(a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8, a.pitch, rows);
    encode_block_avx_pk<SAFE>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  }
  uint8_t *wl = lds[wave];
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + wave_t0) * 64; // == first group * 512
  const uint32_t c2 = (lane & 31) * 2;
  auto staged = [&](uint32_t g) { // coefficients c2, c2+1 of group g: 16 contiguous output bytes
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
    u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
    if constexpr (!SAFE)
      v = ~v; // the fast quantiser staged complemented bytes (encode_block_avx_pk)
    return v;
  };
  if constexpr (!GENERAL)
  { // four unguarded 1 KiB stores
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const uint32_t g = 2 * k + (lane >> 5);
      __builtin_nontemporal_store(staged(g), reinterpret_cast<u32x4_t *>(outw + g * 512 + c2 * 8));
    }
    return;
  }
  // partial last wave and / or pitched strips (mdct_fwd_quant_u8_pitched)
  const uint32_t wave_groups = wave_blocks >> 3; // nblocks % 8 == 0
#pragma unroll 1
  for (int k = 0; k < 4; k++)
  {
    const uint32_t g = 2 * k + (lane >> 5);
    if (g < wave_groups)
    {
      uint8_t *dst = outw + g * 512 + c2 * 8;
      if (!a.out_tight)
      { // group G of the launch -> (block row, group x)
        const uint32_t G = (wave_t0 >> 3) + g, gpr = a.bpr >> 3;
        const uint32_t grow = G / gpr;
        dst = a.to + (size_t)(a.by0 + grow) * a.out_strip + (size_t)(G - grow * gpr) * 512 + c2 * 8;
      }
      __builtin_nontemporal_store(staged(g), reinterpret_cast<u32x4_t *>(dst));
    }
  }
}

// The same product for launches whose waves each lie in one block row (sizeX % 512 == 0, tight output): one workgroup =
// one wave = one 64-block tile, 2-D grid (x = tile in the row, y = block row), wave-uniform addressing (above).
// 8192^2: 28.6-28.8 us against 30.4-30.7 for k_q32_avx<false,false> (profiles/r03_exp_q32_addressing.log).
__global__ __launch_bounds__(64, MDCT_Q32_MINW) void k_q32_tile(U8Args a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x, row = blockIdx.y;
  uint32_t q[64];
  {
    uint2 rows[8];
    load_block_rows_g(a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)tile * 512, a.pitch, lane * 8, rows);
    encode_block_avx_pk<false>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  }
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the wave's 4 KiB of output (:2227-2230): store k, lane l -> coefficients c2 = 2 (l & 31), c2 + 1 of group 2k + (l >> 5),
  // at group * 512 + c2 * 8 == k * 1024 + l * 16
  const gptr_t outw = sgpr_ptr(a.to + ((size_t)(a.by0 + row) * a.bpr + (size_t)tile * 64) * 64);
  const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
    const u32x4_g v = ~u32x4_g{lo.x, lo.y, hi.x, hi.y}; // the fast quantiser staged complemented bytes (encode_block_avx_pk)
    store16_g(outw + k * 1024 + lane * 16, v);
  }
}

// The other tiers / layouts.  Occupancy steering, measured optimum of {default, 3..6} waves/SIMD
// (profiles/r02_u8_tiers_waves_per_eu.log): the SSE encq tier 4 (30.8 us; 31.2 at 3, 31.0 at 5, 32.9 unsteered, 37.6 at 6 --
// with the pair-swapped 16-byte stores; its scattered dword stores before that took 38.8 at 3), the scalar stereo tier 4
// (46.1 vs 49.5); the others are best left to the compiler (stereo/SSE 31.4 us unsteered, 34-36 steered).
#ifdef MDCT_U8_WAVES
#define MDCT_U8_ATTR __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(MDCT_U8_WAVES, MDCT_U8_WAVES)))
#else
// (round 4, tiled kernels, profiles/r04_exp_scalar_waves.log: stereo/SSE 31.9-32.0 us steered to 3 against 33.9 unsteered, 33.4 at 5, 47.6 at 6;
// stereo/scalar 35.5 at 4 or 5, 37-38 at 3, 46 at 6; encq/scalar 31.4-31.8 unsteered, 32.3-33.6 steered)
// (round 5, the scalar tiers without their LDS quotient table, profiles/r05_exp_u8_tier_waves.log: encq/scalar 32.6 us unsteered (74 VGPRs: 6 waves),
// 31.0 at 3, 30.6 at 4, 31.2 at 5; stereo/scalar 30.6-30.9 at 4, 31.7 at 3, 31.3 at 5)
constexpr int u8_waves_lo(int profile, int layout) { return layout == MDCT_LAYOUT_STEREO ? (profile == MDCT_PROFILE_REF_SCALAR ? 4 : 3) : 4; }
constexpr int u8_waves_hi(int profile, int layout) { return u8_waves_lo(profile, layout); }
#define MDCT_U8_ATTR __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(u8_waves_lo(PROFILE, LAYOUT), u8_waves_hi(PROFILE, LAYOUT))))
#endif
// TILED: 2-D grid for launches whose workgroups each lie in one row of blocks (sizeX % 2048 == 0): blockIdx.y = row of
// the launch ((block row, eye) for STEREO), blockIdx.x = 256-block tile of that row; no division, wave-uniform base
// addresses (see sgpr_ptr).  Otherwise the linear form: any shape, partial last workgroup.
template <int PROFILE, int LAYOUT, bool SAFE, bool TILED>
__global__ MDCT_U8_ATTR void k_fwd_quant_u8(U8Args a)
{
  // linear block index within the launch, block coordinates.  STEREO enumerates (block row, eye, block x): simd_dct.cpp:1089-1099.
  uint32_t t, row, bx;
  if constexpr (TILED)
  {
    row = blockIdx.y;
    bx = blockIdx.x * kWG + threadIdx.x;
    t = row * a.bpr + bx;
  }
  else
  {
    t = wg_index() * kWG + threadIdx.x;
    row = t / a.bpr;
    bx = t - row * a.bpr;
  }
  const bool valid = TILED || t < a.nblocks;
  uint32_t by, eye = 0;
  if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
  {
    by = a.by0 + (row >> 1);
    eye = row & 1;
  }
  else
    by = a.by0 + row;

  uint32_t q[64];
  if (valid)
  {
    uint2 rows[8];
    if constexpr (TILED)
    { // base of the wave's first block: wave-uniform; the lane adds 8 bytes per block
      const uint32_t wave_bx0 = blockIdx.x * kWG + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
      const uint8_t *base = a.from + (size_t)by * 8 * a.pitch + (size_t)wave_bx0 * 8;
      if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
        base += (size_t)eye * a.eye_offset;
      load_block_rows_g(base, a.pitch, (threadIdx.x & 63) * 8, rows);
    }
    else
    {
      const uint8_t *src = a.from + (size_t)by * 8 * a.pitch + (size_t)bx * 8;
      if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
        src += (size_t)eye * a.eye_offset;
      load_block_rows(src, a.pitch, rows);
    }
    encode_block_pk<PROFILE, LAYOUT, SAFE>(reinterpret_cast<const PkConsts &>(a.pk), rows, a.qt, q);
  }

  if constexpr (LAYOUT == MDCT_LAYOUT_STEREO)
  {
    // 64 coefficient planes; block t of the launch lands at byte (by0*2*bpr + t) of every plane
    // (:1061-1099), so a full workgroup owns 256 consecutive bytes per plane.  Stage them in LDS
    // as [coef][block] and store 16 B per lane (4 wide stores instead of 64 byte stores per lane).
    // (Two barrier-free forms were measured and lost: a wave-private reorder with 64-byte pieces, 37 vs 32 us,
    // profiles/r03_exp_stereo_wave_private_reorder.log; one wave walking four tiles so that it owns the 256-byte pieces itself,
    // 17 KiB of LDS per wave = 9 waves per CU, 49.3 vs 32.0 us (SSE) and 37.7 vs 34.0 (scalar), profiles/r04_exp_stereo_wave4.log.
    // Workgroups of 128 blocks -- a barrier between two waves instead of four, 128-byte pieces per plane: 32.1-32.6 vs 31.4-31.6 us (SSE),
    // 32.6-33.2 vs 31.8-32.3 (scalar), profiles/r05_exp_stereo_wg128.log.
    // The tier runs 739 vector instructions per wave against q32's 663 -- the reference's x(1/255) and +127.0f are operations of
    // their own -- and 30.3 us against 27.3: it sits at the same vector-issue bound as q32, the barrier is not what it waits for.)
    __shared__ __attribute__((aligned(16))) uint8_t slds[64 * kStereoRowStride];
    const uint32_t wg_t0 = TILED ? blockIdx.y * a.bpr + blockIdx.x * kWG : wg_index() * kWG;
    const bool full_wg = TILED || wg_t0 + kWG <= a.nblocks; // workgroup-uniform
    if (full_wg)
    {
#pragma unroll
      for (int c = 0; c < 64; c++)
        slds[c * kStereoRowStride + threadIdx.x] = (uint8_t)q[c];
      wg_sync();
      uint8_t *out0 = a.to + (size_t)a.by0 * 2 * a.bpr + wg_t0;
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        const uint32_t chunk = threadIdx.x + kWG * k; // 64 planes x 16 chunks of 16 B
        const uint32_t c = chunk >> 4, part = chunk & 15;
        const uint4 v = *reinterpret_cast<const uint4 *>(slds + c * kStereoRowStride + part * 16);
        typedef unsigned int u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
        const u32x4_unaligned w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_unaligned *>(out0 + a.plane_stride * c + part * 16));
      }
    }
    else if (valid)
    {
      const size_t pos = ((size_t)by * 2 + eye) * a.bpr + bx;
#pragma unroll
      for (int i = 0; i < 64; i++)
        a.to[a.plane_stride * i + pos] = (uint8_t)q[i];
    }
  }
  else if constexpr (LAYOUT == MDCT_LAYOUT_BLOCK)
  {
    // Left to itself a lane stores its block's 64 bytes as 4 x 16: every store instruction of the wave then touches a quarter of each
    // of 64 cache lines.  MDCT_BLOCK_STAGE: the pieces go through wave-private LDS (block stride 80 B: conflict-free b128 both ways)
    // and leave as four contiguous 1 KiB stores, like the q32 layout's.
    constexpr int kBlockStride = 80;
    __shared__ __attribute__((aligned(16))) uint8_t blds[(TILED && MDCT_BLOCK_STAGE) ? (kWG / 64) * 64 * kBlockStride : 16];
    if (valid)
    {
      uint8_t *dst = a.to + (size_t)by * a.out_strip + (size_t)bx * 64;
      const gptr_t dst_g = sgpr_ptr(TILED ? a.to + (size_t)by * a.out_strip + (size_t)(blockIdx.x * kWG) * 64 : a.to);
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
          w[j] = pack4_lo8(q[k * 16 + j * 4], q[k * 16 + j * 4 + 1], q[k * 16 + j * 4 + 2], q[k * 16 + j * 4 + 3]);
        if constexpr (TILED && MDCT_BLOCK_STAGE)
          *reinterpret_cast<uint4 *>(blds + (threadIdx.x >> 6) * (64 * kBlockStride) + (threadIdx.x & 63) * kBlockStride + k * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        else if constexpr (TILED)
          *reinterpret_cast<u32x4_g __attribute__((address_space(1))) *>(dst_g + threadIdx.x * 64 + k * 16) = u32x4_g{w[0], w[1], w[2], w[3]};
        else
          *reinterpret_cast<uint4 *>(dst + k * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      if constexpr (TILED && MDCT_BLOCK_STAGE)
      { // the wave's 64 blocks are 4 KiB of contiguous output: read the staged pieces back so that every store instruction covers 1 KiB
        // (lane l of store k takes bytes [1024 k + 16 l, + 16) = piece l & 3 of the wave's block 16 k + (l >> 2)); wave-private, no barrier
        const uint32_t lane = threadIdx.x & 63;
        const uint8_t *wl = blds + (threadIdx.x >> 6) * (64 * kBlockStride);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
          const uint4 v = *reinterpret_cast<const uint4 *>(wl + (16 * k + (lane >> 2)) * kBlockStride + (lane & 3) * 16);
          store16_g(dst_g + (threadIdx.x & ~63u) * 64 + k * 1024 + lane * 16, u32x4_g{v.x, v.y, v.z, v.w});
        }
      }
    }
  }
  else
  { // MDCT_LAYOUT_BLOCK_SSE (:1662-1676): of every 128-byte pair of blocks only the first 64 bytes are written -- for each
    // coefficient row i8 the columns {0,1,4,5} of the even block, then of the odd block.  The two lanes of a pair swap
    // halves (DPP) so that each stores 32 contiguous bytes as 2 x 16 B instead of eight scattered dwords.
    if (valid)
    {
      const uint32_t ab = bx & 1; // sizeX % 16 == 0 for this tier: a pair is two neighbouring lanes of one block row
      uint32_t L[8];
#pragma unroll
      for (int i8 = 0; i8 < 8; i8++)
        L[i8] = pack4_lo8(q[i8 * 8 + 0], q[i8 * 8 + 1], q[i8 * 8 + 4], q[i8 * 8 + 5]);
      // (Splitting the pair's 64 bytes the other way -- rows {0,1,4,5} / {2,3,6,7} per lane, so that every store instruction writes 32 contiguous bytes
      // per pair instead of 16 + 16 -- is much slower: 35.1 vs 24.1 us, same bytes.  A lane's two stores to ADJACENT addresses are what the write path
      // combines; keep them adjacent.  Staging the wave's 32 x 64 bytes through LDS so that a store instruction covers whole 64-byte halves: 24.8-25.1 us,
      // no better either -- unlike the scalar tier's BLOCK layout this one is not held back by its stores.)
      uint32_t w[8]; // this lane's 32 bytes: rows 0..3 (even block's lane) or 4..7 (odd block's lane), both blocks' dwords interleaved
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const uint32_t keep = ab ? L[4 + j] : L[j], give = ab ? L[j] : L[4 + j];
        const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)give, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
        w[2 * j] = ab ? got : keep;
        w[2 * j + 1] = ab ? keep : got;
      }
      uint8_t *pair = a.to + (size_t)by * 8 * a.sizeX + (size_t)(bx >> 1) * 128;
      typedef unsigned int u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
      const u32x4_unaligned v0 = {w[0], w[1], w[2], w[3]}, v1 = {w[4], w[5], w[6], w[7]};
      if constexpr (TILED)
      { // pair p of the workgroup's 128 pairs at 128 p; this lane's 32 bytes at + 32 ab: thread i -> 64 (i >> 1) * 2 + 32 (i & 1) = 64 i - 32 (i & 1) ... spelled out:
        const gptr_t wg_out = sgpr_ptr(a.to + (size_t)by * 8 * a.sizeX + (size_t)(blockIdx.x * kWG) * 64);
        const uint32_t off = (threadIdx.x >> 1) * 128 + ab * 32;
        *reinterpret_cast<u32x4_unaligned __attribute__((address_space(1))) *>(wg_out + off) = v0;
        *reinterpret_cast<u32x4_unaligned __attribute__((address_space(1))) *>(wg_out + off + 16) = v1;
      }
      else
      {
        *reinterpret_cast<u32x4_unaligned *>(pair + ab * 32) = v0;
        *reinterpret_cast<u32x4_unaligned *>(pair + ab * 32 + 16) = v1;
      }
      const bool spill = (by == a.by_last) && ((bx | 1u) == a.bpr - 1) && a.spill_ok;
      if (spill)
      { // the surviving spill of the plane's last pair (:1676): the other columns, one pair further
#pragma unroll
        for (int i8 = 0; i8 < 8; i8++)
          *reinterpret_cast<uint32_t *>(pair + ab * 4 + 128 + i8 * 8) = pack4_lo8(q[i8 * 8 + 2], q[i8 * 8 + 3], q[i8 * 8 + 6], q[i8 * 8 + 7]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// int16 / float32 engine-own kernels: plane layout in and out, 16 B per lane per row, so
// every global access is a fully coalesced 1 KiB wave transaction and no LDS is needed.
// The planes are streamed exactly once: loads and stores are non-temporal (measured
// +6..10 % over plain on a read-N/write-N stream, tools/membench).
// ---------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 ld_stream16(const void *p)
{
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void st_stream16(void *p, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
  const u32x4 v = {x, y, z, w};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}

// Where a lane finds the eight 16-byte rows of its block.  Linear: one pointer per lane (any shape).  Tiled: the wave's 64
// blocks lie in one block row; `src` / `dst` are the wave-uniform addresses of row 0 of its first block (kept in SGPRs,
// sgpr_ptr) and the lane adds a 32-bit offset -- no per-lane 64-bit address arithmetic, 2 address registers instead of 32.
struct RowsLinear
{
  const int16_t *src;
  int16_t *dst;
  size_t pitch_in, pitch_out; // elements
  __device__ __forceinline__ uint4 ld(int r) const { return ld_stream16(src + (size_t)r * pitch_in); }
  __device__ __forceinline__ void st(int r, uint32_t x, uint32_t y, uint32_t z, uint32_t w) const { st_stream16(dst + (size_t)r * pitch_out, x, y, z, w); }
};
struct RowsTiled
{
  const int16_t *src;
  int16_t *dst;
  size_t pitch_in, pitch_out; // elements
  uint32_t lane_off;          // bytes from src
  uint32_t lane_off_out;      // bytes from dst (== lane_off except for the upper lanes of a paired plane's straddling tile when the pitches differ)
  __device__ __forceinline__ uint4 ld(int r) const
  {
    const u32x4_g v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_g __attribute__((address_space(1))) *>(sgpr_ptr(src + (size_t)r * pitch_in) + lane_off));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  __device__ __forceinline__ void st(int r, uint32_t x, uint32_t y, uint32_t z, uint32_t w) const { store16_g(sgpr_ptr(dst + (size_t)r * pitch_out) + lane_off_out, u32x4_g{x, y, z, w}); }
};

__device__ __forceinline__ void unpack_i16x8(const uint4 v, float (&row)[8])
{
  row[0] = (float)(int16_t)(v.x & 0xFFFF); row[1] = (float)(int16_t)(v.x >> 16);
  row[2] = (float)(int16_t)(v.y & 0xFFFF); row[3] = (float)(int16_t)(v.y >> 16);
  row[4] = (float)(int16_t)(v.z & 0xFFFF); row[5] = (float)(int16_t)(v.z >> 16);
  row[6] = (float)(int16_t)(v.w & 0xFFFF); row[7] = (float)(int16_t)(v.w >> 16);
}

// sat_i16(rne(v * 2^-SHIFT)) without a convert: clamp in float (bounds are multiples of
// 2^SHIFT, so clamping commutes with the rounding), then add 1.5 * 2^(23+SHIFT).  The sum's
// ulp is 2^SHIFT, the add rounds to nearest-even in exactly those units, and because the
// magic constant's low 16 mantissa bits are zero and its integer part is even, the low 16
// bits of the result ARE the two's-complement int16.  SHIFT = 6 is the fused round trip's
// 1/64.  Returns the raw bits; callers keep the low half.
template <int SHIFT>
__device__ __forceinline__ uint32_t rne_i16_bits(const DctConsts &C, float v)
{
  constexpr float scale = (float)(1 << SHIFT);
  const float m = __builtin_amdgcn_fmed3f(v, -32768.0f * scale, 32767.0f * scale);
  return __float_as_uint(m + (SHIFT == 0 ? C.magic23 : C.magic29));
}

// The engine-own QUANTISER (round 6): c = sat_i16(rne(y * qf)) with ONE rounding -- fma(y, qf, 1.5 * 2^23) is the exact product rounded to
// nearest-even in units of 1 (for |y qf| < 2^22; beyond, the clamp decides), where rounds 1-5 rounded the product to a float first and that
// float to an integer.  One operation fewer per coefficient pair in the packed forms, and the quantised value is the correctly rounded one.
// The clamp works on the biased value: its bounds 1.5 * 2^23 - 32768 and + 32767 are exact floats.  Same operations as the CPU checker
// (its quant_i16).  Low 16 bits of the result = the two's-complement int16.
constexpr float kMagic23 = 12582912.0f, kQLo = kMagic23 - 32768.0f, kQHi = kMagic23 + 32767.0f;
__device__ __forceinline__ uint32_t quant_i16_bits(const DctConsts &C, float y, float qf)
{
  return __float_as_uint(__builtin_amdgcn_fmed3f(__builtin_fmaf(y, qf, C.magic23), kQLo, kQHi));
}
// the same, result as a float integer (for quantise -> dequantise in registers)
__device__ __forceinline__ float quant_i16_float(const DctConsts &C, float y, float qf)
{
  return __builtin_amdgcn_fmed3f(__builtin_fmaf(y, qf, C.magic23), kQLo, kQHi) - C.magic23;
}

__device__ __forceinline__ uint32_t pack_lo16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); }

// a row of raw forward outputs quantised on the way out (quant_i16_bits)
__device__ __forceinline__ void store_q_i16x8(const DctConsts &C, int16_t *dst, const float (&row)[8], const float *qf)
{
  uint32_t t[8];
#pragma unroll
  for (int c = 0; c < 8; c++)
    t[c] = quant_i16_bits(C, row[c], qf[c]);
  st_stream16(dst, pack_lo16(t[0], t[1]), pack_lo16(t[2], t[3]), pack_lo16(t[4], t[5]), pack_lo16(t[6], t[7]));
}

template <int MODE, bool HAS_LUT, class Rows, class Tables>
__device__ __forceinline__ void i16_block(const DctConsts &C, const Rows rows, const Tables &tb)
{
  float b[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    unpack_i16x8(rows.ld(r), b[r]);

  if constexpr (MODE == MODE_INV)
  {
#pragma unroll
    for (int i = 0; i < 64; i++)
      b[i >> 3][i & 7] = b[i >> 3][i & 7] * tb.dq[i];
  }
  else
    raw_fwd(C, b);

  if constexpr (MODE != MODE_FWD)
  {
    if constexpr (MODE == MODE_ROUNDTRIP && HAS_LUT)
    {
#pragma unroll
      for (int i = 0; i < 64; i++)
        b[i >> 3][i & 7] = quant_i16_float(C, b[i >> 3][i & 7], tb.qf[i]) * tb.dq[i];
    }
    raw_inv(C, b);
  }

  // fused round trip without a table: forward-scale * inverse-scale == 1/64 exactly
  constexpr int SHIFT = (MODE == MODE_ROUNDTRIP && !HAS_LUT) ? 6 : 0;
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    uint32_t t[8];
#pragma unroll
    for (int c = 0; c < 8; c++)
      t[c] = MODE == MODE_FWD ? quant_i16_bits(C, b[r][c], tb.qf[r * 8 + c]) : rne_i16_bits<SHIFT>(C, b[r][c]); // forward: the quantiser is the output stage
    rows.st(r, pack_lo16(t[0], t[1]), pack_lo16(t[2], t[3]), pack_lo16(t[4], t[5]), pack_lo16(t[6], t[7]));
  }
}

// The AAN butterflies on packed fp32 (fused round trip; profiles/r02_exp_i16_packed.log).  Same idea as the u8
// tiers above: rows "horizontally" on 4 register pairs with op_sel / neg modifiers, columns "vertically" on pairs of
// columns.  AAN's flow graph leaves 6 (forward) / 12 (inverse) operations per horizontal transform without a
// partner; they stay scalar.  Every packed or scalar operation is the individually rounded IEEE operation of
// aan_fwd8 / aan_inv8, so the results are bit-identical (and are tested as such against the CPU checker).
// The first ten floats of DctConsts are laid out as the pairs these functions consume.
struct AanPk
{
  f32x2 c707_382;   // (cos(pi/4), cos(3pi/8))
  f32x2 c541_1306;  // (cos(pi/8)-cos(3pi/8), cos(pi/8)+cos(3pi/8))
  f32x2 c1414_1847; // (sqrt 2, 2cos(pi/8))
  f32x2 c1082_2613;
  f32x2 magic;      // (1.5*2^23, 1.5*2^29)
};
static_assert(offsetof(DctConsts, c707) == 0 && offsetof(DctConsts, c382) == 4 && offsetof(DctConsts, c541) == 8 && offsetof(DctConsts, c1306) == 12 &&
                  offsetof(DctConsts, c1414) == 16 && offsetof(DctConsts, c1847) == 20 && offsetof(DctConsts, c1082) == 24 && offsetof(DctConsts, c2613) == 28 &&
                  offsetof(DctConsts, magic23) == 32 && offsetof(DctConsts, magic29) == 36,
              "AanPk views the head of DctConsts");

#define MDCT_LOLO "op_sel:[0,0] op_sel_hi:[0,0]"
#define MDCT_HIHI "op_sel:[1,1] op_sel_hi:[1,1]"
#define MDCT_LOHI "op_sel:[0,1] op_sel_hi:[0,1]" // src0.lo with src1.hi, for both halves
#define MDCT_HILO "op_sel:[1,0] op_sel_hi:[1,0]"
#define MDCT_SUMDIFF "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" // (a.lo + a.hi, a.lo - a.hi) when both sources are a
#define MDCT_XSEL " op_sel:[0,1] op_sel_hi:[1,0]"              // lo = a.lo (+) b.hi, hi = a.hi (+) b.lo (MDCT_X inside a block)

// forward, one line in natural pairs (p0,p1)(p2,p3)(p4,p5)(p6,p7) -> (y0,y4) (y2,y6) (y5,y3) (y1,y7)
__device__ __forceinline__ void aan_fwd_h_stmt(const AanPk &K, f32x2 a01, f32x2 a23, f32x2 a45, f32x2 a67, f32x2 &o04, f32x2 &o26, f32x2 &o53, f32x2 &o17)
{
  f32x2 t01, t23, t76, t54, e01, e32, w, o, z24, z1113;
  MDCT_PKA(t01, a01, a67, MDCT_X);                   // (t0, t1) = (p0+p7, p1+p6)
  MDCT_PKA(t23, a23, a45, MDCT_X);                   // (t2, t3) = (p2+p5, p3+p4)
  MDCT_PKA(t76, a01, a67, MDCT_X " " MDCT_NEG_B);    // (t7, t6) = (p0-p7, p1-p6)
  MDCT_PKA(t54, a23, a45, MDCT_X " " MDCT_NEG_B);    // (t5, t4) = (p2-p5, p3-p4)
  MDCT_PKA(e01, t01, t23, MDCT_X);                   // (e10, e11) = (t0+t3, t1+t2)
  MDCT_PKA(e32, t01, t23, MDCT_X " " MDCT_NEG_B);    // (e13, e12) = (t0-t3, t1-t2)
  MDCT_PKA(o04, e01, e01, MDCT_SUMDIFF);                  // (e10+e11, e10-e11)
  w.x = e32.y + e32.x;                               // s1 = e12 + e13
  w.y = t54.x + t76.y;                               // o11 = t5 + t6
  o.x = t54.y + t54.x;                               // o10 = t4 + t5
  o.y = t76.y + t76.x;                               // o12 = t6 + t7
  f32x2 z5;
  z5.x = (o.x - o.y) * K.c707_382.y;                 // z5 = (o10 - o12) * c382
  z5.y = z5.x;
  MDCT_PKF(z24, o, K.c541_1306, z5, "op_sel:[0,0,0] op_sel_hi:[1,1,0]");                   // (z2, z4) = (c541 o10 + z5, c1306 o12 + z5)
  MDCT_PKF(z1113, w, K.c707_382, t76, "op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[1,0,0]");  // (z11, z13) = (o11 c707 + t7, -o11 c707 + t7)
  MDCT_PKF(o26, w, K.c707_382, e32, "op_sel:[0,0,0] op_sel_hi:[0,0,0] neg_hi:[1,0,0]");    // (s1 c707 + e13, -s1 c707 + e13)
  MDCT_PKA(o53, z1113, z24, MDCT_HILO " neg_hi:[0,1]"); // (z13+z2, z13-z2)
  MDCT_PKA(o17, z1113, z24, MDCT_LOHI " neg_hi:[0,1]"); // (z11+z4, z11-z4)
}

// forward down a pair of columns, in place (direct image of aan_fwd8)
__device__ __forceinline__ void aan_fwd_v_stmt(const AanPk &K, f32x2 (&p)[8])
{
  f32x2 t0, t7, t1, t6, t2, t5, t3, t4, e10, e13, e11, e12, s1, o10, o11, o12, z5, z2, z4, z11, z13;
  MDCT_PKA(t0, p[0], p[7], ""); MDCT_PKA(t7, p[0], p[7], MDCT_NEG_B); MDCT_PKA(t1, p[1], p[6], ""); MDCT_PKA(t6, p[1], p[6], MDCT_NEG_B);
  MDCT_PKA(t2, p[2], p[5], ""); MDCT_PKA(t5, p[2], p[5], MDCT_NEG_B); MDCT_PKA(t3, p[3], p[4], ""); MDCT_PKA(t4, p[3], p[4], MDCT_NEG_B);
  MDCT_PKA(e10, t0, t3, ""); MDCT_PKA(e13, t0, t3, MDCT_NEG_B); MDCT_PKA(e11, t1, t2, ""); MDCT_PKA(e12, t1, t2, MDCT_NEG_B);
  MDCT_PKA(s1, e12, e13, "");
  MDCT_PKA(o10, t4, t5, ""); MDCT_PKA(o11, t5, t6, ""); MDCT_PKA(o12, t6, t7, "");
  MDCT_PKA(z5, o10, o12, MDCT_NEG_B); MDCT_PKM(z5, z5, K.c707_382, MDCT_K_HH);
  MDCT_PKF(z2, o10, K.c541_1306, z5, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
  MDCT_PKF(z4, o12, K.c541_1306, z5, "op_sel:[0,1,0] op_sel_hi:[1,1,1]");
  MDCT_PKF(z11, o11, K.c707_382, t7, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
  MDCT_PKF(z13, o11, K.c707_382, t7, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]");
  MDCT_PKA(p[0], e10, e11, ""); MDCT_PKA(p[4], e10, e11, MDCT_NEG_B);
  MDCT_PKF(p[2], s1, K.c707_382, e13, "op_sel:[0,0,0] op_sel_hi:[1,0,1]");
  MDCT_PKF(p[6], s1, K.c707_382, e13, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]");
  MDCT_PKA(p[5], z13, z2, ""); MDCT_PKA(p[3], z13, z2, MDCT_NEG_B);
  MDCT_PKA(p[1], z11, z4, ""); MDCT_PKA(p[7], z11, z4, MDCT_NEG_B);
}

// inverse down a pair of columns, in place (direct image of aan_inv8)
__device__ __forceinline__ void aan_inv_v_stmt(const AanPk &K, f32x2 (&p)[8])
{
  f32x2 e10, e11, e13, e12, t0, t3, t1, t2, z13, z10, z11, z12, t7, d, z5, o10, o12, t6, t5, t4;
  MDCT_PKA(e10, p[0], p[4], ""); MDCT_PKA(e11, p[0], p[4], MDCT_NEG_B);
  MDCT_PKA(e13, p[2], p[6], "");
  MDCT_PKA(d, p[2], p[6], MDCT_NEG_B); MDCT_PKF(e12, d, K.c1414_1847, e13, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]");
  MDCT_PKA(t0, e10, e13, ""); MDCT_PKA(t3, e10, e13, MDCT_NEG_B); MDCT_PKA(t1, e11, e12, ""); MDCT_PKA(t2, e11, e12, MDCT_NEG_B);
  MDCT_PKA(z13, p[5], p[3], ""); MDCT_PKA(z10, p[5], p[3], MDCT_NEG_B); MDCT_PKA(z11, p[1], p[7], ""); MDCT_PKA(z12, p[1], p[7], MDCT_NEG_B);
  MDCT_PKA(t7, z11, z13, "");
  MDCT_PKA(d, z11, z13, MDCT_NEG_B);
  MDCT_PKA(z5, z10, z12, ""); MDCT_PKM(z5, z5, K.c1414_1847, MDCT_K_HH);
  MDCT_PKF(o10, z12, K.c1082_2613, z5, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]");
  MDCT_PKF(o12, z10, K.c1082_2613, z5, "op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]");
  MDCT_PKA(t6, o12, t7, MDCT_NEG_B);
  MDCT_PKF(t5, d, K.c1414_1847, t6, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]");
  MDCT_PKA(t4, o10, t5, "");
  MDCT_PKA(p[0], t0, t7, ""); MDCT_PKA(p[7], t0, t7, MDCT_NEG_B);
  MDCT_PKA(p[1], t1, t6, ""); MDCT_PKA(p[6], t1, t6, MDCT_NEG_B);
  MDCT_PKA(p[2], t2, t5, ""); MDCT_PKA(p[5], t2, t5, MDCT_NEG_B);
  MDCT_PKA(p[4], t3, t4, ""); MDCT_PKA(p[3], t3, t4, MDCT_NEG_B);
}

// inverse, one line given as (c0,c4) (c2,c6) (c5,c3) (c1,c7) -> (x0,x7) (x1,x6) (x2,x5) (x4,x3)
__device__ __forceinline__ void aan_inv_h_stmt(const AanPk &K, f32x2 i04, f32x2 i26, f32x2 i53, f32x2 i17, f32x2 &o07, f32x2 &o16, f32x2 &o25, f32x2 &o43)
{
  f32x2 e, f, t03, t12, z3, z1, td, u, v;
  MDCT_PKA(e, i04, i04, MDCT_SUMDIFF);                    // (e10, e11) = (c0+c4, c0-c4)
  MDCT_PKA(f, i26, i26, MDCT_SUMDIFF);                    // (e13, c2-c6)
  f.y = __builtin_fmaf(f.y, K.c1414_1847.x, -f.x);   // e12 = (c2-c6)*sqrt2 - e13, fused
  MDCT_PKA(t03, e, f, MDCT_LOLO " neg_hi:[0,1]");       // (t0, t3) = (e10+e13, e10-e13)
  MDCT_PKA(t12, e, f, MDCT_HIHI " neg_hi:[0,1]");       // (t1, t2) = (e11+e12, e11-e12)
  MDCT_PKA(z3, i53, i53, MDCT_SUMDIFF);                   // (z13, z10) = (c5+c3, c5-c3)
  MDCT_PKA(z1, i17, i17, MDCT_SUMDIFF);                   // (z11, z12) = (c1+c7, c1-c7)
  MDCT_PKA(td, z1, z3, MDCT_LOLO " neg_hi:[0,1]");      // (t7, z11-z13)
  const float z5 = (z3.y + z1.y) * K.c1414_1847.y;   // (z10 + z12) * c1847
  const float o10 = __builtin_fmaf(K.c1082_2613.x, z1.y, -z5);
  const float o12 = __builtin_fmaf(-K.c1082_2613.y, z3.y, z5);
  u.x = o12 - td.x;                                  // t6
  u.y = __builtin_fmaf(td.y, K.c1414_1847.x, -u.x);  // t5 = (z11-z13)*sqrt2 - t6, fused
  v.x = o10 + u.y;                                   // t4
  MDCT_PKA(o07, t03, td, MDCT_LOLO " neg_hi:[0,1]");    // (t0+t7, t0-t7)
  MDCT_PKA(o16, t12, u, MDCT_LOLO " neg_hi:[0,1]");     // (t1+t6, t1-t6)
  MDCT_PKA(o25, t12, u, MDCT_HIHI " neg_hi:[0,1]");     // (t2+t5, t2-t5)
  MDCT_PKA(o43, t03, v, MDCT_HILO " neg_hi:[0,1]");     // (t3+t4, t3-t4)
}


// ---------------------------------------------------------------------------------------
// The same four passes as ASM BLOCKS (round 5).  One asm statement per operation costs wait states the hardware does not need: for an asm
// statement that reads a VGPR written by the asm statement right before it the compiler inserts an s_nop (it cannot see inside and assumes
// the gfx940 dst_sel forwarding hazard, which a v_pk_*_f32 does not have) -- 97 of them per wave in k_u8_batch, 60-70 in the int16 round
// trip, each an issue slot.  The column passes are pure packed sequences: one block of 34 instructions each, registers allocated by hand,
// results left in a PERMUTATION of the input registers (renaming is free for the caller).  The row passes keep their 6 / 12 unpaired scalar
// operations as compiler-visible code (an inline-asm operand cannot name half of a register pair) between two blocks of packed operations.
// Same operations, same operands, same order of operands in every subtraction as the _stmt forms above (MDCT_AAN_BLOCKS=0 builds those).
// ---------------------------------------------------------------------------------------
#ifndef MDCT_AAN_BLOCKS
#define MDCT_AAN_BLOCKS 1
#endif
#define MDCT_SUB " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define MDCT_KLO " op_sel:[0,0] op_sel_hi:[1,0]\n\t" // times the constant pair's low half, both halves
#define MDCT_KHI " op_sel:[0,1] op_sel_hi:[1,1]\n\t" // times its high half
// v_pk_fma_f32 d, a, k, c: a * (the constant pair's low / high half, both halves) + c; _NA negates a (-a k + c), _NC negates c (a k - c)
#define MDCT_FKLO " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
#define MDCT_FKHI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define MDCT_FKLO_NA " op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define MDCT_FKHI_NA " op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define MDCT_FKLO_NC " op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"

__device__ __forceinline__ void aan_fwd_v(const AanPk &K, f32x2 (&p)[8])
{
#if !MDCT_AAN_BLOCKS
  aan_fwd_v_stmt(K, p);
#else
  f32x2 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4], r5 = p[5], r6 = p[6], r7 = p[7], T;
  asm("v_pk_add_f32 %8, %0, %7\n\t"            /* T  = t0 = p0 + p7 */
      "v_pk_add_f32 %7, %0, %7" MDCT_SUB       /* r7 = t7 = p0 - p7 */
      "v_pk_add_f32 %0, %1, %6\n\t"            /* r0 = t1 */
      "v_pk_add_f32 %6, %1, %6" MDCT_SUB       /* r6 = t6 */
      "v_pk_add_f32 %1, %2, %5\n\t"            /* r1 = t2 */
      "v_pk_add_f32 %5, %2, %5" MDCT_SUB       /* r5 = t5 */
      "v_pk_add_f32 %2, %3, %4\n\t"            /* r2 = t3 */
      "v_pk_add_f32 %4, %3, %4" MDCT_SUB       /* r4 = t4 */
      "v_pk_add_f32 %3, %8, %2\n\t"            /* r3 = e10 = t0 + t3 */
      "v_pk_add_f32 %2, %8, %2" MDCT_SUB       /* r2 = e13 = t0 - t3 */
      "v_pk_add_f32 %8, %0, %1\n\t"            /* T  = e11 = t1 + t2 */
      "v_pk_add_f32 %1, %0, %1" MDCT_SUB       /* r1 = e12 = t1 - t2 */
      "v_pk_add_f32 %0, %3, %8\n\t"            /* r0 = out0 = e10 + e11 */
      "v_pk_add_f32 %8, %3, %8" MDCT_SUB       /* T  = out4 = e10 - e11 */
      "v_pk_add_f32 %3, %1, %2\n\t"            /* r3 = s1 = e12 + e13 */
      "v_pk_add_f32 %1, %4, %5\n\t"            /* r1 = o10 = t4 + t5            (e12 is dead) */
      "v_pk_add_f32 %4, %5, %6\n\t"            /* r4 = o11 = t5 + t6 */
      "v_pk_add_f32 %5, %6, %7\n\t"            /* r5 = o12 = t6 + t7 */
      "v_pk_fma_f32 %6, %3, %9, %2" MDCT_FKLO_NA /* r6 = out6 = -s1 c707 + e13 */
      "v_pk_fma_f32 %2, %3, %9, %2" MDCT_FKLO    /* r2 = out2 = s1 c707 + e13 */
      "v_pk_add_f32 %3, %1, %5" MDCT_SUB       /* r3 = o10 - o12 */
      "v_pk_mul_f32 %3, %3, %9" MDCT_KHI       /* r3 = z5 = (o10 - o12) c382 */
      "v_pk_fma_f32 %1, %1, %10, %3" MDCT_FKLO   /* r1 = z2 = o10 c541 + z5 */
      "v_pk_fma_f32 %5, %5, %10, %3" MDCT_FKHI   /* r5 = z4 = o12 c1306 + z5 */
      "v_pk_fma_f32 %3, %4, %9, %7" MDCT_FKLO    /* r3 = z11 = o11 c707 + t7 */
      "v_pk_fma_f32 %7, %4, %9, %7" MDCT_FKLO_NA /* r7 = z13 = -o11 c707 + t7 */
      "v_pk_add_f32 %4, %7, %1\n\t"            /* r4 = out5 = z13 + z2 */
      "v_pk_add_f32 %7, %7, %1" MDCT_SUB       /* r7 = out3 = z13 - z2 */
      "v_pk_add_f32 %1, %3, %5\n\t"            /* r1 = out1 = z11 + z4 */
      "v_pk_add_f32 %3, %3, %5 neg_lo:[0,1] neg_hi:[0,1]" /* r3 = out7 = z11 - z4 */
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(T)
      : "s"(K.c707_382), "s"(K.c541_1306));
  p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r7; p[4] = T; p[5] = r4; p[6] = r6; p[7] = r3;
#endif
}

__device__ __forceinline__ void aan_inv_v(const AanPk &K, f32x2 (&p)[8])
{
#if !MDCT_AAN_BLOCKS
  aan_inv_v_stmt(K, p);
#else
  f32x2 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4], r5 = p[5], r6 = p[6], r7 = p[7], T;
  asm("v_pk_add_f32 %8, %0, %4\n\t"            /* T  = e10 = p0 + p4 */
      "v_pk_add_f32 %4, %0, %4" MDCT_SUB       /* r4 = e11 = p0 - p4 */
      "v_pk_add_f32 %0, %2, %6\n\t"            /* r0 = e13 = p2 + p6 */
      "v_pk_add_f32 %6, %2, %6" MDCT_SUB       /* r6 = p2 - p6 */
      "v_pk_add_f32 %2, %5, %3\n\t"            /* r2 = z13 = p5 + p3 */
      "v_pk_add_f32 %3, %5, %3" MDCT_SUB       /* r3 = z10 = p5 - p3 */
      "v_pk_add_f32 %5, %1, %7\n\t"            /* r5 = z11 = p1 + p7 */
      "v_pk_add_f32 %7, %1, %7" MDCT_SUB       /* r7 = z12 = p1 - p7 */
      "v_pk_fma_f32 %6, %6, %9, %0" MDCT_FKLO_NC /* r6 = e12 = (p2 - p6) c1414 - e13 */
      "v_pk_add_f32 %1, %5, %2\n\t"            /* r1 = t7 = z11 + z13 */
      "v_pk_add_f32 %5, %5, %2" MDCT_SUB       /* r5 = z11 - z13 */
      "v_pk_add_f32 %2, %3, %7\n\t"            /* r2 = z10 + z12 */
      "v_pk_mul_f32 %2, %2, %9" MDCT_KHI       /* r2 = z5 = (z10 + z12) c1847 */
      "v_pk_fma_f32 %7, %7, %10, %2" MDCT_FKLO_NC /* r7 = o10 = z12 c1082 - z5 */
      "v_pk_fma_f32 %3, %3, %10, %2" MDCT_FKHI_NA /* r3 = o12 = -z10 c2613 + z5 */
      "v_pk_add_f32 %2, %8, %0\n\t"            /* r2 = t0 = e10 + e13 */
      "v_pk_add_f32 %0, %8, %0" MDCT_SUB       /* r0 = t3 = e10 - e13 */
      "v_pk_add_f32 %3, %3, %1" MDCT_SUB       /* r3 = t6 = o12 - t7 */
      "v_pk_add_f32 %8, %4, %6\n\t"            /* T  = t1 = e11 + e12 */
      "v_pk_add_f32 %6, %4, %6" MDCT_SUB       /* r6 = t2 = e11 - e12 */
      "v_pk_fma_f32 %5, %5, %9, %3" MDCT_FKLO_NC /* r5 = t5 = (z11 - z13) c1414 - t6 */
      "v_pk_add_f32 %4, %2, %1\n\t"            /* r4 = out0 = t0 + t7 */
      "v_pk_add_f32 %2, %2, %1" MDCT_SUB       /* r2 = out7 = t0 - t7 */
      "v_pk_add_f32 %7, %7, %5\n\t"            /* r7 = t4 = o10 + t5 */
      "v_pk_add_f32 %1, %8, %3\n\t"            /* r1 = out1 = t1 + t6 */
      "v_pk_add_f32 %8, %8, %3" MDCT_SUB       /* T  = out6 = t1 - t6 */
      "v_pk_add_f32 %3, %6, %5\n\t"            /* r3 = out2 = t2 + t5 */
      "v_pk_add_f32 %6, %6, %5" MDCT_SUB       /* r6 = out5 = t2 - t5 */
      "v_pk_add_f32 %5, %0, %7\n\t"            /* r5 = out4 = t3 + t4 */
      "v_pk_add_f32 %0, %0, %7 neg_lo:[0,1] neg_hi:[0,1]" /* r0 = out3 = t3 - t4 */
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(T)
      : "s"(K.c1414_1847), "s"(K.c1082_2613));
  p[0] = r4; p[1] = r1; p[2] = r3; p[3] = r0; p[4] = r5; p[5] = r6; p[6] = T; p[7] = r2;
#endif
}

// forward, one line in natural pairs (p0,p1)(p2,p3)(p4,p5)(p6,p7) -> (y0,y4) (y2,y6) (y5,y3) (y1,y7)
__device__ __forceinline__ void aan_fwd_h(const AanPk &K, f32x2 a01, f32x2 a23, f32x2 a45, f32x2 a67, f32x2 &o04, f32x2 &o26, f32x2 &o53, f32x2 &o17)
{
#if !MDCT_AAN_BLOCKS
  aan_fwd_h_stmt(K, a01, a23, a45, a67, o04, o26, o53, o17);
#else
  f32x2 T0, T1;
  asm("v_pk_add_f32 %4, %0, %3" MDCT_XSEL "\n\t"                               /* T0 = (t0, t1) = (p0+p7, p1+p6) */
      "v_pk_add_f32 %5, %1, %2" MDCT_XSEL "\n\t"                               /* T1 = (t2, t3) = (p2+p5, p3+p4) */
      "v_pk_add_f32 %0, %0, %3" MDCT_XSEL " neg_lo:[0,1] neg_hi:[0,1]\n\t"     /* A  = (t7, t6) = (p0-p7, p1-p6) */
      "v_pk_add_f32 %1, %1, %2" MDCT_XSEL " neg_lo:[0,1] neg_hi:[0,1]\n\t"     /* B  = (t5, t4) = (p2-p5, p3-p4) */
      "v_pk_add_f32 %2, %4, %5" MDCT_XSEL "\n\t"                               /* C  = (e10, e11) = (t0+t3, t1+t2) */
      "v_pk_add_f32 %3, %4, %5" MDCT_XSEL " neg_lo:[0,1] neg_hi:[0,1]\n\t"     /* D  = (e13, e12) = (t0-t3, t1-t2) */
      "v_pk_add_f32 %4, %2, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]"      /* T0 = (e10+e11, e10-e11) */
      : "+v"(a01), "+v"(a23), "+v"(a45), "+v"(a67), "=&v"(T0), "=&v"(T1));
  const f32x2 t76 = a01, t54 = a23, e32 = a67;
  o04 = T0;
  f32x2 w, o, z5;
  w.x = e32.y + e32.x;                               // s1 = e12 + e13
  w.y = t54.x + t76.y;                               // o11 = t5 + t6
  o.x = t54.y + t54.x;                               // o10 = t4 + t5
  o.y = t76.y + t76.x;                               // o12 = t6 + t7
  z5.x = (o.x - o.y) * K.c707_382.y;                 // z5 = (o10 - o12) * c382
  z5.y = z5.x;
  f32x2 T;
  asm("v_pk_fma_f32 %1, %1, %9, %7 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"                  /* o  = (z2, z4) = (o10 c541 + z5, o12 c1306 + z5) */
      "v_pk_fma_f32 %2, %0, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[1,0,0]\n\t"   /* T  = (z11, z13) = (o11 c707 + t7, -o11 c707 + t7) */
      "v_pk_fma_f32 %3, %0, %8, %6 op_sel:[0,0,0] op_sel_hi:[0,0,0] neg_hi:[1,0,0]\n\t"   /* o26 = (s1 c707 + e13, -s1 c707 + e13) */
      "v_pk_add_f32 %4, %2, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"             /* o53 = (z13+z2, z13-z2) */
      "v_pk_add_f32 %2, %2, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]"                 /* T -> o17 = (z11+z4, z11-z4) */
      : "+v"(w), "+v"(o), "=&v"(T), "=&v"(o26), "=&v"(o53)
      : "v"(t76), "v"(e32), "v"(z5), "s"(K.c707_382), "s"(K.c541_1306));
  o17 = T;
#endif
}

// inverse, one line given as (c0,c4) (c2,c6) (c5,c3) (c1,c7) -> (x0,x7) (x1,x6) (x2,x5) (x4,x3)
__device__ __forceinline__ void aan_inv_h(const AanPk &K, f32x2 i04, f32x2 i26, f32x2 i53, f32x2 i17, f32x2 &o07, f32x2 &o16, f32x2 &o25, f32x2 &o43)
{
#if !MDCT_AAN_BLOCKS
  aan_inv_h_stmt(K, i04, i26, i53, i17, o07, o16, o25, o43);
#else
  f32x2 td;
  asm("v_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"  /* e  = (e10, e11) = (c0+c4, c0-c4) */
      "v_pk_add_f32 %1, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"  /* f  = (e13, c2-c6) */
      "v_pk_add_f32 %2, %2, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"  /* z3 = (z13, z10) = (c5+c3, c5-c3) */
      "v_pk_add_f32 %3, %3, %3 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"  /* z1 = (z11, z12) = (c1+c7, c1-c7) */
      "v_pk_add_f32 %4, %3, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]"      /* td = (t7, z11-z13) */
      : "+v"(i04), "+v"(i26), "+v"(i53), "+v"(i17), "=&v"(td));
  f32x2 f = i26, u;
  f.y = __builtin_fmaf(f.y, K.c1414_1847.x, -f.x);       // e12 = (c2-c6)*sqrt2 - e13, fused
  const float z5 = (i53.y + i17.y) * K.c1414_1847.y;     // (z10 + z12) * c1847
  const float o10 = __builtin_fmaf(K.c1082_2613.x, i17.y, -z5);
  const float o12 = __builtin_fmaf(-K.c1082_2613.y, i53.y, z5);
  u.x = o12 - td.x;                                      // t6
  u.y = __builtin_fmaf(td.y, K.c1414_1847.x, -u.x);      // t5 = (z11-z13)*sqrt2 - t6, fused
  td.y = o10 + u.y;                                      // t4 (z11-z13 is dead: its half of td carries t4 into the block below)
  f32x2 t03, t12;
  asm("v_pk_add_f32 %4, %6, %7 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"  /* t03 = (t0, t3) = (e10+e13, e10-e13) */
      "v_pk_add_f32 %5, %6, %7 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"  /* t12 = (t1, t2) = (e11+e12, e11-e12) */
      "v_pk_add_f32 %0, %4, %8 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"  /* o07 = (t0+t7, t0-t7) */
      "v_pk_add_f32 %3, %4, %8 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"  /* o43 = (t3+t4, t3-t4) */
      "v_pk_add_f32 %1, %5, %9 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"  /* o16 = (t1+t6, t1-t6) */
      "v_pk_add_f32 %2, %5, %9 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]"      /* o25 = (t2+t5, t2-t5) */
      : "=&v"(o07), "=&v"(o16), "=&v"(o25), "=&v"(o43), "=&v"(t03), "=&v"(t12)
      : "v"(i04), "v"(f), "v"(td), "v"(u));
#endif
}

// quantise -> dequantise of the eight pairs of one column pair, SAT-free: c = rne(y qf) by the fused magic add (quant_i16_bits) and the subtract, z = c dq
__device__ __forceinline__ void quant_dequant8(const AanPk &K, f32x2 (&p)[8], const f32x2 (&qf)[8], const f32x2 (&dq)[8])
{
#define MDCT_Q8(OP, A, B, MOD) OP " %0, %0, " A "0" B MOD OP " %1, %1, " A "1" B MOD OP " %2, %2, " A "2" B MOD OP " %3, %3, " A "3" B MOD OP " %4, %4, " A "4" B MOD OP " %5, %5, " A "5" B MOD OP " %6, %6, " A "6" B MOD OP " %7, %7, " A "7" B MOD
  asm("v_pk_fma_f32 %0, %0, %8, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t" /* y qf + 1.5 2^23, one rounding: rne of the exact product */
      "v_pk_fma_f32 %1, %1, %9, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %2, %2, %10, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %3, %3, %11, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %4, %4, %12, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %5, %5, %13, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %6, %6, %14, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %7, %7, %15, %24 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t"
      "v_pk_add_f32 %0, %0, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %1, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %2, %2, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %3, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %4, %4, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %5, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %6, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %7, %24 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %0, %0, %16 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %1, %1, %17 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %2, %2, %18 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %3, %3, %19 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %4, %4, %20 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %5, %5, %21 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %6, %6, %22 op_sel:[0,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %7, %7, %23 op_sel:[0,0] op_sel_hi:[1,1]"
      : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
      : "s"(qf[0]), "s"(qf[1]), "s"(qf[2]), "s"(qf[3]), "s"(qf[4]), "s"(qf[5]), "s"(qf[6]), "s"(qf[7]), "s"(dq[0]), "s"(dq[1]), "s"(dq[2]), "s"(dq[3]), "s"(dq[4]), "s"(dq[5]),
        "s"(dq[6]), "s"(dq[7]), "v"(K.magic));
#undef MDCT_Q8
}

// Fused round trip on packed fp32: forward rows (h), forward columns (v), [quantise -> dequantise], inverse columns (v),
// inverse rows (h).  With a table, `tb` holds the multipliers in the pair order of the column pass,
// j-major, (j*8 + v)*2 + {0,1} = coefficient (v, A[j]) / (v, B[j]) with A = {0,2,5,1}, B = {4,6,3,7}: the pairs aan_fwd_h
// produces and aan_inv_h consumes (mdct_api.hip: make_own_tables).
// PRIO: the wave raises its issue priority as it advances (forward rows 1, columns 2, inverse rows + stores 3), so that of
// the waves sharing a SIMD the one closest to its stores goes first (shortest remaining work first)
// SAT = false (decided on the host: every table entry >= 8.01): a quantised coefficient cannot leave int16 -- an orthonormal 8x8
// DCT coefficient of int16 samples is at most 8 * 32768 in magnitude -- so the two saturations per coefficient pair are left out.
// tbp: where the plane's OwnTables lie -- in the kernel's argument segment (karg_bytes) or in device memory (const_bytes); read with
// scalar loads either way
typedef const __attribute__((address_space(4))) char *kbytes_t;
template <bool HAS_LUT, class Rows, bool PRIO = false, bool SAT = true>
__device__ __forceinline__ void i16_roundtrip_rows(const DctConsts &C, const Rows rows, kbytes_t tbp);

// The two tables are 128 multiplier pairs = 256 SGPRs if the compiler is left to fetch them when it likes -- it fetches
// them all at the top and spills (252 v_readlane + 124 v_writelane per wave, 1451 vector instructions instead of ~1000).
// So the pairs of column pair j are read from the argument segment through a pointer the compiler cannot see through,
// right where they are used: two s_load_dwordx16 per j, 32 SGPRs live (mdct_api.hip lays the tables out j-major for this).
typedef const __attribute__((address_space(4))) f32x2 *karg_pairs_t;
__device__ __forceinline__ kbytes_t karg_bytes(size_t byte_off) { return (kbytes_t)__builtin_amdgcn_kernarg_segment_ptr() + byte_off; }
__device__ __forceinline__ karg_pairs_t karg_pairs(size_t byte_off) { return (karg_pairs_t)karg_bytes(byte_off); }
// a wave-uniform device address as a constant-address-space pointer (scalar loads; the memory must not change during the launch)
__device__ __forceinline__ kbytes_t const_bytes(const void *p)
{
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (kbytes_t)(((uint64_t)hi << 32) | lo);
}

// c = sat_i16(rne(y * qf)); z = c * dq on the eight pairs of column pair j; tq / td: the pairs' multipliers (scalar loads)
template <bool SAT>
__device__ __forceinline__ void quant_dequant_pairs(const AanPk &K, f32x2 (&p)[8], karg_pairs_t tq, karg_pairs_t td)
{
  if constexpr (!SAT && MDCT_AAN_BLOCKS)
  {
    const f32x2 qf[8] = {tq[0], tq[1], tq[2], tq[3], tq[4], tq[5], tq[6], tq[7]}, dq[8] = {td[0], td[1], td[2], td[3], td[4], td[5], td[6], td[7]};
    quant_dequant8(K, p, qf, dq);
  }
  else
  {
    const f32x2 magic_v = K.magic; // (the fused add's third source must be a VGPR: one SGPR operand per instruction on gfx9)
#pragma unroll
    for (int v = 0; v < 8; v++)
    {
      const f32x2 qf = tq[v], dq = td[v];
      f32x2 m;
      MDCT_PKF(m, p[v], qf, magic_v, "op_sel:[0,0,0] op_sel_hi:[1,1,0]"); // quant_i16_bits on both halves
      if constexpr (SAT)
      {
        m.x = __builtin_amdgcn_fmed3f(m.x, kQLo, kQHi);
        m.y = __builtin_amdgcn_fmed3f(m.y, kQLo, kQHi);
      }
      MDCT_PKA(m, m, K.magic, MDCT_K_LL " " MDCT_NEG_B);
      MDCT_PKM(p[v], m, dq, MDCT_K_LH);
    }
  }
}

// (A wave that walks 2 / 4 / 8 consecutive tiles and issues the next tile's row loads between the row pass and the column pass,
// into the registers the row pass has just freed, was measured on the bench workload: 44.8-45.4 / 52.2 / 51.4 us against 43.8,
// profiles/r04_exp_i16_tiles_per_wave.log -- one tile per wave it stays.)
template <bool HAS_LUT, class Rows, bool PRIO, bool SAT>
__device__ __forceinline__ void i16_roundtrip_rows(const DctConsts &C, const Rows rows, kbytes_t tbp)
{
  const AanPk &K = reinterpret_cast<const AanPk &>(C);
  uint4 in[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    in[r] = rows.ld(r);
  MDCT_PHASE_PRIO(1);
  f32x2 P[4][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const f32x2 a01 = {(float)(int16_t)(in[r].x & 0xFFFF), (float)(int16_t)(in[r].x >> 16)};
    const f32x2 a23 = {(float)(int16_t)(in[r].y & 0xFFFF), (float)(int16_t)(in[r].y >> 16)};
    const f32x2 a45 = {(float)(int16_t)(in[r].z & 0xFFFF), (float)(int16_t)(in[r].z >> 16)};
    const f32x2 a67 = {(float)(int16_t)(in[r].w & 0xFFFF), (float)(int16_t)(in[r].w >> 16)};
    aan_fwd_h(K, a01, a23, a45, a67, P[0][r], P[1][r], P[2][r], P[3][r]);
  }
  MDCT_PHASE_PRIO(2);
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    aan_fwd_v(K, P[j]);
    if constexpr (HAS_LUT)
    { // c = sat_i16(rne(y * qf)); z = c * dq  (quant_i16_float), on both halves
      karg_pairs_t tq = (karg_pairs_t)(tbp + offsetof(OwnTables, qf)) + j * 8, td = (karg_pairs_t)(tbp + offsetof(OwnTables, dq)) + j * 8;
      asm volatile("" : "+s"(tq), "+s"(td));
      quant_dequant_pairs<SAT>(K, P[j], tq, td);
    }
    aan_inv_v(K, P[j]);
  }
  MDCT_PHASE_PRIO(3);
  // without a table forward-scale * inverse-scale == 1/64 exactly and rides in the final rounding (rne_i16_bits<6>)
  constexpr float scale = HAS_LUT ? 1.0f : 64.0f;
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    f32x2 o07, o16, o25, o43;
    aan_inv_h(K, P[0][r], P[1][r], P[2][r], P[3][r], o07, o16, o25, o43);
    auto fin = [&](f32x2 v) {
      f32x2 t;
      v.x = __builtin_amdgcn_fmed3f(v.x, -32768.0f * scale, 32767.0f * scale);
      v.y = __builtin_amdgcn_fmed3f(v.y, -32768.0f * scale, 32767.0f * scale);
      if constexpr (HAS_LUT)
        MDCT_PKA(t, v, K.magic, MDCT_K_LL);
      else
        MDCT_PKA(t, v, K.magic, MDCT_K_HH);
      return t;
    };
    const f32x2 b07 = fin(o07), b16 = fin(o16), b25 = fin(o25), b43 = fin(o43);
    rows.st(r, pack_lo16(__float_as_uint(b07.x), __float_as_uint(b16.x)), pack_lo16(__float_as_uint(b25.x), __float_as_uint(b43.y)),
            pack_lo16(__float_as_uint(b43.x), __float_as_uint(b25.y)), pack_lo16(__float_as_uint(b16.y), __float_as_uint(b07.y)));
  }
}

// Where a single-plane launch finds its tables.  The argument segment is written afresh by the host for every launch, so a wave's
// scalar loads from it miss every cache the first time round; tables that the host has parked in device memory (mdct_api.hip: the table
// cache) are hot in L2 across launches.  Measured on the 8192^2 round trip with a table: 46.4 us from the arguments, 44.8 from device
// memory -- the price of the table disappears (profiles/r04_time_batch_c.log).
typedef const __attribute__((address_space(4))) OwnTables *ktables_t;
__device__ __forceinline__ kbytes_t i16_tables(const I16Args &a) { return a.tb_dev ? const_bytes(a.tb_dev) : karg_bytes(offsetof(I16Args, tb)); }

// The compiler's register/scheduling heuristic is steered per mode with amdgpu_waves_per_eu; the
// values are the measured optimum of {2..6} on MI355X, ROCm 7.2 (profiles/r01_waves_per_eu.log):
// forward 44.0 us with 2 (46.6 with 4), inverse 45.7 us with 4 (47.8 with 2), fused round trip
// 47.4 us with 3 (48.3 with 4, 51.5 with 2).  Left to itself the compiler sometimes serialises the eight row loads
// behind the butterflies (58 vs 49 us on identical work, tools/time_planes.py).
constexpr int i16_waves(int mode) { return mode == MODE_ROUNDTRIP ? 3 : (mode == MODE_FWD ? 2 : 4); }
template <int MODE, bool HAS_LUT, bool SAT = true>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(i16_waves(MODE), i16_waves(MODE)))) void k_i16(I16Args a)
{
  const uint32_t t = wg_index() * kWG + threadIdx.x;
  if (t >= a.nblocks)
    return;
  const uint32_t row = t / a.bpr;
  const uint32_t bx = t - row * a.bpr;
  const size_t by = a.by0 + row;
  const int16_t *src = a.from + by * 8 * a.pitch_in + (size_t)bx * 8;
  int16_t *dst = a.to + by * 8 * a.pitch_out + (size_t)bx * 8;
  const kbytes_t tbp = i16_tables(a);
  if constexpr (MODE == MODE_ROUNDTRIP)
    i16_roundtrip_rows<HAS_LUT, RowsLinear, false, SAT>(a.consts, RowsLinear{src, dst, a.pitch_in, a.pitch_out}, tbp);
  else
    i16_block<MODE, HAS_LUT>(a.consts, RowsLinear{src, dst, a.pitch_in, a.pitch_out}, *(ktables_t)tbp);
}

// The same for launches whose waves each lie in one block row (sizeX % 512 == 0): one workgroup = one wave = one 64-block
// tile, 2-D grid (x = tile in the row, y = block row), wave-uniform addressing (RowsTiled).  One-wave workgroups matter
// here: the per-wave timeline of the 4-wave form (profiles/r03_i16_timeline.md) shows a SIMD's slots idle 45 % of the
// time with 2 of 3 waves resident -- a freed slot waits until the other three waves of its workgroup are done too.
// scheduling steered per mode like k_i16 (measured, profiles/r03_exp_i16_tile_and_priorities.log): forward 42.7 us with 2
// (44.1 with 3, 45.1 with 4; k_i16<FWD> 43.7; the copy kernel 43.4), inverse 43.2 with 2 (45.1 with 4; k_i16<INV> 45.4),
// fused round trip with phase priorities 44.3 with 2, 44.7 with 3, 45.9 with 4 (without priorities 48.4 / 45.8 / 46.0;
// k_i16<ROUNDTRIP> 47.1-48.6)
// WAVES: 2 is the steady-state optimum above; a launch that fills the chip only once or twice over is bound by the latency of a
// generation of waves, not by the steady state, and finishes sooner with more of it resident: 4096^2 (4096 tiles) 13.8-14.0 us at 2
// waves per SIMD, 12.3-12.8 at 4 (8192^2: 43.6 against 45.1; profiles/r04_exp_small_planes_waves.log).  The launcher picks by tile count.
constexpr uint32_t kTileSmallLaunch = 6144; // tiles: up to here 4 waves per SIMD, beyond 2 (7680 x 4320 = 8100 tiles is indifferent)
template <int MODE, bool HAS_LUT, bool SAT = true, int WAVES = 2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_i16_tile(I16Args a)
{
  const size_t by = a.by0 + blockIdx.y;
  const RowsTiled rows{a.from + by * 8 * a.pitch_in + (size_t)blockIdx.x * 512, a.to + by * 8 * a.pitch_out + (size_t)blockIdx.x * 512, a.pitch_in, a.pitch_out, threadIdx.x * 16, threadIdx.x * 16};
  const kbytes_t tbp = i16_tables(a);
  if constexpr (MODE == MODE_ROUNDTRIP)
    i16_roundtrip_rows<HAS_LUT, RowsTiled, true, SAT>(a.consts, rows, tbp); // with phase priorities
  else
    i16_block<MODE, HAS_LUT>(a.consts, rows, *(ktables_t)tbp);
}

// Any number of separately allocated planes in ONE launch (BASELINE.json configs[2]: Y + Cb + Cr with their own tables; configs[3]:
// 256 independent planes).  The reference's only batching affordance is the caller-side row range (simd_dct.cpp:2243-2261); this is
// the engine's: a 1-D grid over the 64-block tiles of all planes in order, one wave per tile with k_i16_tile's wave-uniform
// addressing.  A wave finds its plane without a division (equal shapes: magic multiply; up to 8 different shapes: a compare chain
// on kernel arguments; more: a binary search with scalar loads), reads the plane's 64-byte descriptor and its tables with scalar
// loads, and a row's last tile may be partial (1920- and 3840-wide planes are 240 / 480 blocks per row): lanes beyond the row
// leave at once -- nothing below crosses lanes.
typedef uint32_t u32x16_s __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_s __attribute__((ext_vector_type(4)));
static_assert(offsetof(BatchDesc, from) == 0 && offsetof(BatchDesc, to) == 8 && offsetof(BatchDesc, pitch_in) == 16 && offsetof(BatchDesc, pitch_out) == 24 && offsetof(BatchDesc, bpr) == 32 &&
                  offsetof(BatchDesc, tiles) == 36 && offsetof(BatchDesc, tiles_m) == 40 && offsetof(BatchDesc, tiles_s) == 44 && offsetof(BatchDesc, first) == 48 && offsetof(BatchDesc, table) == 52 &&
                  offsetof(BatchDesc, has_lut) == 56,
              "batch_tile picks the descriptor's fields out of one 16-dword load");

// What tile `w` of a batch launch is: its plane's descriptor, the block row and the tile within the row, where the plane's tables lie.
// Exactly two dependent rounds of scalar loads stand between a wave's start and its first row load: the header (20 dwords, fetched
// as ONE batch through a pointer the compiler cannot see through -- left to itself it loaded field by field, ten round trips), then
// the plane's descriptor (one s_load_dwordx16).  The index arithmetic is batch_plan.h's (batch_plane_of, magic_apply): the code the
// CPU test walks under the sanitizers is the code that runs here.
struct BatchTile
{
  u32x16_s d;      // the plane's BatchDesc
  uint32_t row;    // block row within the plane
  uint32_t tile;   // 64-block tile within the row (the last one possibly partial); b0 / 64 where rows are not paired
  uint32_t b0, s, straddle; // batch_pos (batch_plan.h): first block within the row, blocks of the tile in this row, lanes >= s start the next row
  kbytes_t tables; // the plane's OwnTables
  __device__ __forceinline__ uint32_t bpr() const { return d[8]; }
  __device__ __forceinline__ uint32_t has_lut() const { return d[14] & kDescLut; }
  __device__ __forceinline__ uint64_t from() const { return ((uint64_t)d[1] << 32) | d[0]; }
  __device__ __forceinline__ uint64_t to() const { return ((uint64_t)d[3] << 32) | d[2]; }
  __device__ __forceinline__ size_t pitch_in() const { return ((uint64_t)d[5] << 32) | d[4]; }
  __device__ __forceinline__ size_t pitch_out() const { return ((uint64_t)d[7] << 32) | d[6]; }
};
template <bool PAIRS = false> // PAIRS: the kernel handles kDescPaired planes (k_u8_batch, k_q32_batch)
__device__ __forceinline__ BatchTile batch_tile(uint32_t w)
{
  kbytes_t args = karg_bytes(0);
  asm volatile("" : "+s"(args)); // opaque: the loads below stay two wide loads, issued together
  const u32x16_s h = *(const __attribute__((address_space(4))) u32x16_s *)args;
  const u32x4_s ptrs = *(const __attribute__((address_space(4))) u32x4_s *)(args + offsetof(BatchArgs, descs));
  const uint32_t n = h[0], uniform = h[1], pp_m = h[2], pp_s = h[3], table_bytes = h[4];
  const uint64_t descs_dev = ((uint64_t)ptrs[1] << 32) | ptrs[0], tables_dev = ((uint64_t)ptrs[3] << 32) | ptrs[2];
  const kbytes_t blob = args + offsetof(BatchArgs, blob);
  const kbytes_t descs = descs_dev ? (kbytes_t)descs_dev : blob + table_bytes;
  const kbytes_t tables = tables_dev ? (kbytes_t)tables_dev : blob;
  const uint32_t first8[kBatchChain] = {h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]};
  static_assert(kBatchChain == 8 && offsetof(BatchHead, first8) == 32, "first8 = dwords 8..15 of the header");
  uint32_t p = batch_plane_of(w, n, uniform, pp_m, pp_s, first8, [descs](uint32_t k) {
    return *(const __attribute__((address_space(4))) uint32_t *)(descs + (size_t)k * sizeof(BatchDesc) + offsetof(BatchDesc, first));
  });
  p = __builtin_amdgcn_readfirstlane(p); // (the compare chain may have been evaluated on the vector unit)
  kbytes_t dp = descs + (size_t)p * sizeof(BatchDesc);
  asm volatile("" : "+s"(dp));
  BatchTile t;
  t.d = *(const __attribute__((address_space(4))) u32x16_s *)dp;
  const uint32_t lt = w - t.d[12];
  if constexpr (PAIRS)
  {
    const BatchPos at = batch_pos(lt, t.d[8], t.d[9], t.d[10], t.d[11], t.d[14], t.d[15]);
    t.row = at.row, t.b0 = at.b0, t.s = at.s, t.straddle = at.straddle, t.tile = at.b0 >> 6;
  }
  else
  {
    t.row = magic_apply(lt, t.d[10], t.d[11]);
    t.tile = lt - t.row * t.d[9];
    t.b0 = t.tile * 64, t.s = 64, t.straddle = 0; // (unused by the kernels that call this form)
  }
  t.tables = tables + t.d[13];
  return t;
}

// Scheduling steered per mode like k_i16_tile; measured on the 8K 4:2:0 frame and on batches of one (profiles/r04_exp_batch_variants.log):
// the fused round trip with phase priorities at 3 waves per SIMD (frame 38.0 us against 40.3 at 2, 8192^2 45.3 against 46.0-46.5),
// forward and inverse at 2 (8192^2 forward 44.2-44.4 against 44.8-45.0 at 3 and 46.2 at 4).
constexpr int batch_waves(int mode) { return mode == MODE_ROUNDTRIP ? 3 : 2; }
// (and, like k_i16_tile, 4 waves per SIMD for launches of 2049..6144 tiles: SMALL)
template <int MODE, int LUTMODE, bool SAT = true, bool SMALL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SMALL ? 4 : batch_waves(MODE), SMALL ? 4 : batch_waves(MODE)))) void k_i16_batch(BatchArgs a)
{
  const BatchTile t = batch_tile<true>(blockIdx.x);
  if (!t.straddle && threadIdx.x >= t.s) // (a partial last tile drops its lanes)
    return;
  const size_t pin = t.pitch_in(), pout = t.pitch_out();
  // a paired plane's straddling tile: lanes >= s continue at block 0 of the next block row -- 8 rows down, b0 + s = bpr blocks back
  const uint32_t hop = (t.straddle && threadIdx.x >= t.s) ? 0xFFFFFFFFu : 0u;
  const RowsTiled rows{(const int16_t *)t.from() + (size_t)t.row * 8 * pin + (size_t)t.b0 * 8, (int16_t *)t.to() + (size_t)t.row * 8 * pout + (size_t)t.b0 * 8, pin, pout,
                       threadIdx.x * 16 + (hop & (((uint32_t)pin - t.bpr()) * 16)), threadIdx.x * 16 + (hop & (((uint32_t)pout - t.bpr()) * 16))};
  if constexpr (MODE == MODE_ROUNDTRIP)
  {
    if (LUTMODE == BATCH_ALL_LUT || (LUTMODE == BATCH_MIXED && t.has_lut()))
      i16_roundtrip_rows<true, RowsTiled, true, SAT>(a.consts, rows, t.tables);
    else
      i16_roundtrip_rows<false, RowsTiled, true>(a.consts, rows, nullptr);
  }
  else
    i16_block<MODE, true>(a.consts, rows, *(ktables_t)t.tables);
}

// ---------------------------------------------------------------------------------------
// 8-bit pixels in, 8-bit pixels out: forward -> quantise -> dequantise -> inverse in one pass (BASELINE.json configs[2] as SURVEY.md
// 8(d) states it: u8 planes, 2 bytes per pixel; the reference's pixel type, simd_dct.cpp:2107-2143, and the half of the codec it
// does not have).  Bit for bit the forward half (k_u8_i16_fwd == k_u8_batch<U8_FWD>) followed by the inverse half (k_u8_batch<U8_INV>); the int16 coefficients exist only in registers.
//   rows in      8 x 8 B per lane (a tile row = 512 contiguous bytes per wave load), v_cvt_f32_ubyteN
//   forward      aan_fwd_h per row, aan_fwd_v per column pair -- the packed butterflies of i16_roundtrip_rows; the level shift is
//                "raw DC - 64 * 128" (k_u8_i16_fwd)
//   quantiser    c = sat_i16(rne(y * qf)), z = c * dq on register pairs; SAT = false (host: every table entry >= 1/16 in
//                magnitude, so |y / lut| <= 16 * 2040 stays inside int16) leaves the saturations out
//   inverse      aan_inv_v, aan_inv_h
//   rows out     sat_u8(rne(x)): ONE v_cvt_pk_u8_f32 per pixel (round to nearest even, saturate to [0, 255], NaN -> 0; probed on gfx950,
//                tools/probe_cvt_pk_u8.hip; it issues like any convert, tools/valubench2) -- 8 instructions per row where the magic add, the
//                int16 pairing and v_sat_pk_u8_i16 took 14 (rounds 5: 18 with the clamp).  The level shift of the output (+ 128) rides in the
//                DC term: z00 + shift before the inverse transform (a constant plane is exactly the DC term of the AAN inverse), round 6.
// No LDS, nothing crosses lanes: a partial last tile just drops its lanes.
// ---------------------------------------------------------------------------------------
// four pixels -> one dword of bytes: sat_u8(rne(x)) each (v_cvt_pk_u8_f32 inserts byte `sel` into the dword it is given)
__device__ __forceinline__ uint32_t pack4_sat_u8(float x0, float x1, float x2, float x3)
{
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(x0, 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(x1, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(x2, 2, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(x3, 3, w);
}
// MODE: the fused round trip (pixels in, pixels out), or one half of it on the same tiles -- U8_FWD pixels -> quantised int16 coefficients
// (bit for bit k_u8_i16_fwd), U8_INV int16 coefficients -> pixels: what an encoder / a decoder runs on a frame's
// planes in one launch.  The int16 side is a plane of 16-byte rows per lane (a tile row = 1 KiB), pitch in elements.
enum { U8_RT = 0, U8_FWD = 1, U8_INV = 2 };
template <int MODE, bool SAT, bool PRIO>
// in_off / out_off: the lane's byte offset from src / dst (lane * bytes per block row, plus the hop to the next block row for the upper lanes of a
// straddling tile of a paired plane)
// shifts = (64 * shift, shift): the level shift leaves with the forward DC ("raw DC - 64 * shift") and comes back with the inverse's (z00 + shift)
__device__ __forceinline__ void u8_rows(const DctConsts &C, const f32x2 shifts, const void *src, void *dst, size_t pitch_in, size_t pitch_out, uint32_t in_off, uint32_t out_off, kbytes_t tbp)
{
  const AanPk &K = reinterpret_cast<const AanPk &>(C);
  f32x2 P[4][8];
  if constexpr (MODE != U8_INV)
  {
    uint2 rows[8];
    load_block_rows_g(static_cast<const uint8_t *>(src), pitch_in, in_off, rows);
    MDCT_PHASE_PRIO(1);
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const f32x2 a01 = {ubyte_to_float<0>(rows[r].x), ubyte_to_float<1>(rows[r].x)}, a23 = {ubyte_to_float<2>(rows[r].x), ubyte_to_float<3>(rows[r].x)};
      const f32x2 a45 = {ubyte_to_float<0>(rows[r].y), ubyte_to_float<1>(rows[r].y)}, a67 = {ubyte_to_float<2>(rows[r].y), ubyte_to_float<3>(rows[r].y)};
      aan_fwd_h(K, a01, a23, a45, a67, P[0][r], P[1][r], P[2][r], P[3][r]);
    }
  }
  else
  { // coefficient row v of the block: (c0,c1)(c2,c3)(c4,c5)(c6,c7) in four dwords -> the pairs the column pass works on, (v, A[j]) / (v, B[j])
    const RowsTiled in{static_cast<const int16_t *>(src), nullptr, pitch_in, 0, in_off, 0};
    uint4 rows[8];
#pragma unroll
    for (int v = 0; v < 8; v++)
      rows[v] = in.ld(v);
    MDCT_PHASE_PRIO(1);
#pragma unroll
    for (int v = 0; v < 8; v++)
    {
      const uint4 w = rows[v];
      P[0][v] = f32x2{(float)(int16_t)(w.x & 0xFFFF), (float)(int16_t)(w.z & 0xFFFF)}; // (c0, c4)
      P[1][v] = f32x2{(float)(int16_t)(w.y & 0xFFFF), (float)(int16_t)(w.w & 0xFFFF)}; // (c2, c6)
      P[2][v] = f32x2{(float)(int16_t)(w.z >> 16), (float)(int16_t)(w.y >> 16)};       // (c5, c3)
      P[3][v] = f32x2{(float)(int16_t)(w.x >> 16), (float)(int16_t)(w.w >> 16)};       // (c1, c7)
    }
  }
  MDCT_PHASE_PRIO(2);
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    karg_pairs_t tq = (karg_pairs_t)(tbp + offsetof(OwnTables, qf)) + j * 8, td = (karg_pairs_t)(tbp + offsetof(OwnTables, dq)) + j * 8;
    asm volatile("" : "+s"(tq), "+s"(td)); // the pairs' multipliers by scalar loads, right where they are used (i16_roundtrip_rows)
    if constexpr (MODE != U8_INV)
    {
      aan_fwd_v(K, P[j]);
      if (j == 0)
        P[0][0].x = P[0][0].x - shifts.x; // the level shift is exactly "raw DC minus 64 * 128"
    }
    if constexpr (MODE == U8_RT)
      quant_dequant_pairs<SAT>(K, P[j], tq, td);
    else if constexpr (MODE == U8_FWD)
    { // c = sat_i16(rne(y * qf)), quant_i16_bits on both halves: the fused magic add leaves the int16 in the low half of the word, the clamp (SAT) works on the biased value
      const f32x2 magic_v = K.magic;
#pragma unroll
      for (int v = 0; v < 8; v++)
      {
        f32x2 m;
        MDCT_PKF(m, P[j][v], tq[v], magic_v, "op_sel:[0,0,0] op_sel_hi:[1,1,0]");
        if constexpr (SAT)
        {
          m.x = __builtin_amdgcn_fmed3f(m.x, kQLo, kQHi);
          m.y = __builtin_amdgcn_fmed3f(m.y, kQLo, kQHi);
        }
        P[j][v] = m;
      }
    }
    else
    {
#pragma unroll
      for (int v = 0; v < 8; v++)
        MDCT_PKM(P[j][v], P[j][v], td[v], MDCT_K_LH); // z = c * dq
    }
    if constexpr (MODE != U8_FWD)
    {
      if (j == 0)
        P[0][0].x = P[0][0].x + shifts.y; // the output's level shift: every pixel + shift == the DC term + shift, before the inverse
      aan_inv_v(K, P[j]);
    }
  }
  MDCT_PHASE_PRIO(3);
  if constexpr (MODE == U8_FWD)
  {
    const RowsTiled out{nullptr, static_cast<int16_t *>(dst), 0, pitch_out, 0, out_off};
#pragma unroll
    for (int v = 0; v < 8; v++)
      out.st(v, pack_lo16(__float_as_uint(P[0][v].x), __float_as_uint(P[3][v].x)), pack_lo16(__float_as_uint(P[1][v].x), __float_as_uint(P[2][v].y)),
             pack_lo16(__float_as_uint(P[0][v].y), __float_as_uint(P[2][v].x)), pack_lo16(__float_as_uint(P[1][v].y), __float_as_uint(P[3][v].y)));
    return;
  }
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    f32x2 o07, o16, o25, o43;
    aan_inv_h(K, P[0][r], P[1][r], P[2][r], P[3][r], o07, o16, o25, o43);
    // x0..x7 = o07.x o16.x o25.x o43.y o43.x o25.y o16.y o07.y
    const uint32_t w0 = pack4_sat_u8(o07.x, o16.x, o25.x, o43.y), w1 = pack4_sat_u8(o43.x, o25.y, o16.y, o07.y);
    const u32x2_unaligned_g w = {w0, w1};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x2_unaligned_g __attribute__((address_space(1))) *>(sgpr_ptr(static_cast<uint8_t *>(dst) + (size_t)r * pitch_out) + out_off));
  }
}

#ifndef MDCT_U8B_WAVES
#define MDCT_U8B_WAVES 4 // measured 2..6 on the 8K 4:2:0 frame: 27.2 / 26.2 / 25.6 / 25.9 / 25.7 us (profiles/r05_exp_u8_waves.log): bound by vector issue, not occupancy
#endif
#ifndef MDCT_U8B_WAVES_SMALL
#define MDCT_U8B_WAVES_SMALL 4
#endif
#ifndef MDCT_U8B_PRIO
#define MDCT_U8B_PRIO true
#endif
// GENERAL = false: every plane's table is tame (mdct_api.hip: u8_table_is_tame / u8_table_is_bounded): no saturations in the quantiser.
// (The output stage saturates by itself in every build.)  U8_INV is always GENERAL: its coefficients are the caller's, not a transform's.
template <int MODE, bool GENERAL, bool SMALL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SMALL ? MDCT_U8B_WAVES_SMALL : MDCT_U8B_WAVES, SMALL ? MDCT_U8B_WAVES_SMALL : MDCT_U8B_WAVES))) void k_u8_batch(BatchArgs a)
{
  static_assert(MODE != U8_INV || GENERAL, "the inverse clamps its output");
  const BatchTile t = batch_tile<true>(blockIdx.x);
  const uint32_t lane = threadIdx.x;
  if (!t.straddle && lane >= t.s) // (wave-uniform first: a partial last tile drops its lanes)
    return;
  const size_t pin = t.pitch_in(), pout = t.pitch_out();
  const f32x2 shifts = {a.px[0], a.px[1]};
  // a tile = 64 blocks of one block row: 512 bytes of every pixel row, 512 elements of every coefficient row; in a paired plane's straddling
  // tile the lanes >= s continue at block 0 of the next block row: 8 rows down and b0 + s = bpr blocks back
  constexpr uint32_t in_el = MODE == U8_INV ? 2 : 1, out_el = MODE == U8_FWD ? 2 : 1;
  const char *src = (const char *)t.from() + ((size_t)t.row * 8 * pin + (size_t)t.b0 * 8) * in_el;
  char *dst = (char *)t.to() + ((size_t)t.row * 8 * pout + (size_t)t.b0 * 8) * out_el;
  const uint32_t hop = (t.straddle && lane >= t.s) ? 0xFFFFFFFFu : 0u;
  const uint32_t hop_in = ((uint32_t)pin - t.bpr()) * (8 * in_el), hop_out = ((uint32_t)pout - t.bpr()) * (8 * out_el);
  u8_rows<MODE, GENERAL, MDCT_U8B_PRIO>(a.consts, shifts, src, dst, pin, pout, lane * (8 * in_el) + (hop & hop_in),
                                                                                           lane * (8 * out_el) + (hop & hop_out), t.tables);
}

// The reference's primary product (B1, simd_dct.cpp:2064-2262: k_q32_tile above) on a plane batch: any list of separately allocated 8-bit planes,
// each with its own table and its own output buffer, in ONE launch -- a frame's Y, Cb and Cr where the reference's caller makes three calls
// (main.cpp:543).  One wave = one 64-block tile of one block row of one plane (batch_tile); sizeX % 64 == 0 (the reference's own condition,
// :117), so a partial last tile (3840-wide planes: 7.5 tiles per row) is a whole number of 8-block groups: its idle lanes redo the last block
// and their bytes are never read back.  Block row r of a plane lands at r * pitch_out (8 * sizeX when tight), group g of the row at + 512 g
// (:2227-2230).  Tables: OwnTables::qf holds the 64 multipliers 255 / (lut * 0.95) in pair order, negated for the fast quantiser.
// (steered to exactly 3 / 4 / 5 waves per SIMD the frame takes 26.8-27.3 / 25.5-26.0 / 26.7-27.3 us against 24.5-24.8 as below: profiles/r05_exp_q32_batch_waves.log)
#ifdef MDCT_Q32B_WAVES
#define MDCT_Q32B_ATTR __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MDCT_Q32B_WAVES, MDCT_Q32B_WAVES)))
#else
#define MDCT_Q32B_ATTR __launch_bounds__(64, SAFE ? 1 : MDCT_Q32_MINW)
#endif
template <bool SAFE>
__global__ MDCT_Q32B_ATTR void k_q32_batch(BatchArgs a)
{
  __shared__ __attribute__((aligned(16))) uint8_t wl[64 * kQ32RowStride];
  const BatchTile t = batch_tile<true>(blockIdx.x);
  const uint32_t lane = threadIdx.x;
  const uint32_t nb = t.straddle ? 64u : t.s; // blocks of this tile, a multiple of 8 (t.s: those in block row t.row, a multiple of 16 when the tile straddles)
  const size_t pin = t.pitch_in(), pout = t.pitch_out();
  uint32_t q[64];
  {
    uint2 rows[8];
    // a paired plane's straddling tile: lanes >= s continue at block 0 of the next block row, 8 pixel rows down and bpr blocks back
    const uint32_t hop = (t.straddle && lane >= t.s) ? ((uint32_t)pin - t.bpr()) * 8 : 0u;
    load_block_rows_g((const uint8_t *)t.from() + (size_t)t.row * 8 * pin + (size_t)t.b0 * 8, pin, min(lane, nb - 1) * 8 + hop, rows);
    encode_block_avx_pk_t<SAFE>(reinterpret_cast<const PkConsts &>(a.pk), rows, (karg_pairs_t)t.tables, q);
  }
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the tile's (up to) 4 KiB of output: store k, lane l -> coefficients c2 = 2 (l & 31), c2 + 1 of group 2k + (l >> 5) at group * 512 + c2 * 8
  // (store k = the 16 blocks 16k .. 16k + 15: wholly in row t.row or -- straddling tile, s a multiple of 16 -- wholly in the next one, whose strip starts pout further)
  uint8_t *const strip = (uint8_t *)t.to() + (size_t)t.row * pout + (size_t)t.b0 * 64;
  const uint32_t rd = (lane & 31) * (2 * kQ32RowStride) + (lane >> 5) * 8;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if ((2 * k + (lane >> 5)) * 8 < nb)
    {
      const uint2 lo = *reinterpret_cast<const uint2 *>(wl + rd + k * 16);
      const uint2 hi = *reinterpret_cast<const uint2 *>(wl + rd + k * 16 + kQ32RowStride);
      u32x4_g v = {lo.x, lo.y, hi.x, hi.y};
      if constexpr (!SAFE)
        v = ~v; // the fast quantiser staged complemented bytes (encode_block_avx_pk)
      const gptr_t outw = sgpr_ptr((t.straddle && 16u * k >= t.s) ? strip + pout - (size_t)t.bpr() * 64 : strip);
      store16_g(outw + k * 1024 + lane * 16, v);
    }
}

// 8-bit pixels -> int16 coefficients, one plane (the JPEG-style pair's forward half; the inverse of one plane is a batch of one through
// k_u8_batch<U8_INV>, mdct_api.hip: mdct_inv_i16_u8): u8 rows are 8 B per lane (512 B per wave load), int16 rows 16 B per lane.  The level
// shift costs nothing: it is exactly "raw DC minus 64*128" (all other AAN outputs are differences of exact integer sums, so the offset
// cancels bit for bit).
#ifndef MDCT_U8I16_WAVES
#define MDCT_U8I16_WAVES 4 // measured: 34.5 us with 4 (37.8 with 3, 36.4 with 6) at 8192^2
#endif
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(MDCT_U8I16_WAVES, MDCT_U8I16_WAVES))) void k_u8_i16_fwd(U8I16Args a)
{
  const uint32_t t = wg_index() * kWG + threadIdx.x;
  if (t >= a.nblocks)
    return;
  const uint32_t row = t / a.bpr;
  const uint32_t bx = t - row * a.bpr;
  const size_t by = a.by0 + row;
  const DctConsts &C = a.consts;
  float b[8][8];
  const uint8_t *src = a.px + by * 8 * a.pitch_px + (size_t)bx * 8;
  int16_t *dst = a.coef + by * 8 * a.pitch_coef + (size_t)bx * 8;
  uint2 rows[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    rows[r] = load8(src + (size_t)r * a.pitch_px);
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    b[r][0] = ubyte_to_float<0>(rows[r].x); b[r][1] = ubyte_to_float<1>(rows[r].x);
    b[r][2] = ubyte_to_float<2>(rows[r].x); b[r][3] = ubyte_to_float<3>(rows[r].x);
    b[r][4] = ubyte_to_float<0>(rows[r].y); b[r][5] = ubyte_to_float<1>(rows[r].y);
    b[r][6] = ubyte_to_float<2>(rows[r].y); b[r][7] = ubyte_to_float<3>(rows[r].y);
  }
  raw_fwd(C, b);
  b[0][0] = b[0][0] - a.dc_shift; // 8192 or 0
#pragma unroll
  for (int r = 0; r < 8; r++)
    store_q_i16x8(C, dst + (size_t)r * a.pitch_coef, b[r], a.tb.qf + r * 8);
}

// Pixels (or an int16 plane, I16_IN) -> records in one pass (the encoder's front half): the forward
// transform and quantiser of k_u8_i16_fwd (k_i16<MODE_FWD>), then -- instead of storing the int16 plane and reading it back -- the zig-zag scan and
// run/level compaction of k_scan (scan_records.h) on the values still in registers.  1 B/px in, 3 B/px out
// (k_u8_i16_fwd + k_scan move 3 + 5).  Bit for bit the records mdct_fwd_u8_i16 + mdct_zigzag_rle_i16 produce.
// One wave per workgroup: every wave works alone on 64 consecutive blocks (lane = block).
// CLAMP = false (8-bit pixels, every table entry >= 1/16 in magnitude or no table, decided on the host): |coefficient| <= 8 * 255, so the
// quantised value cannot leave int16 and the 64 saturations per block are left out.
template <bool I16_IN, bool CLAMP = true>
__global__ __launch_bounds__(64) void k_u8_records(U8RecArgs a)
{
  __shared__ __attribute__((aligned(16))) uint8_t lv[64 * kLvRow];
  __shared__ __attribute__((aligned(16))) uint8_t rn[64 * kRnRow];
  const uint32_t lane = threadIdx.x;
  const uint32_t wave_t0 = blockIdx.x * 64; // first block of the wave within the launch
  const uint32_t nvalid = min(64u, a.nblocks - wave_t0);
  const bool valid = lane < nvalid;
  const uint32_t t = wave_t0 + (valid ? lane : 0);
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const DctConsts &C = a.consts;
  // the forward transform on packed fp32 (aan_fwd_h / aan_fwd_v: the same individually rounded operations as raw_fwd);
  // a.tb.qf is in the pair order of the column pass, (j*8 + v)*2 + {0,1} = coefficient (v, A[j]) / (v, B[j]).
  // Rows are converted as they are consumed: the raw rows (16 / 32 registers) stay live, not 64 floats.
  const AanPk &K = reinterpret_cast<const AanPk &>(C);
  f32x2 P[4][8];
  if constexpr (I16_IN)
  { // an int16 plane (pitch in elements): the input of mdct_fwd_i16
    const int16_t *src = reinterpret_cast<const int16_t *>(a.px) + (size_t)(a.by0 + row) * 8 * a.pitch_px + (size_t)bx * 8;
    uint4 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = ld_stream16(src + (size_t)r * a.pitch_px);
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const f32x2 a01 = {(float)(int16_t)(rows[r].x & 0xFFFF), (float)(int16_t)(rows[r].x >> 16)}, a23 = {(float)(int16_t)(rows[r].y & 0xFFFF), (float)(int16_t)(rows[r].y >> 16)};
      const f32x2 a45 = {(float)(int16_t)(rows[r].z & 0xFFFF), (float)(int16_t)(rows[r].z >> 16)}, a67 = {(float)(int16_t)(rows[r].w & 0xFFFF), (float)(int16_t)(rows[r].w >> 16)};
      aan_fwd_h(K, a01, a23, a45, a67, P[0][r], P[1][r], P[2][r], P[3][r]);
    }
  }
  else
  {
    const uint8_t *src = a.px + (size_t)(a.by0 + row) * 8 * a.pitch_px + (size_t)bx * 8;
    uint2 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
      rows[r] = load8(src + (size_t)r * a.pitch_px);
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const f32x2 a01 = {ubyte_to_float<0>(rows[r].x), ubyte_to_float<1>(rows[r].x)}, a23 = {ubyte_to_float<2>(rows[r].x), ubyte_to_float<3>(rows[r].x)};
      const f32x2 a45 = {ubyte_to_float<0>(rows[r].y), ubyte_to_float<1>(rows[r].y)}, a67 = {ubyte_to_float<2>(rows[r].y), ubyte_to_float<3>(rows[r].y)};
      aan_fwd_h(K, a01, a23, a45, a67, P[0][r], P[1][r], P[2][r], P[3][r]);
    }
  }
  int val[64];
  constexpr int kA[4] = {0, 2, 5, 1}, kB[4] = {4, 6, 3, 7};
  const f32x2 magic_v = K.magic;
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    aan_fwd_v(K, P[j]);
    if (j == 0)
      P[0][0].x = P[0][0].x - a.dc_shift; // the level shift is exactly "raw DC minus 64 * 128"
#pragma unroll
    for (int v = 0; v < 8; v++)
    { // the int16 the plane would have held (quant_i16_bits), sign-extended
      f32x2 t;
      MDCT_PKF(t, P[j][v], reinterpret_cast<const f32x2 *>(a.tb.qf)[j * 8 + v], magic_v, "op_sel:[0,0,0] op_sel_hi:[1,1,0]");
      if constexpr (CLAMP)
      {
        t.x = __builtin_amdgcn_fmed3f(t.x, kQLo, kQHi);
        t.y = __builtin_amdgcn_fmed3f(t.y, kQLo, kQHi);
      }
      val[v * 8 + kA[j]] = (int)(int16_t)(__float_as_uint(t.x) & 0xFFFFu);
      val[v * 8 + kB[j]] = (int)(int16_t)(__float_as_uint(t.y) & 0xFFFFu);
    }
  }
  scan_emit<true>(val, lv, rn, lane, nvalid, valid, (size_t)a.by0 * a.bpr + wave_t0, a.levels, a.runs, a.counts);
}

// ---------------------------------------------------------------------------------------
// Pixels (or an int16 plane) -> baseline Huffman rows in ONE kernel: the front half of
// k_u8_records (forward transform + quantiser on packed fp32), the zig-zag run/level compaction straight into the
// workgroup's LDS -- 16-bit entries run << 12 | level, 132 bytes per block (HuffRowCoder16, huffman_rows.h: the first version
// kept k_huffman_rows' dword pairs, 260 bytes per block = 2 waves/SIMD, and was slower than the two stages it replaces), a lane
// writes and later reads only its own row -- and the chunk coder of k_huffman_rows.  The 3 B/px of records that k_u8_records writes and
// k_huffman_rows reads back (two thirds of them padding) never exist: 1 B/px in, ~0.2 B/px out.  One workgroup of WAVES
// waves per block row (= restart interval); the next chunk's pixel rows are loaded while the current one is coded.
// A wave's first block needs the previous wave's last DC as predictor, which is computed in the same pass: LATE_DC.
// Byte for byte the segments of mdct_fwd_u8_records + mdct_huffman_rows.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kFusedRing = 1024; // words of bit stream held in LDS (2 bit/px over a 256-block chunk; denser chunks take several windows)
// PACK (4 waves): the finished row goes on into the contiguous scan in the same launch -- it publishes its stuffed length,
// waits until the rows before it have published theirs (pack_rows.h: chain_*), and copies its own segment (still in L2)
// to its place, stuffed and followed by its restart marker: pixels -> decodable scan, one kernel.
// CLAMP = false (8-bit pixels and a table whose entries are all >= 1.01, decided on the host): no level can leave +-1023 -- an AC
// coefficient of 8-bit pixels is at most 8 * 128 in magnitude, the DC 8 * 255 -- so the 64 saturations per block are left out.
template <bool I16_IN, int WAVES, bool PACK = false, bool CLAMP = true>
__global__ __launch_bounds__(64 * WAVES) void k_px_huffman_rows(PxHuffArgs a)
{
  static_assert(CLAMP || !I16_IN, "an int16 plane can hold anything");
  static_assert(!PACK || WAVES == 4, "the packing tail is written for 256 threads");
  __shared__ uint32_t ac[256], dc[12];
  __shared__ __attribute__((aligned(16))) uint16_t rec_all[WAVES][64 * kRec16Row]; // PACK: later the staging buffer of the copy
  static_assert(!PACK || sizeof(rec_all) >= kPackStageWords * 4, "the packing tail stages a window in the record rows");
  __shared__ uint32_t ring[kFusedRing];
  __shared__ uint32_t tot[2][WAVES];
  __shared__ int dcx[2][2][WAVES];
  __shared__ uint32_t ff_total;
  constexpr uint32_t kChunk = 64 * WAVES;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t row = a.by0 + blockIdx.x;
  uint32_t tag = 0; // PACK: the chain's epoch, read now so that nobody waits for it later
  if constexpr (PACK)
  {
    if (chain_failed(a.work))
    { // an earlier launch on this work array failed: nothing is coded or published until the caller has zeroed it again
      if (tid == 0)
      {
        a.row_off[blockIdx.x] = ~0ull;
        if (blockIdx.x + 1 == a.n_rows)
          a.row_off[a.n_rows] = ~0ull;
      }
      return;
    }
    tag = __builtin_amdgcn_readfirstlane(chain_epoch_tag(a.work));
  }
  for (uint32_t i = tid; i < 256; i += kChunk)
    ac[i] = a.ac[i];
  if (tid < 12)
    dc[tid] = a.dc[tid];
  if (tid == 0)
    ff_total = 0;
  for (uint32_t w = tid; w < kFusedRing; w += kChunk)
    ring[w] = 0;
  HuffRowCoder16<WAVES, kFusedRing> coder;
  coder.ac = ac;
  coder.dc = dc;
  coder.ring = ring;
  coder.tot = tot;
  coder.dcx = dcx;
  coder.out_w = reinterpret_cast<uint32_t *>(a.out + (size_t)row * a.seg_stride);
  coder.eob = a.ac[0x00];
  coder.bpr = a.bpr;
  const DctConsts &C = a.consts;
  const AanPk &K = reinterpret_cast<const AanPk &>(C);
  uint16_t *rec = rec_all[wave] + lane * kRec16Row;
  const uint32_t last_blk = a.bpr - 1;

  typedef typename std::conditional<I16_IN, uint4, uint2>::type row_t;
  row_t rows[8];
  auto fetch = [&](uint32_t bx) { // the 8 rows of block min(bx, last) of this block row (lanes past the row's end redo the last block)
    const uint32_t blk = min(bx, last_blk);
    if constexpr (I16_IN)
    {
      const int16_t *src = reinterpret_cast<const int16_t *>(a.px) + (size_t)row * 8 * a.pitch_px + (size_t)blk * 8;
#pragma unroll
      for (int r = 0; r < 8; r++)
        rows[r] = ld_stream16(src + (size_t)r * a.pitch_px);
    }
    else
    {
      const uint8_t *src = a.px + (size_t)row * 8 * a.pitch_px + (size_t)blk * 8;
#pragma unroll
      for (int r = 0; r < 8; r++)
        rows[r] = load8(src + (size_t)r * a.pitch_px);
    }
  };
  fetch(wave * 64 + lane);
  wg_sync(); // tables and the cleared ring
  for (uint32_t c0 = 0; c0 < a.bpr; c0 += kChunk)
  {
    const uint32_t bx = c0 + wave * 64 + lane;
    const bool live = bx < a.bpr;
    // ---- forward transform and quantiser (as k_u8_records)
    f32x2 P[4][8];
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      f32x2 a01, a23, a45, a67;
      if constexpr (I16_IN)
      {
        a01 = f32x2{(float)(int16_t)(rows[r].x & 0xFFFF), (float)(int16_t)(rows[r].x >> 16)};
        a23 = f32x2{(float)(int16_t)(rows[r].y & 0xFFFF), (float)(int16_t)(rows[r].y >> 16)};
        a45 = f32x2{(float)(int16_t)(rows[r].z & 0xFFFF), (float)(int16_t)(rows[r].z >> 16)};
        a67 = f32x2{(float)(int16_t)(rows[r].w & 0xFFFF), (float)(int16_t)(rows[r].w >> 16)};
      }
      else
      {
        a01 = f32x2{ubyte_to_float<0>(rows[r].x), ubyte_to_float<1>(rows[r].x)};
        a23 = f32x2{ubyte_to_float<2>(rows[r].x), ubyte_to_float<3>(rows[r].x)};
        a45 = f32x2{ubyte_to_float<0>(rows[r].y), ubyte_to_float<1>(rows[r].y)};
        a67 = f32x2{ubyte_to_float<2>(rows[r].y), ubyte_to_float<3>(rows[r].y)};
      }
      aan_fwd_h(K, a01, a23, a45, a67, P[0][r], P[1][r], P[2][r], P[3][r]);
    }
    // quantised levels as the low 16 bits of val[]: the DC like the int16 plane would hold it (sat_i16), the AC coefficients
    // saturated to the +-1023 of baseline categories 1..10 -- exactly what the staged coder does to an int16 record at token time
    uint32_t val[64];
    constexpr int kA[4] = {0, 2, 5, 1}, kB[4] = {4, 6, 3, 7};
    const f32x2 magic_v = K.magic;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      aan_fwd_v(K, P[j]);
      if (j == 0)
        P[0][0].x = P[0][0].x - a.dc_shift; // the level shift is exactly "raw DC minus 64 * 128"
      // the 8 multiplier pairs of column pair j, fetched here (all 64 multipliers held in SGPRs from the top of the kernel spilled)
      karg_pairs_t tq = karg_pairs(offsetof(PxHuffArgs, tb) + offsetof(OwnTables, qf)) + j * 8;
      asm volatile("" : "+s"(tq));
#pragma unroll
      for (int v = 0; v < 8; v++)
      {
        f32x2 c;
        MDCT_PKF(c, P[j][v], tq[v], magic_v, "op_sel:[0,0,0] op_sel_hi:[1,1,0]"); // quant_i16_bits; the clamps (baseline JPEG: AC to +-1023) on the biased value
        const bool is_dc = j == 0 && v == 0;
        if constexpr (CLAMP)
          c = f32x2{__builtin_amdgcn_fmed3f(c.x, is_dc ? kQLo : kMagic23 - 1023.0f, is_dc ? kQHi : kMagic23 + 1023.0f), __builtin_amdgcn_fmed3f(c.y, kMagic23 - 1023.0f, kMagic23 + 1023.0f)};
        val[v * 8 + kA[j]] = __float_as_uint(c.x);
        val[v * 8 + kB[j]] = __float_as_uint(c.y);
      }
    }
    // ---- zig-zag order; AC entries run << 12 | level compacted to the front of the lane's LDS row, a ZRL entry for every
    // 16 zeros in a row.  EVERY coefficient writes run << 12 | level at the current position and only those that need an entry
    // advance it: a zero's write is overwritten by the next entry -- unless it is the 16th zero in a row, and then what it
    // wrote, 15 << 12 | 0, IS the ZRL entry.  (7 vector instructions per coefficient; selecting value and address per
    // coefficient and tracking the last non-zero took 11.)
    uint32_t pos = 0, r12 = 0;
#pragma unroll
    for (int k = 1; k < 64; k++)
    {
      const uint32_t v = val[kZigZag[k]];
      const bool nz = (v & 0xFFFFu) != 0;
      const bool wr = nz || r12 == 0xF000u;
      rec[pos] = (uint16_t)((v & 0xFFFu) | r12);
      pos += wr ? 1u : 0u;
      r12 = wr ? 0u : r12 + 0x1000u;
    }
    // ZRL entries after the last coefficient are dropped (at most three: 62 zeros)
    uint32_t n = pos;
#pragma unroll
    for (int t = 0; t < 3; t++)
      n -= (n > 0 && rec[n - 1] == 0xF000u) ? 1u : 0u;
    const int my_dc = (int)(int16_t)(val[0] & 0xFFFFu);
    const bool need_eob = (val[kZigZag[63]] & 0xFFFFu) == 0;
    if (c0 + kChunk < a.bpr)
      fetch(bx + kChunk); // in flight while this chunk is coded
    coder.chunk(c0, rec, (int)n, live, my_dc, need_eob);
  }
  if ((PACK || a.ff_counts) && coder.ff)
    atomicAdd(&ff_total, coder.ff);
  wg_sync();
  if constexpr (!PACK)
  {
    if (tid == 0)
    {
      uint32_t ff_last;
      a.seg_bytes[row] = coder.finish(&ff_last);
      if (a.ff_counts)
        a.ff_counts[row] = ff_total + ff_last;
    }
  }
  else
  {
    __shared__ uint32_t row_info[2]; // bytes, 0xFF bytes
    __shared__ uint32_t wave_tot[4];
    __shared__ unsigned long long wave_sum[4];
    const uint32_t r = blockIdx.x;
    const bool marker = r + 1 < a.n_rows;
    if (tid == 0)
    {
      uint32_t ff_last;
      const uint32_t nbytes = coder.finish(&ff_last), ff = ff_total + ff_last;
      if (a.seg_bytes)
        a.seg_bytes[row] = nbytes;
      if (a.ff_counts)
        a.ff_counts[row] = ff;
      chain_publish(a.work, r, tag, nbytes + ff + (marker ? 2u : 0u));
      row_info[0] = nbytes;
      row_info[1] = ff;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores of the segment have reached L2 ...
    wg_sync();                                  // ... and so have everybody's: the copy below reads them back from there
    const uint32_t nbytes = row_info[0];
    PackRow<true> prow;
    prow.begin(a.out + (size_t)row * a.seg_stride, a.seg_stride, nbytes); // on its way while the chain is consulted
    const unsigned long long len = (unsigned long long)nbytes + row_info[1] + (marker ? 2u : 0u);
    bool ok;
    const unsigned long long base = chain_base(a.work, r, tag, wave_sum, ok);
    if (tid == 0)
    {
      if (!ok)
        __hip_atomic_fetch_or(a.work + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sticky: this launch and every later one fail visibly
      a.row_off[r] = ok ? base : ~0ull;
      if (marker)
        chain_row_decided(a.work); // AFTER the failure bit: the last row reads the verdicts of all rows, not just their lengths
      else
      { // the last row: every other row has published (chain_base) -- now wait until every one of them has also DECIDED, so that a row which
        // gave up between publishing and setting the failure bit cannot be missed; then fold the verdict into row_off[n_rows]
        const bool failed = !ok || !chain_all_decided(a.work, a.n_rows - 1);
        if (!failed)
          chain_next_epoch(a.work, tag); // (a failed launch keeps its epoch: its stragglers must not look like the next launch's rows)
        a.row_off[a.n_rows] = failed ? ~0ull : base + len;
      }
    }
    if (ok && base + len <= a.capacity) // a row that does not fit is not written: the caller sees row_off[n_rows] > capacity
      prow.finish(a.scan + base, marker, (a.first_rst + r) & 7, reinterpret_cast<uint32_t *>(&rec_all[0][0]), wave_tot);
  }
}

// float32 rows are 32 B per block: if every lane fetched its own 2 x 16 B, each wave load would
// touch 64 x 16 B at a 32-byte stride (half of every cache line per instruction).  When the
// wave's 64 blocks are one contiguous 2 KiB row segment (plane width % 512 == 0) the WIDE form
// instead issues two fully contiguous 1 KiB loads per row -- lane l takes floats [4l, 4l+4) of
// each KiB, i.e. half a row of block l/2 resp. 32 + l/2 -- and lane pairs swap halves with one
// DPP quad_perm [1,0,3,2] per register.  Even lane 2k ends up owning block k, odd lane 2k+1
// block 32+k of the wave; stores mirror this.
__device__ __forceinline__ float swap_pair(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// scheduling steered like the int16 kernels (measured: 89.7 us with 2, 97.7 unsteered, 93-107 with 3..6)
#ifndef MDCT_F32_WAVES
#define MDCT_F32_WAVES 2
#endif
#define MDCT_F32_ATTR __attribute__((amdgpu_waves_per_eu(MDCT_F32_WAVES, MDCT_F32_WAVES)))
template <int MODE, bool WIDE>
__global__ __launch_bounds__(kWG) MDCT_F32_ATTR void k_f32(F32Args a)
{
  const uint32_t t = wg_index() * kWG + threadIdx.x;
  if (t >= a.nblocks)
    return;
  const DctConsts &C = a.consts;
  const uint32_t lane = threadIdx.x & 63;
  const bool odd = lane & 1;
  float b[8][8];
  const float *src;
  float *dst;

  if constexpr (WIDE)
  {
    const uint32_t t0 = t - lane; // the wave's first block; all 64 blocks share a block row
    const uint32_t row = t0 / a.bpr;
    const uint32_t bx0 = t0 - row * a.bpr;
    const size_t by = a.by0 + row;
    src = a.from + by * 8 * a.pitch_in + (size_t)bx0 * 8 + lane * 4;
    dst = a.to + by * 8 * a.pitch_out + (size_t)bx0 * 8 + lane * 4;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const f32x4 A = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + (size_t)r * a.pitch_in));
      const f32x4 B = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + (size_t)r * a.pitch_in + 256));
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const float got = swap_pair(odd ? A[j] : B[j]); // even lanes receive A (right half), odd lanes B (left half)
        b[r][j] = odd ? got : A[j];
        b[r][4 + j] = odd ? B[j] : got;
      }
    }
  }
  else
  {
    const uint32_t row = t / a.bpr;
    const uint32_t bx = t - row * a.bpr;
    const size_t by = a.by0 + row;
    src = a.from + by * 8 * a.pitch_in + (size_t)bx * 8;
    dst = a.to + by * 8 * a.pitch_out + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const f32x4 lo = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + (size_t)r * a.pitch_in));
      const f32x4 hi = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + (size_t)r * a.pitch_in + 4));
      b[r][0] = lo.x; b[r][1] = lo.y; b[r][2] = lo.z; b[r][3] = lo.w;
      b[r][4] = hi.x; b[r][5] = hi.y; b[r][6] = hi.z; b[r][7] = hi.w;
    }
  }

  if constexpr (MODE == MODE_FWD)
  {
    raw_fwd(C, b);
#pragma unroll
    for (int i = 0; i < 64; i++)
      b[i >> 3][i & 7] = b[i >> 3][i & 7] * a.scale[i];
  }
  else
  {
#pragma unroll
    for (int i = 0; i < 64; i++)
      b[i >> 3][i & 7] = b[i >> 3][i & 7] * a.scale[i];
    raw_inv(C, b);
  }

#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    if constexpr (WIDE)
    {
      f32x4 A, B;
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const float got = swap_pair(odd ? b[r][j] : b[r][4 + j]); // even sends its right half, odd its left
        A[j] = odd ? got : b[r][j];
        B[j] = odd ? b[r][4 + j] : got;
      }
      __builtin_nontemporal_store(A, reinterpret_cast<f32x4 *>(dst + (size_t)r * a.pitch_out));
      __builtin_nontemporal_store(B, reinterpret_cast<f32x4 *>(dst + (size_t)r * a.pitch_out + 256));
    }
    else
    {
      const f32x4 lo = {b[r][0], b[r][1], b[r][2], b[r][3]};
      const f32x4 hi = {b[r][4], b[r][5], b[r][6], b[r][7]};
      __builtin_nontemporal_store(lo, reinterpret_cast<f32x4 *>(dst + (size_t)r * a.pitch_out));
      __builtin_nontemporal_store(hi, reinterpret_cast<f32x4 *>(dst + (size_t)r * a.pitch_out + 4));
    }
  }
}

// The WIDE form as one-wave tiles on a 2-D grid (x = 64-block tile of the row, y = block row; plane width % 512 == 0):
// wave-uniform row bases in SGPRs, the lane adds (lane * 16) bytes for each of the row's two contiguous 1 KiB halves.
#ifndef MDCT_F32_TILE_WAVES
#define MDCT_F32_TILE_WAVES 2
#endif
template <int MODE, bool PRIO = false>
__device__ __forceinline__ void f32_tile_body(const F32Args &a);
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MDCT_F32_TILE_WAVES, MDCT_F32_TILE_WAVES))) void k_f32_tile(F32Args a)
{
  f32_tile_body<MODE>(a);
}
template <int MODE, bool PRIO>
__device__ __forceinline__ void f32_tile_body(const F32Args &a)
{
  typedef float f32x4_g __attribute__((ext_vector_type(4)));
  const DctConsts &C = a.consts;
  const uint32_t lane = threadIdx.x;
  const bool odd = lane & 1;
  const size_t by = a.by0 + blockIdx.y;
  const float *src = a.from + by * 8 * a.pitch_in + (size_t)blockIdx.x * 512; // wave-uniform: 64 blocks x 8 floats
  float *dst = a.to + by * 8 * a.pitch_out + (size_t)blockIdx.x * 512;
  float b[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const gptr_t rp = sgpr_ptr(src + (size_t)r * a.pitch_in);
    const f32x4_g A = __builtin_nontemporal_load(reinterpret_cast<const f32x4_g __attribute__((address_space(1))) *>(rp + lane * 16));
    const f32x4_g B = __builtin_nontemporal_load(reinterpret_cast<const f32x4_g __attribute__((address_space(1))) *>(rp + lane * 16 + 1024));
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const float got = swap_pair(odd ? A[j] : B[j]); // even lanes receive A (right half), odd lanes B (left half)
      b[r][j] = odd ? got : A[j];
      b[r][4 + j] = odd ? B[j] : got;
    }
  }
  MDCT_PHASE_PRIO(2);
  if constexpr (MODE == MODE_FWD)
  {
    raw_fwd(C, b);
#pragma unroll
    for (int i = 0; i < 64; i++)
      b[i >> 3][i & 7] = b[i >> 3][i & 7] * a.scale[i];
  }
  else
  {
#pragma unroll
    for (int i = 0; i < 64; i++)
      b[i >> 3][i & 7] = b[i >> 3][i & 7] * a.scale[i];
    raw_inv(C, b);
  }
  MDCT_PHASE_PRIO(3);
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    f32x4_g A, B;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const float got = swap_pair(odd ? b[r][j] : b[r][4 + j]); // even sends its right half, odd its left
      A[j] = odd ? got : b[r][j];
      B[j] = odd ? b[r][4 + j] : got;
    }
    const gptr_t rp = sgpr_ptr(dst + (size_t)r * a.pitch_out);
    __builtin_nontemporal_store(A, reinterpret_cast<f32x4_g __attribute__((address_space(1))) *>(rp + lane * 16));
    __builtin_nontemporal_store(B, reinterpret_cast<f32x4_g __attribute__((address_space(1))) *>(rp + lane * 16 + 1024));
  }
}

// Parks one quantiser table in device memory (mdct_api.hip: the table cache): the 512 bytes arrive in the argument segment and one
// wave writes them to the slot, 8 bytes per lane -- an upload that is ordered on the stream like any launch and needs no host buffer.
__global__ __launch_bounds__(64) void k_park_table(OwnTables tb, OwnTables *slot)
{
  static_assert(sizeof(OwnTables) == 64 * sizeof(uint2), "one uint2 per lane");
  reinterpret_cast<uint2 *>(slot)[threadIdx.x] = reinterpret_cast<const uint2 *>(&tb)[threadIdx.x];
}

// Shader-clock probe (diagnostics): one wave per XCD-sized grid slot spins for `ticks` of the constant 100 MHz counter and reports how many
// shader cycles (s_memtime) passed meanwhile -- launched on a second stream beside a workload it tells at which clock the chip runs that
// workload (bench.py turns instruction counts and measured issue cycles into a time with it: valu_floor_ms).
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out, unsigned int ticks)
{
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks) // bounded: the 100 MHz counter always advances
  {
    __builtin_amdgcn_s_sleep(4);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0)
  {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// read-N / write-N stream copy, 8 x 16 B per lane, non-temporal: the box's measured HBM roofline
// (tools/membench: this shape is the fastest of those tried, ~6.2 TB/s).
constexpr int kCopyUnroll = 8;
__global__ __launch_bounds__(kWG) void k_stream_copy(const u32x4 *__restrict__ from, u32x4 *__restrict__ to, size_t n16)
{
  const size_t base = (size_t)wg_index() * kWG * kCopyUnroll + threadIdx.x;
  u32x4 v[kCopyUnroll];
  if (base + (size_t)(kCopyUnroll - 1) * kWG < n16)
  {
#pragma unroll
    for (int u = 0; u < kCopyUnroll; u++)
      v[u] = __builtin_nontemporal_load(from + base + (size_t)u * kWG);
#pragma unroll
    for (int u = 0; u < kCopyUnroll; u++)
      __builtin_nontemporal_store(v[u], to + base + (size_t)u * kWG);
  }
  else
  {
    for (int u = 0; u < kCopyUnroll; u++)
      if (base + (size_t)u * kWG < n16)
        to[base + (size_t)u * kWG] = from[base + (size_t)u * kWG];
  }
}

// ---------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t nblocks) { return (nblocks + kWG - 1) / kWG; }

template <int PROFILE, int LAYOUT>
static hipError_t launch_u8_pl(const U8Args &a, bool safe, hipStream_t s)
{
  const uint32_t launch_rows = a.nblocks / a.bpr; // rows of blocks in the launch: (block row, eye) pairs for STEREO
  if (a.bpr % kWG == 0 && launch_rows <= 65535u && a.out_tight)
  { // every workgroup inside one row of blocks: 2-D grid, wave-uniform addressing (TILED)
    const dim3 g(a.bpr / kWG, launch_rows);
    if (safe)
      hipLaunchKernelGGL((k_fwd_quant_u8<PROFILE, LAYOUT, true, true>), g, dim3(kWG), 0, s, a);
    else
      hipLaunchKernelGGL((k_fwd_quant_u8<PROFILE, LAYOUT, false, true>), g, dim3(kWG), 0, s, a);
    return hipGetLastError();
  }
  if (safe)
    hipLaunchKernelGGL((k_fwd_quant_u8<PROFILE, LAYOUT, true, false>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  else
    hipLaunchKernelGGL((k_fwd_quant_u8<PROFILE, LAYOUT, false, false>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_fwd_quant_u8(const U8Args &a, int layout, int profile, bool safe, hipStream_t s)
{
  if (a.nblocks == 0)
    return hipSuccess;
  if (layout == MDCT_LAYOUT_Q32 && profile == MDCT_PROFILE_REF_AVX)
  {
    const bool general = a.nblocks % 64 != 0 || !a.out_tight;
    const uint32_t launch_rows = a.nblocks / a.bpr;
    if (!safe && !general && a.bpr % 64 == 0 && launch_rows <= 65535u)
    { // every wave inside one block row: the tile kernel with wave-uniform addressing
      hipLaunchKernelGGL(k_q32_tile, dim3(a.bpr / 64, launch_rows), dim3(64), 0, s, a);
      return hipGetLastError();
    }
    const dim3 g(grid_for(a.nblocks)), b(kWG);
    if (safe && general)
      hipLaunchKernelGGL((k_q32_avx<true, true>), g, b, 0, s, a);
    else if (safe)
      hipLaunchKernelGGL((k_q32_avx<true, false>), g, b, 0, s, a);
    else if (general)
      hipLaunchKernelGGL((k_q32_avx<false, true>), g, b, 0, s, a);
    else
      hipLaunchKernelGGL((k_q32_avx<false, false>), g, b, 0, s, a);
    return hipGetLastError();
  }
  if (layout == MDCT_LAYOUT_STEREO && profile == MDCT_PROFILE_REF_SSE)
    return launch_u8_pl<MDCT_PROFILE_REF_SSE, MDCT_LAYOUT_STEREO>(a, safe, s);
  if (layout == MDCT_LAYOUT_STEREO && profile == MDCT_PROFILE_REF_SCALAR)
    return launch_u8_pl<MDCT_PROFILE_REF_SCALAR, MDCT_LAYOUT_STEREO>(a, false, s);
  if (layout == MDCT_LAYOUT_BLOCK && profile == MDCT_PROFILE_REF_SCALAR)
    return launch_u8_pl<MDCT_PROFILE_REF_SCALAR, MDCT_LAYOUT_BLOCK>(a, false, s);
  if (layout == MDCT_LAYOUT_BLOCK_SSE && profile == MDCT_PROFILE_REF_SSE)
    return launch_u8_pl<MDCT_PROFILE_REF_SSE, MDCT_LAYOUT_BLOCK_SSE>(a, safe, s);
  return hipErrorInvalidValue;
}

#ifndef MDCT_I16_TILED
#define MDCT_I16_TILED 1
#endif
template <int MODE>
static hipError_t launch_i16_m(const I16Args &a, bool has_lut, bool lut_bounded, hipStream_t s)
{
  const uint32_t launch_rows = a.nblocks / a.bpr;
  constexpr bool RT = MODE == MODE_ROUNDTRIP; // the only mode with a build without saturations
  if (MDCT_I16_TILED && a.bpr % 64 == 0 && launch_rows <= 65535u)
  {
    const dim3 g(a.bpr / 64, launch_rows);
    const uint64_t tiles = (uint64_t)g.x * g.y;
    const bool small = tiles > 2048 && tiles <= kTileSmallLaunch; // (up to 2048 tiles the chip holds the whole launch at 2 waves per SIMD already)
    if (has_lut && RT && lut_bounded)
    {
      if (small)
        hipLaunchKernelGGL((k_i16_tile<MODE, true, !RT, 4>), g, dim3(64), 0, s, a);
      else
        hipLaunchKernelGGL((k_i16_tile<MODE, true, !RT, 2>), g, dim3(64), 0, s, a);
    }
    else if (has_lut)
    {
      if (small)
        hipLaunchKernelGGL((k_i16_tile<MODE, true, true, 4>), g, dim3(64), 0, s, a);
      else
        hipLaunchKernelGGL((k_i16_tile<MODE, true, true, 2>), g, dim3(64), 0, s, a);
    }
    else if (small)
      hipLaunchKernelGGL((k_i16_tile<MODE, false, true, 4>), g, dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL((k_i16_tile<MODE, false, true, 2>), g, dim3(64), 0, s, a);
    return hipGetLastError();
  }
  if (has_lut && RT && lut_bounded)
    hipLaunchKernelGGL((k_i16<MODE, true, !RT>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  else if (has_lut)
    hipLaunchKernelGGL((k_i16<MODE, true>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  else
    hipLaunchKernelGGL((k_i16<MODE, false>), dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_i16(const I16Args &a, int mode, bool has_lut, hipStream_t s, bool lut_bounded)
{
  if (a.nblocks == 0)
    return hipSuccess;
  switch (mode)
  {
  case MODE_FWD: return launch_i16_m<MODE_FWD>(a, has_lut, lut_bounded, s);
  case MODE_INV: return launch_i16_m<MODE_INV>(a, has_lut, lut_bounded, s);
  case MODE_ROUNDTRIP: return launch_i16_m<MODE_ROUNDTRIP>(a, has_lut, lut_bounded, s);
  }
  return hipErrorInvalidValue;
}

template <bool SMALL>
static hipError_t launch_i16_batch_w(const BatchArgs &a, uint32_t total, int mode, int lutmode, bool sat, hipStream_t s)
{
  const dim3 g(total), b(64);
  if (mode == MODE_FWD)
    hipLaunchKernelGGL((k_i16_batch<MODE_FWD, BATCH_ALL_LUT, true, SMALL>), g, b, 0, s, a);
  else if (mode == MODE_INV)
    hipLaunchKernelGGL((k_i16_batch<MODE_INV, BATCH_ALL_LUT, true, SMALL>), g, b, 0, s, a);
  else if (mode != MODE_ROUNDTRIP)
    return hipErrorInvalidValue;
  else if (lutmode == BATCH_NO_LUT)
    hipLaunchKernelGGL((k_i16_batch<MODE_ROUNDTRIP, BATCH_NO_LUT, true, SMALL>), g, b, 0, s, a);
  else if (lutmode == BATCH_ALL_LUT && !sat)
    hipLaunchKernelGGL((k_i16_batch<MODE_ROUNDTRIP, BATCH_ALL_LUT, false, SMALL>), g, b, 0, s, a);
  else if (lutmode == BATCH_ALL_LUT)
    hipLaunchKernelGGL((k_i16_batch<MODE_ROUNDTRIP, BATCH_ALL_LUT, true, SMALL>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_i16_batch<MODE_ROUNDTRIP, BATCH_MIXED, true, SMALL>), g, b, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_i16_batch(const BatchArgs &a, uint32_t total, int mode, int lutmode, bool sat, hipStream_t s)
{
  if (total == 0)
    return hipSuccess;
  return total > 2048 && total <= kTileSmallLaunch ? launch_i16_batch_w<true>(a, total, mode, lutmode, sat, s) : launch_i16_batch_w<false>(a, total, mode, lutmode, sat, s);
}

template <int MODE, bool GENERAL>
static hipError_t launch_u8_batch_m(const BatchArgs &a, uint32_t total, hipStream_t s)
{
  const dim3 g(total), b(64);
  if (total > 2048 && total <= kTileSmallLaunch)
    hipLaunchKernelGGL((k_u8_batch<MODE, GENERAL, true>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_u8_batch<MODE, GENERAL, false>), g, b, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_q32_batch(const BatchArgs &a, uint32_t total, bool safe, hipStream_t s)
{
  if (total == 0)
    return hipSuccess;
  if (safe)
    hipLaunchKernelGGL(k_q32_batch<true>, dim3(total), dim3(64), 0, s, a);
  else
    hipLaunchKernelGGL(k_q32_batch<false>, dim3(total), dim3(64), 0, s, a);
  return hipGetLastError();
}

// mode: U8_RT / U8_FWD / U8_INV
hipError_t launch_u8_batch(const BatchArgs &a, uint32_t total, int mode, bool general, hipStream_t s)
{
  if (total == 0)
    return hipSuccess;
  switch (mode)
  {
  case U8_RT: return general ? launch_u8_batch_m<U8_RT, true>(a, total, s) : launch_u8_batch_m<U8_RT, false>(a, total, s);
  case U8_FWD: return general ? launch_u8_batch_m<U8_FWD, true>(a, total, s) : launch_u8_batch_m<U8_FWD, false>(a, total, s);
  case U8_INV: return launch_u8_batch_m<U8_INV, true>(a, total, s);
  }
  return hipErrorInvalidValue;
}

// mdct_init: the first use of any kernel of a code object loads it onto the device (~1.5 ms for this file's); asking for a kernel's attributes
// does the same without launching anything, so the cost sits in mdct_init instead of in the caller's first transform
hipError_t preload_kernels()
{
  hipFuncAttributes attr;
  const hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_clock_probe));
  if (e != hipSuccess)
    (void)hipGetLastError();
  return e;
}

hipError_t launch_clock_probe(unsigned long long *out, unsigned int ticks, unsigned int waves, hipStream_t s)
{
  hipLaunchKernelGGL(k_clock_probe, dim3(waves), dim3(64), 0, s, out, ticks);
  return hipGetLastError();
}

hipError_t launch_park_table(const OwnTables &tb, OwnTables *slot, hipStream_t s)
{
  hipLaunchKernelGGL(k_park_table, dim3(1), dim3(64), 0, s, tb, slot);
  return hipGetLastError();
}

hipError_t launch_u8_i16_fwd(const U8I16Args &a, hipStream_t s)
{
  if (a.nblocks == 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_u8_i16_fwd, dim3(grid_for(a.nblocks)), dim3(kWG), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_u8_records(const U8RecArgs &a, bool i16_in, hipStream_t s, bool clamp)
{
  if (a.nblocks == 0)
    return hipSuccess;
  if (i16_in)
    hipLaunchKernelGGL(k_u8_records<true>, dim3((a.nblocks + 63) / 64), dim3(64), 0, s, a);
  else if (clamp)
    hipLaunchKernelGGL(k_u8_records<false>, dim3((a.nblocks + 63) / 64), dim3(64), 0, s, a);
  else
    hipLaunchKernelGGL((k_u8_records<false, false>), dim3((a.nblocks + 63) / 64), dim3(64), 0, s, a);
  return hipGetLastError();
}

// 4 waves per workgroup (= per block row); 2- and 8-wave builds (chunks of 128 / 512 blocks) measured 125 / 126 us against 100
// (profiles/r03_d_time_all_entry_points.log) and are not kept
hipError_t launch_px_huffman(const PxHuffArgs &a, bool i16_in, bool pack, bool clamp, uint32_t n_rows, hipStream_t s)
{
  if (n_rows == 0)
    return hipSuccess;
  const dim3 grid(n_rows), wg(256);
  if (i16_in)
  {
    if (pack)
      hipLaunchKernelGGL((k_px_huffman_rows<true, 4, true>), grid, wg, 0, s, a);
    else
      hipLaunchKernelGGL((k_px_huffman_rows<true, 4, false>), grid, wg, 0, s, a);
  }
  else if (clamp)
  {
    if (pack)
      hipLaunchKernelGGL((k_px_huffman_rows<false, 4, true>), grid, wg, 0, s, a);
    else
      hipLaunchKernelGGL((k_px_huffman_rows<false, 4, false>), grid, wg, 0, s, a);
  }
  else
  {
    if (pack)
      hipLaunchKernelGGL((k_px_huffman_rows<false, 4, true, false>), grid, wg, 0, s, a);
    else
      hipLaunchKernelGGL((k_px_huffman_rows<false, 4, false, false>), grid, wg, 0, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_f32(const F32Args &a, int mode, hipStream_t s)
{
  if (a.nblocks == 0)
    return hipSuccess;
  const bool wide = a.bpr % 64 == 0; // every wave = 64 blocks of one block row, 2 KiB contiguous per pixel row
  const uint32_t launch_rows = a.nblocks / a.bpr;
  if (wide && launch_rows <= 65535u)
  {
    const dim3 g2(a.bpr / 64, launch_rows);
    if (mode == MODE_FWD)
      hipLaunchKernelGGL(k_f32_tile<MODE_FWD>, g2, dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL(k_f32_tile<MODE_INV>, g2, dim3(64), 0, s, a);
    return hipGetLastError();
  }
  const dim3 g(grid_for(a.nblocks)), b(kWG);
  if (mode == MODE_FWD && wide)
    hipLaunchKernelGGL((k_f32<MODE_FWD, true>), g, b, 0, s, a);
  else if (mode == MODE_FWD)
    hipLaunchKernelGGL((k_f32<MODE_FWD, false>), g, b, 0, s, a);
  else if (wide)
    hipLaunchKernelGGL((k_f32<MODE_INV, true>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_f32<MODE_INV, false>), g, b, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_stream_copy(const void *from, void *to, size_t bytes, int cus, hipStream_t s)
{
  const size_t n16 = bytes / 16;
  if (n16 == 0)
    return hipSuccess;
  (void)cus;
  const size_t per_wg = (size_t)kWG * kCopyUnroll;
  const size_t grid = (n16 + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(k_stream_copy, dim3((uint32_t)grid), dim3(kWG), 0, s, (const u32x4 *)from, (u32x4 *)to, n16);
  return hipGetLastError();
}

} // namespace mdct
