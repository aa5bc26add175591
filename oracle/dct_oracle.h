/*
 * dct_oracle.h -- CPU restatement of the reference's 8x8 block-DCT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (simd_dct_amd/, include/,
 * the C-ABI library or the shim) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status
 *   PINNED   (bit-exact vs the reference built from /root/reference with
 *             `-O2 -ffp-contract=off`, see oracle/Makefile target `ref`, and vs the
 *             SHA-256 known answers of SURVEY.md 8(c) / Appendix B):
 *               orc_q32_avx            B1  simd_dct.cpp:2064-2262 (AVX2 == AVX-512VL tier)
 *               orc_stereo_sse         B2  simd_dct.cpp:896-1103  (SSE4.1 == SSSE3 == SSE2 tier)
 *               orc_encq_sse           B3  simd_dct.cpp:1540-1704 (SSE4.1 == SSSE3 tier)
 *               orc_stereo_scalar      B4  simd_dct.cpp:177-298
 *               orc_encq_scalar        B5  simd_dct.cpp:300-395
 *   UNPINNED ("parity unpinned": the reference has no int16 / float32-out / inverse
 *             path at all; these restate the engine's OWN arithmetic definition):
 *               orc_fwd_i16, orc_inv_i16, orc_roundtrip_i16, orc_fwd_u8_i16, orc_inv_i16_u8, orc_roundtrip_u8,
 *               orc_fwd_f32, orc_inv_f32, orc_fwd_f64ref
 *             (and, restating published definitions of ITU-T T.81 rather than engine arithmetic:
 *               orc_zigzag_table, orc_zigzag_rle_i16, orc_zigzag_rle_q32, orc_split420_u8)
 *
 * All arithmetic is IEEE-754 binary32, one rounding per written operation, no FMA
 * (compile with -ffp-contract=off, never -ffast-math).
 */
#ifndef DCT_ORACLE_H
#define DCT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1-D 8-point kernels, in place on p[0], p[stride], ... p[7*stride].
 * which: 0 = K_AVX  (simd_dct.cpp:2158-2184, k=3 sign quirk, pairwise association)
 *        1 = K_SSE  (simd_dct.cpp:434-654 / 672-892, k=1 sign quirk)
 *        2 = K_TRUE (simd_dct.cpp:138-172, left-to-right association, correct DCT-II)
 *        3 = K_OWN  (engine's own: scaled AAN butterfly + scale factors)  [unpinned]   */
void orc_dct8(float *p, ptrdiff_t stride, int which);
/* inverse of K_OWN, orthonormal coefficients in [unpinned] */
void orc_idct8_own(float *p, ptrdiff_t stride);
/* the raw AAN butterflies (5 mul + 29 add) and their 2-D scale tables [unpinned] */
extern const double orc_aan_scale[8];
void orc_aan_fwd8(float *p, ptrdiff_t stride);
void orc_aan_inv8(float *p, ptrdiff_t stride);
void orc_aan_tables(float *fwd64, float *inv64);

/* The five reference behaviours, with the reference's exact call semantics
 * (top-half loop, inclusive endY, untouched output bytes stay untouched).
 * Return 0 = sdr_Success, 1 = sdr_InvalidParameter, 2 = sdr_NotSupported. */
int orc_q32_avx(const uint8_t *from, uint8_t *to, const float *lut, size_t sizeX, size_t sizeY, size_t startY, size_t endY);
int orc_stereo_sse(const uint8_t *from, uint8_t *to, const float *lut, size_t sizeX, size_t sizeY, size_t startY, size_t endY);
int orc_encq_sse(const uint8_t *from, uint8_t *to, const float *lut, size_t sizeX, size_t sizeY, size_t startY, size_t endY);
int orc_stereo_scalar(const uint8_t *from, uint8_t *to, const float *lut, size_t sizeX, size_t sizeY, size_t startY, size_t endY);
int orc_encq_scalar(const uint8_t *from, uint8_t *to, const float *lut, size_t sizeX, size_t sizeY, size_t startY, size_t endY);

/* Engine-own variants [unpinned].  Planes are row-major with pitches in ELEMENTS,
 * block rows [by0, by1) (units of 8 pixel rows) over the FULL plane, coefficient (v,u)
 * of block (by,bx) stored in place at (by*8+v, bx*8+u).
 * lut == NULL: no quantisation; otherwise coefficients are divided by lut[i] before
 * rounding (fwd) and multiplied by it again (inv).  The exact arithmetic -- AAN butterflies,
 * where each scale factor and each rounding sits -- is defined in dct_oracle.c next to
 * own_tables().                                                                        */
int orc_fwd_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                size_t sizeX, size_t sizeY, size_t by0, size_t by1);
int orc_inv_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                size_t sizeX, size_t sizeY, size_t by0, size_t by1);
int orc_roundtrip_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                      size_t sizeX, size_t sizeY, size_t by0, size_t by1);
int orc_fwd_u8_i16(const uint8_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift,
                   size_t sizeX, size_t sizeY, size_t by0, size_t by1);
int orc_inv_i16_u8(const int16_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift,
                   size_t sizeX, size_t sizeY, size_t by0, size_t by1);
/* fused: orc_fwd_u8_i16 then orc_inv_i16_u8 per block, pitches in bytes (configs[2] as SURVEY.md 8(d) defines it: u8 in, u8 out) */
int orc_roundtrip_u8(const uint8_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift,
                     size_t sizeX, size_t sizeY, size_t by0, size_t by1);
int orc_fwd_f32(const float *from, float *to, size_t pitch_in, size_t pitch_out,
                size_t sizeX, size_t sizeY, size_t by0, size_t by1);
int orc_inv_f32(const float *from, float *to, size_t pitch_in, size_t pitch_out,
                size_t sizeX, size_t sizeY, size_t by0, size_t by1);
/* double-precision orthonormal DCT-II by the defining cosine sum (config 5 yardstick) */
int orc_fwd_f64ref(const float *from, double *to, size_t pitch_in, size_t pitch_out,
                   size_t sizeX, size_t sizeY, size_t by0, size_t by1);

/* the engine-own quantiser c = sat_i16(rne(y * qf)) -- ONE rounding of the exact product -- and the 8-bit output stage sat_u8(rne(x)), element-wise [unpinned] */
void orc_quant_i16(const float *y, const float *qf, int16_t *out, size_t n);
void orc_sat_u8_rne(const float *x, uint8_t *out, size_t n);

/* u8 forward+quantise with the engine's full-plane, half-open block-row range and
 * explicit pitches -- same arithmetic as B1 (profile 0) -- used to check the native
 * C-ABI (mdct_fwd_quant_u8) rather than the reference-semantics shim. */
int orc_q32_native(const uint8_t *from, uint8_t *to, size_t pitch_in, const float *lut,
                   size_t sizeX, size_t sizeY, size_t by0, size_t by1);

/* Stages either side of the codec core, engine-own, defined against ITU-T T.81:
 * zig-zag scan (Figure A.6) + run/level pairs of an int16 coefficient plane or of the reference's
 * q32 byte layout, and the 4:2:0 split / subsample that feeds BASELINE.json configs[2]. */
void orc_zigzag_table(uint8_t *zz64);
int orc_zigzag_rle_i16(const int16_t *coef, size_t pitch, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                       int16_t *levels, uint8_t *runs, uint8_t *counts);
int orc_zigzag_rle_q32(const uint8_t *q32, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                       int16_t *levels, uint8_t *runs, uint8_t *counts);
int orc_zigzag_rle_u8(const uint8_t *coef, int layout /* 0 q32, 1 stereo, 2 block */, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                      int16_t *levels, uint8_t *runs, uint8_t *counts);
/* Baseline Huffman coding (T.81 Annex C, F.1.2, tables K.3-K.6) of the records above: one unstuffed, 1-padded
 * segment per block row at out + by*seg_stride (seg_stride >= 208*sizeX/8 + 8), its length in seg_bytes[by];
 * the DC predictor restarts in every row.  orc_huffman_spec: BITS/HUFFVAL as a DHT segment carries them
 * (which: 0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma); orc_huffman_tables: size << 16 | code per symbol. */
int orc_huffman_spec(int which, uint8_t *bits16, uint8_t *vals, int *nvals);
void orc_huffman_tables(int chroma, uint32_t *dc256, uint32_t *ac256);
int orc_huffman_rows(const int16_t *levels, const uint8_t *runs, const uint8_t *counts, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                     int chroma, uint8_t *out, size_t seg_stride, uint32_t *seg_bytes);
int orc_jpeg_pack_rows(const uint8_t *seg, const uint32_t *seg_bytes, size_t seg_stride, size_t n_rows, int first_rst, uint8_t *out, size_t capacity, uint64_t *row_off);
int orc_split420_u8(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, int16_t *y, int16_t *cb, int16_t *cr,
                    size_t pitch_y, size_t pitch_c);

#ifdef __cplusplus
}
#endif
#endif
