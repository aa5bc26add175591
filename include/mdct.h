/*
 * mdct.h -- C-ABI of the MI355X-native 8x8 block-DCT engine (libmdct_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * Every entry point names the reference interface it replaces
 * (file:line under rainerzufalldererste/simd_dct `src/`).
 *
 * Pointers `from` / `to` are DEVICE pointers (HBM) unless a function says otherwise;
 * quantisation tables (`lut`) are always HOST pointers to 64 floats, indexed v*8+u.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are
 * asynchronous on that stream; nothing is allocated or synchronised inside a launch
 * (safe for hipGraph capture), except where stated.
 *
 * Return value: the reference's own status enum, simd_dct.h:22-27:
 *   0 MDCT_SUCCESS, 1 MDCT_INVALID_PARAMETER, 2 MDCT_NOT_SUPPORTED.
 * HIP failures map to MDCT_NOT_SUPPORTED with detail in mdct_last_error().
 */
#ifndef MDCT_H
#define MDCT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* simd_dct.h:22-27 (enum simdDctResult) */
enum
{
  MDCT_SUCCESS = 0,
  MDCT_INVALID_PARAMETER = 1,
  MDCT_NOT_SUPPORTED = 2
};

/* Arithmetic profile of the u8 forward+quantise path: which reference tier the
 * bytes must be identical to (behaviours B1..B5). */
enum
{
  MDCT_PROFILE_REF_AVX = 0,    /* B1: simd_dct.cpp:2064-2262 (AVX2 == AVX-512VL)        */
  MDCT_PROFILE_REF_SSE = 1,    /* B2/B3: simd_dct.cpp:896-1103, :1540-1704 (SSE tiers)  */
  MDCT_PROFILE_REF_SCALAR = 2  /* B4/B5: simd_dct.cpp:177-298, :300-395 (scalar tiers)  */
};

/* Output layout of the u8 forward+quantise path. */
enum
{
  MDCT_LAYOUT_Q32 = 0,        /* 8 blocks interleaved: to[group*512 + coef*8 + blk], simd_dct.cpp:2221-2230 */
  MDCT_LAYOUT_STEREO = 1,     /* 64 coefficient planes, (block row, eye, block x) order, simd_dct.cpp:1061-1099 */
  MDCT_LAYOUT_BLOCK = 2,      /* 64 B per block, coefficients transposed (u*8+v), simd_dct.cpp:347-362 */
  MDCT_LAYOUT_BLOCK_SSE = 3   /* the SSE encq tiers' half-written pair layout, simd_dct.cpp:1662-1676 */
};

typedef struct mdct_device_info
{
  int device;             /* HIP device ordinal in use */
  int compute_units;      /* 256 on MI355X */
  int wavefront_size;     /* 64 */
  int lds_bytes_per_cu;   /* 163840 */
  int is_gfx950;          /* 1 when the device is CDNA4 */
  size_t hbm_bytes;       /* total global memory */
  char name[128];         /* gcnArchName */
} mdct_device_info;

/* Replaces _DetectCPUFeatures() (simd_platform.c:68-178): selects the HIP device for the
 * calling thread and probes it.  Idempotent.  Unlike the reference's CPU flags
 * (simd_dct.cpp:78-85 read them, nobody sets them), every entry point below calls this
 * lazily with the current device, so forgetting it cannot downgrade the path.  It is where
 * the one-time costs belong: it also loads the library's code objects onto the device, ~1.5 ms
 * that the caller's first transform would pay otherwise (first launch after mdct_init: 0.08 ms). */
int mdct_init(int device);
int mdct_get_device_info(mdct_device_info *info);
/* Thread-local detail for the last non-zero return in this thread ("" if none). */
const char *mdct_last_error(void);

/* ---- u8 plane -> u8 quantised coefficients (the reference's three products) ----------
 * Replaces the tier functions behind simd_dct.h:29-31, i.e.
 *   simdDCT_EncodeQuantize32ReorderBuffer_AVX2_Float        simd_dct.cpp:2064 (layout Q32,   profile REF_AVX)
 *   simdDCT_EncodeQuantizeReorderStereoBuffer_SSE41_Float   simd_dct.cpp:896  (layout STEREO, profile REF_SSE)
 *   simdDCT_EncodeQuantizeReorderStereoBuffer_NoSimd_Float  simd_dct.cpp:177  (layout STEREO, profile REF_SCALAR)
 *   simdDCT_EncodeQuantizeBuffer_NoSimd_Float               simd_dct.cpp:300  (layout BLOCK,  profile REF_SCALAR)
 *   simdDCT_EncodeQuantizeBuffer_SSE41_Float                simd_dct.cpp:1540 (layout BLOCK_SSE, profile REF_SSE)
 * with a sane range: block rows [by0, by1) in units of 8 pixel rows.
 *   Q32 / BLOCK / BLOCK_SSE: the plane is sizeX x sizeY, by in [0, sizeY/8).
 *   STEREO: `from` holds two stacked sizeX x sizeY/2 images; by in [0, sizeY/16) and
 *           each block row is transformed for both eyes.
 * `pitch_in` is the input row pitch in bytes (>= sizeX; the reference has pitch == sizeX).
 * The output buffer is sizeX*sizeY bytes with the reference's addressing; bytes the
 * reference would not write are not written.
 * Constraints: sizeX % 64 == 0 for Q32, sizeX % 16 == 0 for STEREO / BLOCK_SSE,
 * sizeX % 8 == 0 for BLOCK; sizeY % 8 == 0 (STEREO: % 16).  No alignment requirement on
 * `from` / pitch_in (like the reference, simd_dct.cpp:2109); 8-byte aligned rows are fastest. */
int mdct_fwd_quant_u8(const uint8_t *from, uint8_t *to, size_t pitch_in, const float *lut,
                      size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                      int layout, int profile, void *stream);

/* The same with an OUTPUT pitch (north_star: "plane in/out, width/height/stride"; the reference has
 * none, its strips are tight, simd_dct.cpp:2227-2230): block row `by` of the Q32 or BLOCK layout starts
 * at to + by * pitch_out instead of to + by * 8 * sizeX.  pitch_out >= 8 * sizeX, multiple of 16 bytes;
 * layouts Q32 and BLOCK only (the other two have no row strips). */
int mdct_fwd_quant_u8_pitched(const uint8_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                              size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                              int layout, int profile, void *stream);

/* ---- engine-own variants (no reference counterpart; BASELINE.json configs 2-5) --------
 * Planes are row-major, pitches in ELEMENTS, coefficient (v,u) of block (by,bx) lives
 * at (by*8+v, bx*8+u); coefficients are those of the orthonormal 2-D DCT-II.
 * Arithmetic: float32, scaled Arai-Agui-Nakajima butterflies (5 mul + 29 add per 8 points, 4 of the
 * multiplies fused into the additions they feed: 30 operations per pass, each rounded once); scale
 * factors folded into the (de)quantiser multipliers; rne(dct / lut) is ONE rounding of the exact product
 * (a fused multiply-add against 1.5 * 2^23).  Exact definition: DESIGN.md 4.2 (and the CPU checker under oracle/).
 * (The reference-pinned uint8 tiers above never fuse anything.)
 * `lut` (HOST, 64 floats, finite and non-zero) may be NULL = no quantisation:
 *   fwd:       coef = sat_i16(rne(dct / lut[i]))
 *   inv:       x    = sat_i16(rne(idct(coef * lut[i])))
 *   roundtrip: fwd -> (quantise -> dequantise when lut) -> inv, fused, one pass over HBM;
 *              without a table it returns the input bit-exactly.
 * Rows must be 16-byte aligned (pitch*sizeof(elem) % 16 == 0, base 16-byte aligned).
 * Tables: a table's multipliers are parked in device memory, where they stay hot in L2 across launches (the kernel argument
 * segment is cold in every cache on every launch).  First sight of a table (per device, per distinct content) enqueues a one-wave
 * upload kernel on `stream` right before the launch -- asynchronous and stream-ordered like the launch itself: no host block, no
 * synchronisation, no staging buffer.  Up to 256 tables per device stay resident, the least recently used one is evicted (its upload
 * waits, on the device, for the launches that still read the old content; other streams that meet a table whose upload is in flight
 * wait for it with hipStreamWaitEvent).  While `stream` is capturing, tables travel in the kernel arguments and the cache is not
 * touched -- a replayed graph must not depend on what a slot holds later: same results, nothing allocated, copied or synchronised
 * inside a capture.  mdct_table_cache_stats (below) counts what happened. */
int mdct_fwd_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                 size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
int mdct_inv_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                 size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
int mdct_roundtrip_i16(const int16_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut,
                       size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
/* 8-bit pixels <-> int16 coefficients, the JPEG-style pair (3 algorithmic bytes per pixel):
 *   fwd: coef = sat_i16(rne(dct(px - (level_shift ? 128 : 0)) / lut[i]))
 *   inv: px   = sat_u8(rne(idct(coef * lut[i] [+ 128 on the DC term when level_shift])))   -- the output's level shift rides in the
 *               DC term (a constant plane IS the DC term), so the shifted value takes part in the inverse's roundings
 * pitch of the u8 plane in bytes, of the int16 plane in elements; coefficient rows 16-byte
 * aligned, no alignment requirement on the pixel plane.  lut may be NULL. */
int mdct_fwd_u8_i16(const uint8_t *from, int16_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift,
                    size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
int mdct_inv_i16_u8(const int16_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift,
                    size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
int mdct_fwd_f32(const float *from, float *to, size_t pitch_in, size_t pitch_out,
                 size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
int mdct_inv_f32(const float *from, float *to, size_t pitch_in, size_t pitch_out,
                 size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);

/* Multi-plane call on int16 planes (Y + Cb + Cr with per-plane tables): == mdct_roundtrip_i16_batch below -- descriptors and
 * tables by value in the kernel arguments (no allocation, no sync, capture-safe).  ONE launch while the list fits the
 * 3584-byte argument blob (64 bytes per plane + 512 per distinct table: 3 planes with 3 tables, 48 planes sharing one);
 * longer lists take several launches -- mdct_batch_create / mdct_batch_run is one launch for any list.  `planes` is a HOST array. */
typedef struct mdct_plane_i16
{
  const int16_t *from;
  int16_t *to;
  size_t pitch_in, pitch_out; /* elements */
  size_t sizeX, sizeY;
  const float *lut;           /* HOST pointer to 64 floats, or NULL */
} mdct_plane_i16;
int mdct_roundtrip_i16_planes(const mdct_plane_i16 *planes, int n_planes, void *stream);

/* Plane batches: ANY number of separately allocated planes (sizes, pitches and tables of their own) per call --
 * BASELINE.json configs[3] ("256 independent 4096x4096 planes") and configs[2] (Y + Cb + Cr); the reference's only
 * batching affordance is the caller-side row range startY/endY (simd_dct.cpp:2243-2261).  Semantics per plane exactly
 * those of mdct_fwd_i16 / mdct_inv_i16 / mdct_roundtrip_i16 over the whole plane (to shard a plane, describe the
 * strip: offset pointers, smaller sizeY).  Every plane is cut into 64-block tiles of one block row (the last tile of
 * a row may be partial: any sizeX % 8 == 0) and the launch is one 1-D grid over the tiles of all planes.
 * `planes` is a HOST array; nothing is allocated, copied or synchronised: descriptors and tables travel in the
 * kernel arguments, as many planes per launch as fit (~50 planes sharing one table; capture-safe).  All planes are
 * validated before anything is launched. */
int mdct_fwd_i16_batch(const mdct_plane_i16 *planes, int n_planes, void *stream);
int mdct_inv_i16_batch(const mdct_plane_i16 *planes, int n_planes, void *stream);
int mdct_roundtrip_i16_batch(const mdct_plane_i16 *planes, int n_planes, void *stream);
/* The same with the descriptors and tables in device memory: mdct_batch_create allocates and uploads them once
 * (synchronous; not capture-safe), mdct_batch_run is then ONE launch for the whole list whatever its length (more
 * only beyond 2^26 - 1 tiles = 4.29e9 blocks: a tile is a 64-thread workgroup and one launch holds < 2^32 threads),
 * asynchronous on `stream`, capture-safe, repeatable.  The batch refers to the planes'
 * memory, not to the `planes` array or the tables, which may be freed after creation. */
enum
{
  MDCT_MODE_FWD = 0,
  MDCT_MODE_INV = 1,
  MDCT_MODE_ROUNDTRIP = 2
};
typedef struct mdct_batch mdct_batch;
int mdct_batch_create(mdct_batch **batch, int mode, const mdct_plane_i16 *planes, int n_planes);
int mdct_batch_run(const mdct_batch *batch, void *stream);
int mdct_batch_launches(const mdct_batch *batch); /* kernel launches one run takes (host function) */
int mdct_batch_destroy(mdct_batch *batch);

/* ---- 8-bit pixels in, 8-bit pixels out: the fused round trip (BASELINE.json configs[2], "3-plane 4:2:0 with per-plane JPEG
 * quant tables, fwd+inv"; SURVEY.md 8(d): u8 planes, 2 bytes per pixel).  The reference's pixel type is uint8 everywhere
 * (simd_dct.cpp:2107-2143) and it stops after the quantiser; this is forward -> quantise -> dequantise -> inverse in one
 * pass over HBM, bit for bit
 *     mdct_fwd_u8_i16(from -> coef, lut, level_shift)  followed by  mdct_inv_i16_u8(coef -> to, lut, level_shift)
 * without the int16 plane in between:
 *     c = sat_i16(rne(dct(px - shift) / lut[i])),  px' = sat_u8(rne(idct(c * lut[i], DC term + shift))),  shift = level_shift ? 128 : 0.
 * lut == NULL: no quantisation table (the coefficients are still rounded to int16, as the two calls would).
 * Pitches in BYTES; no alignment requirement on the planes (like the reference, simd_dct.cpp:2109); sizeX, sizeY multiples of 8.
 * One 64-block tile of one block row per wave, the last tile of a row may be partial (any sizeX % 8 == 0).
 * IN PLACE is allowed (to == from with pitch_out == pitch_in; also for mdct_roundtrip_i16 and the round-trip batches): a lane has read
 * all 64 samples of its block before it stores the first, and no other lane touches that block.
 * mdct_roundtrip_u8: block rows [by0, by1) of one plane.  _batch: any number of separately allocated planes, each with its own
 * table, in one launch -- descriptors and tables by value in the kernel arguments exactly like mdct_roundtrip_i16_batch
 * (no allocation, no synchronisation, capture-safe; Y + Cb + Cr with three tables fit one launch), or device-resident through
 * mdct_batch_create_u8 / mdct_batch_run / mdct_batch_destroy.  Tables are parked in device memory like those of the int16 calls. */
typedef struct mdct_plane_u8
{
  const uint8_t *from;
  uint8_t *to;
  size_t pitch_in, pitch_out; /* bytes */
  size_t sizeX, sizeY;
  const float *lut;           /* HOST pointer to 64 floats (finite, non-zero), or NULL */
} mdct_plane_u8;
int mdct_roundtrip_u8(const uint8_t *from, uint8_t *to, size_t pitch_in, size_t pitch_out, const float *lut, int level_shift,
                      size_t sizeX, size_t sizeY, size_t by0, size_t by1, void *stream);
int mdct_roundtrip_u8_batch(const mdct_plane_u8 *planes, int n_planes, int level_shift, void *stream);
/* device-resident form: run with mdct_batch_run, free with mdct_batch_destroy (above) */
int mdct_batch_create_u8(mdct_batch **batch, const mdct_plane_u8 *planes, int n_planes, int level_shift);

/* The reference's primary product on a plane list (replaces a caller's loop of simdDCT_EncodeQuantize32ReorderBuffer calls over the planes of
 * a frame, main.cpp:543 / simd_dct.cpp:113-133): per plane exactly
 *     mdct_fwd_quant_u8_pitched(from, to, pitch_in, pitch_out, lut, sizeX, sizeY, 0, sizeY / 8, MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX)
 * -- every block row, the reference's AVX2-tier arithmetic bit for bit (simd_dct.cpp:2064-2262) -- for any number of separately allocated
 * planes with their own tables in ONE launch.  pitch_in: bytes between pixel rows; pitch_out: bytes between the 8 * sizeX-byte output
 * strips of consecutive block rows (>= 8 * sizeX, a multiple of 16; 8 * sizeX = the reference's tight layout).  lut is required (HOST
 * pointer to 64 floats); sizeX % 64 == 0 and sizeY % 8 == 0 as in the reference's dispatcher (:117), else MDCT_NOT_SUPPORTED.  A table
 * that needs the exact x86 convert emulation (an entry beyond 2^17 in magnitude after scaling, inf, NaN) switches the whole call to it.
 * Kernel-argument form (no allocation, capture-safe) and device-resident form (run with mdct_batch_run) as for the other batches.
 * (Every block row of every plane is transformed: a caller that wants what ONE reference call does to its buffer -- the top sizeY / 2 rows
 * only, simd_dct.cpp:2245 -- describes that half as the plane, sizeY / 2 high; a row sub-range is a plane that starts further down.)
 * NOT in place (like the reference: a group's 512 output bytes cover pixels other lanes still have to read, simd_dct.cpp:2227-2230): a plane whose
 * output strips overlap its own input is refused with MDCT_INVALID_PARAMETER. */
int mdct_fwd_quant32_u8_batch(const mdct_plane_u8 *planes, int n_planes, void *stream);
int mdct_batch_create_q32(mdct_batch **batch, const mdct_plane_u8 *planes, int n_planes);

/* The two halves of that round trip on plane batches -- what an encoder (pixels -> quantised int16 coefficients) and a decoder
 * (coefficients -> pixels) run on the planes of a frame in ONE launch: per plane exactly mdct_fwd_u8_i16 / mdct_inv_i16_u8 over the
 * whole plane (3 bytes per pixel over HBM), on the tile kernel of the round trip.  pitch_px in bytes (no alignment requirement on the
 * pixel planes), pitch_coef in elements with 16-byte aligned coefficient rows.  Kernel-argument form (no allocation, capture-safe) and
 * device-resident form (mode = MDCT_MODE_FWD or MDCT_MODE_INV; run with mdct_batch_run) as for the other batches. */
typedef struct mdct_plane_u8_i16
{
  uint8_t *px;                /* 8-bit pixel plane: read by the forward, written by the inverse */
  int16_t *coef;              /* int16 coefficient plane: written by the forward, read by the inverse */
  size_t pitch_px, pitch_coef;
  size_t sizeX, sizeY;
  const float *lut;           /* HOST pointer to 64 floats (finite, non-zero), or NULL */
} mdct_plane_u8_i16;
int mdct_fwd_u8_i16_batch(const mdct_plane_u8_i16 *planes, int n_planes, int level_shift, void *stream);
int mdct_inv_i16_u8_batch(const mdct_plane_u8_i16 *planes, int n_planes, int level_shift, void *stream);
int mdct_batch_create_u8_i16(mdct_batch **batch, int mode, const mdct_plane_u8_i16 *planes, int n_planes, int level_shift);

/* ---- the stages either side of the transform (no reference counterpart: its pipeline starts from a
 * ready-made plane, main.cpp:475-493, and ends at the reorder store, simd_dct.cpp:2221-2230) ----------
 * After the quantiser: zig-zag scan (ITU-T T.81 Figure A.6) and run/level pairs (T.81 F.1.2.2) of every
 * block, one fixed-stride record per block, block index = by * (sizeX/8) + bx over the whole plane:
 *   levels[blk*64 + i], runs[blk*64 + i]  i-th non-zero coefficient in scan order (all 64 positions,
 *                                         DC included) and the number of zeros that precede it;
 *                                         zero beyond counts[blk]
 *   counts[blk]                           number of pairs; the coefficients after the last pair are
 *                                         zero (end of block)
 * runs == NULL (counts ignored): plain scan, levels[blk*64 + k] = coefficient at scan position k.
 * Sources: an int16 coefficient plane as written by mdct_fwd_i16 / mdct_fwd_u8_i16 (pitch in elements),
 * or the reference's q32 byte layout as written by mdct_fwd_quant_u8(MDCT_LAYOUT_Q32) / simd_dct.h:31,
 * whose bytes carry a +127 bias (simd_dct.cpp:2224): level = byte - 127.  Only block rows [by0, by1)
 * are read and only their records written.  Arrays 16-byte aligned. */
int mdct_zigzag_rle_i16(const int16_t *coef, size_t pitch, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                        int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream);
int mdct_zigzag_rle_q32(const uint8_t *q32, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                        int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream);
/* The same from any of the reference's three intact byte layouts (all carry the +127 bias):
 *   MDCT_LAYOUT_Q32     as above
 *   MDCT_LAYOUT_STEREO  the 64 coefficient planes of simd_dct.h:30 -- the streams the layout was made for;
 *                       block index = stream position (by*2 + eye) * (sizeX/8) + bx, [by0, by1) in units of
 *                       16 pixel rows (sizeY/16 of them), sizeX % 16 == 0
 *   MDCT_LAYOUT_BLOCK   the scalar encq tier's 64 bytes per block with coefficients stored transposed (u*8+v)
 * MDCT_LAYOUT_BLOCK_SSE stores only half of every block (simd_dct.cpp:1662-1676) and is refused. */
int mdct_zigzag_rle_u8(const uint8_t *coef, int layout, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                       int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream);
/* The encoder's front half fused: 8-bit pixels -> the records of mdct_fwd_u8_i16 followed by mdct_zigzag_rle_i16
 * (bit for bit), without the int16 plane in between: 1 byte in + 3 bytes out per pixel instead of 3 + 5.
 * pitch in bytes; no alignment requirement on the pixel plane; levels and runs 16-byte aligned; lut may be NULL. */
int mdct_fwd_u8_records(const uint8_t *px, size_t pitch, const float *lut, int level_shift, size_t sizeX, size_t sizeY,
                        size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream);
/* the same from an int16 plane (e.g. the planes of mdct_split420_u8): mdct_fwd_i16 followed by mdct_zigzag_rle_i16;
 * pitch in elements, rows 16-byte aligned */
int mdct_fwd_i16_records(const int16_t *from, size_t pitch, const float *lut, size_t sizeX, size_t sizeY,
                         size_t by0, size_t by1, int16_t *levels, uint8_t *runs, uint8_t *counts, void *stream);
/* Entropy stage: baseline Huffman coding of those records (ITU-T T.81 Annex C code construction, F.1.2.1 DC
 * difference categories, F.1.2.2 RRRRSSSS with ZRL / EOB, the typical tables of Annex K.3.3 -- `chroma` selects
 * Tables K.4 / K.6 instead of K.3 / K.5).  One independently decodable segment per block row: the row is a
 * restart interval of sizeX/8 blocks (E.1.4: DC predictor 0 at its start), byte-aligned, last byte padded with
 * 1-bits (F.1.2.3), written at out + by * seg_stride with its length in seg_bytes[by].  The bytes are NOT
 * stuffed (B.1.1.5) and carry no markers: the container writer does both (simd_dct_amd/jfif.py writes a JFIF
 * file any decoder opens; the tests decode it with libjpeg).  Levels are those of 8-bit baseline JPEG (DC
 * differences within +-2047, AC within +-1023; larger values saturate).  A count above 64 is read as 64; a
 * record that then is not a block (a zero AC level, scan positions beyond 63) is coded as its DC coefficient
 * alone, so the worst case below holds for any input.  seg_stride: multiple of 4,
 * >= 208 * (sizeX/8) + 8 (the worst case of F.1.2); sizeX/8 <= 65535. */
int mdct_huffman_rows(const int16_t *levels, const uint8_t *runs, const uint8_t *counts, size_t sizeX, size_t sizeY,
                      size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride, uint32_t *seg_bytes, void *stream);
/* The whole encoder front to back in ONE kernel: pixels (or an int16 plane) -> the row segments mdct_fwd_u8_records
 * (mdct_fwd_i16_records) followed by mdct_huffman_rows would produce, byte for byte -- the records exist only in LDS,
 * so 1 byte in and ~0.2 bytes out per pixel instead of 1 + 3 and 3 + 0.2.  Arguments as for those two calls
 * (pitch: bytes for pixels, elements for the int16 plane, whose rows must be 16-byte aligned; lut may be NULL;
 * seg_stride as above; sizeX/8 <= 65535).  ff_counts (may be NULL): ff_counts[by] = number of 0xFF bytes in row by's
 * segment, counted while the segment is written -- hand it to mdct_jpeg_pack_rows_counted and the packing needs no
 * counting pass of its own. */
int mdct_fwd_u8_huffman_rows(const uint8_t *px, size_t pitch, const float *lut, int level_shift, size_t sizeX, size_t sizeY,
                             size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride, uint32_t *seg_bytes, uint32_t *ff_counts,
                             void *stream);
int mdct_fwd_i16_huffman_rows(const int16_t *from, size_t pitch, const float *lut, size_t sizeX, size_t sizeY,
                              size_t by0, size_t by1, int chroma, uint8_t *out, size_t seg_stride, uint32_t *seg_bytes, uint32_t *ff_counts,
                              void *stream);
/* smallest legal seg_stride for a plane sizeX wide: 208 * (sizeX/8) + 8 (host function) */
size_t mdct_huffman_seg_stride(size_t sizeX);
/* The row segments -> one contiguous scan, ready to follow an SOS header: every row byte-stuffed (B.1.1.5: a zero
 * byte after each 0xFF) and, between consecutive rows, the restart marker FF D0+m with m = (first_rst + row) mod 8
 * (E.1.4; first_rst = 0 for a scan that starts at the image's first row).  row_offsets: n_rows + 1 device uint64;
 * on completion row_offsets[r] is where row r starts in `out` and row_offsets[n_rows] the total length.  A row
 * that would end beyond out_capacity is not written: compare row_offsets[n_rows] with the capacity.
 * (n_rows * seg_stride * 2 always suffices; in practice the total is ~0.2 bytes per pixel.) */
int mdct_jpeg_pack_rows(const uint8_t *segments, const uint32_t *seg_bytes, size_t seg_stride, size_t n_rows, int first_rst,
                        uint8_t *out, size_t out_capacity, uint64_t *row_offsets, void *stream);
/* the same when the producer of the segments has counted their 0xFF bytes (ff_counts[r], mdct_fwd_*_huffman_rows): two launches instead of three */
int mdct_jpeg_pack_rows_counted(const uint8_t *segments, const uint32_t *seg_bytes, const uint32_t *ff_counts, size_t seg_stride, size_t n_rows,
                                int first_rst, uint8_t *out, size_t out_capacity, uint64_t *row_offsets, void *stream);
/* Pixels (or an int16 plane) -> the finished scan in ONE launch: mdct_fwd_*_huffman_rows and mdct_jpeg_pack_rows_counted in a single
 * kernel.  A row's workgroup codes its segment into seg_work (scratch, addressed like `out` of mdct_fwd_u8_huffman_rows:
 * row by at seg_work + by * seg_stride, so by1 * seg_stride bytes), publishes its stuffed length, waits for the rows before it and copies
 * the segment -- still in L2 -- to its place in `out`.  out / out_capacity / row_offsets (by1 - by0 + 1 entries) / first_rst as for
 * mdct_jpeg_pack_rows; by0 < by1.
 * row_work: by1 - by0 + 2 device uint64 that the CALLER ZEROES ONCE (hipMemset) before the first call; every call leaves them ready
 * for the next one, also for replays of a captured launch and for another number of rows.  Calls that share a row_work must
 * not overlap (same stream, or ordered by events); concurrent calls take one row_work each.
 * Up to 16384 rows every row adds up the lengths of all rows before it in the same launch; taller planes take two launches
 * internally (the fused coder, then the counted packing), same bytes, same arguments -- they use row_work only as scratch that they
 * leave zeroed and neither consult nor set the failure word below (their launches have no cross-row wait that could time out).
 * FAILURE INDICATOR: row_offsets[by1 - by0] == UINT64_MAX.  Should a row not hear from all of its predecessors within ~1 s (cannot
 * happen while rows are dispatched in order), it sets a sticky word in row_work, the launch ends with UINT64_MAX there, and so does
 * every later call on the same row_work (without coding anything) until the caller zeroes row_work again.  A total above
 * out_capacity means the scan did not fit (rows that would end beyond it are not written). */
int mdct_fwd_u8_jpeg_scan(const uint8_t *px, size_t pitch, const float *lut, int level_shift, size_t sizeX, size_t sizeY, size_t by0, size_t by1,
                          int chroma, uint8_t *seg_work, size_t seg_stride, uint64_t *row_work, int first_rst, uint8_t *out, size_t out_capacity,
                          uint64_t *row_offsets, void *stream);
int mdct_fwd_i16_jpeg_scan(const int16_t *from, size_t pitch, const float *lut, size_t sizeX, size_t sizeY, size_t by0, size_t by1, int chroma,
                           uint8_t *seg_work, size_t seg_stride, uint64_t *row_work, int first_rst, uint8_t *out, size_t out_capacity,
                           uint64_t *row_offsets, void *stream);
/* BITS (16 counts) and HUFFVAL of the table as a DHT marker segment carries them (host function).
 * which: 0 DC luminance (K.3), 1 AC luminance (K.5), 2 DC chrominance (K.4), 3 AC chrominance (K.6). */
int mdct_huffman_spec(int which, uint8_t *bits16, uint8_t *vals, int *nvals);
/* the scan order used above: zz[k] = natural index v*8+u of scan position k (host function) */
void mdct_zigzag_table(uint8_t *zz64);
/* Before the transform (feeds mdct_roundtrip_i16_planes / BASELINE.json configs[2]): interleaved 8-bit
 * Y Cb Cr (3 bytes per pixel, pitch in bytes) -> three int16 planes level-shifted by -128; Y at full
 * resolution, Cb / Cr subsampled 2x2 by the rounded box average (a+b+c+d+2) >> 2 (JFIF centred siting).
 * sizeX, sizeY multiples of 16; pitches of the int16 planes in elements. */
int mdct_split420_u8(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, int16_t *y, int16_t *cb, int16_t *cr,
                     size_t pitch_y, size_t pitch_c, void *stream);
/* The same split into 8-BIT planes, not level-shifted (feeds the 8-bit plane batches -- mdct_roundtrip_u8_batch, mdct_fwd_u8_i16_batch,
 * mdct_fwd_quant32_u8_batch: BASELINE.json configs[2] as SURVEY.md 8(d) states it -- which shift by themselves): y = Y, cb / cr =
 * (a+b+c+d+2) >> 2; pitches in bytes, no alignment requirement.  3 bytes in + 1.5 bytes out per pixel. */
int mdct_split420_u8_planes(const uint8_t *ycc, size_t pitch, size_t sizeX, size_t sizeY, uint8_t *y, uint8_t *cb, uint8_t *cr,
                            size_t pitch_y, size_t pitch_c, void *stream);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------------------------
 * The reference has no communication; its only parallelism hook is the caller-side row range
 * startY/endY (simd_dct.cpp:2245-2255).  Here: every rank transforms its block-row shard in place
 * in a full-size output buffer (any entry point above with [by0, by1) = mdct_shard_rows(...)), then
 * one all-gather makes every rank's buffer complete.  RCCL is loaded on first use (librccl.so.1).
 *
 * mdct_shard_rows: contiguous, balanced, half-open shard of `n_rows` block rows for `rank`; shards are
 * in rank order and differ by at most one row (same arithmetic as simd_dct_amd/sharding.py). */
void mdct_shard_rows(size_t n_rows, int world, int rank, size_t *b0, size_t *b1);
/* The stereo layout (simd_dct.cpp:1061-1099) scatters a block-row shard over all 64 coefficient
 * planes: rank `rank` owns `piece_bytes` bytes at `first_offset + k * plane_stride`, k = 0..63.
 * Pure arithmetic (no device); mdct_allgather_stereo moves exactly these pieces. */
int mdct_stereo_shard_piece(size_t sizeX, size_t sizeY, int world, int rank,
                            size_t *first_offset, size_t *plane_stride, size_t *piece_bytes);
#define MDCT_UNIQUE_ID_BYTES 128
typedef struct mdct_comm mdct_comm;
/* rank 0 creates the id and hands its 128 bytes to the other ranks out of band (file, pipe, MPI, ...) */
int mdct_comm_get_unique_id(void *id128);
/* collective over all `world` ranks; binds the calling thread's current HIP device (call mdct_init first) */
int mdct_comm_init(mdct_comm **comm, int rank, int world, const void *id128);
int mdct_comm_destroy(mdct_comm *comm);
int mdct_comm_rank(const mdct_comm *comm);
int mdct_comm_world(const mdct_comm *comm);
/* Row-strip layouts (Q32, BLOCK, int16 / float32 planes): `buf` holds n_rows strips of row_bytes bytes
 * (row_bytes = 8 * pitch in bytes); rank r has filled the strips of mdct_shard_rows(n_rows, world, r).
 * Equal shards: ONE in-place ncclAllGather; ragged: one grouped broadcast per rank.  Asynchronous on
 * `stream` (the stream the kernels ran on: no extra synchronisation needed).
 * (Diagnostics: the environment variable MDCT_FORCE_RAGGED_GATHER=1 selects the grouped-broadcast form for equal shards too -- same
 * bytes; it lets a one-GPU box drive that form through the real RCCL, tests/test_comm.py.) */
int mdct_allgather_rows(mdct_comm *comm, void *buf, size_t row_bytes, size_t n_rows, void *stream);
/* Stereo coefficient-planar output of a sizeX x sizeY call sharded by mdct_shard_rows(sizeY / 16, ...):
 * 64 strided pieces per rank, gathered as 64 collectives inside one RCCL group. */
int mdct_allgather_stereo(mdct_comm *comm, uint8_t *buf, size_t sizeX, size_t sizeY, void *stream);

/* Diagnostics of the calling thread's current device's table cache (see "Tables" above), cumulative since the process started;
 * writes min(n, MDCT_TABLE_STAT_COUNT) counters (host function, no device work). */
enum
{
  MDCT_TABLE_STAT_HITS = 0,           /* launches that found their table resident */
  MDCT_TABLE_STAT_UPLOADS = 1,        /* first sights: upload kernel enqueued on the caller's stream */
  MDCT_TABLE_STAT_EVICTIONS = 2,      /* uploads that replaced the least recently used table */
  MDCT_TABLE_STAT_FROM_ARGUMENTS = 3, /* tables that travelled in the kernel arguments instead (capturing stream, no free slot, failure) */
  MDCT_TABLE_STAT_STREAM_WAITS = 4,   /* hipStreamWaitEvent on another stream's upload still in flight */
  MDCT_TABLE_STAT_UNFENCEABLE = 5,    /* eviction candidates passed over because a reader stream could not be fenced (destroyed stream, > 64 streams) */
  MDCT_TABLE_STAT_COUNT = 6
};
int mdct_table_cache_stats(uint64_t *stats, int n);

/* Shader-clock probe (diagnostics): `waves` one-wave workgroups each spin for `ticks_100MHz` ticks of the constant 100 MHz counter and write
 * (shader cycles elapsed, ticks elapsed) to out[2 * w], out[2 * w + 1] (device memory, 16 * waves bytes).  Launched on a second stream
 * beside a workload it reports the clock the chip holds under that workload: cycles * 100 / ticks MHz.  ticks_100MHz <= 10^7 (0.1 s). */
int mdct_clock_probe(uint64_t *out, uint32_t ticks_100MHz, uint32_t waves, void *stream);

/* Measured-roofline helper for bench tools: a read-N/write-N 16 B/lane stream copy on
 * the same stream (what "HBM roofline" means on this box). */
int mdct_stream_copy(const void *from, void *to, size_t bytes, void *stream);

/* Timing helpers so callers without a HIP binding (ctypes, cgo, JNI) can time the stream the
 * kernels run on with HIP events.  mdct_timer_* are NOT capture-safe. */
typedef struct mdct_timer mdct_timer;
mdct_timer *mdct_timer_create(void);
void mdct_timer_destroy(mdct_timer *t);
int mdct_timer_start(mdct_timer *t, void *stream);
int mdct_timer_stop(mdct_timer *t, void *stream);
/* blocks until the stop event has completed; returns elapsed milliseconds (< 0 on error) */
double mdct_timer_elapsed_ms(mdct_timer *t);
/* waits for the stop event by POLLING it (hipEventQuery in a loop on the calling thread): returns within a microsecond or two of the
 * event, where a blocking wait pays the interrupt wake-up (tens of microseconds -- 7 % of a 20-launch timed region) */
int mdct_timer_wait_spin(mdct_timer *t);
int mdct_stream_synchronize(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDCT_H */
