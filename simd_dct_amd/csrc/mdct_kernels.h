// mdct_kernels.h -- launch interface between the C-ABI layer (mdct_api.hip) and the
// gfx950 kernels (mdct_kernels.hip).  Internal; the public boundary is include/mdct.h.
#ifndef MDCT_KERNELS_H
#define MDCT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "mdct.h"
#include "batch_plan.h"

namespace mdct
{

enum { MODE_FWD = 0, MODE_INV = 1, MODE_ROUNDTRIP = 2 };

// simd_dct.cpp:140-146: sqrt(2)*cos(k*pi/16) for k = 1,2,3,5,6,7 and 1/sqrt(8), as the
// reference's float literals.  Passed by value in every argument block (see mdct_kernels.hip).
struct DctConsts
{
  // engine-own AAN butterfly (the CPU checker uses the same literals): cos(pi/4), cos(3pi/8), cos(pi/8)-+cos(3pi/8),
  // sqrt(2), 2cos(pi/8), 2(cos(pi/8)-+cos(3pi/8))
  float c707 = 0.707106781186547524f;
  float c382 = 0.382683432365089772f;
  float c541 = 0.541196100146196985f;
  float c1306 = 1.306562964876376528f;
  float c1414 = 1.414213562373095049f;
  float c1847 = 1.847759065022573512f;
  float c1082 = 1.082392200292393968f;
  float c2613 = 2.613125929752753056f;
  // rounding constants (also kept out of the instruction stream): 1.5*2^23 and 1.5*2^29
  float magic23 = 12582912.0f;
  float magic29 = 805306368.0f;
  float a = 1.3870398453221474618216191915664f;
  float b = 1.3065629648763765278566431734272f;
  float c = 1.1758756024193587169744671046113f;
  float d = 0.78569495838710218127789736765722f;
  float e = 0.54119610014619698439972320536639f;
  float f = 0.27589937928294301233595756366937f;
  float n = 0.35355339059327376220042218105242f;
};

// 64 quantiser multipliers, passed BY VALUE in the kernarg segment so the kernel reads
// them with scalar loads (wave-uniform, no VGPR or LDS cost).
struct QuantTable
{
  float q[64];
};

// engine-own paths: forward multiplier (AAN scale, times 1/lut when quantising) and
// inverse multiplier (AAN scale, times lut when dequantising), index v*8+u
struct OwnTables
{
  float qf[64];
  float dq[64];
};

// B1 on packed fp32: the butterfly constants as the register pairs the kernel consumes
// (Ca,Cf) (Cc,Cd) (Cb,Ce) (Cn, rounding constant 1.5*2^23 + 128), see encode_block_avx_pk
struct PkConstsArg
{
  float af[2], cd[2], be[2], nm[2];
  float da[2], fd[2], fc[2], ca[2]; // scalar tiers (K_TRUE): (Cd,Ca) (Cf,Cd) (Cf,Cc) (Cc,Ca)
  float bias[2];                    // SSE tiers: (1/255, 127.0f); scalar tiers: (255.0f, pred(0.5)), with nm = (Cn, 127/255)
  float div[2];                     // scalar tiers: (rn(1/255), rn(1/255 - rn(1/255))): px / 255.f without the division (mdct_kernels.hip)
};

struct U8Args
{
  const uint8_t *from;
  uint8_t *to;
  QuantTable qt;       // in the packed kernels' register-pair order (mdct_api.hip: pair_order)
  PkConstsArg pk;      // butterfly constants as the packed kernels consume them (mdct_api.hip)
  DctConsts consts;
  size_t pitch;        // input row pitch, bytes
  size_t sizeX;        // plane width, bytes (output addressing)
  size_t out_strip;    // Q32 / BLOCK: bytes between the output strips of consecutive block rows (8*sizeX when tight)
  size_t eye_offset;   // STEREO: byte offset of the second image
  size_t plane_stride; // STEREO: bytes per coefficient plane (sizeX*sizeY/64)
  uint32_t bpr;        // blocks per block row (sizeX/8)
  uint32_t by0;        // first block row of the launch
  uint32_t by_last;    // BLOCK_SSE: last block row of the launch
  uint32_t nblocks;    // blocks in the launch
  uint32_t spill_ok;   // BLOCK_SSE: trailing spill stays inside the buffer
  uint32_t out_tight;  // out_strip == 8*sizeX: a launch's Q32 / BLOCK output is one contiguous slab
};

struct I16Args
{
  const int16_t *from;
  int16_t *to;
  const OwnTables *tb_dev; // the same tables in device memory (mdct_api.hip: table cache), or nullptr: read `tb` from the argument segment
  OwnTables tb;
  DctConsts consts;
  size_t pitch_in, pitch_out; // elements
  uint32_t bpr, by0, nblocks;
};

struct U8I16Args
{ // one plane, 8-bit pixels -> quantised int16 coefficients (k_u8_i16_fwd)
  const uint8_t *px;
  int16_t *coef;
  OwnTables tb;
  DctConsts consts;
  size_t pitch_px, pitch_coef; // bytes / elements
  uint32_t bpr, by0, nblocks;
  float dc_shift;              // 64*128 when level-shifting, else 0
};

struct U8RecArgs
{ // pixels -> quantised coefficients -> zig-zag + run/level records in one pass (k_u8_records)
  const uint8_t *px; // 8-bit pixels, or an int16 plane (k_u8_records<true>)
  int16_t *levels; // [block][64]
  uint8_t *runs;   // [block][64]
  uint8_t *counts; // [block]
  OwnTables tb;
  DctConsts consts;
  size_t pitch_px; // bytes (int16 plane: elements)
  uint32_t bpr, by0, nblocks;
  float dc_shift;  // 64*128 when level-shifting, else 0
};

struct PxHuffArgs
{ // pixels -> quantised coefficients -> baseline Huffman rows in one pass (k_px_huffman_rows); records exist only in LDS
  const uint8_t *px; // 8-bit pixels, or an int16 plane (k_px_huffman_rows<true, ..>)
  uint8_t *out;      // row segments, seg_stride apart
  uint32_t *seg_bytes;
  uint32_t *ff_counts; // nullable: 0xFF bytes per row segment (what mdct_jpeg_pack_rows_counted needs instead of a counting pass)
  size_t seg_stride;
  size_t pitch_px;   // bytes (int16 plane: elements)
  OwnTables tb;      // qf in the pair order of the column pass (as U8RecArgs)
  DctConsts consts;
  uint32_t bpr, by0;
  float dc_shift;    // 64*128 when level-shifting, else 0
  uint32_t dc[12];   // size << 16 | code per DC category
  uint32_t ac[256];  // size << 16 | code per RRRRSSSS
  // k_px_huffman_rows<.., PACK> only: the row goes on into the contiguous scan in the same launch (pack_rows.h)
  uint8_t *scan;
  unsigned long long capacity;
  unsigned long long *row_off; // [n_rows + 1]
  unsigned long long *work;    // [n_rows + 2], zeroed once by the caller
  uint32_t n_rows, first_rst;
};

struct F32Args
{
  const float *from;
  float *to;
  float scale[64]; // AAN forward or inverse scale table
  DctConsts consts;
  size_t pitch_in, pitch_out; // elements
  uint32_t bpr, by0, nblocks;
};

// ---- plane batches: any number of separately allocated int16 planes in one launch (k_i16_batch) ----------------------
// One workgroup = one wave = one 64-block tile of one block row of one plane; the launch is a 1-D grid over the tiles of
// all planes in order.  BatchDesc, the division-free index arithmetic and the host-side layout: batch_plan.h.
// Descriptors and tables reach the kernel either embedded in the argument block (no allocation: mdct_*_i16_batch chunk their planes
// so that a chunk fits `blob`) or from device memory (mdct_batch_*: uploaded once at creation, one launch for any number of planes).
constexpr int kBatchBlob = 3584;
struct BatchHead
{ // 64 bytes: one s_load_dwordx16 at the top of every wave
  uint32_t n;              // planes in the launch
  uint32_t uniform;        // every plane has the same tile grid: plane = tile index / tiles per plane (pp_m, pp_s: its MagicDiv)
  uint32_t pp_m, pp_s;
  uint32_t table_bytes;    // embedded form: descriptors start at blob + table_bytes
  uint32_t pad[3];
  uint32_t first8[kBatchChain]; // `first` of planes 0..7 (UINT32_MAX beyond n)
};
static_assert(sizeof(BatchHead) == 64, "one s_load_dwordx16");
struct BatchArgs
{
  BatchHead head;
  const BatchDesc *descs;  // device memory, or nullptr: embedded
  const OwnTables *tables; // device memory, or nullptr: embedded (blob)
  DctConsts consts;
  alignas(8) float px[4];  // k_u8_batch only: (64 * shift, shift, 0, 0), shift = 128 with the level shift, else 0
  PkConstsArg pk;          // k_q32_batch only: the reference's AVX2-tier constants as register pairs (U8Args::pk)
  alignas(64) unsigned char blob[kBatchBlob]; // [tables][descriptors]
};
static_assert(offsetof(BatchArgs, descs) == 64 && offsetof(BatchArgs, tables) == 72, "the batch kernels read head + pointers as 20 consecutive dwords");
static_assert(offsetof(BatchArgs, px) % 8 == 0, "k_u8_batch reads px as two register pairs");
static_assert(sizeof(BatchArgs) <= 4096, "kernel argument block");
enum { BATCH_NO_LUT = 0, BATCH_ALL_LUT = 1, BATCH_MIXED = 2 };
// total: tiles in the launch; lutmode / sat matter for MODE_ROUNDTRIP only
hipError_t launch_i16_batch(const BatchArgs &a, uint32_t total, int mode, int lutmode, bool sat, hipStream_t s);
// the fused 8-bit round trip on the same descriptors (pointers / pitches in bytes); general: some plane's table needs the saturating
// quantiser or the clamping output stage (mdct_api.hip: u8_table_is_tame)
// mode: 0 the round trip, 1 pixels -> int16 coefficients, 2 int16 coefficients -> pixels (mdct_kernels.hip: U8_RT / U8_FWD / U8_INV)
hipError_t launch_u8_batch(const BatchArgs &a, uint32_t total, int mode, bool general, hipStream_t s);
// the reference's q32 product on the same descriptors (pitch_out = bytes between the block rows' output strips); safe: some table needs the exact convert emulation
hipError_t launch_q32_batch(const BatchArgs &a, uint32_t total, bool safe, hipStream_t s);

hipError_t launch_fwd_quant_u8(const U8Args &a, int layout, int profile, bool safe, hipStream_t s);
// lut_bounded / luts_bounded: every entry of every table >= 8.01 in magnitude (a quantised coefficient cannot leave int16)
hipError_t launch_i16(const I16Args &a, int mode, bool has_lut, hipStream_t s, bool lut_bounded = false);
hipError_t launch_u8_i16_fwd(const U8I16Args &a, hipStream_t s);
hipError_t launch_u8_records(const U8RecArgs &a, bool i16_in, hipStream_t s, bool clamp = true);
hipError_t launch_px_huffman(const PxHuffArgs &a, bool i16_in, bool pack, bool clamp, uint32_t n_rows, hipStream_t s);
hipError_t launch_f32(const F32Args &a, int mode, hipStream_t s);
hipError_t launch_park_table(const OwnTables &tb, OwnTables *slot, hipStream_t s); // table cache upload (mdct_api.hip)
hipError_t launch_clock_probe(unsigned long long *out, unsigned int ticks, unsigned int waves, hipStream_t s); // diagnostics
hipError_t preload_kernels();        // mdct_init: load this file's code object now (mdct_kernels.hip)
hipError_t preload_stage_kernels();  // the same for stages.hip
hipError_t launch_stream_copy(const void *from, void *to, size_t bytes, int cus, hipStream_t s);

} // namespace mdct
#endif
