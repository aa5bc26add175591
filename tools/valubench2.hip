// valubench2.hip -- per-instruction issue cost on gfx950 for the instructions these kernels
// are made of: 8 waves/SIMD, 64-instruction straight-line bodies (loop overhead < 3 %).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valubench2.hip -o tools/valubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(INS) R8(INS) R8(INS) R8(INS) R8(INS) R8(INS) R8(INS) R8(INS) R8(INS)

#define DEF(NAME, ASMSTR)                                                                     \
  __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned *lds_dummy)      \
  {                                                                                            \
    __shared__ unsigned lds[256 * 8];                                                          \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const float c = 1.0001f;                                                                   \
    unsigned addr = threadIdx.x;                                                               \
    (void)lds; (void)addr;                                                                     \
    for (int i = 0; i < iters; i++)                                                            \
    {                                                                                          \
      asm volatile(ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR                     \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(addr)); \
    }                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)");                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
  }

#define I8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define U8(OP) OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"
#define T8(OP) OP " %0, %0, %8, %8\n" OP " %1, %1, %8, %8\n" OP " %2, %2, %8, %8\n" OP " %3, %3, %8, %8\n" OP " %4, %4, %8, %8\n" OP " %5, %5, %8, %8\n" OP " %6, %6, %8, %8\n" OP " %7, %7, %8, %8\n"
#define S8(OP, MOD) OP " %0, %0 " MOD "\n" OP " %1, %1 " MOD "\n" OP " %2, %2 " MOD "\n" OP " %3, %3 " MOD "\n" OP " %4, %4 " MOD "\n" OP " %5, %5 " MOD "\n" OP " %6, %6 " MOD "\n" OP " %7, %7 " MOD "\n"
#define L8 "ds_write_b8 %9, %0\n ds_write_b8 %9, %1 offset:64\n ds_write_b8 %9, %2 offset:128\n ds_write_b8 %9, %3 offset:192\n ds_write_b8 %9, %4 offset:256\n ds_write_b8 %9, %5 offset:320\n ds_write_b8 %9, %6 offset:384\n ds_write_b8 %9, %7 offset:448\n"
#define L32 "ds_write_b32 %9, %0\n ds_write_b32 %9, %1 offset:256\n ds_write_b32 %9, %2 offset:512\n ds_write_b32 %9, %3 offset:768\n ds_write_b32 %9, %4 offset:1024\n ds_write_b32 %9, %5 offset:1280\n ds_write_b32 %9, %6 offset:1536\n ds_write_b32 %9, %7 offset:1792\n"
#define D8 "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" \
           "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"

DEF(k_add, I8("v_add_f32"))
DEF(k_mul, I8("v_mul_f32"))
DEF(k_sub, I8("v_sub_f32"))
DEF(k_fma, T8("v_fma_f32"))
DEF(k_addu32, I8("v_add_u32"))
DEF(k_addu16, I8("v_add_u16"))
DEF(k_med3f, T8("v_med3_f32"))
DEF(k_med3i, T8("v_med3_i32"))
DEF(k_perm, T8("v_perm_b32"))
DEF(k_cvt_f32_i32, U8("v_cvt_f32_i32"))
DEF(k_cvt_f32_i32_sdwa, S8("v_cvt_f32_i32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"))
DEF(k_cvt_f32_ubyte0, U8("v_cvt_f32_ubyte0"))
DEF(k_cvt_f32_ubyte2, U8("v_cvt_f32_ubyte2"))
DEF(k_cvt_i32_f32, U8("v_cvt_i32_f32"))
DEF(k_rndne, U8("v_rndne_f32"))
DEF(k_mov, U8("v_mov_b32"))
DEF(k_lshl_or, T8("v_lshl_or_b32"))
DEF(k_and_or, T8("v_and_or_b32"))
DEF(k_bfe, T8("v_bfe_i32"))
DEF(k_dpp_mov, D8)
DEF(k_ds_write_b8, L8)
DEF(k_ds_write_b32, L32)

// packed fp32 (register pairs): the instructions the u8 tiers and the fused round trips are mostly made of
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DEFP(NAME, ASMSTR)                                                                     \
  __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned *lds_dummy)      \
  {                                                                                            \
    const float t = threadIdx.x;                                                               \
    f32x2 a0 = {t, t + 8}, a1 = {t + 1, t + 9}, a2 = {t + 2, t + 10}, a3 = {t + 3, t + 11}, a4 = {t + 4, t + 12}, a5 = {t + 5, t + 13}, a6 = {t + 6, t + 14}, a7 = {t + 7, t + 15}; \
    const f32x2 c = {1.0001f, 0.9999f};                                                        \
    for (int i = 0; i < iters; i++)                                                            \
    {                                                                                          \
      asm volatile(ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR ASMSTR                     \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); \
    }                                                                                          \
    const f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                     \
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                           \
    (void)lds_dummy;                                                                           \
  }
DEFP(k_pk_add, I8("v_pk_add_f32"))
DEFP(k_pk_mul, I8("v_pk_mul_f32"))
DEFP(k_pk_fma, T8("v_pk_fma_f32"))
#define PX8 "v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n" \
            "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
DEFP(k_pk_add_opsel, PX8)
DEF(k_sat_pk_u8_i16, U8("v_sat_pk_u8_i16"))
DEF(k_cvt_u32_f32, U8("v_cvt_u32_f32"))
#define C8 "v_cvt_pk_u8_f32 %0, %0, 1, %0\n v_cvt_pk_u8_f32 %1, %1, 1, %1\n v_cvt_pk_u8_f32 %2, %2, 1, %2\n v_cvt_pk_u8_f32 %3, %3, 1, %3\n v_cvt_pk_u8_f32 %4, %4, 1, %4\n v_cvt_pk_u8_f32 %5, %5, 1, %5\n v_cvt_pk_u8_f32 %6, %6, 1, %6\n v_cvt_pk_u8_f32 %7, %7, 1, %7\n"
DEF(k_cvt_pk_u8_f32, C8) // float -> byte (RNE, saturating), inserted into a dword: the one-instruction output stage measured for the 8-bit round trip

// the clock the chip holds while a benchmark kernel runs: one wave spins for `ticks` of the constant 100 MHz counter and reports the
// shader cycles (s_memtime) that passed meanwhile; launched just before the benchmark kernel on a second stream
__global__ __launch_bounds__(64) void k_clock(unsigned long long *out, unsigned ticks)
{
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks)
  {
    __builtin_amdgcn_s_sleep(4);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0)
  {
    out[0] = __builtin_amdgcn_s_memtime() - c0;
    out[1] = r1 - r0;
  }
}

static int g_waves = 8;
struct Cost
{
  const char *name;
  double ns, cycles;
};
static double g_last_cycles = 0;
static Cost g_costs[64];
static int g_ncosts = 0;

template <typename K>
void run(const char *name, K kern, float *out, int per_body)
{
  const int iters = 2048, waves_per_simd = g_waves;
  const int grid = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, nullptr);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 3; r++)
  {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, nullptr);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double instr_per_simd = (double)waves_per_simd * iters * per_body;
  const double ns = best * 1e6 / instr_per_simd;
  // the clock during this kernel: the probe first (it takes one wave slot), then the kernel beside it
  static hipStream_t s2 = nullptr;
  static unsigned long long *probe = nullptr;
  if (!s2)
  {
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipMalloc(&probe, 16);
  }
  hipDeviceSynchronize();
  const unsigned ticks = (unsigned)(best * 1e5 * 0.6); // 60 % of the kernel's duration, in 10 ns ticks
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, s2, probe, ticks > 100 ? ticks : 100);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, nullptr);
  hipDeviceSynchronize();
  unsigned long long pr[2] = {0, 1};
  hipMemcpy(pr, probe, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)pr[0] / ((double)pr[1] * 10.0);
  printf("%-22s %d waves/SIMD %8.3f ms   %5.2f ns/instr/SIMD  = %5.2f cycles at the measured %.2f GHz\n", name, waves_per_simd, best, ns, ns * ghz, ghz);
  g_last_cycles = ns * ghz;
  if (g_waves == 8 && g_ncosts < 64)
    g_costs[g_ncosts++] = Cost{name, ns, g_last_cycles};
}

int main()
{
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
#define RUN(K) run(#K, K, out, 64)
  RUN(k_add); RUN(k_mul); RUN(k_sub); RUN(k_fma); RUN(k_addu32); RUN(k_addu16); RUN(k_med3f); RUN(k_med3i); RUN(k_perm);
  RUN(k_cvt_f32_i32); RUN(k_cvt_f32_i32_sdwa); RUN(k_cvt_f32_ubyte0); RUN(k_cvt_f32_ubyte2); RUN(k_cvt_i32_f32); RUN(k_rndne);
  RUN(k_mov); RUN(k_lshl_or); RUN(k_and_or); RUN(k_bfe); RUN(k_dpp_mov); RUN(k_ds_write_b8); RUN(k_ds_write_b32);
  RUN(k_pk_add); RUN(k_pk_mul); RUN(k_pk_fma); RUN(k_pk_add_opsel); RUN(k_sat_pk_u8_i16); RUN(k_cvt_u32_f32); RUN(k_cvt_pk_u8_f32);
  // the classes bench.py's vector-issue floor is built from (tools/isa_classes.py sorts a kernel's instructions into them), at the
  // occupancy the product kernels run at as well
  for (int w : {4, 3, 2})
  {
    g_waves = w;
    RUN(k_add); RUN(k_pk_add); RUN(k_pk_mul); RUN(k_perm); RUN(k_cvt_f32_ubyte0); RUN(k_med3f); RUN(k_cvt_pk_u8_f32); RUN(k_sat_pk_u8_i16);
  }
  auto cost = [&](const char *n, bool cyc) {
    for (int i = 0; i < g_ncosts; i++)
      if (!strcmp(g_costs[i].name, n))
        return cyc ? g_costs[i].cycles : g_costs[i].ns;
    return 0.0;
  };
  auto min3 = [](double a, double b, double c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); };
  // per wave-instruction per SIMD at 8 waves per SIMD, the cheapest member of each class: a FLOOR.  In shader CYCLES (the clock each
  // benchmark ran at was measured beside it), so that a kernel's floor can be priced at the clock the chip holds under THAT kernel.
  for (int cyc = 0; cyc < 2; cyc++)
    printf("VALU_ISSUE_COSTS_%s {\"plain\": %.4f, \"packed\": %.4f, \"other\": %.4f, \"note\": \"8 waves per SIMD, independent chains; plain = v_add/sub/mul_f32, v_mov, v_add_u32; "
           "packed = v_pk_add/mul/fma_f32; other = converts, v_med3, v_perm, v_rndne, 3-operand integer ops, v_sat_pk_u8_i16\"}\n", cyc ? "CYCLES" : "NS",
           min3(cost("k_add", cyc), cost("k_mul", cyc), cost("k_mov", cyc)), min3(cost("k_pk_add", cyc), cost("k_pk_mul", cyc), cost("k_pk_add_opsel", cyc)),
           min3(cost("k_perm", cyc), cost("k_cvt_f32_ubyte0", cyc), cost("k_med3f", cyc)));
  return 0;
}
