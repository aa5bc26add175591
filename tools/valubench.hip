// valubench.hip -- issue rate of plain vs packed fp32 VALU ops on gfx950 (wave64).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 1.0001f;
  const f2 c2 = {1.0001f, 0.9999f};
  for (int i = 0; i < iters; i++)
  {
    if (MODE == 0)
    { // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    else if (MODE == 1)
    { // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    else if (MODE == 2)
    { // 8 independent v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
    }
    else if (MODE == 3)
    { // 8 independent v_pk_mul_f32
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
    }
    else if (MODE == 4)
    { // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    else if (MODE == 5)
    { // 8 independent v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
    }
    else if (MODE == 6)
    { // pk_add with op_sel / neg modifiers (the butterfly forms)
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n v_pk_add_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %2, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel:[0,1] op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %4, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n v_pk_add_f32 %5, %5, %6 op_sel:[0,1] op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %6, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n v_pk_add_f32 %7, %7, %0 op_sel:[0,1] op_sel_hi:[1,0]\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
    }
    else if (MODE == 7)
    { // integer / conversion ops typical of the quantiser
      asm volatile("v_rndne_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_med3_i32 %2, %2, 0, %8\n v_cvt_f32_i32 %3, %3\n"
                   "v_rndne_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_med3_i32 %6, %6, 0, %8\n v_cvt_f32_i32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
  }
  if (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6)
  {
    a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y;
    a4 = p4.x + p4.y; a5 = p5.x + p5.y; a6 = p6.x + p6.y; a7 = p7.x + p7.y;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, float *out, int waves_per_simd)
{
  const int iters = 4096;
  const int grid = 256 * waves_per_simd; // 256 CUs x (waves_per_simd*4 waves)/4 waves per WG
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)grid * 4 * iters * 8; // wave-instructions
  const double per_simd_per_s = instr / 1024.0 / (ms * 1e-3);
  printf("%-28s waves/SIMD %d: %7.3f ms  %6.2f G wave-instr/s/SIMD  -> %.2f cycles/instr @2.4GHz\n", name, waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main()
{
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4, 8})
  {
    run<0>("v_add_f32", out, w);
    run<1>("v_mul_f32", out, w);
    run<4>("v_fma_f32", out, w);
    run<2>("v_pk_add_f32", out, w);
    run<3>("v_pk_mul_f32", out, w);
    run<5>("v_pk_fma_f32", out, w);
    run<6>("v_pk_add_f32 op_sel/neg", out, w);
    run<7>("rndne/cvt/med3 mix", out, w);
  }
  return 0;
}
