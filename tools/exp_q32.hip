// exp_q32.hip -- where does the q32 (u8 -> u8, 8-block interleave) kernel's time go?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Iinclude -Isimd_dct_amd/csrc tools/exp_q32.hip -o tools/exp_q32
#include "../simd_dct_amd/csrc/mdct_kernels.hip"
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
using namespace mdct;

// REPS x (convert + DCT + quantise [+ LDS reorder]) on the same registers, one load, one store
template <int REPS, bool WITH_LDS>
__global__ __launch_bounds__(256, 6) void v_compute(U8Args a)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)row * 8 * a.pitch + (size_t)bx * 8;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  uint8_t *wl = lds[threadIdx.x >> 6];
  uint32_t q[64];
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 1
  for (int i = 0; i < REPS; i++)
  {
    encode_block<MDCT_PROFILE_REF_AVX, MDCT_LAYOUT_Q32, false>(a.consts, src + (acc.x & 1) * 8, a.pitch, a.qt, nullptr, q);
    if (WITH_LDS)
    {
#pragma unroll
      for (int c = 0; c < 64; c++)
        wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t c2 = (lane & 31) * 2;
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        const uint32_t g = 2 * k + (lane >> 5);
        const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
        const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
        acc.x ^= lo.x; acc.y ^= lo.y; acc.z ^= hi.x; acc.w ^= hi.y;
      }
    }
    else
    {
#pragma unroll
      for (int c = 0; c < 64; c += 4)
      {
        acc.x ^= q[c]; acc.y ^= q[c + 1]; acc.z ^= q[c + 2]; acc.w ^= q[c + 3];
      }
    }
  }
  *reinterpret_cast<uint4 *>(a.to + (size_t)t * 64) = acc;
}

// Staggered start: the first resident generation of waves all issue their loads at t = 0 and then
// all compute together, and the convoy persists (every later wave starts when its predecessor
// ends).  Delay first-generation wave k of each SIMD by k x (one wave's compute time) so that the
// SIMD always has one wave computing while the others load.
template <int STEP>
__global__ __launch_bounds__(256, 6) void v_stagger(U8Args a, uint32_t first_gen_blocks)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  if (__builtin_amdgcn_readfirstlane(t) < first_gen_blocks)
  {
    const uint32_t wid = __builtin_amdgcn_s_getreg(0x1804) & 15; // HW_REG_HW_ID[3:0]: wave slot within the SIMD
    for (uint32_t k = 0; k < wid; k++)
      __builtin_amdgcn_s_sleep(STEP);
  }
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const uint8_t *src = a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  uint8_t *wl = lds[threadIdx.x >> 6];
  uint32_t q[64];
  encode_block<MDCT_PROFILE_REF_AVX, MDCT_LAYOUT_Q32, false>(a.consts, src, a.pitch, a.qt, nullptr, q);
#pragma unroll
  for (int c = 0; c < 64; c++)
    wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + (t - lane)) * 64;
  const uint32_t c2 = (lane & 31) * 2;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const uint32_t g = 2 * k + (lane >> 5);
    const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
    const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(outw + g * 512 + c2 * 8));
  }
}

// persistent waves, register prefetch: the raw bytes of tile i are dead after the converts, so the
// loads of tile i+1 reuse those 16 VGPRs and are in flight during the butterflies of tile i.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v_pipe(U8Args a, uint32_t ntiles)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][64 * kQ32RowStride];
  const uint32_t lane = threadIdx.x & 63;
  uint8_t *wl = lds[threadIdx.x >> 6];
  const uint32_t total_waves = gridDim.x * 4;
  uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  auto src_of = [&](uint32_t tl) {
    const uint32_t t = tl * 64 + lane;
    const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
    return a.from + (size_t)(a.by0 + row) * 8 * a.pitch + (size_t)bx * 8;
  };
  uint2 rows[8];
  load_block_rows(src_of(tile), a.pitch, rows);
  for (;;)
  {
    float b[8][8];
    encode_rows<MDCT_PROFILE_REF_AVX, MDCT_LAYOUT_Q32, false>(a.consts, rows, a.qt, nullptr, b);
    const uint32_t next = tile + total_waves;
    const bool more = next < ntiles;
    if (more)
      load_block_rows(src_of(next), a.pitch, rows);
    uint32_t q[64];
    transform_quantise<MDCT_PROFILE_REF_AVX, MDCT_LAYOUT_Q32, false>(a.consts, b, a.qt, q);
#pragma unroll
    for (int c = 0; c < 64; c++)
      wl[c * kQ32RowStride + lane] = (uint8_t)q[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t *outw = a.to + ((size_t)a.by0 * a.bpr + (size_t)tile * 64) * 64;
    const uint32_t c2 = (lane & 31) * 2;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const uint32_t g = 2 * k + (lane >> 5);
      const uint2 lo = *reinterpret_cast<const uint2 *>(wl + c2 * kQ32RowStride + g * 8);
      const uint2 hi = *reinterpret_cast<const uint2 *>(wl + (c2 + 1) * kQ32RowStride + g * 8);
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(outw + g * 512 + c2 * 8));
    }
    if (!more) break;
    // the next iteration's LDS writes must not overtake this iteration's LDS reads (same wave:
    // LDS operations of one wave execute in order)
    tile = next;
  }
}

int main()
{
  const size_t W = 8192, H = 8192, bytes = W * H;
  const int NS = 4;
  std::vector<uint8_t *> A(NS), B(NS);
  std::vector<uint8_t> host(W * H);
  for (size_t i = 0; i < W * H; i++) host[i] = (uint8_t)((i * 2654435761u) >> 24);
  for (int i = 0; i < NS; i++)
  {
    hipMalloc(&A[i], bytes + 64);
    hipMalloc(&B[i], bytes);
    hipMemcpy(A[i], host.data(), bytes, hipMemcpyHostToDevice);
  }
  U8Args a;
  memset(&a, 0, sizeof(a));
  a.consts = DctConsts();
  for (int i = 0; i < 64; i++) a.qt.q[i] = 255.0f / ((0.1f + 0.01f * i) * 2000 * 0.95f);
  a.pitch = W; a.sizeX = W; a.bpr = W / 8; a.by0 = 0; a.nblocks = (uint32_t)(W / 8 * H / 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct V { const char *name; std::function<void(int)> f; std::vector<float> t; };
  std::vector<V> vs;
  auto args = [&](int s) { U8Args x = a; x.from = A[s]; x.to = B[s]; return x; };
  vs.push_back({"product q32", [&](int s) { launch_fwd_quant_u8(args(s), MDCT_LAYOUT_Q32, MDCT_PROFILE_REF_AVX, false, 0); }, {}});
  vs.push_back({"compute x1 +lds", [&](int s) { hipLaunchKernelGGL((v_compute<1, true>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"compute x3 +lds", [&](int s) { hipLaunchKernelGGL((v_compute<3, true>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"compute x1 nolds", [&](int s) { hipLaunchKernelGGL((v_compute<1, false>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  vs.push_back({"compute x3 nolds", [&](int s) { hipLaunchKernelGGL((v_compute<3, false>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s)); }, {}});
  const uint32_t ntiles = a.nblocks / 64;
  const uint32_t fg = 6 * 1024 * 64; // first resident generation: 6 waves x 1024 SIMDs x 64 blocks
  vs.push_back({"stagger step 0 (control)", [&](int s) { hipLaunchKernelGGL((v_stagger<0>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s), 0u); }, {}});
  vs.push_back({"stagger 12 x64 cyc/slot", [&](int s) { hipLaunchKernelGGL((v_stagger<12>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s), fg); }, {}});
  vs.push_back({"stagger 25 x64 cyc/slot", [&](int s) { hipLaunchKernelGGL((v_stagger<25>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s), fg); }, {}});
  vs.push_back({"stagger 47 x64 cyc/slot", [&](int s) { hipLaunchKernelGGL((v_stagger<47>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s), fg); }, {}});
  vs.push_back({"stagger 80 x64 cyc/slot", [&](int s) { hipLaunchKernelGGL((v_stagger<80>), dim3(a.nblocks / 256), dim3(256), 0, 0, args(s), fg); }, {}});
  vs.push_back({"pipe 6w grid 1536", [&](int s) { hipLaunchKernelGGL((v_pipe<6>), dim3(1536), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"pipe 6w grid 1024", [&](int s) { hipLaunchKernelGGL((v_pipe<6>), dim3(1024), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"pipe 6w grid 2048", [&](int s) { hipLaunchKernelGGL((v_pipe<6>), dim3(2048), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"pipe 5w grid 1280", [&](int s) { hipLaunchKernelGGL((v_pipe<5>), dim3(1280), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"pipe 4w grid 1024", [&](int s) { hipLaunchKernelGGL((v_pipe<4>), dim3(1024), dim3(256), 0, 0, args(s), ntiles); }, {}});
  vs.push_back({"pipe 6w grid 4096 (1 tile)", [&](int s) { hipLaunchKernelGGL((v_pipe<6>), dim3(4096), dim3(256), 0, 0, args(s), ntiles); }, {}});
  // correctness of the pipelined variants against the product kernel
  {
    std::vector<uint8_t> ref(bytes), got(bytes);
    vs[0].f(0); hipMemcpy(ref.data(), B[0], bytes, hipMemcpyDeviceToHost);
    for (size_t k = 5; k < vs.size(); k++)
    {
      hipMemset(B[0], 0x55, bytes);
      vs[k].f(0); hipMemcpy(got.data(), B[0], bytes, hipMemcpyDeviceToHost);
      size_t bad = 0;
      for (size_t i = 0; i < bytes; i++) bad += got[i] != ref[i];
      if (bad) printf("!! %s: %zu mismatching bytes\n", vs[k].name, bad);
    }
  }
  for (auto &v : vs) for (int i = 0; i < 300; i++) v.f(i % NS);
  hipDeviceSynchronize();
  for (int round = 0; round < 7; round++)
    for (auto &v : vs)
    {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 40; i++) v.f(i % NS);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      v.t.push_back(ms / 40);
    }
  for (auto &v : vs)
  {
    std::sort(v.t.begin(), v.t.end());
    printf("%-20s median %7.2f us  min %7.2f us\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3);
  }
  return 0;
}
