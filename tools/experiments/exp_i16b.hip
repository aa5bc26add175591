// exp_i16b.hip -- round-2 A/B harness: the engine-own AAN butterflies on packed fp32.
// Same idea as the u8 tiers: the row pass runs "horizontally" on 4 register pairs with op_sel /
// neg modifiers, the column pass "vertically" on pairs of columns; here the flow graph is Arai-Agui-Nakajima's,
// which leaves 6 (forward) / 12 (inverse) scalar operations per horizontal transform that have no partner.
// Every packed or scalar operation is the individually rounded IEEE operation of aan_fwd8 / aan_inv8, so the
// bytes must equal the product kernel's (checked before timing).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Iinclude -Isimd_dct_amd/csrc tools/exp_i16b.hip -o tools/exp_i16b
#include "../simd_dct_amd/csrc/mdct_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <functional>
#include <vector>
using namespace mdct;

struct AanPk
{
  f32x2 c707_382;   // (cos(pi/4), cos(3pi/8))
  f32x2 c541_1306;  // (cos(pi/8)-cos(3pi/8), cos(pi/8)+cos(3pi/8))
  f32x2 c1414_1847; // (sqrt 2, 2cos(pi/8))
  f32x2 c1082_2613;
  f32x2 magic;      // (1.5*2^23, 1.5*2^29)
};

#define X_LOLO "op_sel:[0,0] op_sel_hi:[0,0]"
#define X_HIHI "op_sel:[1,1] op_sel_hi:[1,1]"
#define X_LOHI "op_sel:[0,1] op_sel_hi:[0,1]" // src0.lo with src1.hi, for both halves
#define X_HILO "op_sel:[1,0] op_sel_hi:[1,0]"
#define SUMDIFF "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" // (a.lo + a.hi, a.lo - a.hi) when both sources are a

// forward, one line in natural pairs (p0,p1)(p2,p3)(p4,p5)(p6,p7) -> (y0,y4) (y2,y6) (y5,y3) (y1,y7)
__device__ __forceinline__ void aan_fwd_h(const AanPk &K, f32x2 a01, f32x2 a23, f32x2 a45, f32x2 a67, f32x2 &o04, f32x2 &o26, f32x2 &o53, f32x2 &o17)
{
  f32x2 t01, t23, t76, t54, e01, e32, w, o, z13, z24, z1113;
  MDCT_PKA(t01, a01, a67, MDCT_X);                   // (t0, t1) = (p0+p7, p1+p6)
  MDCT_PKA(t23, a23, a45, MDCT_X);                   // (t2, t3) = (p2+p5, p3+p4)
  MDCT_PKA(t76, a01, a67, MDCT_X " " MDCT_NEG_B);    // (t7, t6) = (p0-p7, p1-p6)
  MDCT_PKA(t54, a23, a45, MDCT_X " " MDCT_NEG_B);    // (t5, t4) = (p2-p5, p3-p4)
  MDCT_PKA(e01, t01, t23, MDCT_X);                   // (e10, e11) = (t0+t3, t1+t2)
  MDCT_PKA(e32, t01, t23, MDCT_X " " MDCT_NEG_B);    // (e13, e12) = (t0-t3, t1-t2)
  MDCT_PKA(o04, e01, e01, SUMDIFF);                  // (e10+e11, e10-e11)
  w.x = e32.y + e32.x;                               // e12 + e13
  w.y = t54.x + t76.y;                               // o11 = t5 + t6
  o.x = t54.y + t54.x;                               // o10 = t4 + t5
  o.y = t76.y + t76.x;                               // o12 = t6 + t7
  f32x2 z5;
  z5.x = (o.x - o.y) * K.c707_382.y;                 // z5 = (o10 - o12) * c382
  MDCT_PKM(z13, w, K.c707_382, MDCT_K_LL);           // (z1, z3) = (e12+e13, o11) * c707
  MDCT_PKM(z24, o, K.c541_1306, MDCT_K_LH);          // (c541 o10, c1306 o12)
  MDCT_PKA(z24, z24, z5, "op_sel:[0,0] op_sel_hi:[1,0]"); // (z2, z4) = (.. + z5, .. + z5)
  MDCT_PKA(z1113, t76, z13, X_LOHI " neg_hi:[0,1]"); // (z11, z13) = (t7+z3, t7-z3)
  MDCT_PKA(o26, e32, z13, X_LOLO " neg_hi:[0,1]");   // (e13+z1, e13-z1)
  MDCT_PKA(o53, z1113, z24, X_HILO " neg_hi:[0,1]"); // (z13+z2, z13-z2)
  MDCT_PKA(o17, z1113, z24, X_LOHI " neg_hi:[0,1]"); // (z11+z4, z11-z4)
}

// forward down a pair of columns, in place (direct image of aan_fwd8)
__device__ __forceinline__ void aan_fwd_v(const AanPk &K, f32x2 (&p)[8])
{
  f32x2 t0, t7, t1, t6, t2, t5, t3, t4, e10, e13, e11, e12, z1, o10, o11, o12, z5, z2, z4, z3, z11, z13;
  MDCT_PKA(t0, p[0], p[7], ""); MDCT_PKA(t7, p[0], p[7], MDCT_NEG_B); MDCT_PKA(t1, p[1], p[6], ""); MDCT_PKA(t6, p[1], p[6], MDCT_NEG_B);
  MDCT_PKA(t2, p[2], p[5], ""); MDCT_PKA(t5, p[2], p[5], MDCT_NEG_B); MDCT_PKA(t3, p[3], p[4], ""); MDCT_PKA(t4, p[3], p[4], MDCT_NEG_B);
  MDCT_PKA(e10, t0, t3, ""); MDCT_PKA(e13, t0, t3, MDCT_NEG_B); MDCT_PKA(e11, t1, t2, ""); MDCT_PKA(e12, t1, t2, MDCT_NEG_B);
  MDCT_PKA(z1, e12, e13, ""); MDCT_PKM(z1, z1, K.c707_382, MDCT_K_LL);
  MDCT_PKA(o10, t4, t5, ""); MDCT_PKA(o11, t5, t6, ""); MDCT_PKA(o12, t6, t7, "");
  MDCT_PKA(z5, o10, o12, MDCT_NEG_B); MDCT_PKM(z5, z5, K.c707_382, MDCT_K_HH);
  MDCT_PKM(z2, o10, K.c541_1306, MDCT_K_LL); MDCT_PKA(z2, z2, z5, "");
  MDCT_PKM(z4, o12, K.c541_1306, MDCT_K_HH); MDCT_PKA(z4, z4, z5, "");
  MDCT_PKM(z3, o11, K.c707_382, MDCT_K_LL);
  MDCT_PKA(z11, t7, z3, ""); MDCT_PKA(z13, t7, z3, MDCT_NEG_B);
  MDCT_PKA(p[0], e10, e11, ""); MDCT_PKA(p[4], e10, e11, MDCT_NEG_B);
  MDCT_PKA(p[2], e13, z1, ""); MDCT_PKA(p[6], e13, z1, MDCT_NEG_B);
  MDCT_PKA(p[5], z13, z2, ""); MDCT_PKA(p[3], z13, z2, MDCT_NEG_B);
  MDCT_PKA(p[1], z11, z4, ""); MDCT_PKA(p[7], z11, z4, MDCT_NEG_B);
}

// inverse down a pair of columns, in place (direct image of aan_inv8)
__device__ __forceinline__ void aan_inv_v(const AanPk &K, f32x2 (&p)[8])
{
  f32x2 e10, e11, e13, e12, t0, t3, t1, t2, z13, z10, z11, z12, t7, o11, z5, o10, o12, t6, t5, t4, m;
  MDCT_PKA(e10, p[0], p[4], ""); MDCT_PKA(e11, p[0], p[4], MDCT_NEG_B);
  MDCT_PKA(e13, p[2], p[6], "");
  MDCT_PKA(e12, p[2], p[6], MDCT_NEG_B); MDCT_PKM(e12, e12, K.c1414_1847, MDCT_K_LL); MDCT_PKA(e12, e12, e13, MDCT_NEG_B);
  MDCT_PKA(t0, e10, e13, ""); MDCT_PKA(t3, e10, e13, MDCT_NEG_B); MDCT_PKA(t1, e11, e12, ""); MDCT_PKA(t2, e11, e12, MDCT_NEG_B);
  MDCT_PKA(z13, p[5], p[3], ""); MDCT_PKA(z10, p[5], p[3], MDCT_NEG_B); MDCT_PKA(z11, p[1], p[7], ""); MDCT_PKA(z12, p[1], p[7], MDCT_NEG_B);
  MDCT_PKA(t7, z11, z13, "");
  MDCT_PKA(o11, z11, z13, MDCT_NEG_B); MDCT_PKM(o11, o11, K.c1414_1847, MDCT_K_LL);
  MDCT_PKA(z5, z10, z12, ""); MDCT_PKM(z5, z5, K.c1414_1847, MDCT_K_HH);
  MDCT_PKM(m, z12, K.c1082_2613, MDCT_K_LL); MDCT_PKA(o10, m, z5, MDCT_NEG_B);
  MDCT_PKM(m, z10, K.c1082_2613, MDCT_K_HH); MDCT_PKA(o12, z5, m, MDCT_NEG_B);
  MDCT_PKA(t6, o12, t7, MDCT_NEG_B); MDCT_PKA(t5, o11, t6, MDCT_NEG_B); MDCT_PKA(t4, o10, t5, "");
  MDCT_PKA(p[0], t0, t7, ""); MDCT_PKA(p[7], t0, t7, MDCT_NEG_B);
  MDCT_PKA(p[1], t1, t6, ""); MDCT_PKA(p[6], t1, t6, MDCT_NEG_B);
  MDCT_PKA(p[2], t2, t5, ""); MDCT_PKA(p[5], t2, t5, MDCT_NEG_B);
  MDCT_PKA(p[4], t3, t4, ""); MDCT_PKA(p[3], t3, t4, MDCT_NEG_B);
}

// inverse, one line given as (c0,c4) (c2,c6) (c5,c3) (c1,c7) -> (x0,x7) (x1,x6) (x2,x5) (x4,x3)
__device__ __forceinline__ void aan_inv_h(const AanPk &K, f32x2 i04, f32x2 i26, f32x2 i53, f32x2 i17, f32x2 &o07, f32x2 &o16, f32x2 &o25, f32x2 &o43)
{
  f32x2 e, f, t03, t12, z3, z1, td, u, v;
  MDCT_PKA(e, i04, i04, SUMDIFF);                    // (e10, e11) = (c0+c4, c0-c4)
  MDCT_PKA(f, i26, i26, SUMDIFF);                    // (e13, c2-c6)
  f.y = (f.y * K.c1414_1847.x) - f.x;                // e12 = (c2-c6)*sqrt2 - e13
  MDCT_PKA(t03, e, f, X_LOLO " neg_hi:[0,1]");       // (t0, t3) = (e10+e13, e10-e13)
  MDCT_PKA(t12, e, f, X_HIHI " neg_hi:[0,1]");       // (t1, t2) = (e11+e12, e11-e12)
  MDCT_PKA(z3, i53, i53, SUMDIFF);                   // (z13, z10) = (c5+c3, c5-c3)
  MDCT_PKA(z1, i17, i17, SUMDIFF);                   // (z11, z12) = (c1+c7, c1-c7)
  MDCT_PKA(td, z1, z3, X_LOLO " neg_hi:[0,1]");      // (t7, z11-z13)
  td.y = td.y * K.c1414_1847.x;                      // o11
  const float z5 = (z3.y + z1.y) * K.c1414_1847.y;   // (z10 + z12) * c1847
  const float o10 = (K.c1082_2613.x * z1.y) - z5;
  const float o12 = z5 - (K.c1082_2613.y * z3.y);
  u.x = o12 - td.x;                                  // t6
  u.y = td.y - u.x;                                  // t5
  v.x = o10 + u.y;                                   // t4
  MDCT_PKA(o07, t03, td, X_LOLO " neg_hi:[0,1]");    // (t0+t7, t0-t7)
  MDCT_PKA(o16, t12, u, X_LOLO " neg_hi:[0,1]");     // (t1+t6, t1-t6)
  MDCT_PKA(o25, t12, u, X_HIHI " neg_hi:[0,1]");     // (t2+t5, t2-t5)
  MDCT_PKA(o43, t03, v, X_HILO " neg_hi:[0,1]");     // (t3+t4, t3-t4)
}

// fused round trip without a table: forward rows (h), forward columns (v), inverse columns (v), inverse rows (h);
// the 1/64 rides in the final rounding (rne_i16_bits<6>)
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void v_rt_pk(I16Args a, AanPk K)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.nblocks)
    return;
  const uint32_t row = t / a.bpr, bx = t - row * a.bpr;
  const size_t by = a.by0 + row;
  const int16_t *src = a.from + by * 8 * a.pitch_in + (size_t)bx * 8;
  int16_t *dst = a.to + by * 8 * a.pitch_out + (size_t)bx * 8;
  uint4 in[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    in[r] = ld_stream16(src + (size_t)r * a.pitch_in);
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
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    aan_fwd_v(K, P[j]);
    aan_inv_v(K, P[j]);
  }
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    f32x2 o07, o16, o25, o43;
    aan_inv_h(K, P[0][r], P[1][r], P[2][r], P[3][r], o07, o16, o25, o43);
    // sat_i16(rne(x / 64)): clamp to multiples of 64, add 1.5*2^29, keep the low 16 bits (rne_i16_bits<6>)
    auto fin = [&](f32x2 v) {
      f32x2 t;
      v.x = __builtin_amdgcn_fmed3f(v.x, -32768.0f * 64.0f, 32767.0f * 64.0f);
      v.y = __builtin_amdgcn_fmed3f(v.y, -32768.0f * 64.0f, 32767.0f * 64.0f);
      MDCT_PKA(t, v, K.magic, MDCT_K_HH);
      return t;
    };
    const f32x2 b07 = fin(o07), b16 = fin(o16), b25 = fin(o25), b43 = fin(o43);
    st_stream16(dst + (size_t)r * a.pitch_out, pack_lo16(__float_as_uint(b07.x), __float_as_uint(b16.x)), pack_lo16(__float_as_uint(b25.x), __float_as_uint(b43.y)),
                pack_lo16(__float_as_uint(b43.x), __float_as_uint(b25.y)), pack_lo16(__float_as_uint(b16.y), __float_as_uint(b07.y)));
  }
}

int main()
{
  const size_t W = 8192, H = 8192, bytes = W * H * 2;
  const int NS = 4;
  std::vector<int16_t *> A(NS), B(NS);
  std::vector<int16_t> host(W * H);
  for (size_t i = 0; i < W * H; i++) host[i] = (int16_t)(((i * 2654435761u) >> 20) & 0xFFF) - 2048; // 12-bit noise
  for (size_t i = 0; i < 64 * 8; i++) host[i] = (i & 1) ? 32767 : -32768;                            // saturating blocks
  for (int i = 0; i < NS; i++)
  {
    if (hipMalloc(&A[i], bytes) != hipSuccess || hipMalloc(&B[i], bytes) != hipSuccess) { puts("alloc failed"); return 1; }
    hipMemcpy(A[i], host.data(), bytes, hipMemcpyHostToDevice);
  }
  I16Args a;
  memset(&a, 0, sizeof(a));
  a.consts = DctConsts();
  a.pitch_in = a.pitch_out = W;
  a.bpr = W / 8;
  a.by0 = 0;
  a.nblocks = (uint32_t)(W / 8 * H / 8);
  for (int i = 0; i < 64; i++) { a.tb.qf[i] = 1.0f; a.tb.dq[i] = 1.0f / 64.0f; } // unused by the no-table round trip
  const DctConsts &C = a.consts;
  AanPk K;
  K.c707_382 = f32x2{C.c707, C.c382}; K.c541_1306 = f32x2{C.c541, C.c1306}; K.c1414_1847 = f32x2{C.c1414, C.c1847}; K.c1082_2613 = f32x2{C.c1082, C.c2613};
  K.magic = f32x2{C.magic23, C.magic29};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t st[2];
  hipStreamCreate(&st[0]); hipStreamCreate(&st[1]);
  struct V { const char *name; std::function<void(int)> f; std::vector<float> t; };
  std::vector<V> vs;
  auto args = [&](int s) { I16Args x = a; x.from = A[s]; x.to = B[s]; return x; };
  const uint32_t nwg = a.nblocks / 256;
  vs.push_back({"product k_i16<RT>", [&](int s) { launch_i16(args(s), MODE_ROUNDTRIP, false, 0); }, {}});
  vs.push_back({"stream copy", [&](int s) { launch_stream_copy(A[s], B[s], bytes, 256, 0); }, {}});
  vs.push_back({"packed RT 2w", [&](int s) { hipLaunchKernelGGL((v_rt_pk<2>), dim3(nwg), dim3(256), 0, 0, args(s), K); }, {}});
  vs.push_back({"packed RT 3w", [&](int s) { hipLaunchKernelGGL((v_rt_pk<3>), dim3(nwg), dim3(256), 0, 0, args(s), K); }, {}});
  vs.push_back({"packed RT 4w", [&](int s) { hipLaunchKernelGGL((v_rt_pk<4>), dim3(nwg), dim3(256), 0, 0, args(s), K); }, {}});
  vs.push_back({"packed RT 5w", [&](int s) { hipLaunchKernelGGL((v_rt_pk<5>), dim3(nwg), dim3(256), 0, 0, args(s), K); }, {}});
  vs.push_back({"packed RT 6w", [&](int s) { hipLaunchKernelGGL((v_rt_pk<6>), dim3(nwg), dim3(256), 0, 0, args(s), K); }, {}});
  vs.push_back({"packed RT 3w, 2 streams", [&](int s) { hipLaunchKernelGGL((v_rt_pk<3>), dim3(nwg), dim3(256), 0, st[s & 1], args(s), K); }, {}});
  vs.push_back({"product RT, 2 streams", [&](int s) { launch_i16(args(s), MODE_ROUNDTRIP, false, st[s & 1]); }, {}});
  {
    std::vector<int16_t> ref(W * H), got(W * H);
    vs[0].f(0); hipMemcpy(ref.data(), B[0], bytes, hipMemcpyDeviceToHost);
    size_t notid = 0;
    for (size_t i = 0; i < W * H; i++) notid += ref[i] != host[i];
    printf("product round trip differs from the input in %zu values\n", notid);
    for (size_t k = 2; k < vs.size(); k++)
    {
      hipMemset(B[0], 0x55, bytes);
      vs[k].f(0);
      if (hipMemcpy(got.data(), B[0], bytes, hipMemcpyDeviceToHost) != hipSuccess) { printf("!! %s: launch failed\n", vs[k].name); return 1; }
      size_t bad = 0;
      for (size_t i = 0; i < W * H; i++) bad += got[i] != ref[i];
      printf("%-24s %s (%zu mismatching values)\n", vs[k].name, bad ? "!! MISMATCH" : "bit-exact", bad);
    }
    fflush(stdout);
  }
  for (auto &v : vs) for (int i = 0; i < 400; i++) v.f(i % NS);
  hipDeviceSynchronize();
  for (int round = 0; round < 7; round++)
    for (auto &v : vs)
    {
      for (int i = 0; i < 40; i++) v.f(i % NS);
      hipDeviceSynchronize();
      const auto c0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 200; i++) v.f(i % NS);
      hipDeviceSynchronize();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
      v.t.push_back((float)(ms / 200));
    }
  for (auto &v : vs)
  {
    std::sort(v.t.begin(), v.t.end());
    printf("%-24s median %7.2f us  min %7.2f us\n", v.name, v.t[v.t.size() / 2] * 1e3, v.t[0] * 1e3);
  }
  return 0;
}
