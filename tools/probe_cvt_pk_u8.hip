// probe: rounding / saturation / NaN behaviour of v_cvt_pk_u8_f32 on gfx950 (is it RNE?)   hipcc --offload-arch=gfx950 tools/probe_cvt_pk_u8.hip -o tools/probe_cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n)
{
  const int i = threadIdx.x;
  if (i < n)
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main()
{
  const float v[] = {0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 126.5f, 127.5f, 254.5f, 255.0f, 255.49f, 255.5f, 256.0f, 300.0f, 1e10f, -0.4f, -0.5f, -0.6f, -1.0f, -1e10f, NAN, INFINITY, -INFINITY, 0.99999994f, 1.4999999f};
  const int n = sizeof(v) / sizeof(v[0]);
  float *d;
  unsigned *o, h[64];
  hipMalloc(&d, sizeof(v));
  hipMalloc(&o, sizeof(h));
  hipMemcpy(d, v, sizeof(v), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n);
  hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++)
    printf("%14.8g -> %u   (rne+sat would be %d)\n", v[i], h[i], std::isnan(v[i]) ? 0 : (int)fminf(fmaxf(nearbyintf(v[i]), 0.f), 255.f));
  return 0;
}
