/* sanitize_driver.c -- runs every oracle entry point on small planes under ASan/UBSan
 * (tests/test_oracle.py builds it with -fsanitize=address,undefined).  Test infrastructure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dct_oracle.h"

int main(void)
{
  static const size_t dims[][2] = {{64, 8}, {64, 16}, {128, 48}, {192, 64}};
  float lut[64];
  for (int i = 0; i < 64; i++)
    lut[i] = 0.1f + 0.37f * (float)i;
  unsigned seed = 12345u;
  for (size_t d = 0; d < sizeof(dims) / sizeof(dims[0]); d++)
  {
    const size_t W = dims[d][0], H = dims[d][1], n = W * H;
    /* exact-size heap buffers so that any out-of-bounds access trips the sanitizer */
    uint8_t *in = malloc(n), *out = malloc(n);
    int16_t *i16 = malloc(n * 2), *o16 = malloc(n * 2);
    float *f32 = malloc(n * 4), *of32 = malloc(n * 4);
    double *f64 = malloc(n * 8);
    for (size_t i = 0; i < n; i++)
    {
      seed = seed * 1664525u + 1013904223u;
      in[i] = (uint8_t)(seed >> 24);
      i16[i] = (int16_t)(seed >> 16);
      f32[i] = (float)in[i] - 77.25f;
    }
    const size_t ranges[][2] = {{0, H}, {0, 2 * H}, {16, 32}, {H, H}, {0, 0}};
    for (size_t r = 0; r < 5; r++)
    {
      memset(out, 0xA5, n);
      if (orc_q32_avx(in, out, lut, W, H, ranges[r][0], ranges[r][1])) return 10;
      if (orc_stereo_scalar(in, out, lut, W, H, ranges[r][0], ranges[r][1]) && H % 16 == 0) return 11;
      if (orc_encq_scalar(in, out, lut, W, H, ranges[r][0], ranges[r][1])) return 12;
      if (H % 16 == 0)
      {
        if (orc_stereo_sse(in, out, lut, W, H, ranges[r][0], ranges[r][1])) return 13;
      }
      if (orc_encq_sse(in, out, lut, W, H, ranges[r][0], ranges[r][1])) return 14;
    }
    if (orc_q32_native(in, out, W, lut, W, H, 0, H / 8)) return 20;
    const float *tables[2] = {NULL, lut};
    for (int t = 0; t < 2; t++)
    {
      if (orc_fwd_i16(i16, o16, W, W, tables[t], W, H, 0, H / 8)) return 30;
      if (orc_inv_i16(i16, o16, W, W, tables[t], W, H, 0, H / 8)) return 31;
      if (orc_roundtrip_i16(i16, o16, W, W, tables[t], W, H, 0, H / 8)) return 32;
      if (orc_fwd_u8_i16(in, o16, W, W, tables[t], t, W, H, 0, H / 8)) return 33;
      if (orc_inv_i16_u8(i16, out, W, W, tables[t], t, W, H, 0, H / 8)) return 34;
    }
    if (orc_fwd_f32(f32, of32, W, W, W, H, 0, H / 8)) return 40;
    if (orc_inv_f32(f32, of32, W, W, W, H, 0, H / 8)) return 41;
    if (orc_fwd_f64ref(f32, f64, W, W, W, H, 0, H / 8)) return 42;
    free(in); free(out); free(i16); free(o16); free(f32); free(of32); free(f64);
  }
  puts("sanitize ok");
  return 0;
}
