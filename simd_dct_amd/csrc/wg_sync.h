// wg_sync.h -- the workgroup barrier these kernels use wherever LDS is handed from one wave to another.
// __syncthreads() on ROCm 7.2 fences the local address space at workgroup scope, for which the compiler considers LDS
// operations totally ordered and may emit s_barrier WITHOUT s_waitcnt lgkmcnt(0): a ds_write still queued behind its SIMD's
// other LDS requests when its wave passes the barrier can then be overtaken by an LDS operation another SIMD's wave issues
// after the barrier.  That happened (huffman_rows.h, the window loop; profiles/r03_soak_jpeg_scan.log), so the wait is explicit.
#pragma once
#include <hip/hip_runtime.h>

namespace mdct
{
__device__ __forceinline__ void wg_sync()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}
} // namespace mdct
