// amax.h -- the largest magnitude of a tensor, recorded by the kernel that writes the tensor, for the two-plane fp16 form of the
// split launches (convx.hip): the consumer scales the tensor by a power of two chosen from this number.
//
// A record is an array of floats: rec[0] = the number n of entries (an int's bit pattern), rec[1 .. n] = one maximum per block
// of the producing launch.  Every block of the producer stores its own entry (no atomics: 4 000 device-scope atomics on one
// address cost a pooling launch 30 us), every block of a consumer reduces the n entries (an L2-resident read of a few KB).
// Nothing is zeroed between steps: a launch rewrites all its n entries and the count.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"   // AMAX_REC

namespace frcnn {

// every thread of the block calls this once, with the largest magnitude it stored (block of up to 1024 threads, 1-D grid)
__device__ __forceinline__ void amax_store_block(float m, float* rec) {
  __shared__ float amax_part[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) amax_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 1; i < nw; ++i) m = fmaxf(m, amax_part[i]);
    rec[1 + blockIdx.x] = m;
    if (blockIdx.x == 0) rec[0] = __builtin_bit_cast(float, (int)gridDim.x);
  }
}

// -> the tensor's largest magnitude, in every thread of the block
__device__ __forceinline__ float amax_load_block(const float* __restrict__ rec) {
  __shared__ float amax_all[16];
  const int n = __builtin_bit_cast(int, rec[0]);
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, rec[1 + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) amax_all[threadIdx.x >> 6] = m;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  m = amax_all[0];
  for (int i = 1; i < nw; ++i) m = fmaxf(m, amax_all[i]);
  // (kernels that scale TWO tensors call this twice in a row: no wave may overwrite its slot for the second record while a slow
  // wave is still reading the first one's -- ADVICE r5)
  __syncthreads();
  return m;
}

}  // namespace frcnn
