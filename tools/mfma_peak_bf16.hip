// Sustained v_mfma_f32_32x32x16_bf16 rate on this chip (no memory traffic): the practical roof of the split-bf16 kernels.
// usage: mfma_peak_bf16.out   (prints TFLOP/s for 1/2 waves per SIMD, zero and random operands)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 ua, ub;
  unsigned t = threadIdx.x * 2654435761u + seed;
  ua = make_uint4(seed ? (t & 0x3FFF3FFFu) | 0x3F003F00u : 0, seed ? ((t >> 3) & 0x3FFF3FFFu) | 0x3F003F00u : 0, seed ? ((t >> 5) & 0x3FFF3FFFu) | 0x3F003F00u : 0, seed ? ((t >> 7) & 0x3FFF3FFFu) | 0x3F003F00u : 0);
  ub = make_uint4(ua.y ^ (seed ? 0x80000000u : 0), ua.z, ua.w, ua.x);
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run(int blocks, int iters, unsigned seed, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(d, 10, seed);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, iters, seed);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * NACC * 32768.0;
  printf("NACC=%d blocks=%d (%.0f waves/SIMD) iters=%d %s: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, blocks / 256.0, iters, seed ? "random" : "zero", ms, fl / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 4);
  for (unsigned seed : {0u, 7u}) {
    run<4>(256, 20000, seed, d);
    run<4>(512, 20000, seed, d);
    run<8>(256, 20000, seed, d);
    run<4>(512, 200000, seed, d);
  }
  return 0;
}
