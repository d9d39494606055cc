// What limits an LDS-fed fp32 MFMA loop?  Waves per SIMD x barrier x staging ablation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE bit0: __syncthreads twice per chunk; bit1: stage 18 KB into LDS per chunk from global
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* src, int chunks, int lds_floats) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
  for (int i = tid; i < lds_floats; i += 256) sm[i] = src[i];
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int aoff = h * 128 + (wave >> 1) * 64 + li;
  const int boff = 72 * 128 + h * 256 + (wave & 1) * 64 + li;
  for (int c = 0; c < chunks; ++c) {
    if (MODE & 2) {
      float4 v[9];
      const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)(c & 15) * 9216);
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = s4[tid + i * 256];
#pragma unroll
      for (int i = 0; i < 9; ++i) reinterpret_cast<float4*>(sm)[tid + i * 256] = v[i];
    }
    if (MODE & 1) __syncthreads();
#pragma unroll
    for (int kp = 0; kp < 36; ++kp) {
      float a0 = sm[aoff + kp * 256], a1 = sm[aoff + kp * 256 + 32];
      float b0 = sm[boff + (kp % 9) * 3 + (kp / 9) * 512], b1 = sm[boff + (kp % 9) * 3 + (kp / 9) * 512 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (MODE & 1) __syncthreads();
  }
  float s = 0;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
void run(int blocks, int chunks, size_t lds, float* d, float* src) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, src, 2, 11264);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, src, chunks, 11264);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * chunks * 144 * 4096.0;
  printf("MODE=%d blocks=%4d (%.2f/CU) lds=%3zuKB: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, blocks / 256.0, lds / 1024, ms, flops / ms / 1e9);
}
int main() {
  float *d, *src; hipMalloc(&d, 4096 * 256 * 4); hipMalloc(&src, 16 * 9216 * 4 + 65536); { size_t n = 16 * 9216 + 16384; float* hsrc = (float*)malloc(n * 4); unsigned st = 12345u; for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; hsrc[i] = (getenv("ZERO") ? 0.f : ((st >> 8) * (1.0f / 8388608.0f) - 1.0f)); } hipMemcpy(src, hsrc, n * 4, hipMemcpyHostToDevice); }
  size_t l3 = 45 * 1024, l2 = 70 * 1024, l1 = 100 * 1024;   // 3, 2, 1 blocks per CU
  run<0>(256, 400, l1, d, src); run<0>(512, 400, l2, d, src); run<0>(768, 400, l3, d, src);
  run<1>(256, 400, l1, d, src); run<1>(512, 400, l2, d, src); run<1>(768, 400, l3, d, src);
  run<3>(256, 400, l1, d, src); run<3>(512, 400, l2, d, src); run<3>(768, 400, l3, d, src);
  run<3>(720, 400, l3, d, src);
  run<0>(720, 16, l3, d, src); run<1>(720, 16, l3, d, src); run<3>(720, 16, l3, d, src); run<3>(720, 32, l3, d, src); run<3>(768, 16, l3, d, src);
  return 0;
}
