// Probe of the v_mfma_f32_32x32x2_f32 operand / result lane maps on the real chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {  // A[32][2], B[2][32], D[32][32]
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = c[r];
  }
}
int main() {
  float hA[64], hB[64], hD[1024];
  for (int i = 0; i < 64; ++i) { hA[i] = 1 + i * 0.5f; hB[i] = 3 - i * 0.25f + (i % 7); }
  float *dA, *dB, *dD;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double w = (double)hA[i * 2] * hB[j] + (double)hA[i * 2 + 1] * hB[32 + j];
    worst = fmax(worst, fabs(w - hD[i * 32 + j]));
  }
  printf("mfma 32x32x2 f32 mapping worst err %g  D[0][0..3]=%g %g %g %g\n", worst, hD[0], hD[1], hD[2], hD[3]);
  return 0;
}
