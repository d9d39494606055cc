// Semantics probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in (lane, j) for a given per-lane address.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int idx = mode == 0 ? l * 4 : (mode == 1 ? (l & 15) * 64 + (l >> 4) * 4 : ((l & 3) * 4 + (l >> 2) * 100 * 4));
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + idx));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (lane supplies element index: %s)\n", mode, mode == 0 ? "l*4" : mode == 1 ? "(l&15)*64+(l>>4)*4" : "(l&3)*4+(l>>2)*400");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
