// Probe for conv_wgradx's partial-sum fold: do fp32 atomic adds WITHOUT the sc1 (device-scope) bit execute in the XCD's own
// L2 and stay coherent between the CUs of that XCD?  Every block reads its XCC id from the hardware register and adds 1.0f
// to each element of that XCC's private array; afterwards element sums must equal the number of blocks that ran on the XCC.
// Prints the XCC histogram, whether blockIdx & 7 == XCC id, the verdict and the atomic throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define XCC_ID_REG (20 | (0 << 6) | (3 << 11))   // hwreg(HW_REG_XCC_ID, 0, 4)

// `share` blocks of an XCC add into the same array (32: all of them, 1: every block has its own)
__global__ void probe(float* buf, int n, int* xcc_of_block, int share) {
  const unsigned xcc = __builtin_amdgcn_s_getreg(XCC_ID_REG) & 15u;
  if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = (int)xcc;
  float* dst = buf + ((size_t)(xcc & 7) * 32 + (blockIdx.x >> 3) / share) * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

int main() {
  const int n = 9 * 64 * 64, blocks = 256;
  float* buf; int* xb;
  hipMalloc(&buf, sizeof(float) * 8 * 32 * n); hipMalloc(&xb, sizeof(int) * blocks);
  hipMemset(buf, 0, sizeof(float) * 8 * 32 * n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<<<blocks, 256>>>(buf, n, xb, 32);
  hipDeviceSynchronize();
  std::vector<float> h(8 * 32 * n); std::vector<int> hx(blocks);
  hipMemcpy(h.data(), buf, sizeof(float) * 8 * 32 * n, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), xb, sizeof(int) * blocks, hipMemcpyDeviceToHost);
  int cnt[16] = {}, agree = 0;
  for (int b = 0; b < blocks; ++b) { cnt[hx[b] & 15]++; agree += (hx[b] == (b & 7)); }
  printf("blocks per XCC:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]); printf("   blockIdx&7 == XCC for %d of %d blocks\n", agree, blocks);
  int bad = 0;
  for (int x = 0; x < 8; ++x) for (int i = 0; i < n; ++i) bad += (h[(size_t)x * 32 * n + i] != (float)cnt[x]);
  printf("elements whose sum differs from the block count of their XCC: %d of %d\n", bad, 8 * n);
  for (int share : {32, 8, 4, 2, 1}) {
    hipMemset(buf, 0, sizeof(float) * 8 * 32 * n);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) probe<<<blocks, 256>>>(buf, n, xb, share);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%d blocks x %d atomics, %2d blocks per array: %.1f us per launch (%.0f G atomics/s)\n", blocks, n, share, ms * 100,
           blocks * (double)n / (ms * 1e-4) * 1e-9);
  }
  return bad != 0;
}
