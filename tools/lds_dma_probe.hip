// How fast can a CU bring operands in?  Sustained bytes per clock and CU of LDS-DMA (global_load_lds, 16 B per lane) and of
// global_load_dwordx4 into registers, for the access patterns the GEMM / convolution kernels use:
//   private  : every block walks its own window (L2-resident 64 KB, or 4 MB = streamed from HBM)
//   shared   : all blocks walk the SAME window at the same time (an operand tile that many blocks read: same lines, same
//              L2 channel at the same moment)
//   pieces   : a wave instruction gathers 16 rows x 64 B at a row stride of 55 296 B (fp32 weights [n][k], 16 k per step)
// usage: tools/bin/lds_dma_probe      (prints GB/s and B/clk/CU at the 2.4 GHz nominal clock)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0: LDS-DMA, 1: registers
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ src, size_t window, int shared, int pieces, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = shared ? src : src + ((size_t)blockIdx.x * window) % (((size_t)1 << 31) - window);
  size_t off, stride;
  if (pieces) {   // lane -> row (lane >> 2) of this wave's 16 rows, 16-byte piece (lane & 3); a step advances 64 B along the row
    off = (size_t)(wave * 16 + (lane >> 2)) * 55296 + (lane & 3) * 16;
    stride = 64;
  } else {
    off = (size_t)wave * 1024 + lane * 16;
    stride = (size_t)nw * 1024;
  }
  const size_t wrap = pieces ? 55296 - 64 : window;
  size_t adv = 0;
  float4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + adv),
                                         (__attribute__((address_space(3))) void*)(sm + (wave * 8 + u) * 1024), 16, 0, 0);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(base + off + adv);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      adv += stride;
      if (adv >= wrap) adv -= wrap;
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 0) { __syncthreads(); acc.x = *reinterpret_cast<float*>(sm + threadIdx.x * 4); }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int MODE>
void run(const char* what, int blocks, int threads, size_t window, int shared, int pieces, int iters, const char* src, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = MODE == 0 ? (size_t)(threads / 64) * 8192 : 0;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  probe<MODE><<<blocks, threads, lds>>>(src, window, shared, pieces, 64, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<MODE><<<blocks, threads, lds>>>(src, window, shared, pieces, iters, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * (threads / 64) * iters * 8 * 1024.0;
  printf("%-9s %-22s blocks=%4d waves/block=%2d window=%7zu KB: %8.3f ms %8.1f GB/s  %6.1f B/clk/CU\n", MODE == 0 ? "LDS-DMA" : "registers", what,
         blocks, threads / 64, window >> 10, ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.4e9 / 256.0);
}
int main() {
  const size_t total = (size_t)1 << 31;   // 2 GB
  char* src; (void)hipMalloc(&src, total); (void)hipMemset(src, 1, total);
  float* out; (void)hipMalloc(&out, 4);
  const int it = 2000;
  for (int threads : {256, 512}) {
    run<0>("private", 256, threads, (size_t)64 << 10, 0, 0, it, src, out);
    run<1>("private", 256, threads, (size_t)64 << 10, 0, 0, it, src, out);
    run<0>("private (HBM)", 256, threads, (size_t)4 << 20, 0, 0, it, src, out);
    run<0>("shared by all", 256, threads, (size_t)64 << 10, 1, 0, it, src, out);
    run<1>("shared by all", 256, threads, (size_t)64 << 10, 1, 0, it, src, out);
    run<0>("shared by all, 2 MB", 256, threads, (size_t)2 << 20, 1, 0, it, src, out);
    run<0>("shared by all, 32 MB", 256, threads, (size_t)32 << 20, 1, 0, it, src, out);
    run<0>("64 B pieces, private", 256, threads, (size_t)8 << 20, 0, 1, it, src, out);
    run<0>("64 B pieces, shared", 256, threads, (size_t)8 << 20, 1, 1, it, src, out);
    run<1>("64 B pieces, shared", 256, threads, (size_t)8 << 20, 1, 1, it, src, out);
  }
  return 0;
}
