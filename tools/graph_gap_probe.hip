// Per-launch cost of a chain of dependent tiny kernels in one stream: plain launches vs the same chain replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void busy(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x + 256 * blockIdx.x] = v; }
int main() {
  float* d; hipMalloc(&d, 1 << 24);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 200;
  for (int mode = 0; mode < 2; ++mode) {   // 0: tiny kernels, 1: ~20 us kernels filling the GPU
    auto chain = [&]() { for (int i = 0; i < N; ++i) { if (mode == 0) tiny<<<1, 64, 0, s>>>(d); else busy<<<2048, 256, 0, s>>>(d, 2000); } };
    chain(); hipStreamSynchronize(s);
    hipEventRecord(e0, s); chain(); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d plain : %.2f us per launch\n", mode, 1e3 * ms / N);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d graph : %.2f us per launch\n", mode, 1e3 * ms / N);
  }
  return 0;
}
