// common.h -- shared host-side plumbing for libfrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/frcnn_hip.h"

namespace frcnn {

void set_error(const char* fmt, ...);

#define FR_HIP(expr)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      frcnn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                       __LINE__);                                                       \
      return FRCNN_ERR_HIP;                                                             \
    }                                                                                   \
  } while (0)

#define FR_CHECK(cond, ...)                  \
  do {                                       \
    if (!(cond)) {                           \
      frcnn::set_error(__VA_ARGS__);         \
      return FRCNN_ERR_ARG;                  \
    }                                        \
  } while (0)

#define FR_TRY(expr)            \
  do {                          \
    int r_ = (expr);            \
    if (r_ != FRCNN_OK) return r_; \
  } while (0)

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

// ---- per-kernel-class HIP-event profiler (bench.py's roofline leg) -----------------
// When enabled, every launch made through FR_LAUNCH is bracketed by two events on the
// launch stream; frcnn_prof_collect() synchronises and aggregates per class.
enum KClass {
  KC_CONV_IGEMM_K3 = 0,   // conv_igemm<3,...> fwd + dgrad (the dominant kernel)
  KC_CONV_IGEMM_OTHER,    // 1x1 / 5x5 / 7x7 instantiations
  KC_CONV_WGRAD_K3,
  KC_CONV_WGRAD_OTHER,
  KC_GEMM,                // cnet Linear
  KC_ELEMWISE,            // pool / prelu-bwd / pack / zero / scale
  KC_ROI,
  KC_RPN,
  KC_NMS,
  KC_OPTIM,
  KC_IMAGE,               // BatchIterator:processImage kernels (image.hip)
  KC_CONV_X3,             // conv_x3_kernel: 3x3 fwd + dgrad in the split-bf16 operand form (convx.hip) -- the dominant kernel
  KC_CONV_WGRADX,         // conv_wgradx_kernel (+ its slab fold): 3x3 weight gradient in the same form (wgradx.hip)
  KC_COUNT
};

bool prof_enabled(int klass);
void prof_before(int klass, hipStream_t s);
void prof_after(int klass, double flops, double bytes, hipStream_t s);

#ifdef FR_SKIPTEST   /* tools only: FRCNN_EXP_SKIP=<substring> drops matching launches (critical-path sensitivity) */
#define FR_SKIP_(kernel) (getenv("FRCNN_EXP_SKIP") && strstr(#kernel, getenv("FRCNN_EXP_SKIP")))
#else
#define FR_SKIP_(kernel) false
#endif
#define FR_LAUNCH(klass, flops, bytes, stream, kernel, grid, block, shmem, ...)        \
  do {                                                                                 \
    if (FR_SKIP_(kernel)) break;                                                       \
    if (frcnn::prof_enabled(klass)) frcnn::prof_before((klass), (stream));                  \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);               \
    if (frcnn::prof_enabled(klass)) frcnn::prof_after((klass), (flops), (bytes), (stream)); \
  } while (0)

#define FR_LAUNCH_CHECK()                                                          \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      frcnn::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),  \
                       __FILE__, __LINE__);                                        \
      return FRCNN_ERR_HIP;                                                        \
    }                                                                              \
  } while (0)

}  // namespace frcnn

// counter-based RNG of the throughput runs (dropout masks): element i of a stream draws splitmix64(seed*FNV + i)
#ifdef __HIPCC__
__host__ __device__ __forceinline__ unsigned long long frcnn_splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ float frcnn_keep_mask(unsigned long long seed, unsigned long long i, float p) {
  const unsigned long long r = frcnn_splitmix64(seed * 0x100000001B3ull + i);
  const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.f;  // keep with probability 1-p
}
#endif
