// elem.hip -- bandwidth-bound pieces of the proposal network and the optimiser:
//   nn.PReLU / nn.SpatialDropout / nn.SpatialMaxPooling(2,2,2,2):ceil() forward+backward
//   (models/model_utilities.lua:9-12,23), bias/slope gradient reductions, flat-buffer ops
//   (objective.lua:49,200) and optim.rmsprop (main.lua:133).
// No MFMA here: these are gather / argmax / streaming kernels; the design rule is coalesced
// 16-byte accesses where the layout allows and one wave-level reduction + one atomic per block.
#include <algorithm>

#include <mutex>
#include <unordered_map>

#include "kernels.h"
#include "amax.h"

namespace frcnn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    r = threadIdx.x < (blockDim.x >> 6) ? sh[threadIdx.x] : 0.f;
    r = wave_sum(r);
  }
  __syncthreads();
  return r;
}

// the same in fp64 (the PReLU slope gradient is ONE number summed over a whole layer with cancellation: its partial sums are
// carried in double so that the result is good to fp32 rounding whatever the layer size)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < 64) {
    r = threadIdx.x < (blockDim.x >> 6) ? sh[threadIdx.x] : 0.0;
    r = wave_sum_d(r);
  }
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------- flat buffer ops
int fill_zero(void* p, size_t bytes, hipStream_t s) {
  if (prof_enabled(KC_ELEMWISE)) prof_before(KC_ELEMWISE, s);
  FR_HIP(hipMemsetAsync(p, 0, bytes, s));
  if (prof_enabled(KC_ELEMWISE)) prof_after(KC_ELEMWISE, 0, (double)bytes, s);
  return FRCNN_OK;
}

__global__ void scale_kernel(float* __restrict__ x, long n, float sc) {
  long n4 = n >> 2;
  float4* x4 = reinterpret_cast<float4*>(x);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = x4[i];
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    x4[i] = v;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    x[i] *= sc;
}
int scale_inplace(float* x, long n, float sc, hipStream_t s) {
  FR_CHECK(((uintptr_t)x & 15) == 0, "scale_inplace: buffer must be 16-byte aligned");
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n / 4, 256)), 2048);
  FR_LAUNCH(KC_ELEMWISE, 0, n * 8.0, s, scale_kernel, dim3(grid), dim3(256), 0, x, n, sc);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ void add_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] += x[i];
}
int add_inplace(float* y, const float* x, long n, hipStream_t s) {
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 2048);
  FR_LAUNCH(KC_ELEMWISE, 0, n * 12.0, s, add_kernel, dim3(grid), dim3(256), 0, y, x, n);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ void fill_kernel(float* x, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = v;
}
int fill_value(float* x, long n, float v, hipStream_t s) {
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 2048);
  FR_LAUNCH(KC_ELEMWISE, 0, n * 4.0, s, fill_kernel, dim3(grid), dim3(256), 0, x, n, v);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- activation (tests)
__global__ void act_forward_kernel(const float* __restrict__ x, int C, long hw, const float* slope,
                                   const float* scale, float* __restrict__ y) {
  const float a = slope ? *slope : 1.f;
  long total = (long)C * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (slope) v = v > 0.f ? v : a * v;
    if (scale) v *= scale[i / hw];
    y[i] = v;
  }
}
int act_forward(const float* x, int C, long hw, const float* slope, const float* scale, float* y,
                hipStream_t s) {
  long total = (long)C * hw;
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(total, 256)), 2048);
  FR_LAUNCH(KC_ELEMWISE, 0, total * 8.0, s, act_forward_kernel, dim3(grid), dim3(256), 0, x, C, hw, slope,
            scale, y);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- max pool (fused activation)
// One thread per output element; output rows are contiguous so loads of the two input rows are
// 8-byte-per-lane strided reads (coalesced across the wave).
__global__ void maxpool_act_forward_kernel(const float* __restrict__ x, int C, int H, int W, int Ho,
                                           int Wo, const float* slope, const float* scale,
                                           float* __restrict__ out, unsigned char* __restrict__ idx, float* amax) {
  const float a = slope ? *slope : 1.f;
  long total = (long)C * Ho * Wo;
  float am = 0.f;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int ox = (int)(t % Wo);
    long r = t / Wo;
    int oy = (int)(r % Ho);
    int c = (int)(r / Ho);
    const float sc = scale ? scale[c] : 1.f;
    const float* xp = x + (size_t)c * H * W;
    float best = -3.402823466e+38f;
    int bi = 0;
    bool any = false;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      int y = oy * 2 + dy;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        int xx = ox * 2 + dx;
        if (y < H && xx < W) {
          float v = xp[(size_t)y * W + xx];
          if (slope) v = v > 0.f ? v : a * v;
          if (scale) v *= sc;
          if (!any || v > best) { best = v; bi = dy * 2 + dx; any = true; }
        }
      }
    }
    out[t] = best;
    idx[t] = (unsigned char)bi;
    am = fmaxf(am, fabsf(best));
  }
  if (amax) amax_store_block(am, amax);
}
int maxpool_act_forward(const float* x, int C, int H, int W, const float* slope, const float* scale,
                        float* out, unsigned char* idx, hipStream_t s, float* amax) {
  int Ho = (H - 2 + 1) / 2 + 1, Wo = (W - 2 + 1) / 2 + 1;  // ceil((n-2)/2)+1
  long total = (long)C * Ho * Wo;
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(total, 256)), 4096);
  FR_LAUNCH(KC_ELEMWISE, 0, (double)C * H * W * 4.0 + total * 5.0, s, maxpool_act_forward_kernel, dim3(grid),
            dim3(256), 0, x, C, H, W, Ho, Wo, slope, scale, out, idx, amax);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// gx[c][y][x] = (argmax of window == this position ? gpool : 0) * scale[c] * prelu'(x)
// One block per (channel, slab of rows): the bias-gradient and slope-gradient partial sums are
// reduced in the block and leave through one atomic each.
// 512 threads: two waves per SIMD at 40 registers.  In the backward phase this pass runs beside conv_wgradx, whose one block per CU
// holds 392 of a SIMD's 512 registers: a 1024-thread block (4 x 40 registers per SIMD) does not fit beside it and waited for
// weight-gradient blocks to retire -- 60-70 us live for a 15-us pass on the dependent chain (profiles/r05_step_timeline.txt).
// Same-box A/B of the step: 2.824 (1024) / 2.799 (768) / 2.796 ms (512).
#ifndef ACT_BWD_THREADS
#define ACT_BWD_THREADS 512
#endif
template <bool POOLED, bool VEC>
__global__ __launch_bounds__(ACT_BWD_THREADS) void act_backward_kernel(const float* __restrict__ gin, const unsigned char* __restrict__ idx,
                                    const float* __restrict__ x, int C, int H, int W, int Ho, int Wo,
                                    const float* slope, const float* scale, float* __restrict__ gx,
                                    float* gbias, float* gslope, int chunks, float* part_b, float* part_a, float* amax) {
  __shared__ float sh[16];
  __shared__ double shd[16];
  float am = 0.f;   // largest magnitude written (amax.h)
  const int c = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const long hw = (long)H * W;
  long per = cdivl(hw, chunks);
  if (VEC) per = (per + 3) & ~3L;
  const long beg = chunk * per, end = beg + per < hw ? beg + per : hw;
  const float a = slope ? *slope : 1.f;
  const float sc = scale ? scale[c] : 1.f;
  const bool has_slope = slope != nullptr;
  float sb = 0.f;
  double sa = 0.0;
  if (VEC) {
    // 4 consecutive elements of one row per thread and load: 16-byte loads/stores (W % 4 == 0, Wo even).
    // Two such groups are in flight per thread (all loads issued before the first use) and the indices are
    // 32-bit (a channel plane has < 2^31 elements): the kernel is a pure HBM stream and was latency-bound.
    const float* xc = x + (size_t)c * hw;
    const float* gc = POOLED ? gin + (size_t)c * Ho * Wo : gin + (size_t)c * hw;
    const unsigned char* ic = POOLED ? idx + (size_t)c * Ho * Wo : nullptr;
    float* oc = gx + (size_t)c * hw;
    const unsigned ibeg = (unsigned)beg, iend = (unsigned)end, uW = (unsigned)W, stride = 4u * blockDim.x;
    auto fetch = [&](unsigned i, bool ok, float* g, float4& xv4) {
      if (!ok) { g[0] = g[1] = g[2] = g[3] = 0.f; xv4 = make_float4(1.f, 1.f, 1.f, 1.f); return; }
      if (POOLED) {
        const unsigned y = i / uW, xx = i - y * uW;
        const unsigned po = (y >> 1) * (unsigned)Wo + (xx >> 1);
        const float2 gp = *reinterpret_cast<const float2*>(gc + po);
        const unsigned short ib = *reinterpret_cast<const unsigned short*>(ic + po);
        const unsigned char code = (unsigned char)((y & 1) * 2);
        g[0] = ((ib & 0xff) == code) ? gp.x : 0.f;
        g[1] = ((ib & 0xff) == code + 1) ? gp.x : 0.f;
        g[2] = ((ib >> 8) == code) ? gp.y : 0.f;
        g[3] = ((ib >> 8) == code + 1) ? gp.y : 0.f;
      } else {
        const float4 gv = *reinterpret_cast<const float4*>(gc + i);
        g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
      }
      xv4 = *reinterpret_cast<const float4*>(xc + i);
    };
    auto finish = [&](unsigned i, const float* g, const float4& xv4) {
      const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gj = g[j];
        if (scale) gj *= sc;
        r[j] = gj;
        if (has_slope && !(xv[j] > 0.f)) { r[j] = a * gj; sa += (double)xv[j] * (double)gj; }
        sb += r[j];
        am = fmaxf(am, fabsf(r[j]));
      }
      *reinterpret_cast<float4*>(oc + i) = make_float4(r[0], r[1], r[2], r[3]);
    };
    for (unsigned i = ibeg + 4u * threadIdx.x; i < iend; i += 2u * stride) {
      const unsigned i2 = i + stride;
      const bool ok2 = i2 < iend;
      float g0[4], g1[4];
      float4 x0, x1;
      fetch(i, true, g0, x0);
      fetch(i2, ok2, g1, x1);
      finish(i, g0, x0);
      if (ok2) finish(i2, g1, x1);
    }
  } else {
    for (long i = beg + threadIdx.x; i < end; i += blockDim.x) {
      float g;
      if (POOLED) {
        int y = (int)(i / W), xx = (int)(i - (long)y * W);
        int oy = y >> 1, ox = xx >> 1;
        long po = ((long)c * Ho + oy) * Wo + ox;
        g = (idx[po] == (unsigned char)((y & 1) * 2 + (xx & 1))) ? gin[po] : 0.f;
      } else {
        g = gin[(size_t)c * hw + i];
      }
      if (scale) g *= sc;
      float xv = x[(size_t)c * hw + i];
      float r = g;
      if (has_slope) {
        if (!(xv > 0.f)) { r = a * g; sa += (double)xv * (double)g; }
      }
      gx[(size_t)c * hw + i] = r;
      sb += r;
      am = fmaxf(am, fabsf(r));
    }
  }
  if (amax) amax_store_block(am, amax);
  float tb = block_sum(sb, sh);
  if (part_b) {   // deterministic mode: partials to scratch, folded in index order by fold_partials_kernel
    if (threadIdx.x == 0) part_b[blockIdx.x] = tb;
    float ta = (slope && gslope) ? (float)block_sum_d(sa, shd) : 0.f;
    if (threadIdx.x == 0) part_a[blockIdx.x] = ta;
    return;
  }
  if (threadIdx.x == 0 && gbias) unsafeAtomicAdd(gbias + c, tb);
  if (slope && gslope) {
    float ta = (float)block_sum_d(sa, shd);
    if (threadIdx.x == 0) unsafeAtomicAdd(gslope, ta);
  }
}

// gbias[c] += sum_chunk part_b[c*chunks + chunk] (index order); *gslope += sum_b part_a[b] (fixed strided order + fixed tree)
__global__ void fold_partials_kernel(const float* __restrict__ part_b, const float* __restrict__ part_a, int C, int chunks,
                                     float* gbias, float* gslope) {
  __shared__ float sh[16];
  if (gbias && part_b)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float v = 0.f;
      for (int k = 0; k < chunks; ++k) v += part_b[(size_t)c * chunks + k];
      gbias[c] += v;
    }
  if (gslope && part_a) {
    float v = 0.f;
    for (int b = threadIdx.x; b < C * chunks; b += blockDim.x) v += part_a[b];
    const float t = block_sum(v, sh);
    if (threadIdx.x == 0) *gslope += t;
  }
}
static int fold_partials(const float* part_b, const float* part_a, int C, int chunks, float* gbias, float* gslope, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, (double)C * chunks * 8.0, s, fold_partials_kernel, dim3(1), dim3(256), 0, part_b, part_a, C, chunks,
            gbias, gslope);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

static bool g_deterministic = getenv("FRCNN_DETERMINISTIC") && atoi(getenv("FRCNN_DETERMINISTIC")) != 0;   // default of the option
bool deterministic() { return g_deterministic; }
void set_deterministic(bool on) { g_deterministic = on; }
static std::mutex g_det_mu;
static std::unordered_map<hipStream_t, std::pair<float*, size_t>> g_det_ws;
int det_workspace(hipStream_t s, size_t floats, float** out) {
  std::lock_guard<std::mutex> lk(g_det_mu);
  auto& e = g_det_ws[s];
  if (e.second < floats) {
    if (e.first) FR_HIP(hipFree(e.first));
    e.first = nullptr; e.second = 0;
    size_t n = std::max<size_t>(floats, 1 << 16);
    FR_HIP(hipMalloc((void**)&e.first, n * 4));
    e.second = n;
  }
  *out = e.first;
  return FRCNN_OK;
}

// Blocks of 1024 threads, ~512-768 of them: every block ends with ONE atomic on the single slope-gradient
// address, and 2048 of those serialise for ~15 us in the memory-side atomic unit (measured: a 26 MB and a
// 138 MB launch both took 35 us; without the atomic 12-21 us).
static int act_bwd_chunks(int C, long hw) {
  long want = cdivl(512, C);
  long maxc = cdivl(hw, 4096);
  return (int)std::max<long>(1, std::min<long>(want, maxc));
}

int maxpool_act_backward(const float* gpool, const unsigned char* idx, const float* x, int C, int H,
                         int W, const float* slope, const float* scale, float* gx, float* gbias,
                         float* gslope, hipStream_t s, float* amax) {
  int Ho = (H - 2 + 1) / 2 + 1, Wo = (W - 2 + 1) / 2 + 1;
  int chunks = act_bwd_chunks(C, (long)H * W);
  // (more blocks than a record has entries: the magnitude of gx is taken by a pass of its own below, as every other record producer does)
  float* const amax_after = (amax && (long)C * chunks > AMAX_MAX_BLOCKS) ? amax : nullptr;
  if (amax_after) amax = nullptr;
  float *pb = nullptr, *pa = nullptr;
  if (deterministic()) { FR_TRY(det_workspace(s, (size_t)2 * C * chunks, &pb)); pa = pb + (size_t)C * chunks; }
  const bool vec = (W % 4 == 0) && (Wo % 2 == 0) && (((uintptr_t)gpool | (uintptr_t)x | (uintptr_t)gx) % 16 == 0) &&
                   ((uintptr_t)idx % 2 == 0);
  if (vec)
    FR_LAUNCH(KC_ELEMWISE, 0, (double)C * H * W * 9.25, s, (act_backward_kernel<true, true>), dim3(C * chunks),
              dim3(ACT_BWD_THREADS), 0, gpool, idx, x, C, H, W, Ho, Wo, slope, scale, gx, gbias, gslope, chunks, pb, pa, amax);
  else
    FR_LAUNCH(KC_ELEMWISE, 0, (double)C * H * W * 9.25, s, (act_backward_kernel<true, false>), dim3(C * chunks),
              dim3(ACT_BWD_THREADS), 0, gpool, idx, x, C, H, W, Ho, Wo, slope, scale, gx, gbias, gslope, chunks, pb, pa, amax);
  FR_LAUNCH_CHECK();
  if (pb) FR_TRY(fold_partials(gbias ? pb : nullptr, (slope && gslope) ? pa : nullptr, C, chunks, gbias, gslope, s));
  if (amax_after) FR_TRY(tensor_absmax(gx, (long)C * H * W, amax_after, s));
  return FRCNN_OK;
}

int act_backward(const float* gy, const float* x, int C, long hw, const float* slope,
                 const float* scale, float* gx, float* gbias, float* gslope, hipStream_t s, float* amax) {
  int chunks = act_bwd_chunks(C, hw);
  // (more blocks than a record has entries: the magnitude of gx is taken by a pass of its own below, as every other record producer does)
  float* const amax_after = (amax && (long)C * chunks > AMAX_MAX_BLOCKS) ? amax : nullptr;
  if (amax_after) amax = nullptr;
  float *pb = nullptr, *pa = nullptr;
  if (deterministic()) { FR_TRY(det_workspace(s, (size_t)2 * C * chunks, &pb)); pa = pb + (size_t)C * chunks; }
  const bool vec = (hw % 4 == 0) && (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx) % 16 == 0);
  if (vec)
    FR_LAUNCH(KC_ELEMWISE, 0, (double)C * hw * 12.0, s, (act_backward_kernel<false, true>), dim3(C * chunks),
              dim3(ACT_BWD_THREADS), 0, gy, (const unsigned char*)nullptr, x, C, 1, (int)hw, 1, 1, slope, scale, gx,
              gbias, gslope, chunks, pb, pa, amax);
  else
    FR_LAUNCH(KC_ELEMWISE, 0, (double)C * hw * 12.0, s, (act_backward_kernel<false, false>), dim3(C * chunks),
              dim3(ACT_BWD_THREADS), 0, gy, (const unsigned char*)nullptr, x, C, 1, (int)hw, 1, 1, slope, scale, gx,
              gbias, gslope, chunks, pb, pa, amax);
  FR_LAUNCH_CHECK();
  if (pb) FR_TRY(fold_partials(gbias ? pb : nullptr, (slope && gslope) ? pa : nullptr, C, chunks, gbias, gslope, s));
  if (amax_after) FR_TRY(tensor_absmax(gx, (long)C * hw, amax_after, s));
  return FRCNN_OK;
}

// (16-byte loads where the chunk allows: the first layer's bias gradient of vgg_large sums 64 x 600 x 1000 values -- 120 us one value
// per load and thread, round 6)
__global__ void channel_sum_kernel(const float* __restrict__ g, long hw, float* gbias, int chunks, float* part_b) {
  __shared__ float sh[16];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  long per = cdivl(hw, chunks);
  per = (per + 3) & ~3L;
  const long beg = chunk * per, end = beg + per < hw ? beg + per : hw;
  const float* gc = g + (size_t)c * hw;
  float sb = 0.f;
  if (beg < end) {
    if ((reinterpret_cast<uintptr_t>(gc + beg) & 15) == 0) {
      const long n4 = (end - beg) >> 2;
      const float4* g4 = reinterpret_cast<const float4*>(gc + beg);
      float s0 = 0.f, s1 = 0.f;
      long i = threadIdx.x;
      for (; i + blockDim.x < n4; i += 2 * blockDim.x) {
        const float4 a = g4[i], b = g4[i + blockDim.x];
        s0 += (a.x + a.y) + (a.z + a.w); s1 += (b.x + b.y) + (b.z + b.w);
      }
      if (i < n4) { const float4 a = g4[i]; s0 += (a.x + a.y) + (a.z + a.w); }
      sb = s0 + s1;
      for (long j = beg + (n4 << 2) + threadIdx.x; j < end; j += blockDim.x) sb += gc[j];
    } else {
      for (long i = beg + threadIdx.x; i < end; i += blockDim.x) sb += gc[i];
    }
  }
  float tb = block_sum(sb, sh);
  if (part_b) { if (threadIdx.x == 0) part_b[blockIdx.x] = tb; return; }
  if (threadIdx.x == 0) unsafeAtomicAdd(gbias + c, tb);
}
int channel_sum(const float* g, int C, long hw, float* gbias, hipStream_t s) {
  int chunks = act_bwd_chunks(C, hw);
  float* pb = nullptr;
  if (deterministic()) FR_TRY(det_workspace(s, (size_t)C * chunks, &pb));
  FR_LAUNCH(KC_ELEMWISE, 0, (double)C * hw * 4.0, s, channel_sum_kernel, dim3(C * chunks), dim3(256), 0, g,
            hw, gbias, chunks, pb);
  FR_LAUNCH_CHECK();
  if (pb) FR_TRY(fold_partials(pb, nullptr, C, chunks, gbias, nullptr, s));
  return FRCNN_OK;
}

// gb[o] += sum_r g[r][o]  (row-major R x O; nn.Linear bias gradient): 64 columns x 16 row groups per block
__global__ __launch_bounds__(1024) void channel_sum_cols_kernel(const float* __restrict__ g, int R, int O, float* gb) {
  __shared__ float sh[16 * 64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int o = blockIdx.x * 64 + tx;
  float sacc = 0.f;
  if (o < O) for (int r = ty; r < R; r += 16) sacc += g[(size_t)r * O + o];
  sh[ty * 64 + tx] = sacc;
  __syncthreads();
  if (ty == 0 && o < O) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k * 64 + tx];
    gb[o] += t;
  }
}
int channel_sum_cols(const float* g, int R, int O, float* gb, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * O * 4.0, s, channel_sum_cols_kernel, dim3(cdiv(O, 64)), dim3(64, 16), 0, g, R,
            O, gb);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- sparse anchor-head backward helpers
// delta_outputs[1..4] are non-zero only at the sampled anchors (objective.lua:91-134), so the
// backward of an anchor head can run on P gathered positions instead of the whole map.
// dst[c][p] = src[c][pos[p]]  (and dst_act = prelu(dst) when slope != null)
__global__ void gather_positions_kernel(const float* __restrict__ src, int C, long hw, const int* __restrict__ pos, int P,
                                        float* __restrict__ dst, const float* slope, float* __restrict__ dst_act) {
  const long total = (long)C * P;
  const float a = slope ? *slope : 1.f;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t / P), pi = (int)(t - (long)c * P);
    const float v = src[(size_t)c * hw + pos[pi]];
    dst[t] = v;
    if (dst_act) dst_act[t] = v > 0.f ? v : a * v;
  }
}
int gather_positions(const float* src, int C, long hw, const int* pos, int P, float* dst, const float* slope,
                     float* dst_act, hipStream_t s) {
  long total = (long)C * P;
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(total, 256)), 2048);
  FR_LAUNCH(KC_ELEMWISE, 0, total * 8.0, s, gather_positions_kernel, dim3(grid), dim3(256), 0, src, C, hw, pos, P, dst,
            slope, dst_act);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// col[p][(c,ky,kx)] = X[c][y_p + ky][x_p + kx]   (valid convolution: always in bounds); pos = y*Wo + x
__global__ void im2col_positions_kernel(const float* __restrict__ X, int C, int H, int W, int k, int Wo,
                                        const int* __restrict__ pos, int P, float* __restrict__ col) {
  const int ckk = C * k * k;
  const long total = (long)P * ckk;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int pi = (int)(t / ckk), q = (int)(t - (long)pi * ckk);
    const int kx = q % k, ky = (q / k) % k, c = q / (k * k);
    const int y = pos[pi] / Wo, x = pos[pi] - y * Wo;
    col[t] = X[((size_t)c * H + y + ky) * W + x + kx];
  }
}
int im2col_positions(const float* X, int C, int H, int W, int k, int Wo, const int* pos, int P, float* col, hipStream_t s) {
  long total = (long)P * C * k * k;
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(total, 256)), 4096);
  FR_LAUNCH(KC_ELEMWISE, 0, total * 8.0, s, im2col_positions_kernel, dim3(grid), dim3(256), 0, X, C, H, W, k, Wo, pos, P, col);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// gX[c][y_p + ky][x_p + kx] += col[p][(c,ky,kx)]  (neighbourhoods of different anchors overlap: atomics)
__global__ void col2im_positions_add_kernel(const float* __restrict__ col, int C, int H, int W, int k, int Wo,
                                            const int* __restrict__ pos, int P, float* __restrict__ gX) {
  const int ckk = C * k * k;
  const long total = (long)P * ckk;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int pi = (int)(t / ckk), q = (int)(t - (long)pi * ckk);
    const int kx = q % k, ky = (q / k) % k, c = q / (k * k);
    const int y = pos[pi] / Wo, x = pos[pi] - y * Wo;
    unsafeAtomicAdd(gX + ((size_t)c * H + y + ky) * W + x + kx, col[t]);
  }
}
// deterministic variant: one thread per element of gX gathers, in position order, every patch entry that lands on it
__global__ void col2im_positions_gather_kernel(const float* __restrict__ col, int C, int H, int W, int k, int Wo,
                                               const int* __restrict__ pos, int P, float* __restrict__ gX) {
  const int ckk = C * k * k;
  const long total = (long)C * H * W;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(t % W), yy = (int)((t / W) % H), c = (int)(t / ((long)W * H));
    float v = 0.f;
    bool any = false;
    for (int pi = 0; pi < P; ++pi) {
      const int y = pos[pi] / Wo, x = pos[pi] - y * Wo;
      const int ky = yy - y, kx = xx - x;
      if (ky >= 0 && ky < k && kx >= 0 && kx < k) { v += col[(size_t)pi * ckk + (c * k + ky) * k + kx]; any = true; }
    }
    if (any) gX[t] += v;
  }
}
int col2im_positions_add(const float* col, int C, int H, int W, int k, int Wo, const int* pos, int P, float* gX,
                         hipStream_t s) {
  if (deterministic()) {
    long n = (long)C * H * W;
    int g = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 8192);
    FR_LAUNCH(KC_ELEMWISE, 0, n * 8.0, s, col2im_positions_gather_kernel, dim3(g), dim3(256), 0, col, C, H, W, k, Wo, pos, P, gX);
    FR_LAUNCH_CHECK();
    return FRCNN_OK;
  }
  long total = (long)P * C * k * k;
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(total, 256)), 4096);
  FR_LAUNCH(KC_ELEMWISE, 0, total * 8.0, s, col2im_positions_add_kernel, dim3(grid), dim3(256), 0, col, C, H, W, k, Wo, pos,
            P, gX);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- RNG (throughput runs)
__global__ void bernoulli_kernel(float* m, long n, float p, unsigned long long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    m[i] = frcnn_keep_mask(seed, (unsigned long long)i, p);
}
__global__ void bernoulli_multi_kernel(DropoutJobs j) {
  const int k = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < j.C[k]; i += gridDim.x * blockDim.x)
    j.ptr[k][i] = frcnn_keep_mask(j.seed[k], (unsigned long long)i, j.p[k]);
}
// the SpatialDropout keep vectors of all blocks of a forward pass in ONE launch
int dropout_channel_masks(const DropoutJobs& j, hipStream_t s) {
  if (j.n <= 0) return FRCNN_OK;
  int maxC = 0;
  for (int k = 0; k < j.n; ++k) maxC = std::max(maxC, j.C[k]);
  FR_LAUNCH(KC_ELEMWISE, 0, maxC * 4.0 * j.n, s, bernoulli_multi_kernel, dim3(cdiv(maxC, 256), j.n), dim3(256), 0, j);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int dropout_channel_mask(float* scale, int C, float p, unsigned long long seed, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, C * 4.0, s, bernoulli_kernel, dim3(cdiv(C, 256)), dim3(256), 0, scale, (long)C,
            p, seed);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int dropout_mask(float* mask, long n, float p, unsigned long long seed, hipStream_t s) {
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 1024);
  FR_LAUNCH(KC_ELEMWISE, 0, n * 4.0, s, bernoulli_kernel, dim3(grid), dim3(256), 0, mask, n, p, seed);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- optim.rmsprop
// m = alpha*m + (1-alpha)*g*g ; x -= lr * g / (sqrt(m) + eps).  5 streams of n floats: HBM-bound.
// SCALE: g is first multiplied by gscale and written back (gradient:div(n), objective.lua:200, folded into the
// optimiser's pass over the flat vectors: 6 streams instead of 2 + 5).
// one element of the update: shared by the whole-vector kernel and the slice kernel, so that an update applied slice by slice
// (frcnn_scale_rmsprop_slice: beside the backward pass, as slices of the gradient become final) leaves the same bits as one
// pass over the whole vector wherever a slice boundary falls inside a 16-byte group
template <bool SCALE>
__device__ __forceinline__ void rmsprop_one(float& x, float& g, float& m, float lr, float alpha, float oma, float eps, float gscale) {
#pragma clang fp contract(off)   // every operation rounded by itself: the same bits from either instantiation and either kernel
  if (SCALE) g *= gscale;
  m = alpha * m + oma * (g * g);
  x = x - lr * g / (sqrtf(m) + eps);
}
template <bool SCALE>
__device__ __forceinline__ void rmsprop_four(float4* x4, float4* g4, float4* m4, long i, float lr, float alpha, float oma, float eps, float gscale) {
  float4 xv = x4[i], gv = g4[i], mv = m4[i];
  rmsprop_one<SCALE>(xv.x, gv.x, mv.x, lr, alpha, oma, eps, gscale);
  rmsprop_one<SCALE>(xv.y, gv.y, mv.y, lr, alpha, oma, eps, gscale);
  rmsprop_one<SCALE>(xv.z, gv.z, mv.z, lr, alpha, oma, eps, gscale);
  rmsprop_one<SCALE>(xv.w, gv.w, mv.w, lr, alpha, oma, eps, gscale);
  if (SCALE) g4[i] = gv;
  m4[i] = mv;
  x4[i] = xv;
}
template <bool SCALE>
__device__ __forceinline__ void rmsprop_scalar(float* x, float* g, float* m, long i, float lr, float alpha, float oma, float eps, float gscale) {
  float xi = x[i], gi = g[i], mi = m[i];
  rmsprop_one<SCALE>(xi, gi, mi, lr, alpha, oma, eps, gscale);
  if (SCALE) g[i] = gi;
  m[i] = mi;
  x[i] = xi;
}

template <bool SCALE>
__global__ void rmsprop_kernel(float* __restrict__ x, float* __restrict__ g, float* __restrict__ m,
                               long n, float lr, float alpha, float eps, float gscale,
                               const double* __restrict__ gcount) {
  // gcount: the divisor of gradient:div (objective.lua:200) read from the device -- the all-reduced example count of a
  // data-parallel step, which no host has seen yet (a count of 0 leaves the gradient as it is)
  if (SCALE && gcount) { const double c = *gcount; gscale = c > 0.0 ? (float)(1.0 / c) : 1.0f; }
  const long n4 = n >> 2;
  float4* x4 = reinterpret_cast<float4*>(x);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  const float oma = 1.0f - alpha;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    rmsprop_four<SCALE>(x4, g4, m4, i, lr, alpha, oma, eps, gscale);
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    rmsprop_scalar<SCALE>(x, g, m, i, lr, alpha, oma, eps, gscale);
}

// elements [lo, hi) of 16-byte aligned vectors: 16-byte groups where the slice covers them whole, the up to three elements
// in front of the first and behind the last such group one by one (thread t < 8 of block 0)
template <bool SCALE>
__global__ void rmsprop_slice_kernel(float* __restrict__ x, float* __restrict__ g, float* __restrict__ m, long lo, long hi,
                                     float lr, float alpha, float eps, float gscale) {
  const long a = min(hi, (lo + 3) & ~3L), b = max(a, hi & ~3L);   // [lo, a) ragged | [a, b) whole groups | [b, hi) ragged
  const long n4 = (b - a) >> 2;
  float4* x4 = reinterpret_cast<float4*>(x + a);
  float4* g4 = reinterpret_cast<float4*>(g + a);
  float4* m4 = reinterpret_cast<float4*>(m + a);
  const float oma = 1.0f - alpha;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    rmsprop_four<SCALE>(x4, g4, m4, i, lr, alpha, oma, eps, gscale);
  if (blockIdx.x == 0 && threadIdx.x < 8) {
    const int t = threadIdx.x;
    const long i = t < 4 ? lo + t : b + (t - 4);
    if (t < 4 ? i < a : i < hi) rmsprop_scalar<SCALE>(x, g, m, i, lr, alpha, oma, eps, gscale);
  }
}
int rmsprop_slice(float* x, float* g, float* m, long lo, long hi, float lr, float alpha, float eps, float gscale, bool scale_first,
                  hipStream_t s) {
  FR_CHECK((((uintptr_t)x | (uintptr_t)g | (uintptr_t)m) & 15) == 0, "rmsprop_slice: the vectors must be 16-byte aligned");
  FR_CHECK(lo >= 0 && lo <= hi, "rmsprop_slice: bad range [%ld, %ld)", lo, hi);
  if (lo == hi) return FRCNN_OK;
  const long n = hi - lo;
  // half the wave slots at most: a slice update runs BESIDE other launches (the backward pass) and must not keep them waiting for slots
  static const int max_grid = getenv("FRCNN_SLICE_GRID") ? atoi(getenv("FRCNN_SLICE_GRID")) : 1024;
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n / 4, 256)), max_grid);
  if (scale_first)
    FR_LAUNCH(KC_OPTIM, 0, n * 24.0, s, rmsprop_slice_kernel<true>, dim3(grid), dim3(256), 0, x, g, m, lo, hi, lr, alpha, eps, gscale);
  else
    FR_LAUNCH(KC_OPTIM, 0, n * 20.0, s, rmsprop_slice_kernel<false>, dim3(grid), dim3(256), 0, x, g, m, lo, hi, lr, alpha, eps, 1.f);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

int rmsprop_step(float* x, float* g, float* m, long n, float lr, float alpha, float eps, float gscale,
                 bool scale_first, hipStream_t s, const double* gcount_dev) {
  FR_CHECK((((uintptr_t)x | (uintptr_t)g | (uintptr_t)m) & 15) == 0, "rmsprop_step: buffers must be 16-byte aligned");
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n / 4, 256)), 2048);
  if (scale_first)
    FR_LAUNCH(KC_OPTIM, 0, n * 24.0, s, rmsprop_kernel<true>, dim3(grid), dim3(256), 0, x, g, m, n, lr, alpha, eps, gscale, gcount_dev);
  else
    FR_LAUNCH(KC_OPTIM, 0, n * 20.0, s, rmsprop_kernel<false>, dim3(grid), dim3(256), 0, x, g, m, n, lr, alpha, eps, 1.f, (const double*)nullptr);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
