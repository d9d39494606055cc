// roi.hip -- nn.SpatialAdaptiveMaxPooling(kw,kh) applied to the strided sub-window that
// extract_roi_pooling_input() cuts out of the last feature map (objective.lua:5-13,117-118,
// 137-138,182-185; Detector.lua:96-97), batched: ONE launch pools every ROI of an image instead
// of the reference's per-ROI module call.  Gather/argmax work, no MFMA; reads hit the 2.2 MB map
// in L2, writes are R x C*kh*kw contiguous rows (the cnet input batch).
#include "kernels.h"

namespace frcnn {

// wins[r] = {row_lo,row_hi,col_lo,col_hi}: 1-based inclusive, exactly the idx table of
// objective.lua:11.  Cell (i,j) covers rows [floor(i*h/kh), ceil((i+1)*h/kh)) of the window;
// first max wins (strict >) in row-major scan order.  idx = flat y*W+x in map coordinates.
__global__ void roi_pool_forward_kernel(const float* __restrict__ fmap, int C, int H, int W,
                                        const int* __restrict__ wins, int R, int kh, int kw,
                                        float* __restrict__ out, int* __restrict__ idx) {
  const long total = (long)R * C * kh * kw;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int j = (int)(t % kw);
    long q = t / kw;
    int i = (int)(q % kh);
    q /= kh;
    int c = (int)(q % C);
    int r = (int)(q / C);
    const int* wn = wins + 4 * r;
    const int r0 = wn[0] - 1, c0 = wn[2] - 1;
    const int h = wn[1] - wn[0] + 1, w = wn[3] - wn[2] + 1;
    const int ys = (i * h) / kh, ye = ((i + 1) * h + kh - 1) / kh;
    const int xs = (j * w) / kw, xe = ((j + 1) * w + kw - 1) / kw;
    const float* ip = fmap + (size_t)c * H * W;
    float best = -3.402823466e+38f;
    int bi = -1;
    for (int y = ys; y < ye; ++y)
      for (int x = xs; x < xe; ++x) {
        float v = ip[(size_t)(r0 + y) * W + (c0 + x)];
        if (v > best) { best = v; bi = (r0 + y) * W + (c0 + x); }
      }
    out[t] = best;
    if (idx) idx[t] = bi;   // (inference has no backward pass: Detector.lua:96-98 drops the indices)
  }
}

// The same, one block per (ROI, channel slice): a thread owns ONE cell of the kh x kw grid, works out its window once and
// walks the channels (blockDim / (kh kw) of them at a time) -- the generic kernel above spends most of its instructions on four
// 64-bit divisions and the window arithmetic per OUTPUT (7.7 M outputs for 560 ROIs: 73 us, 7x the bytes it writes).
__global__ __launch_bounds__(256) void roi_pool_forward_cells_kernel(const float* __restrict__ fmap, int C, int H, int W,
                                                                     const int* __restrict__ wins, int kh, int kw,
                                                                     float* __restrict__ out, int* __restrict__ idx) {
  const int r = blockIdx.x, cells = kh * kw, groups = blockDim.x / cells;
  const int tid = threadIdx.x;
  if (tid >= groups * cells) return;
  const int cg = tid / cells, cell = tid - cg * cells;
  const int i = cell / kw, j = cell - i * kw;
  const int* wn = wins + 4 * r;
  const int r0 = wn[0] - 1, c0 = wn[2] - 1;
  const int h = wn[1] - wn[0] + 1, w = wn[3] - wn[2] + 1;
  const int ys = (i * h) / kh, ye = ((i + 1) * h + kh - 1) / kh;
  const int xs = (j * w) / kw, xe = ((j + 1) * w + kw - 1) / kw;
  const int HW = H * W;
  const int base = (r0 + ys) * W + c0 + xs, nx = xe - xs, ny = ye - ys;
  // The kernel is bound by the number of scattered load instructions (every lane another address), not by bytes: a row of
  // a cell is read in pieces of FOUR consecutive columns (one unaligned 16-byte load; the last piece of a plane is pulled
  // back so that it ends with the plane, elements outside the cell are masked, and an element seen twice cannot win
  // twice: the comparison is strict) instead of one dword per column.
  if (HW >= 4) {
    for (int c = blockIdx.y * groups + cg; c < C; c += gridDim.y * groups) {
      const float* ip = fmap + (size_t)c * HW;
      float best = -3.402823466e+38f;
      int bi = -1;
      for (int y = 0, o = base; y < ny; ++y, o += W)
        for (int x = 0; x < nx; x += 4) {
          const int q = min(o + x, HW - 4);
          float v[4];
          __builtin_memcpy(v, ip + q, 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int at = q + e;
            if (at >= o && at < o + nx && v[e] > best) { best = v[e]; bi = at; }
          }
        }
      const size_t t = ((size_t)r * C + c) * cells + cell;
      out[t] = best;
      if (idx) idx[t] = bi;
    }
    return;
  }
  for (int c = blockIdx.y * groups + cg; c < C; c += gridDim.y * groups) {
    const float* ip = fmap + (size_t)c * HW;
    float best = -3.402823466e+38f;
    int bi = -1;
    for (int y = 0, o = base; y < ny; ++y, o += W)
      for (int x = 0; x < nx; ++x) {
        const float v = ip[o + x];
        if (v > best) { best = v; bi = o + x; }
      }
    const size_t t = ((size_t)r * C + c) * cells + cell;
    out[t] = best;
    if (idx) idx[t] = bi;
  }
}

int roi_pool_forward(const float* fmap, int C, int H, int W, const int* wins, int R, int kh, int kw,
                     float* out, int* idx, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  long total = (long)R * C * kh * kw;
  if (kh * kw <= 256) {
    const int groups = 256 / (kh * kw);
    const int slices = std::max(1, std::min(cdiv(C, groups), (int)cdivl(4096, R)));   // enough blocks to fill the chip
    FR_LAUNCH(KC_ROI, 0, total * 8.0, s, roi_pool_forward_cells_kernel, dim3(R, slices), dim3(256), 0, fmap, C, H, W,
              wins, kh, kw, out, idx);
    FR_LAUNCH_CHECK();
    return FRCNN_OK;
  }
  int grid = (int)std::min<long>(cdivl(total, 256), 4096);
  FR_LAUNCH(KC_ROI, 0, total * 8.0, s, roi_pool_forward_kernel, dim3(grid), dim3(256), 0, fmap, C, H, W,
            wins, R, kh, kw, out, idx);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// delta_outputs[5][idx]:add(amp:backward(...))  (objective.lua:182-185): windows of different
// ROIs overlap, so this is a scatter-ADD.  One workgroup per channel keeps that channel's H x W plane
// in LDS, folds all R x kh x kw contributions with LDS atomics (ds_add_f32) and adds the plane to
// HBM once -- fp32 atomics straight to HBM cost ~190 us for 560 ROIs, this ~25 us.
__global__ void roi_pool_backward_lds_kernel(float* __restrict__ gmap, int C, int HW, const float* __restrict__ gout,
                                             const int* __restrict__ idx, int R, int cell) {
  extern __shared__ float plane[];
  const int c = blockIdx.x;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) plane[i] = 0.f;
  __syncthreads();
  const long D = (long)C * cell;
  const int total = R * cell;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = e / cell, j = e - r * cell;
    const long t = (long)r * D + (long)c * cell + j;
    const int bi = idx[t];
    const float g = gout[t];
    if (bi >= 0 && g != 0.f) atomicAdd(&plane[bi], g);
  }
  __syncthreads();
  float* gp = gmap + (size_t)c * HW;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const float v = plane[i];
    if (v != 0.f) gp[i] += v;
  }
}

// Deterministic variants: the contributions are accumulated in 64-bit fixed point (value * 2^44, rounded to nearest): integer
// addition is exact, so the sum does not depend on the order in which the atomics land.  |g| < 2^18, resolution 6e-14.
#define ROI_FIX_SCALE 17592186044416.0   /* 2^44 */
__device__ __forceinline__ long long roi_to_fix(float g) { return __double2ll_rn((double)g * ROI_FIX_SCALE); }
__global__ void roi_pool_backward_lds_det_kernel(float* __restrict__ gmap, int C, int HW, const float* __restrict__ gout,
                                                 const int* __restrict__ idx, int R, int cell) {
  extern __shared__ unsigned long long fplane[];
  const int c = blockIdx.x;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) fplane[i] = 0ull;
  __syncthreads();
  const long D = (long)C * cell;
  const int total = R * cell;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = e / cell, j = e - r * cell;
    const long t = (long)r * D + (long)c * cell + j;
    const int bi = idx[t];
    const float g = gout[t];
    if (bi >= 0 && g != 0.f) atomicAdd(&fplane[bi], (unsigned long long)roi_to_fix(g));
  }
  __syncthreads();
  float* gp = gmap + (size_t)c * HW;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const long long v = (long long)fplane[i];
    if (v != 0) gp[i] += (float)((double)v / ROI_FIX_SCALE);
  }
}
__global__ void roi_pool_backward_det_kernel(unsigned long long* __restrict__ fix, int C, long HW,
                                             const float* __restrict__ gout, const int* __restrict__ idx,
                                             long total, int cell) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int c = (int)((t / cell) % C);
    int bi = idx[t];
    float g = gout[t];
    if (bi >= 0 && g != 0.f) atomicAdd(fix + (size_t)c * HW + bi, (unsigned long long)roi_to_fix(g));
  }
}
__global__ void roi_fix_apply_kernel(const unsigned long long* __restrict__ fix, long n, float* __restrict__ gmap) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const long long v = (long long)fix[t];
    if (v != 0) gmap[t] += (float)((double)v / ROI_FIX_SCALE);
  }
}

// delta_outputs[5][idx]:add(amp:backward(...))  (objective.lua:182-185): windows of different
// ROIs overlap, so this is a scatter-ADD; fp32 atomics in L2.
__global__ void roi_pool_backward_kernel(float* __restrict__ gmap, int C, long HW,
                                         const float* __restrict__ gout, const int* __restrict__ idx,
                                         long total, int cell) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int c = (int)((t / cell) % C);
    int bi = idx[t];
    float g = gout[t];
    if (bi >= 0 && g != 0.f) unsafeAtomicAdd(gmap + (size_t)c * HW + bi, g);
  }
}

int roi_pool_backward(float* gmap, int C, int H, int W, const float* gout, const int* idx, int R,
                      int kh, int kw, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  long total = (long)R * C * kh * kw;
  if (deterministic()) {
    if ((size_t)H * W * 8 <= 64 * 1024) {
      FR_LAUNCH(KC_ROI, 0, total * 12.0, s, roi_pool_backward_lds_det_kernel, dim3(C), dim3(256), (size_t)H * W * 8, gmap, C,
                H * W, gout, idx, R, kh * kw);
    } else {
      float* ws = nullptr;
      const long n = (long)C * H * W;
      FR_TRY(det_workspace(s, (size_t)n * 2, &ws));
      FR_HIP(hipMemsetAsync(ws, 0, (size_t)n * 8, s));
      int grid = (int)std::min<long>(cdivl(total, 256), 4096);
      FR_LAUNCH(KC_ROI, 0, total * 12.0, s, roi_pool_backward_det_kernel, dim3(grid), dim3(256), 0, (unsigned long long*)ws, C,
                (long)H * W, gout, idx, total, kh * kw);
      FR_LAUNCH(KC_ROI, 0, n * 16.0, s, roi_fix_apply_kernel, dim3((int)std::min<long>(cdivl(n, 256), 4096)), dim3(256), 0,
                (const unsigned long long*)ws, n, gmap);
    }
    FR_LAUNCH_CHECK();
    return FRCNN_OK;
  }
  if ((size_t)H * W * 4 <= 64 * 1024) {
    FR_LAUNCH(KC_ROI, 0, total * 12.0, s, roi_pool_backward_lds_kernel, dim3(C), dim3(1024), (size_t)H * W * 4, gmap, C,
              H * W, gout, idx, R, kh * kw);
    FR_LAUNCH_CHECK();
    return FRCNN_OK;
  }
  int grid = (int)std::min<long>(cdivl(total, 256), 4096);
  FR_LAUNCH(KC_ROI, 0, total * 12.0, s, roi_pool_backward_kernel, dim3(grid), dim3(256), 0, gmap, C,
            (long)H * W, gout, idx, total, kh * kw);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
