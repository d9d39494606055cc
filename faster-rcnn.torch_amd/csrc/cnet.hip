// cnet.hip -- the small row-wise layers of the classification network
// (models/model_utilities.lua:76-124): nn.BatchNormalization, nn.PReLU + nn.Dropout, nn.LogSoftMax,
// and the two criteria of objective.lua:170-177 fused with their gradients.  R (number of ROIs)
// is a few hundred at most, so these are latency-bound: one launch each, fp64 accumulators for
// the per-feature statistics (cheap at this size), coalesced along the feature dimension.
#include "kernels.h"

namespace frcnn {

#define BN_EPS 1e-5
#define BN_MOM 0.1

// Block = BN_CB features x BN_RG row groups (1024 threads): lanes run along the feature dimension, the row
// groups split the R rows and meet in LDS (fixed order, fp64 partial sums).  16 features per block -> 64 blocks
// for the 1024-wide layer (64 features per block left 240 of the 256 CUs idle: 18-20 us per launch).
#define BN_CB 16
#define BN_RG 64
template <int CB = BN_CB, int RG = BN_RG>
__device__ __forceinline__ double bn_reduce(double v, double* sh, int tx, int ty) {
  sh[ty * CB + tx] = v;
  __syncthreads();
  // tree over the row groups, same order for every feature
  for (int stride = RG / 2; stride > 0; stride >>= 1) {
    if (ty < stride) sh[ty * CB + tx] += sh[(ty + stride) * CB + tx];
    __syncthreads();
  }
  const double s = sh[tx];
  __syncthreads();
  return s;
}
// The fused kernels' column sums: lanes = 16 row groups x 4 features, so four lane exchanges add a wave's row groups and the 16
// waves meet in LDS -- two barriers instead of the nine of the tree above (fixed order: the same sums on every run).
template <int NV>
__device__ __forceinline__ void fb_reduce(double (&v)[NV], double* sh, int tx, int ty) {
  const int tid = ty * 4 + tx, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane < 4) sh[(k * 16 + wave) * 4 + lane] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[(k * 16 + w) * 4 + tx];
    v[k] = t;
  }
  __syncthreads();
}
// the fused kernels: 4 features x 256 row groups per block -- 256 blocks for the 1024-wide layer, two or three rows per thread
// (with 16 x 64 the 64 blocks of the fused forward took 42 us: nine rows per thread, three dependent passes)
#define FB_CB 4   // (fb_reduce is written for 4 features x 256 row groups)
#define FB_RG 256
#define FB_KEEP 4   // rows a thread keeps in registers (batches of up to 1024 rows)

__global__ __launch_bounds__(1024) void bn_forward_kernel(const float* __restrict__ x, int R, int n,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running, int training, float* __restrict__ xhat,
                                                          float* __restrict__ invstd, float* __restrict__ y) {
  __shared__ double sh[BN_RG * BN_CB];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int j = blockIdx.x * BN_CB + tx;
  const bool ok = j < n;
  double mean, var;
  if (training) {
    double s = 0.0;
    if (ok) for (int r = ty; r < R; r += BN_RG) s += x[(size_t)r * n + j];
    mean = bn_reduce(s, sh, tx, ty) / R;
    double q = 0.0;
    if (ok) for (int r = ty; r < R; r += BN_RG) { double d = x[(size_t)r * n + j] - mean; q += d * d; }
    var = bn_reduce(q, sh, tx, ty);
    const double unb = R > 1 ? var / (R - 1) : var / R;
    var /= R;
    if (running && ok && ty == 0) {
      running[j] = (float)((1.0 - BN_MOM) * running[j] + BN_MOM * mean);
      running[n + j] = (float)((1.0 - BN_MOM) * running[n + j] + BN_MOM * unb);
    }
  } else {
    mean = ok ? running[j] : 0.0;
    var = ok ? running[n + j] : 1.0;
  }
  if (!ok) return;
  const double is = 1.0 / sqrt(var + BN_EPS);
  if (ty == 0) invstd[j] = (float)is;
  const double g = gamma[j], b = beta[j];
  for (int r = ty; r < R; r += BN_RG) {
    const double xh = (x[(size_t)r * n + j] - mean) * is;
    xhat[(size_t)r * n + j] = (float)xh;
    y[(size_t)r * n + j] = (float)(xh * g + b);
  }
}
int bn_forward(const float* x, int R, int n, const float* gamma, const float* beta, float* running,
               int training, float* xhat, float* invstd, float* y, hipStream_t s) {
  FR_CHECK(training || running, "bn_forward: evaluate mode needs running statistics");
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 16.0, s, bn_forward_kernel, dim3(cdiv(n, BN_CB)), dim3(BN_CB, BN_RG), 0, x, R, n,
            gamma, beta, running, training, xhat, invstd, y);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ __launch_bounds__(1024) void bn_backward_kernel(const float* __restrict__ gy, const float* __restrict__ xhat,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           int R, int n, int training, float* __restrict__ gx, float* ggamma,
                                                           float* gbeta) {
  __shared__ double sh[BN_RG * BN_CB];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int j = blockIdx.x * BN_CB + tx;
  const bool ok = j < n;
  double sg = 0.0, sgx = 0.0;
  if (ok)
    for (int r = ty; r < R; r += BN_RG) {
      const double g = gy[(size_t)r * n + j];
      sg += g;
      sgx += g * xhat[(size_t)r * n + j];
    }
  sg = bn_reduce(sg, sh, tx, ty);
  sgx = bn_reduce(sgx, sh, tx, ty);
  if (!ok) return;
  if (ty == 0) {
    ggamma[j] = (float)((double)ggamma[j] + sgx);
    gbeta[j] = (float)((double)gbeta[j] + sg);
  }
  const double is = invstd[j], gm = gamma[j];
  for (int r = ty; r < R; r += BN_RG) {
    const double g = gy[(size_t)r * n + j];
    const double xh = xhat[(size_t)r * n + j];
    const double v = training ? (g - sg / R - xh * sgx / R) * gm * is : g * gm * is;
    gx[(size_t)r * n + j] = (float)v;
  }
}
int bn_backward(const float* gy, const float* xhat, const float* invstd, const float* gamma, int R,
                int n, int training, float* gx, float* ggamma, float* gbeta, hipStream_t s) {
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 16.0, s, bn_backward_kernel, dim3(cdiv(n, BN_CB)), dim3(BN_CB, BN_RG), 0, gy,
            xhat, invstd, gamma, R, n, training, gx, ggamma, gbeta);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------ fused forms (round 4)
// Every launch of this chain costs ~10 us of the step (measured: six of them removed -> 2.945 to 2.86 ms): the fold of the
// product's split-K slabs, the batch normalisation and the activation are one kernel, same arithmetic in the same order as the
// separate ones (bit-identical results; FRCNN_CNET_FUSE=0 runs those).
__device__ __forceinline__ float fold_value(const float* __restrict__ x, const GemmFold& src, size_t total, size_t idx, int j) {
  if (src.nSplit == 0) return x[idx];
  float v = src.bias ? src.bias[j] : 0.f;
  for (int sI = 0; sI < src.nSplit; ++sI) v += src.slab[(size_t)sI * total + idx];
  return v;
}

template <bool GEN>
__global__ __launch_bounds__(1024) void cnet_act_forward_kernel(const float* __restrict__ x, GemmFold src, int R, int n,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running, int training, float* __restrict__ lin,
                                                                float* __restrict__ xhat, float* __restrict__ invstd,
                                                                float* __restrict__ pre, const float* slope, float* __restrict__ mask,
                                                                float inv_keep, float p, unsigned long long seed,
                                                                float* __restrict__ post) {
  __shared__ double sh[FB_RG * FB_CB];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int j = blockIdx.x * FB_CB + tx;
  const bool ok = j < n;
  const size_t total = (size_t)R * n;
  const bool bn = gamma != nullptr;
  // pass 0: the folded product.  A thread's (up to FB_KEEP) values stay in registers for the passes that follow -- ONE trip to
  // memory instead of three dependent ones; taller batches re-read what the thread itself wrote.
  const float* v = src.nSplit ? lin : x;
  const bool keep = R <= FB_RG * FB_KEEP;
  float xv[FB_KEEP];
  double s0 = 0.0;
  if (ok) {
#pragma unroll
    for (int i = 0; i < FB_KEEP; ++i) {
      const int r = ty + i * FB_RG;
      xv[i] = 0.f;
      if (keep && r < R) {
        const size_t idx = (size_t)r * n + j;
        xv[i] = fold_value(x, src, total, idx, j);
        if (src.nSplit) lin[idx] = xv[i];
        s0 += xv[i];
      }
    }
    if (!keep && (src.nSplit || (bn && training)))
      for (int r = ty; r < R; r += FB_RG) {
        const size_t idx = (size_t)r * n + j;
        const float t = fold_value(x, src, total, idx, j);
        if (src.nSplit) lin[idx] = t;
        s0 += t;
      }
  }
  double mean = 0.0, is = 1.0, g = 1.0, b = 0.0;
  if (bn) {
    double var;
    if (training) {
      { double t[1] = {s0}; fb_reduce(t, sh, tx, ty); mean = t[0] / R; }
      double q = 0.0;
      if (ok && keep) {
#pragma unroll
        for (int i = 0; i < FB_KEEP; ++i)
          if (ty + i * FB_RG < R) { double d = xv[i] - mean; q += d * d; }
      } else if (ok) {
        for (int r = ty; r < R; r += FB_RG) { double d = v[(size_t)r * n + j] - mean; q += d * d; }
      }
      { double t[1] = {q}; fb_reduce(t, sh, tx, ty); var = t[0]; }
      const double unb = R > 1 ? var / (R - 1) : var / R;
      var /= R;
      if (running && ok && ty == 0) {
        running[j] = (float)((1.0 - BN_MOM) * running[j] + BN_MOM * mean);
        running[n + j] = (float)((1.0 - BN_MOM) * running[n + j] + BN_MOM * unb);
      }
    } else {
      mean = ok ? running[j] : 0.0;
      var = ok ? running[n + j] : 1.0;
    }
    if (!ok) return;
    is = 1.0 / sqrt(var + BN_EPS);
    if (ty == 0) invstd[j] = (float)is;
    g = gamma[j]; b = beta[j];
  }
  if (!ok) return;
  const float a = *slope;
  auto emit = [&](size_t idx, float y) {
    if (bn) {
      const double xh = ((double)y - mean) * is;
      xhat[idx] = (float)xh;
      y = (float)(xh * g + b);
      pre[idx] = y;
    }
    y = y > 0.f ? y : a * y;
    if (GEN) {
      const float mk = frcnn_keep_mask(seed, (unsigned long long)idx, p);
      mask[idx] = mk;
      y = y * (mk * inv_keep);
    } else if (mask) {
      y = y * (mask[idx] * inv_keep);
    }
    post[idx] = y;
  };
  if (keep) {
#pragma unroll
    for (int i = 0; i < FB_KEEP; ++i)
      if (ty + i * FB_RG < R) emit((size_t)(ty + i * FB_RG) * n + j, xv[i]);
  } else {
    for (int r = ty; r < R; r += FB_RG) emit((size_t)r * n + j, v[(size_t)r * n + j]);
  }
}
// (no batch normalisation: nothing couples the rows -- a plain element-wise pass, coalesced along the features)
template <bool GEN>
__global__ void cnet_act_forward_flat_kernel(const float* __restrict__ x, GemmFold src, long total, int n, float* __restrict__ lin,
                                             const float* slope, float* __restrict__ mask, float inv_keep, float p,
                                             unsigned long long seed, float* __restrict__ post) {
  const float a = *slope;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float y = fold_value(x, src, (size_t)total, (size_t)i, (int)(i % n));
    if (src.nSplit) lin[i] = y;
    y = y > 0.f ? y : a * y;
    if (GEN) {
      const float mk = frcnn_keep_mask(seed, (unsigned long long)i, p);
      mask[i] = mk;
      y = y * (mk * inv_keep);
    } else if (mask) {
      y = y * (mask[i] * inv_keep);
    }
    post[i] = y;
  }
}
int cnet_act_forward(const float* x, GemmFold src, int R, int n, const float* gamma, const float* beta, float* running,
                     int training, float* lin, float* xhat, float* invstd, float* pre, const float* slope, float* mask, bool gen,
                     float inv_keep, float p, unsigned long long seed, float* post, hipStream_t s) {
  FR_CHECK(!gamma || training || running, "cnet_act_forward: evaluate mode needs running statistics");
  if (!gamma) {
    const long total = (long)R * n;
    const int grid = (int)std::min<long>(std::max<long>(1, cdivl(total, 256)), 2048);
    if (gen)
      FR_LAUNCH(KC_ELEMWISE, 0, total * 16.0, s, cnet_act_forward_flat_kernel<true>, dim3(grid), dim3(256), 0, x, src, total, n, lin, slope,
                mask, inv_keep, p, seed, post);
    else
      FR_LAUNCH(KC_ELEMWISE, 0, total * 16.0, s, cnet_act_forward_flat_kernel<false>, dim3(grid), dim3(256), 0, x, src, total, n, lin, slope,
                mask, inv_keep, p, seed, post);
    FR_LAUNCH_CHECK();
    return FRCNN_OK;
  }
  if (gen)
    FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 20.0, s, cnet_act_forward_kernel<true>, dim3(cdiv(n, FB_CB)), dim3(FB_CB, FB_RG), 0, x,
              src, R, n, gamma, beta, running, training, lin, xhat, invstd, pre, slope, mask, inv_keep, p, seed, post);
  else
    FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 20.0, s, cnet_act_forward_kernel<false>, dim3(cdiv(n, FB_CB)), dim3(FB_CB, FB_RG), 0, x,
              src, R, n, gamma, beta, running, training, lin, xhat, invstd, pre, slope, mask, inv_keep, p, seed, post);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// gx = BN'( PReLU'(pre) * mask/keep * gy ): prelu_dropout_backward + bn_backward in one launch.  The intermediate gradient
// stays in registers between the two passes (taller batches: parked in gx, each thread re-reads what it wrote); the slope
// gradient leaves through one atomic per block (deterministic mode: per-block partials folded in block order).
__global__ __launch_bounds__(1024) void cnet_act_bn_backward_kernel(const float* __restrict__ gy, GemmFold src,
                                                                    const float* __restrict__ pre, const float* __restrict__ xhat,
                                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                    const float* slope, const float* __restrict__ mask, float inv_keep,
                                                                    int R, int n, int training, float* __restrict__ gx, float* ggamma,
                                                                    float* gbeta, float* gslope, float* part) {
  __shared__ double sh[FB_RG * FB_CB];
  __shared__ float shs[16];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int j = blockIdx.x * FB_CB + tx;
  const bool ok = j < n;
  const size_t total = (size_t)R * n;
  const float a = *slope;
  double sg = 0.0, sgx = 0.0;
  float sa = 0.f;
  const bool keep = R <= FB_RG * FB_KEEP;   // (the thread's gradients and xhat values stay in registers: see the forward kernel)
  float tv[FB_KEEP], xh[FB_KEEP];
  auto first = [&](size_t idx, float& t, float& xhv) {
    float g = fold_value(gy, src, total, idx, j);
    if (mask) g = g * (mask[idx] * inv_keep);
    const float xv = pre[idx];
    t = g;
    if (!(xv > 0.f)) { t = a * g; sa += xv * g; }
    xhv = xhat[idx];
    sg += t;
    sgx += (double)t * xhv;
  };
  if (ok && keep) {
#pragma unroll
    for (int i = 0; i < FB_KEEP; ++i) {
      tv[i] = 0.f; xh[i] = 0.f;
      if (ty + i * FB_RG < R) first((size_t)(ty + i * FB_RG) * n + j, tv[i], xh[i]);
    }
  } else if (ok) {
    for (int r = ty; r < R; r += FB_RG) {
      const size_t idx = (size_t)r * n + j;
      float t, xhv;
      first(idx, t, xhv);
      gx[idx] = t;   // parked for the second pass
    }
  }
  { double t[2] = {sg, sgx}; fb_reduce(t, sh, tx, ty); sg = t[0]; sgx = t[1]; }
  // slope gradient: one number per block
  {
    const int tid = ty * FB_CB + tx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sa += __shfl_down(sa, o, 64);
    if ((tid & 63) == 0) shs[tid >> 6] = sa;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int i = 0; i < 16; ++i) t += shs[i];
      if (part) part[blockIdx.x] = t; else unsafeAtomicAdd(gslope, t);
    }
  }
  if (!ok) return;
  if (ty == 0) {
    ggamma[j] = (float)((double)ggamma[j] + sgx);
    gbeta[j] = (float)((double)gbeta[j] + sg);
  }
  const double is = invstd[j], gm = gamma[j];
  auto second = [&](size_t idx, double g, double xhv) {
    const double v = training ? (g - sg / R - xhv * sgx / R) * gm * is : g * gm * is;
    gx[idx] = (float)v;
  };
  if (keep) {
#pragma unroll
    for (int i = 0; i < FB_KEEP; ++i)
      if (ty + i * FB_RG < R) second((size_t)(ty + i * FB_RG) * n + j, tv[i], xh[i]);
  } else {
    for (int r = ty; r < R; r += FB_RG) {
      const size_t idx = (size_t)r * n + j;
      second(idx, gx[idx], xhat[idx]);
    }
  }
}
__global__ void slope_fold_kernel(const float* __restrict__ part, int n, float* gslope);
int cnet_act_bn_backward(const float* gy, GemmFold src, const float* pre, const float* xhat, const float* invstd,
                         const float* gamma, const float* slope, const float* mask, float inv_keep, int R, int n, int training,
                         float* gx, float* ggamma, float* gbeta, float* gslope, hipStream_t s) {
  const int grid = cdiv(n, FB_CB);
  float* part = nullptr;
  if (deterministic()) FR_TRY(det_workspace(s, (size_t)grid, &part));
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 28.0, s, cnet_act_bn_backward_kernel, dim3(grid), dim3(FB_CB, FB_RG), 0, gy, src, pre, xhat,
            invstd, gamma, slope, mask, inv_keep, R, n, training, gx, ggamma, gbeta, gslope, part);
  if (part) FR_LAUNCH(KC_ELEMWISE, 0, grid * 4.0, s, slope_fold_kernel, dim3(1), dim3(1), 0, (const float*)part, grid, gslope);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// y = prelu(x) * (mask * inv_keep)   (nn.PReLU then nn.Dropout v2; mask==null -> identity)
// GEN: the keep mask is drawn here (same counter-based stream as dropout_mask()) and stored for the backward pass
template <bool GEN>
__global__ void prelu_dropout_forward_kernel(const float* __restrict__ x, long n, const float* slope,
                                             float* __restrict__ mask, float inv_keep, float p,
                                             unsigned long long seed, float* __restrict__ y) {
  const float a = *slope;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    v = v > 0.f ? v : a * v;
    if (GEN) {
      const float mk = frcnn_keep_mask(seed, (unsigned long long)i, p);
      mask[i] = mk;
      v = v * (mk * inv_keep);
    } else if (mask) {
      v = v * (mask[i] * inv_keep);
    }
    y[i] = v;
  }
}
int prelu_dropout_forward(const float* x, long n, const float* slope, const float* mask, float inv_keep,
                          float* y, hipStream_t s) {
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 1024);
  FR_LAUNCH(KC_ELEMWISE, 0, n * 12.0, s, prelu_dropout_forward_kernel<false>, dim3(grid), dim3(256), 0, x, n, slope,
            const_cast<float*>(mask), inv_keep, 0.f, 0ull, y);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int prelu_dropout_forward_gen(const float* x, long n, const float* slope, float* mask_out, float p,
                              unsigned long long seed, float* y, hipStream_t s) {
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 1024);
  FR_LAUNCH(KC_ELEMWISE, 0, n * 12.0, s, prelu_dropout_forward_kernel<true>, dim3(grid), dim3(256), 0, x, n, slope,
            mask_out, 1.0f / (1.0f - p), p, seed, y);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ void prelu_dropout_backward_kernel(const float* __restrict__ gy, const float* __restrict__ x, long n,
                                              const float* slope, const float* __restrict__ mask,
                                              float inv_keep, float* __restrict__ gx, float* gslope, float* part) {
  __shared__ float sh[4];
  const float a = *slope;
  float sa = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float g = gy[i];
    if (mask) g = g * (mask[i] * inv_keep);
    const float xv = x[i];
    float r = g;
    if (!(xv > 0.f)) { r = a * g; sa += xv * g; }
    gx[i] = r;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sa += __shfl_down(sa, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sa;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = sh[0] + sh[1] + sh[2] + sh[3];
    if (part) part[blockIdx.x] = t;   // deterministic mode: folded in block order by slope_fold_kernel
    else unsafeAtomicAdd(gslope, t);
  }
}
__global__ void slope_fold_kernel(const float* __restrict__ part, int n, float* gslope) {
  float v = 0.f;
  for (int i = 0; i < n; ++i) v += part[i];
  *gslope += v;
}
int prelu_dropout_backward(const float* gy, const float* x, long n, const float* slope,
                           const float* mask, float inv_keep, float* gx, float* gslope, hipStream_t s) {
  int grid = (int)std::min<long>(std::max<long>(1, cdivl(n, 256)), 256);
  float* part = nullptr;
  if (deterministic()) FR_TRY(det_workspace(s, (size_t)grid, &part));
  FR_LAUNCH(KC_ELEMWISE, 0, n * 16.0, s, prelu_dropout_backward_kernel, dim3(grid), dim3(256), 0, gy, x, n,
            slope, mask, inv_keep, gx, gslope, part);
  if (part) FR_LAUNCH(KC_ELEMWISE, 0, grid * 4.0, s, slope_fold_kernel, dim3(1), dim3(1), 0, (const float*)part, grid, gslope);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// nn.LogSoftMax per row (max-shifted), fp32 result
// one wave per row (lanes over the classes, fp64 max / sum by lane exchanges): with config/imagenet.lua's 201 classes a
// thread-per-row loop of 201 fp64 exponentials took 100 us per call
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
// (fold variant: the logits are a deferred product -- folded once into `x`'s buffer by the row's own wave, then read back)
__global__ void log_softmax_rows_fold_kernel(float* __restrict__ x, GemmFold src, int R, int n, float* __restrict__ y,
                                             float* __restrict__ y2) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  float* xr = x + (size_t)r * n;
  const size_t total = (size_t)R * n;
  if (src.nSplit)
    for (int i = lane; i < n; i += 64) xr[i] = fold_value(x, src, total, (size_t)r * n + i, i);
  double m = -1.0e300;
  for (int i = lane; i < n; i += 64) m = xr[i] > m ? (double)xr[i] : m;
  m = wave_max_f64(m);
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += exp((double)xr[i] - m);
  s = wave_sum_f64(s);
  const double lse = m + log(s);
  for (int i = lane; i < n; i += 64) {
    const float v = (float)((double)xr[i] - lse);
    y[(size_t)r * n + i] = v;
    if (y2) y2[(size_t)r * n + i] = v;
  }
}
int log_softmax_rows_fold(const float* x, GemmFold src, int R, int n, float* y, float* y2, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 8.0, s, log_softmax_rows_fold_kernel, dim3(cdiv(R, 4)), dim3(256), 0, const_cast<float*>(x),
            src, R, n, y, y2);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ void log_softmax_rows_kernel(const float* __restrict__ x, int R, int n, float* __restrict__ y) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + (size_t)r * n;
  double m = -1.0e300;
  for (int i = lane; i < n; i += 64) m = xr[i] > m ? (double)xr[i] : m;
  m = wave_max_f64(m);
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += exp((double)xr[i] - m);
  s = wave_sum_f64(s);
  const double lse = m + log(s);
  for (int i = lane; i < n; i += 64) y[(size_t)r * n + i] = (float)((double)xr[i] - lse);
}
int log_softmax_rows(const float* x, int R, int n, float* y, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 8.0, s, log_softmax_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, x, R,
            n, y);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------ both heads, one launch each way
// forward: the two heads' weights (4 + nc <= 32 rows of nf) are staged TRANSPOSED in LDS once per block ([k][33]: conflict-free both
// ways), then one wave per row: lane = (output n = lane & 31, half h of the features), nf / 2 multiply-adds against LDS, one lane
// exchange, LogSoftMax across the class lanes.  (A first version let every lane walk its own weight row in global memory with
// fp64 multiply-adds: 29 us -- no faster than the four launches it replaced.)
#define HEADS_PITCH 33
__global__ __launch_bounds__(256) void cnet_heads_forward_kernel(const float* __restrict__ x, int R, int nf, const float* __restrict__ Wb,
                                                                 const float* __restrict__ bb, const float* __restrict__ Wc,
                                                                 const float* __restrict__ bc, int nc, float* __restrict__ bbox_out,
                                                                 float* __restrict__ logits, float* __restrict__ lsm,
                                                                 float* __restrict__ cls_out) {
  extern __shared__ float wt[];   // [nf][HEADS_PITCH], then the four waves' rows [4][nf]
  float* xs = wt + (size_t)nf * HEADS_PITCH + (threadIdx.x >> 6) * nf;
  const int no = 4 + nc;
  for (int k = threadIdx.x; k < nf; k += 256) {   // (row by row: coalesced reads, no division, every load independent)
    float* d = wt + k * HEADS_PITCH;
#pragma unroll
    for (int n = 0; n < 4; ++n) d[n] = Wb[(size_t)n * nf + k];
#pragma unroll 8
    for (int n = 0; n < nc; ++n) d[4 + n] = Wc[(size_t)n * nf + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const bool isb = n < 4, isc = n >= 4 && n < no;
  const int nn = n < no ? n : 0, kh = nf / 2;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
    // the row comes to LDS in one coalesced sweep (64 dependent global loads per lane in the loop below made it 17 us)
    for (int k = lane * 4; k < nf; k += 256) *reinterpret_cast<float4*>(xs + k) = *reinterpret_cast<const float4*>(x + (size_t)r * nf + k);
    __builtin_amdgcn_wave_barrier();
    const float* xr = xs + h * kh;
    const float* wk = wt + (size_t)(h * kh) * HEADS_PITCH + nn;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int k = 0; k < kh; k += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + k);   // (one address per lane half: an LDS broadcast)
      a0 = fmaf(xv.x, wk[(k + 0) * HEADS_PITCH], a0); a1 = fmaf(xv.y, wk[(k + 1) * HEADS_PITCH], a1);
      a2 = fmaf(xv.z, wk[(k + 2) * HEADS_PITCH], a2); a3 = fmaf(xv.w, wk[(k + 3) * HEADS_PITCH], a3);
    }
    float v = (a0 + a1) + (a2 + a3);
    v += __shfl_xor(v, 32, 64);
    v += isb ? bb[n] : isc ? bc[n - 4] : 0.f;
    if (isb && h == 0) bbox_out[(size_t)r * 4 + n] = v;
    // nn.LogSoftMax over the class lanes (max-shifted, fp64: log_softmax_rows_kernel); both halves hold the same values
    const bool mine = isc && h == 0;
    double m = mine ? (double)v : -1.0e300;
    m = wave_max_f64(m);
    double e = mine ? exp((double)v - m) : 0.0;
    e = wave_sum_f64(e);
    const double lse = m + log(e);
    if (mine) {
      const size_t o = (size_t)r * nc + n - 4;
      const float l = (float)((double)v - lse);
      logits[o] = v; lsm[o] = l;
      if (cls_out) cls_out[o] = l;
    }
  }
}
// backward: the weights in LDS as they are ([4 + nc][nf]); one wave per row: lanes compute the LogSoftMax gradient of their class,
// park the row's 4 + nc gradients in LDS, then lanes = features: gfeat[k] = sum_n g[n] W[n][k]
__global__ __launch_bounds__(256) void cnet_heads_backward_kernel(const float* __restrict__ g_bbox, const float* __restrict__ g_cls,
                                                                  const float* __restrict__ lsm, int R, int nf,
                                                                  const float* __restrict__ Wb, const float* __restrict__ Wc, int nc,
                                                                  float* __restrict__ glog, float* __restrict__ gfeat, HeadsPostAct post) {
  extern __shared__ float wt[];   // [4 + nc][nf], then the four waves' gradients [4][64]
  __shared__ float shs[4];
  const float pa = post.pre ? *post.slope : 1.f;
  float sa = 0.f;   // slope gradient of the layer below (post.pre != null: its Dropout + PReLU backward applied to the stored row)
  const int no = 4 + nc;
  for (int k = threadIdx.x; k < nf; k += 256) {
#pragma unroll
    for (int n = 0; n < 4; ++n) wt[n * nf + k] = Wb[(size_t)n * nf + k];
#pragma unroll 8
    for (int n = 0; n < nc; ++n) wt[(4 + n) * nf + k] = Wc[(size_t)n * nf + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sg = wt + (size_t)no * nf + wave * 64;
  const bool isb = lane < 4, isc = lane >= 4 && lane < no;
  for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
    const size_t o = (size_t)r * nc + (isc ? lane - 4 : 0);
    const double gy = isc ? (double)g_cls[o] : 0.0;
    const double sum = wave_sum_f64(gy);
    float g = 0.f;
    if (isc) {
      g = (float)(gy - exp((double)lsm[o]) * sum);   // gx = gy - exp(lsm) * sum_j gy_j
      glog[o] = g;
    } else if (isb) {
      g = g_bbox[(size_t)r * 4 + lane];
    }
    sg[lane] = g;
    __builtin_amdgcn_wave_barrier();   // (a wave reads only what it wrote itself)
    for (int k = lane; k < nf; k += 64) {
      float a0 = 0.f, a1 = 0.f;
      int n = 0;
      for (; n + 2 <= no; n += 2) {
        a0 = fmaf(sg[n], wt[n * nf + k], a0);
        a1 = fmaf(sg[n + 1], wt[(n + 1) * nf + k], a1);
      }
      if (n < no) a0 = fmaf(sg[n], wt[n * nf + k], a0);
      float g = a0 + a1;
      if (post.pre) {   // prelu_dropout_backward_kernel's arithmetic
        const size_t idx = (size_t)r * nf + k;
        if (post.mask) g = g * (post.mask[idx] * post.inv_keep);
        const float xv = post.pre[idx];
        if (!(xv > 0.f)) { sa += xv * g; g = pa * g; }
      }
      gfeat[(size_t)r * nf + k] = g;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (post.pre) {   // one atomic per block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sa += __shfl_down(sa, o, 64);
    if (lane == 0) shs[wave] = sa;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(post.gslope, (shs[0] + shs[1]) + (shs[2] + shs[3]));
  }
}
bool cnet_heads_fused_eligible(int nf, int nc) { return 4 + nc <= 32 && nf % 8 == 0 && nf <= 1024; }
int cnet_heads_forward(const float* x, int R, int nf, const float* Wb, const float* bb, const float* Wc, const float* bc, int nc,
                       float* bbox_out, float* logits, float* lsm, float* cls_out, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  FR_CHECK(cnet_heads_fused_eligible(nf, nc), "cnet_heads_forward: %d classes", nc);
  FR_CHECK(((uintptr_t)x & 15) == 0, "cnet_heads_forward: input alignment");
  const size_t lds = ((size_t)nf * HEADS_PITCH + 4 * (size_t)nf) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnet_heads_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * nf * 4.0, s, cnet_heads_forward_kernel, dim3(std::min(cdiv(R, 4), 256)), dim3(256), lds, x, R, nf, Wb,
            bb, Wc, bc, nc, bbox_out, logits, lsm, cls_out);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int cnet_heads_backward(const float* g_bbox, const float* g_cls, const float* lsm, int R, int nf, const float* Wb, const float* Wc,
                        int nc, float* glog, float* gfeat, hipStream_t s, const HeadsPostAct* post) {
  if (R <= 0) return FRCNN_OK;
  FR_CHECK(cnet_heads_fused_eligible(nf, nc), "cnet_heads_backward: %d classes", nc);
  const size_t lds = ((size_t)(4 + nc) * nf + 256) * 4;
  static size_t attr_lds = 0;   // (the kernel also has a few bytes of static LDS: ask for what the launch needs, not for the CU's 160 KB)
  if (lds > attr_lds) {
    FR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cnet_heads_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * nf * 4.0, s, cnet_heads_backward_kernel, dim3(std::min(cdiv(R, 4), 256)), dim3(256), lds, g_bbox, g_cls,
            lsm, R, nf, Wb, Wc, nc, glog, gfeat, post ? *post : HeadsPostAct{});
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// gx = gy - exp(lsm) * sum_j gy_j
__global__ void log_softmax_backward_kernel(const float* __restrict__ gy, const float* __restrict__ lsm, int R,
                                            int n, float* __restrict__ gx) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  double sum = 0.0;
  for (int i = lane; i < n; i += 64) sum += gy[(size_t)r * n + i];
  sum = wave_sum_f64(sum);
  for (int i = lane; i < n; i += 64)
    gx[(size_t)r * n + i] = (float)((double)gy[(size_t)r * n + i] - exp((double)lsm[(size_t)r * n + i]) * sum);
}
int log_softmax_backward(const float* gy, const float* lsm, int R, int n, float* gx, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * n * 12.0, s, log_softmax_backward_kernel, dim3(cdiv(R, 4)), dim3(256), 0,
            gy, lsm, R, n, gx);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// objective.lua:170-177 in one workgroup:
//   crout[npos.., :] = 0 ; creg_loss = SmoothL1(crout, crtarget) * 10 ; crdelta = grad * 10
//   ccls_loss = ClassNLL(ccout, cctarget) (sizeAverage=true) ; ccdelta = -1/R at the target
// loss2[0] += creg_loss ; loss2[1] += ccls_loss   (fp64 accumulators, objective.lua:57-58)
__global__ void cnet_losses_kernel(float* __restrict__ crout, const float* __restrict__ crtarget,
                                   const float* __restrict__ ccout, const float* __restrict__ cctarget, int R,
                                   int npos, int ncls, float* __restrict__ crdelta, float* __restrict__ ccdelta,
                                   double* loss2) {
  __shared__ double shr[256], shc[256];
  double sr = 0.0, sc = 0.0;
  for (int i = threadIdx.x; i < R * 4; i += blockDim.x) {
    float v = crout[i];
    if (i >= npos * 4) { v = 0.f; crout[i] = 0.f; }
    const float z = v - crtarget[i];
    const float az = fabsf(z);
    sr += az < 1.0f ? 0.5 * (double)z * (double)z : (double)az - 0.5;
    crdelta[i] = (az < 1.0f ? z : (z > 0.f ? 1.0f : -1.0f)) * 10.0f;
  }
  for (int i = threadIdx.x; i < R * ncls; i += blockDim.x) ccdelta[i] = 0.f;
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const int t = (int)cctarget[r] - 1;
    sc -= (double)ccout[(size_t)r * ncls + t];
    ccdelta[(size_t)r * ncls + t] = (float)(-1.0 / R);
  }
  shr[threadIdx.x] = sr;
  shc[threadIdx.x] = sc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { shr[threadIdx.x] += shr[threadIdx.x + o]; shc[threadIdx.x] += shc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss2[0] += (double)(float)shr[0] * 10.0;
    loss2[1] += (double)(float)(shc[0] / R);
  }
}
int cnet_losses(float* crout, const float* crtarget, const float* ccout, const float* cctarget, int R,
                int npos, int ncls, float* crdelta, float* ccdelta, double* loss2, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * (ncls + 8) * 8.0, s, cnet_losses_kernel, dim3(1), dim3(256), 0, crout,
            crtarget, ccout, cctarget, R, npos, ncls, crdelta, ccdelta, loss2);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// torch.sort(cprob, 1, true)[1] (Detector.lua:110): class = argmax (first max), confidence = max
__global__ void cnet_decode_kernel(const float* __restrict__ lsm, int R, int ncls, int* __restrict__ cls,
                                   float* __restrict__ conf) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  int best = 0;
  for (int j = 1; j < ncls; ++j)
    if (lsm[(size_t)r * ncls + j] > lsm[(size_t)r * ncls + best]) best = j;
  cls[r] = best + 1;
  conf[r] = lsm[(size_t)r * ncls + best];
}
int cnet_decode(const float* cls_lsm, int R, int ncls, int* cls_out, float* conf_out, hipStream_t s) {
  if (R <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, (double)R * ncls * 4.0, s, cnet_decode_kernel, dim3(cdiv(R, 64)), dim3(64), 0,
            cls_lsm, R, ncls, cls_out, conf_out);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
