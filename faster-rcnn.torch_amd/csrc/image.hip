// image.hip -- device side of BatchIterator:processImage (BatchIterator.lua:101-164): the
// arithmetic the reference delegates to the torch `image` and `nn` packages, on a decoded 3xHxW float frame
// that is already resident in HBM.
//   image.rgb2yuv                         (utilities.lua load_image, color_space 'yuv')
//   image.scale, 'bilinear'               (BatchIterator.lua:51)   two separable passes, row then column
//   image.crop + image.hflip + image.vflip (BatchIterator.lua:57-80) one gather
//   img[i]:add(-mean), img[i]:div(std)    (BatchIterator.lua:146-160) fixed-order fp64 reductions
//   nn.SpatialContrastiveNormalization(1, image.gaussian1D(w)) on channel 1 (BatchIterator.lua:162)
// All of it is HBM-bound streaming / small-stencil work (a 3x450x800 frame is 4.3 MB): no MFMA, coalesced
// accesses along x, LDS tiles with halo for the two stencil passes.  The float expressions keep the order of
// the C originals (no FMA contraction) so that results can be compared element-wise with the CPU restatement.
#include <algorithm>

#include "kernels.h"

#pragma clang fp contract(off)

namespace frcnn {

// ---------------------------------------------------------------- rgb2yuv
__global__ void rgb2yuv_kernel(const float* __restrict__ rgb, float* __restrict__ yuv, long hw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const float r = rgb[i], g = rgb[hw + i], b = rgb[2 * hw + i];
    yuv[i] = 0.299f * r + 0.587f * g + 0.114f * b;
    yuv[hw + i] = -0.14713f * r - 0.28886f * g + 0.436f * b;
    yuv[2 * hw + i] = 0.615f * r - 0.51499f * g - 0.10001f * b;
  }
}
int image_rgb2yuv(const float* rgb, float* yuv, int H, int W, hipStream_t s) {
  const long hw = (long)H * W;
  if (hw <= 0) return FRCNN_OK;
  int grid = (int)std::min<long>(cdivl(hw, 256), 4096);
  FR_LAUNCH(KC_IMAGE, 0, hw * 24.0, s, rgb2yuv_kernel, dim3(grid), dim3(256), 0, rgb, yuv, hw);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- rgb2hsv / rgb2lab
// The other two conversions load_image offers (utilities.lua:212-215).  image/generic/image.c does them per pixel in `real`
// (float) with the C library's double pow(); here pow() is double too, so the two agree to the rounding of the final cast.
__global__ void rgb2hsv_kernel(const float* __restrict__ rgb, float* __restrict__ hsv, long hw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const float r = rgb[i], g = rgb[hw + i], b = rgb[2 * hw + i];
    const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
    float h = 0.f, sat = 0.f;
    if (mx != mn) {   // (achromatic pixels keep h = s = 0)
      const float d = mx - mn;
      if (mx == r) h = (g - b) / d + (g < b ? 6.f : 0.f);
      else if (mx == g) h = (b - r) / d + 2.f;
      else h = (r - g) / d + 4.f;
      h /= 6.f;
      sat = d / mx;
    }
    hsv[i] = h; hsv[hw + i] = sat; hsv[2 * hw + i] = mx;
  }
}
__device__ __forceinline__ float srgb_expand(float c) {
  return c <= 0.04045f ? (float)((double)c / 12.92) : (float)pow(((double)c + 0.055) / 1.055, 2.4);
}
__device__ __forceinline__ double lab_f(double t) {
  const double eps = 216.0 / 24389.0, kappa = 24389.0 / 27.0;
  return t > eps ? pow(t, 1.0 / 3.0) : (kappa * t + 16.0) / 116.0;
}
__global__ void rgb2lab_kernel(const float* __restrict__ rgb, float* __restrict__ lab, long hw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    // (the expanded channels are `real`; X, Y, Z and the three cube roots stay double until the final stores)
    const float r = srgb_expand(rgb[i]), g = srgb_expand(rgb[hw + i]), b = srgb_expand(rgb[2 * hw + i]);
    double x = 0.412453 * r + 0.357580 * g + 0.180423 * b;   // linear sRGB -> XYZ
    const double y = 0.212671 * r + 0.715160 * g + 0.072169 * b;
    double z = 0.019334 * r + 0.119193 * g + 0.950227 * b;
    x /= 0.950456; z /= 1.088754;                              // D65 white point
    const double fx = lab_f(x), fy = lab_f(y), fz = lab_f(z);
    lab[i] = (float)(116.0 * fy - 16.0);
    lab[hw + i] = (float)(500.0 * (fx - fy));
    lab[2 * hw + i] = (float)(200.0 * (fy - fz));
  }
}
int image_rgb2hsv(const float* rgb, float* hsv, int H, int W, hipStream_t s) {
  const long hw = (long)H * W;
  if (hw <= 0) return FRCNN_OK;
  int grid = (int)std::min<long>(cdivl(hw, 256), 4096);
  FR_LAUNCH(KC_IMAGE, 0, hw * 24.0, s, rgb2hsv_kernel, dim3(grid), dim3(256), 0, rgb, hsv, hw);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}
int image_rgb2lab(const float* rgb, float* lab, int H, int W, hipStream_t s) {
  const long hw = (long)H * W;
  if (hw <= 0) return FRCNN_OK;
  int grid = (int)std::min<long>(cdivl(hw, 256), 4096);
  FR_LAUNCH(KC_IMAGE, 0, hw * 24.0, s, rgb2lab_kernel, dim3(grid), dim3(256), 0, rgb, lab, hw);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- image.scale ('bilinear')
// One output sample per thread along one axis: src[(o*src_len + s)*inner + i] -> dst[(o*dst_len + d)*inner + i].
// Up-scaling interpolates with scale (src_len-1)/(dst_len-1) and copies the last sample; down-scaling is a box
// filter over [d*scale, (d+1)*scale) with fractional end weights, divided by the accumulated weight.
__global__ void scale_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, long outer, int src_len,
                                  int dst_len, long inner) {
  const long total = outer * dst_len * inner;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long i = t % inner;
    const long od = t / inner;
    const int d = (int)(od % dst_len);
    const long o = od / dst_len;
    const float* sp = src + o * src_len * inner + i;
    float v;
    if (dst_len > src_len) {
      if (src_len == 1 || d == dst_len - 1) {
        v = sp[(long)(src_len - 1) * inner];
      } else {
        const float scale = (float)(src_len - 1) / (float)(dst_len - 1);
        float sf = (float)d * scale;
        const long si = (long)sf;
        sf -= (float)si;
        v = (1.f - sf) * sp[si * inner] + sf * sp[(si + 1) * inner];
      }
    } else if (dst_len < src_len) {
      const float scale = (float)src_len / (float)dst_len;
      float s0f = (float)d * scale;
      const long s0 = (long)s0f;
      s0f -= (float)s0;
      float s1f = (float)(d + 1) * scale;
      const long s1 = (long)s1f;
      s1f -= (float)s1;
      float acc = (1.f - s0f) * sp[s0 * inner];
      float n = 1.f - s0f;
      for (long si = s0 + 1; si < s1; ++si) {
        acc += sp[si * inner];
        n += 1.f;
      }
      if (s1 < src_len) {
        acc += s1f * sp[s1 * inner];
        n += s1f;
      }
      v = acc / n;
    } else {
      v = sp[(long)d * inner];
    }
    dst[t] = v;
  }
}

// The row pass with image.rgb2yuv applied to every source sample on the fly (same arithmetic as rgb2yuv_kernel, so
// the result equals converting first): one thread produces the three channels of one output sample.
// SRC 0: planar float RGB [3][H][W];  SRC 1: interleaved 8-bit RGB [H][W][3] as decoders deliver it, converted with
// v * (1/255) exactly like image.load(fn, 3, 'float').  YUV 0: the sample stays RGB.
template <int SRC, int YUV>
__device__ __forceinline__ void yuv_at(const void* __restrict__ src, long hw, long i, float* o) {
  float R, G, B;
  if (SRC == 0) {
    const float* r = (const float*)src;
    R = r[i]; G = r[hw + i]; B = r[2 * hw + i];
  } else {
    const unsigned char* u = (const unsigned char*)src + 3 * i;
    const float k = 1.0f / 255.0f;
    R = (float)u[0] * k; G = (float)u[1] * k; B = (float)u[2] * k;
  }
  if (YUV) {
    o[0] = 0.299f * R + 0.587f * G + 0.114f * B;
    o[1] = -0.14713f * R - 0.28886f * G + 0.436f * B;
    o[2] = 0.615f * R - 0.51499f * G - 0.10001f * B;
  } else {
    o[0] = R; o[1] = G; o[2] = B;
  }
}
template <int SRC, int YUV>
__global__ void scale_rows_rgb2yuv_kernel(const void* __restrict__ src, float* __restrict__ dst, int H, int W, int dW) {
  const long total = (long)H * dW, hw = (long)H * W, ohw = (long)H * dW;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int d = (int)(t % dW);
    const long row = (t / dW) * W;
    float v[3], a[3], b[3];
    if (dW > W) {
      if (W == 1 || d == dW - 1) {
        yuv_at<SRC, YUV>(src, hw, row + W - 1, v);
      } else {
        const float scale = (float)(W - 1) / (float)(dW - 1);
        float sf = (float)d * scale;
        const long si = (long)sf;
        sf -= (float)si;
        yuv_at<SRC, YUV>(src, hw, row + si, a);
        yuv_at<SRC, YUV>(src, hw, row + si + 1, b);
        for (int c = 0; c < 3; ++c) v[c] = (1.f - sf) * a[c] + sf * b[c];
      }
    } else if (dW < W) {
      const float scale = (float)W / (float)dW;
      float s0f = (float)d * scale;
      const long s0 = (long)s0f;
      s0f -= (float)s0;
      float s1f = (float)(d + 1) * scale;
      const long s1 = (long)s1f;
      s1f -= (float)s1;
      yuv_at<SRC, YUV>(src, hw, row + s0, a);
      float acc[3] = {(1.f - s0f) * a[0], (1.f - s0f) * a[1], (1.f - s0f) * a[2]};
      float n = 1.f - s0f;
      for (long si = s0 + 1; si < s1; ++si) {
        yuv_at<SRC, YUV>(src, hw, row + si, a);
        for (int c = 0; c < 3; ++c) acc[c] += a[c];
        n += 1.f;
      }
      if (s1 < W) {
        yuv_at<SRC, YUV>(src, hw, row + s1, a);
        for (int c = 0; c < 3; ++c) acc[c] += s1f * a[c];
        n += s1f;
      }
      for (int c = 0; c < 3; ++c) v[c] = acc[c] / n;
    } else {
      yuv_at<SRC, YUV>(src, hw, row + d, v);
    }
    for (int c = 0; c < 3; ++c) dst[c * ohw + t] = v[c];
  }
}

int image_scale(const float* src, int C, int H, int W, float* dst, int dH, int dW, float* tmp, int rgb2yuv,
                hipStream_t s) {
  FR_CHECK(C > 0 && H > 0 && W > 0 && dH > 0 && dW > 0, "image_scale: empty image (%dx%dx%d -> %dx%d)", C, H, W, dH, dW);
  FR_CHECK(!rgb2yuv || C == 3, "image_scale: the fused rgb2yuv conversion needs 3 channels");
  if (rgb2yuv) {
    const long total = (long)H * dW;
    int grid = (int)std::min<long>(cdivl(total, 256), 8192);
    FR_LAUNCH(KC_IMAGE, 0, ((double)C * H * W + C * total) * 4.0, s, (scale_rows_rgb2yuv_kernel<0, 1>), dim3(grid), dim3(256), 0,
              (const void*)src, tmp, H, W, dW);
    FR_LAUNCH_CHECK();
  } else
  // rows: [C*H][W] -> [C*H][dW]
  {
    const long total = (long)C * H * dW;
    int grid = (int)std::min<long>(cdivl(total, 256), 8192);
    FR_LAUNCH(KC_IMAGE, 0, ((double)C * H * W + total) * 4.0, s, scale_axis_kernel, dim3(grid), dim3(256), 0, src, tmp,
              (long)C * H, W, dW, 1L);
    FR_LAUNCH_CHECK();
  }
  // columns: [C][H][dW] -> [C][dH][dW]
  {
    const long total = (long)C * dH * dW;
    int grid = (int)std::min<long>(cdivl(total, 256), 8192);
    FR_LAUNCH(KC_IMAGE, 0, ((double)C * H * dW + total) * 4.0, s, scale_axis_kernel, dim3(grid), dim3(256), 0,
              (const float*)tmp, dst, (long)C, H, dH, (long)dW);
    FR_LAUNCH_CHECK();
  }
  return FRCNN_OK;
}

// The same two passes for a frame that is still the decoder's 8-bit interleaved RGB: the conversion to float (and to
// YUV when asked) happens on the fly in the row pass -- 6 MB instead of 25 MB cross PCIe and no full-resolution
// float frame is ever written.
int image_scale_u8(const unsigned char* src_hwc, int H, int W, float* dst, int dH, int dW, float* tmp, int rgb2yuv,
                   hipStream_t s) {
  FR_CHECK(H > 0 && W > 0 && dH > 0 && dW > 0, "image_scale_u8: empty image (%dx%d -> %dx%d)", H, W, dH, dW);
  {
    const long total = (long)H * dW;
    int grid = (int)std::min<long>(cdivl(total, 256), 8192);
    const double bytes = 3.0 * H * W + 12.0 * total;
    if (rgb2yuv)
      FR_LAUNCH(KC_IMAGE, 0, bytes, s, (scale_rows_rgb2yuv_kernel<1, 1>), dim3(grid), dim3(256), 0, (const void*)src_hwc, tmp, H, W, dW);
    else
      FR_LAUNCH(KC_IMAGE, 0, bytes, s, (scale_rows_rgb2yuv_kernel<1, 0>), dim3(grid), dim3(256), 0, (const void*)src_hwc, tmp, H, W, dW);
    FR_LAUNCH_CHECK();
  }
  {
    const long total = 3L * dH * dW;
    int grid = (int)std::min<long>(cdivl(total, 256), 8192);
    FR_LAUNCH(KC_IMAGE, 0, (3.0 * H * dW + total) * 4.0, s, scale_axis_kernel, dim3(grid), dim3(256), 0, (const float*)tmp, dst, 3L,
              H, dH, (long)dW);
    FR_LAUNCH_CHECK();
  }
  return FRCNN_OK;
}

// ---------------------------------------------------------------- crop + flips (one gather)
__global__ void crop_flip_kernel(const float* __restrict__ src, int H, int W, int x0, int y0, int w, int h, int hf,
                                 int vf, float* __restrict__ dst, long total) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int x = (int)(t % w);
    const long cy = t / w;
    const int y = (int)(cy % h);
    const long c = cy / h;
    const int sx = x0 + (hf ? w - 1 - x : x);
    const int sy = y0 + (vf ? h - 1 - y : y);
    dst[t] = src[(c * H + sy) * W + sx];
  }
}
int image_crop_flip(const float* src, int C, int H, int W, int x0, int y0, int w, int h, int hflip, int vflip,
                    float* dst, hipStream_t s) {
  FR_CHECK(x0 >= 0 && y0 >= 0 && w > 0 && h > 0 && x0 + w <= W && y0 + h <= H,
           "image_crop_flip: window (%d,%d)+%dx%d outside the %dx%d image", x0, y0, w, h, W, H);
  const long total = (long)C * h * w;
  int grid = (int)std::min<long>(cdivl(total, 256), 8192);
  FR_LAUNCH(KC_IMAGE, 0, total * 8.0, s, crop_flip_kernel, dim3(grid), dim3(256), 0, src, H, W, x0, y0, w, h, hflip,
            vflip, dst, total);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- per-channel centring / scaling
// TH accumulates mean and (unbiased) standard deviation of a float tensor in double.  Here: IMG_NB blocks per
// channel write fp64 partials (fixed tree inside the block), one thread per channel folds them in index order
// -> deterministic.  mode 0: sum(x); mode 1: sum((x - m)^2) with m = stat[c].
#define IMG_NB 64
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  __syncthreads();
  return r;
}
// fixed-order fold of a channel's IMG_NB partials (every block of a consumer kernel repeats it: same order, same value)
__device__ __forceinline__ double fold_partials(const double* __restrict__ part, int c) {
  double t = 0.0;
  for (int i = 0; i < IMG_NB; ++i) t += part[c * IMG_NB + i];
  return t;
}
// One pass over a channel per launch, grid (IMG_NB, C), fp64 partial sums out:
//   STEP 0: out = sum(x)
//   STEP 1: mean = fold(in)/n;  x += (float)(-mean)  (centring);  out = sum(x)      [apply = 0: only out = sum(x)]
//   STEP 2: m = fold(in)/n;  out = sum((x - m)^2)
//   STEP 3: sd = sqrt(fold(in)/(n-1));  x /= (float)sd when sd > 1e-8
template <int STEP>
__global__ void channel_pass_kernel(float* __restrict__ img, long hw, int apply, const double* __restrict__ in,
                                    double* __restrict__ out) {
  __shared__ double sh[4];
  const int c = blockIdx.y;
  float* p = img + (long)c * hw;
  double m = 0.0;
  float f = 0.f;
  if (STEP == 1 && apply) f = (float)(-(fold_partials(in, c) / (double)hw));
  if (STEP == 2) m = fold_partials(in, c) / (double)hw;
  if (STEP == 3) {
    const double sd = sqrt(fold_partials(in, c) / (double)(hw - 1));
    if (!(sd > 1e-8)) return;
    f = (float)sd;
  }
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    float x = p[i];
    if (STEP == 1 && apply) { x = x + f; p[i] = x; }
    if (STEP == 3) { p[i] = x / f; continue; }
    const double xd = (double)x;
    acc += STEP == 2 ? (xd - m) * (xd - m) : xd;
  }
  if (STEP == 3) return;
  const double r = block_sum_f64(acc, sh);
  if (threadIdx.x == 0) out[c * IMG_NB + blockIdx.x] = r;
}

size_t image_normalize_workspace_bytes(int C) { return (size_t)C * IMG_NB * 2 * sizeof(double); }

int image_normalize(float* img, int C, int H, int W, int centering, int scaling, void* ws, size_t ws_bytes,
                    hipStream_t s) {
  const long hw = (long)H * W;
  FR_CHECK(C > 0 && C <= 64 && hw > 0, "image_normalize: bad shape %dx%dx%d", C, H, W);
  FR_CHECK(ws_bytes >= image_normalize_workspace_bytes(C), "image_normalize: workspace too small");
  FR_CHECK(!scaling || hw > 1, "image_normalize: std of a single pixel");
  double* pa = (double*)ws;
  double* pb = pa + (size_t)C * IMG_NB;
  const dim3 grid(IMG_NB, C);
  const double bytes = C * hw * 4.0;
  if (centering)   // sum(x), then centring fused with the sum of the centred values (the mean that std() recomputes)
    FR_LAUNCH(KC_IMAGE, 0, bytes, s, channel_pass_kernel<0>, grid, dim3(256), 0, img, hw, 0, (const double*)pb, pa);
  if (centering || scaling)
    FR_LAUNCH(KC_IMAGE, 0, bytes * (centering ? 2 : 1), s, channel_pass_kernel<1>, grid, dim3(256), 0, img, hw, centering,
              (const double*)pa, pb);
  if (scaling) {
    FR_LAUNCH(KC_IMAGE, 0, bytes, s, channel_pass_kernel<2>, grid, dim3(256), 0, img, hw, 0, (const double*)pb, pa);
    FR_LAUNCH(KC_IMAGE, 0, bytes * 2, s, channel_pass_kernel<3>, grid, dim3(256), 0, img, hw, 0, (const double*)pa, pb);
  }
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- contrastive normalisation of one plane
// nn.SpatialSubtractiveNormalization / nn.SpatialDivisiveNormalization with a 1-D kernel k (K taps, unit sum)
// and one input plane: estimator(x)[y][x] = sum_jy k[jy] * (sum_jx k[jx] * x0[y+jy-p][x+jx-p]) with x0 = x
// zero-padded by p = K/2, horizontal pass first, float accumulation in tap order; coef = estimator(ones).
//   MODE 0: out = in - estimator(in) / coef
//   MODE 1: out = in / max_thr(sqrt(estimator(in^2)) / coef)        max_thr(v) = v > thr ? v : thr
// Tile CN_T x CN_T outputs per 256-thread block; the (CN_T+2p) x (CN_T+2p) input patch and the horizontally
// filtered rows live in LDS.
#define CN_T 32
#define CN_MAXK 15
struct CnArgs {
  const float* in;
  float* out;
  int H, W, K;
  float thr;
  float k[CN_MAXK];
};
template <int MODE>
__global__ __launch_bounds__(256) void contrastive_kernel(CnArgs a) {
  __shared__ float patch[(CN_T + CN_MAXK - 1) * (CN_T + CN_MAXK)];
  __shared__ float hor[(CN_T + CN_MAXK - 1) * (CN_T + 1)];
  const int K = a.K, p = K / 2, PW = CN_T + 2 * p, PP = CN_T + CN_MAXK;
  const int x0 = blockIdx.x * CN_T, y0 = blockIdx.y * CN_T;
  for (int t = threadIdx.x; t < PW * PW; t += 256) {
    const int r = t / PW, c = t % PW;
    const int gy = y0 + r - p, gx = x0 + c - p;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.in[(long)gy * a.W + gx];
    patch[r * PP + c] = MODE ? v * v : v;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < PW * CN_T; t += 256) {   // horizontal pass on every patch row
    const int r = t / CN_T, c = t % CN_T;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) acc += a.k[j] * patch[r * PP + c + j];
    hor[r * (CN_T + 1) + c] = acc;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < CN_T * CN_T; t += 256) {
    const int r = t / CN_T, c = t % CN_T;
    const int gy = y0 + r, gx = x0 + c;
    if (gy >= a.H || gx >= a.W) continue;
    float est = 0.f;
    for (int j = 0; j < K; ++j) est += a.k[j] * hor[(r + j) * (CN_T + 1) + c];
    // coef: the same estimator on a plane of ones (the rows / columns that fall into the padding contribute 0)
    float hx = 0.f;
    for (int j = 0; j < K; ++j) {
      const int sx = gx + j - p;
      hx += a.k[j] * ((sx >= 0 && sx < a.W) ? 1.f : 0.f);
    }
    float coef = 0.f;
    for (int j = 0; j < K; ++j) {
      const int sy = gy + j - p;
      coef += a.k[j] * ((sy >= 0 && sy < a.H) ? hx : 0.f);
    }
    const float x = a.in[(long)gy * a.W + gx];
    float o;
    if (MODE == 0) {
      o = x - est / coef;
    } else {
      float sd = sqrtf(est) / coef;
      sd = sd > a.thr ? sd : a.thr;
      o = x / sd;
    }
    a.out[(long)gy * a.W + gx] = o;
  }
}

int image_contrastive_norm(const float* in, int H, int W, const float* kernel_host, int K, float threshold, float* out,
                           float* tmp, hipStream_t s) {
  FR_CHECK(H > 0 && W > 0, "image_contrastive_norm: empty plane");
  FR_CHECK(K >= 1 && K <= CN_MAXK && (K & 1), "image_contrastive_norm: kernel width %d (odd, <= %d)", K, CN_MAXK);
  FR_CHECK(in != tmp && out != tmp, "image_contrastive_norm: tmp must not alias in/out");
  CnArgs a;
  double sum = 0.0;
  for (int j = 0; j < K; ++j) sum += (double)kernel_host[j];   // kernel:div(kernel:sum() * nInputPlane), nInputPlane = 1
  for (int j = 0; j < CN_MAXK; ++j) a.k[j] = j < K ? kernel_host[j] / (float)sum : 0.f;
  a.H = H; a.W = W; a.K = K; a.thr = threshold;
  const dim3 grid(cdiv(W, CN_T), cdiv(H, CN_T));
  a.in = in; a.out = tmp;
  FR_LAUNCH(KC_IMAGE, 0, (double)H * W * 8.0, s, contrastive_kernel<0>, grid, dim3(256), 0, a);
  a.in = tmp; a.out = out;
  FR_LAUNCH(KC_IMAGE, 0, (double)H * W * 8.0, s, contrastive_kernel<1>, grid, dim3(256), 0, a);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
