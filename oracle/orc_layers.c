/*
 * orc_layers.c -- CPU restatement of the Torch7 `nn` layer math the reference delegates to
 * ([ext]: torch/nn/cunn are not vendored; semantics listed in oracle/ASSUMPTIONS.md).
 * TEST INFRASTRUCTURE ONLY (see frcnn_oracle.h).  Storage is fp32 like the reference's
 * FloatTensor/CudaTensor; dot products accumulate in double (SURVEY 8d "CPU restatement with
 * fp64 accumulation") so that the oracle is the low-noise side of every tolerance check.
 */
#include "frcnn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;
void orc_set_threads(int n) {
  g_threads = n;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}
int orc_get_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

/* nn.SpatialConvolution:updateOutput -- cross-correlation, zero pad, stride 1
 * (model_utilities.lua:8,31,33): out[o,y,x] = b[o] + sum W[o,c,ky,kx]*in[c,y-p+ky,x-p+kx] */
void orc_conv2d_fwd(const float *in, int C, int H, int W, const float *wt, const float *bias,
                    int O, int kh, int kw, int pad, float *out) {
  int Ho = H + 2 * pad - kh + 1, Wo = W + 2 * pad - kw + 1;
#pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * (size_t)Ho * Wo);
#pragma omp for schedule(dynamic, 1)
    for (int o = 0; o < O; ++o) {
      double b = bias ? (double)bias[o] : 0.0;
      for (long t = 0; t < (long)Ho * Wo; ++t) acc[t] = b;
      for (int c = 0; c < C; ++c) {
        const float *ip = in + (size_t)c * H * W;
        for (int ky = 0; ky < kh; ++ky) {
          for (int kx = 0; kx < kw; ++kx) {
            double w = wt[(((size_t)o * C + c) * kh + ky) * kw + kx];
            int y0 = pad - ky > 0 ? pad - ky : 0;
            int y1 = H + pad - ky < Ho ? H + pad - ky : Ho;
            int x0 = pad - kx > 0 ? pad - kx : 0;
            int x1 = W + pad - kx < Wo ? W + pad - kx : Wo;
            for (int y = y0; y < y1; ++y) {
              const float *row = ip + (size_t)(y - pad + ky) * W + (kx - pad);
              double *arow = acc + (size_t)y * Wo;
              for (int x = x0; x < x1; ++x) arow[x] += w * (double)row[x];
            }
          }
        }
      }
      float *op = out + (size_t)o * Ho * Wo;
      for (long t = 0; t < (long)Ho * Wo; ++t) op[t] = (float)acc[t];
    }
    free(acc);
  }
}

/* nn.SpatialConvolution:updateGradInput */
void orc_conv2d_bwd_input(const float *gout, int O, int Ho, int Wo, const float *wt, int C,
                          int kh, int kw, int pad, int H, int W, float *gin) {
#pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * (size_t)H * W);
#pragma omp for schedule(dynamic, 1)
    for (int c = 0; c < C; ++c) {
      for (long t = 0; t < (long)H * W; ++t) acc[t] = 0.0;
      for (int o = 0; o < O; ++o) {
        const float *gp = gout + (size_t)o * Ho * Wo;
        for (int ky = 0; ky < kh; ++ky) {
          for (int kx = 0; kx < kw; ++kx) {
            double w = wt[(((size_t)o * C + c) * kh + ky) * kw + kx];
            /* in[c, y-p+ky, x-p+kx] receives w * gout[o,y,x] */
            int y0 = pad - ky > 0 ? pad - ky : 0;
            int y1 = H + pad - ky < Ho ? H + pad - ky : Ho;
            int x0 = pad - kx > 0 ? pad - kx : 0;
            int x1 = W + pad - kx < Wo ? W + pad - kx : Wo;
            for (int y = y0; y < y1; ++y) {
              const float *grow = gp + (size_t)y * Wo;
              double *arow = acc + (size_t)(y - pad + ky) * W + (kx - pad);
              for (int x = x0; x < x1; ++x) arow[x] += w * (double)grow[x];
            }
          }
        }
      }
      float *op = gin + (size_t)c * H * W;
      for (long t = 0; t < (long)H * W; ++t) op[t] = (float)acc[t];
    }
    free(acc);
  }
}

/* nn.SpatialConvolution:accGradParameters (accumulating, scale 1) */
void orc_conv2d_bwd_weight(const float *in, int C, int H, int W, const float *gout, int O,
                           int kh, int kw, int pad, float *gw, float *gb) {
  int Ho = H + 2 * pad - kh + 1, Wo = W + 2 * pad - kw + 1;
#pragma omp parallel for schedule(dynamic, 1)
  for (int o = 0; o < O; ++o) {
    const float *gp = gout + (size_t)o * Ho * Wo;
    if (gb) {
      double s = 0;
      for (long t = 0; t < (long)Ho * Wo; ++t) s += gp[t];
      gb[o] = (float)((double)gb[o] + s);
    }
    for (int c = 0; c < C; ++c) {
      const float *ip = in + (size_t)c * H * W;
      for (int ky = 0; ky < kh; ++ky) {
        for (int kx = 0; kx < kw; ++kx) {
          int y0 = pad - ky > 0 ? pad - ky : 0;
          int y1 = H + pad - ky < Ho ? H + pad - ky : Ho;
          int x0 = pad - kx > 0 ? pad - kx : 0;
          int x1 = W + pad - kx < Wo ? W + pad - kx : Wo;
          double s = 0;
          for (int y = y0; y < y1; ++y) {
            const float *row = ip + (size_t)(y - pad + ky) * W + (kx - pad);
            const float *grow = gp + (size_t)y * Wo;
            double rs = 0;
            for (int x = x0; x < x1; ++x) rs += (double)grow[x] * (double)row[x];
            s += rs;
          }
          size_t wi = (((size_t)o * C + c) * kh + ky) * kw + kx;
          gw[wi] = (float)((double)gw[wi] + s);
        }
      }
    }
  }
}

/* nn.PReLU() with ONE shared slope (model_utilities.lua:9,32,86): y = x>0 ? x : a*x (fp32) */
void orc_prelu_fwd(const float *x, long n, float a, float *y) {
#pragma omp parallel for
  for (long i = 0; i < n; ++i) y[i] = x[i] > 0.0f ? x[i] : a * x[i];
}

double orc_prelu_bwd(const float *x, const float *gy, long n, float a, float *gx) {
  double ga = 0;
#pragma omp parallel for reduction(+ : ga)
  for (long i = 0; i < n; ++i) {
    if (x[i] > 0.0f) {
      gx[i] = gy[i];
    } else {
      gx[i] = a * gy[i];
      ga += (double)x[i] * (double)gy[i];
    }
  }
  return ga;
}

/* nn.SpatialMaxPooling(2,2,2,2):ceil() (model_utilities.lua:23): oH = ceil((H-2)/2)+1,
 * border windows clipped to the input, first max wins (strict >). idx = flat y*W+x. */
void orc_maxpool2x2_ceil_fwd(const float *in, int C, int H, int W, float *out, int32_t *idx) {
  int Ho = (int)ceil((H - 2) / 2.0) + 1, Wo = (int)ceil((W - 2) / 2.0) + 1;
#pragma omp parallel for
  for (int c = 0; c < C; ++c) {
    const float *ip = in + (size_t)c * H * W;
    for (int oy = 0; oy < Ho; ++oy) {
      for (int ox = 0; ox < Wo; ++ox) {
        float best = -FLT_MAX;
        int bi = -1;
        for (int dy = 0; dy < 2; ++dy) {
          int y = oy * 2 + dy;
          if (y >= H) continue;
          for (int dx = 0; dx < 2; ++dx) {
            int x = ox * 2 + dx;
            if (x >= W) continue;
            float v = ip[(size_t)y * W + x];
            if (v > best) { best = v; bi = y * W + x; }
          }
        }
        out[((size_t)c * Ho + oy) * Wo + ox] = best;
        idx[((size_t)c * Ho + oy) * Wo + ox] = bi;
      }
    }
  }
}

void orc_maxpool2x2_ceil_bwd(const float *gout, const int32_t *idx, int C, int H, int W,
                             float *gin) {
  int Ho = (int)ceil((H - 2) / 2.0) + 1, Wo = (int)ceil((W - 2) / 2.0) + 1;
  memset(gin, 0, sizeof(float) * (size_t)C * H * W);
#pragma omp parallel for
  for (int c = 0; c < C; ++c) {
    for (long t = 0; t < (long)Ho * Wo; ++t) {
      int32_t bi = idx[(size_t)c * Ho * Wo + t];
      if (bi >= 0) gin[(size_t)c * H * W + bi] += gout[(size_t)c * Ho * Wo + t];
    }
  }
}

/* nn.SpatialAdaptiveMaxPooling(kw,kh) on the strided sub-window win (1-based inclusive):
 * output cell (i,j) covers rows [floor(i*h/kh), ceil((i+1)*h/kh)) of the window, first max
 * wins in row-major scan with strict >.  (objective.lua:117-118, Detector.lua:96-97) */
void orc_adaptive_max_pool_fwd(const float *fmap, int C, int H, int W, const int *win, int kh,
                               int kw, float *out, int32_t *idx) {
  int r0 = win[0] - 1, c0 = win[2] - 1;
  int h = win[1] - win[0] + 1, w = win[3] - win[2] + 1;
  (void)H;
#pragma omp parallel for
  for (int c = 0; c < C; ++c) {
    const float *ip = fmap + (size_t)c * H * W;
    for (int i = 0; i < kh; ++i) {
      int ys = (int)floor((double)i * h / kh), ye = (int)ceil((double)(i + 1) * h / kh);
      for (int j = 0; j < kw; ++j) {
        int xs = (int)floor((double)j * w / kw), xe = (int)ceil((double)(j + 1) * w / kw);
        float best = -FLT_MAX;
        int bi = -1;
        for (int y = ys; y < ye; ++y)
          for (int x = xs; x < xe; ++x) {
            float v = ip[(size_t)(r0 + y) * W + (c0 + x)];
            if (v > best) { best = v; bi = (r0 + y) * W + (c0 + x); }
          }
        out[((size_t)c * kh + i) * kw + j] = best;
        idx[((size_t)c * kh + i) * kw + j] = bi;
      }
    }
  }
}

void orc_adaptive_max_pool_bwd(float *gmap, int C, int H, int W, int kh, int kw,
                               const float *gout, const int32_t *idx) {
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < kh * kw; ++t) {
      int32_t bi = idx[(size_t)c * kh * kw + t];
      if (bi >= 0) gmap[(size_t)c * H * W + bi] += gout[(size_t)c * kh * kw + t];
    }
}

/* nn.Linear: y = x W^T + b, W is O x I (model_utilities.lua:82,99,103) */
void orc_linear_fwd(const float *x, int R, int I, const float *wt, const float *b, int O,
                    float *y) {
#pragma omp parallel for schedule(static)
  for (int o = 0; o < O; ++o) {
    const float *wr = wt + (size_t)o * I;
    for (int r = 0; r < R; ++r) {
      const float *xr = x + (size_t)r * I;
      double s = b ? (double)b[o] : 0.0;
      for (int i = 0; i < I; ++i) s += (double)xr[i] * (double)wr[i];
      y[(size_t)r * O + o] = (float)s;
    }
  }
}

void orc_linear_bwd(const float *x, const float *gy, int R, int I, const float *wt, int O,
                    float *gx, float *gw, float *gb) {
  if (gx) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
      double *acc = (double *)calloc(I, sizeof(double));
      for (int o = 0; o < O; ++o) {
        double g = gy[(size_t)r * O + o];
        if (g == 0.0) continue;
        const float *wr = wt + (size_t)o * I;
        for (int i = 0; i < I; ++i) acc[i] += g * (double)wr[i];
      }
      for (int i = 0; i < I; ++i) gx[(size_t)r * I + i] = (float)acc[i];
      free(acc);
    }
  }
  if (gw) {
#pragma omp parallel for schedule(static)
    for (int o = 0; o < O; ++o) {
      double *acc = (double *)calloc(I, sizeof(double));
      double sb = 0;
      for (int r = 0; r < R; ++r) {
        double g = gy[(size_t)r * O + o];
        sb += g;
        if (g == 0.0) continue;
        const float *xr = x + (size_t)r * I;
        for (int i = 0; i < I; ++i) acc[i] += g * (double)xr[i];
      }
      float *gwr = gw + (size_t)o * I;
      for (int i = 0; i < I; ++i) gwr[i] = (float)((double)gwr[i] + acc[i]);
      if (gb) gb[o] = (float)((double)gb[o] + sb);
      free(acc);
    }
  }
}

/* nn.LogSoftMax, max-shifted, fp32 result */
void orc_log_softmax(const float *x, int R, int n, float *y) {
  for (int r = 0; r < R; ++r) {
    const float *xr = x + (size_t)r * n;
    double m = xr[0];
    for (int i = 1; i < n; ++i) if (xr[i] > m) m = xr[i];
    double s = 0;
    for (int i = 0; i < n; ++i) s += exp((double)xr[i] - m);
    double lse = m + log(s);
    for (int i = 0; i < n; ++i) y[(size_t)r * n + i] = (float)((double)xr[i] - lse);
  }
}

/* optim.rmsprop [ext]: m = alpha*m + (1-alpha)*g^2 ; x = x - lr * g / (sqrt(m) + eps) (fp32 ops) */
void orc_rmsprop(float *x, const float *g, float *m, long n, float lr, float alpha, float eps) {
#pragma omp parallel for
  for (long i = 0; i < n; ++i) {
    float mi = alpha * m[i] + (1.0f - alpha) * (g[i] * g[i]);
    m[i] = mi;
    x[i] = x[i] - lr * g[i] / (sqrtf(mi) + eps);
  }
}
