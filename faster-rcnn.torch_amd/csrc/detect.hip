// detect.hip -- the glue of Detector:detect (Detector.lua:17-141) kept ON THE DEVICE between its big steps, so that a
// frame needs two read-backs instead of a dozen:
//   roi_windows   extract_roi_pooling_input (objective.lua:5-13) for every NMS candidate: Localizer:inputToFeatureRect
//                 (Localizer.lua:41-67) in the reference's double arithmetic, clip, 1-based window -- one thread per ROI
//   detect_post   Detector.lua:106-122: class test (class != background and p > 0.2), r2 = Anchors.anchorToInput(r, bbox)
//                 in double, ORDERED compaction of the survivors into the box / class arrays the per-class NMS reads;
//                 the survivor count stays on the device (frcnn_nms_device_n reads it there)
//   detect_gather one record per winner (class, candidate row, confidence, p, anchor rect, decoded rect, anchor index)
// Gather / scan work on a few thousand rows: latency-bound, no MFMA.  Double arithmetic follows the host mirror
// operation by operation; FMA contraction is off for this translation unit (products and sums separately rounded, as
// Lua numbers are).  exp() is the device library's double exp: it may differ from the host libm in the last bit (both
// are within an ulp), far inside the 1e-3 bar the decoded rects are compared at.
#pragma clang fp contract(off)
#include "kernels.h"

namespace frcnn {

struct LocLayers { int n; int l[24][6]; };   // {kW, kH, dW, dH, padW, padH} per layer, input first

__device__ __forceinline__ double lua_mod(double a, double b) { return a - floor(a / b) * b; }   // Lua 5.1: a - floor(a/b)*b

// rect [n][4] double (input space); pick (optional) 1-based rows of rect; wins [k][4] = {row_lo, row_hi, col_lo, col_hi}
__global__ void roi_windows_kernel(const double* __restrict__ rect, const long long* __restrict__ pick, int k, LocLayers L,
                                   int fmH, int fmW, int* __restrict__ wins) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= k) return;
  const double* q = rect + 4 * (pick ? (size_t)(pick[r] - 1) : (size_t)r);
  double minX = q[0], minY = q[1], maxX = q[2], maxY = q[3];
  for (int i = 0; i < L.n; ++i) {   // Localizer.lua:44-64 (the dH / dW mix-ups are the reference's)
    const double kW = L.l[i][0], kH = L.l[i][1], dW = L.l[i][2], dH = L.l[i][3], padW = L.l[i][4], padH = L.l[i][5];
    if (dW < kW) { minX -= kW - dW; minY -= kH - dH; maxX += kW - dW; maxY += kH - dH; }
    minX += padW; minY += padH; maxX += padW; maxY += padH;
    minX = minX / dH;
    minY = minY / dH;
    const double ax = maxX - kW;
    maxX = fmax(lua_mod(ax, dW) == 0.0 ? ax / dW + 1.0 : ceil(ax / dW) + 1.0, minX + 1.0);
    const double ay = maxY - kH;
    maxY = fmax(lua_mod(ay, dH) == 0.0 ? ay / dW + 1.0 : ceil(ay / dH) + 1.0, minY + 1.0);
  }
  minX = floor(minX); minY = floor(minY); maxX = ceil(maxX); maxY = ceil(maxY);   // :66 snapToInt
  // Rect.clip to [0, W] x [0, H] (Rect.lua:73-80), then objective.lua:11
  minX = fmin(fmax(minX, 0.0), (double)fmW); minY = fmin(fmax(minY, 0.0), (double)fmH);
  maxX = fmax(fmin(maxX, (double)fmW), 0.0); maxY = fmax(fmin(maxY, (double)fmH), 0.0);
  int* w = wins + 4 * r;
  w[0] = (int)fmin(minY + 1.0, maxY); w[1] = (int)maxY; w[2] = (int)fmin(minX + 1.0, maxX); w[3] = (int)maxX;
}

int roi_windows(const double* rect, const long long* pick, int k, const int* layers, int nlayers, int fmH, int fmW, int* wins,
                hipStream_t s) {
  if (k <= 0) return FRCNN_OK;
  FR_CHECK(nlayers >= 0 && nlayers <= 24, "roi_windows: %d localizer layers (at most 24)", nlayers);
  LocLayers L;
  L.n = nlayers;
  for (int i = 0; i < nlayers; ++i)
    for (int j = 0; j < 6; ++j) L.l[i][j] = layers[6 * i + j];
  FR_LAUNCH(KC_ROI, 0, k * 48.0, s, roi_windows_kernel, dim3(cdiv(k, 64)), dim3(64), 0, rect, pick, k, L, fmH, fmW, wins);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// One block, ordered compaction.  For candidate r (row pick[r] of the match arrays): keep iff cls != bgclass and
// exp(conf) > min_conf.  Survivor j (in candidate order): bb[j] = {float(r2), conf}, kc[j] = class, keep_row[j] = r,
// r2[j] = decoded rect in double; *K_dev = number of survivors.
#define DP_THREADS 1024
__global__ __launch_bounds__(DP_THREADS) void detect_post_kernel(const int* __restrict__ cls, const float* __restrict__ conf,
                                                                 const float* __restrict__ bbox, const double* __restrict__ rect,
                                                                 const long long* __restrict__ pick, int R, int bgclass,
                                                                 double min_conf, float* __restrict__ bb, int* __restrict__ kc,
                                                                 int* __restrict__ keep_row, double* __restrict__ r2out,
                                                                 int* __restrict__ K_dev) {
  __shared__ int wsum[DP_THREADS / 64];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r0 = 0; r0 < R; r0 += DP_THREADS) {
    const int r = r0 + threadIdx.x;
    bool keep = false;
    if (r < R) keep = cls[r] != bgclass && exp((double)conf[r]) > min_conf;   // Detector.lua:115
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (keep) {
      const int j = off + before;
      const double* a = rect + 4 * (size_t)(pick[r] - 1);   // the anchor's input rect (Detector.lua:106)
      const double aw = a[2] - a[0], ah = a[3] - a[1];
      const float* t = bbox + 4 * (size_t)r;
      // Anchors.anchorToInput (Anchors.lua:245-252): products and sums separately rounded, as Lua numbers are
      double x0 = (double)t[0] * aw; x0 = x0 + a[0];
      double y0 = (double)t[1] * ah; y0 = y0 + a[1];
      const double ew = exp((double)t[2]) * aw, eh = exp((double)t[3]) * ah;
      const double x1 = x0 + ew, y1 = y0 + eh;     // Rect.fromXYWidthHeight
      r2out[4 * (size_t)j] = x0; r2out[4 * (size_t)j + 1] = y0; r2out[4 * (size_t)j + 2] = x1; r2out[4 * (size_t)j + 3] = y1;
      float* b = bb + 5 * (size_t)j;               // r.r2:totensor() (FloatTensor) + the confidence column
      b[0] = (float)x0; b[1] = (float)y0; b[2] = (float)x1; b[3] = (float)y1; b[4] = conf[r];
      kc[j] = cls[r];
      keep_row[j] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < DP_THREADS / 64; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *K_dev = base;
}

int detect_post(const int* cls, const float* conf, const float* bbox, const double* rect, const long long* pick, int R,
                int bgclass, double min_conf, float* bb, int* kc, int* keep_row, double* r2, int* K_dev, hipStream_t s) {
  if (R <= 0) { FR_HIP(hipMemsetAsync(K_dev, 0, sizeof(int), s)); return FRCNN_OK; }
  FR_LAUNCH(KC_ELEMWISE, 0, R * 64.0, s, detect_post_kernel, dim3(1), dim3(DP_THREADS), 0, cls, conf, bbox, rect, pick, R, bgclass,
            min_conf, bb, kc, keep_row, r2, K_dev);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// rec[q][16] (double) for winner q < *nwin_dev (pick order of the per-class NMS):
//   0 class, 1 candidate row (1-based, among the NMS candidates), 2 confidence (log-prob), 3 p (log-prob of the anchor),
//   4-7 anchor rect, 8-11 r2, 12-15 anchor index {layer, aspect, y, x}
__global__ void detect_gather_kernel(const long long* __restrict__ wpick, const int* __restrict__ nwin_dev, int cap,
                                     const int* __restrict__ keep_row, const int* __restrict__ kc, const float* __restrict__ bb,
                                     const double* __restrict__ r2, const long long* __restrict__ pick,
                                     const float* __restrict__ mp, const double* __restrict__ rect, const int* __restrict__ midx,
                                     double* __restrict__ rec) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= min(*nwin_dev, cap)) return;
  const int j = (int)(wpick[q] - 1);
  const int r = keep_row[j];
  const size_t i = (size_t)(pick[r] - 1);
  double* o = rec + 16 * (size_t)q;
  o[0] = kc[j]; o[1] = r + 1; o[2] = bb[5 * (size_t)j + 4]; o[3] = mp[i];
  for (int t = 0; t < 4; ++t) { o[4 + t] = rect[4 * i + t]; o[8 + t] = r2[4 * (size_t)j + t]; o[12 + t] = midx[4 * i + t]; }
}

int detect_gather(const long long* wpick, const int* nwin_dev, int cap, const int* keep_row, const int* kc, const float* bb,
                  const double* r2, const long long* pick, const float* mp, const double* rect, const int* midx, double* rec,
                  hipStream_t s) {
  if (cap <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_ELEMWISE, 0, cap * 128.0, s, detect_gather_kernel, dim3(cdiv(cap, 64)), dim3(64), 0, wpick, nwin_dev, cap, keep_row,
            kc, bb, r2, pick, mp, rect, midx, rec);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
