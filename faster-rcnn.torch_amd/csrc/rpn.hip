// rpn.hip -- the anchor-level work of the reference, batched into single launches:
//  * rpn_scan : Detector.lua:39-66 -- for every (layer, y, x, aspect): 2-way LogSoftMax,
//               exp(c1) > 0.95, Anchors:get (Anchors.lua:60-67), Anchors.anchorToInput
//               (Anchors.lua:245-252), Rect:overlaps(image) (Rect.lua:90-93); survivors are
//               compacted IN SCAN ORDER (the order NMS ids refer to).
//  * rpn_loss : objective.lua:91-140 -- sparse CrossEntropy(2-way) + SmoothL1 x10 on the sampled
//               anchors, gradients scattered into delta_outputs, plus the cnet targets
//               (objective.lua:155-159).
// Gather / compaction kernels: coalesced reads of the 18-channel head maps (637 KB at 800x450),
// wave ballots for the ordered compaction, no MFMA.  Scalar geometry is done in fp64 because the
// reference does it in Lua numbers (doubles) on fp32 table/tensor values.
#include "kernels.h"

namespace frcnn {

struct ScanArgs {
  const float* map[4];
  int H[4], W[4];
  int start[5];  // prefix of 3*H*W per layer
};

__global__ void rpn_scan_kernel(ScanArgs a, const float* __restrict__ aw, const float* __restrict__ ah,
                                double img_w, double img_h, double thr, unsigned char* __restrict__ flag,
                                float* __restrict__ dp, double* __restrict__ drect) {
  const int total = a.start[4];
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < total; n += gridDim.x * blockDim.x) {
    int l = n >= a.start[3] ? 3 : (n >= a.start[2] ? 2 : (n >= a.start[1] ? 1 : 0));
    int r = n - a.start[l];
    int asp = r % 3;
    int pix = r / 3;
    const int W = a.W[l];
    const long hw = (long)a.H[l] * W;
    const int y = pix / W, x = pix - y * W;
    const float* m = a.map[l] + (size_t)(asp * 6) * hw + pix;
    const float v0 = m[0], v1 = m[hw];
    // nn.LogSoftMax (max-shifted), fp32 result like the reference's CudaTensor
    const double mx = v0 > v1 ? (double)v0 : (double)v1;
    const double lse = mx + log(exp((double)v0 - mx) + exp((double)v1 - mx));
    const float c1 = (float)((double)v0 - lse);
    unsigned char f = 0;
    if (exp((double)c1) > thr) {  // Detector.lua:54
      const float* wt = aw + (((size_t)l * 3 + asp) * 200 + x) * 2;
      const float* ht = ah + (((size_t)l * 3 + asp) * 200 + y) * 2;
      const double ax0 = wt[0], ax1 = wt[1], ay0 = ht[0], ay1 = ht[1];
      const double awd = ax1 - ax0, ahd = ay1 - ay0;
      const double t0 = m[2 * hw], t1 = m[3 * hw], t2 = m[4 * hw], t3 = m[5 * hw];
      const double rx = t0 * awd + ax0;
      const double ry = t1 * ahd + ay0;
      const double rw = exp(t2) * awd;
      const double rh = exp(t3) * ahd;
      const double rx1 = rx + rw, ry1 = ry + rh;
      if (rx < img_w && rx1 > 0.0 && ry < img_h && ry1 > 0.0) {  // Rect.lua:90-93 vs (0,0,W,H)
        f = 1;
        dp[n] = c1;
        double* d = drect + 4 * (size_t)n;
        d[0] = rx; d[1] = ry; d[2] = rx1; d[3] = ry1;
      }
    }
    flag[n] = f;
  }
}

// single workgroup, ordered compaction by wave ballots
__global__ __launch_bounds__(1024) void rpn_compact_kernel(ScanArgs a, const unsigned char* __restrict__ flag,
                                                           const float* __restrict__ dp,
                                                           const double* __restrict__ drect, int cap,
                                                           float* __restrict__ match_p, int* __restrict__ match_idx,
                                                           double* __restrict__ match_rect,
                                                           float* __restrict__ match_box, int* __restrict__ count) {
  __shared__ int wave_cnt[16];
  __shared__ int base_sh;
  const int total = a.start[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_sh = 0;
  __syncthreads();
  for (int n0 = 0; n0 < total; n0 += 1024) {
    const int n = n0 + threadIdx.x;
    const bool f = n < total && flag[n];
    const unsigned long long b = __ballot(f);
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int off = base_sh;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int pos = off + __popcll(b & ((1ull << lane) - 1ull));
    if (f && pos < cap) {
      int l = n >= a.start[3] ? 3 : (n >= a.start[2] ? 2 : (n >= a.start[1] ? 1 : 0));
      int r = n - a.start[l];
      int asp = r % 3, pix = r / 3;
      int y = pix / a.W[l], x = pix - y * a.W[l];
      match_p[pos] = dp[n];
      int* mi = match_idx + 4 * (size_t)pos;
      mi[0] = l + 1; mi[1] = asp + 1; mi[2] = y + 1; mi[3] = x + 1;  // 1-based like Anchors:get
      const double* d = drect + 4 * (size_t)n;
      double* o = match_rect + 4 * (size_t)pos;
      float* ob = match_box + 4 * (size_t)pos;
      for (int t = 0; t < 4; ++t) { o[t] = d[t]; ob[t] = (float)d[t]; }  // Detector.lua:74-79
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int s = 0;
      for (int w = 0; w < 16; ++w) s += wave_cnt[w];
      base_sh += s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = base_sh;
}

static void fill_scan_args(const RpnLayers& L, ScanArgs* a) {
  a->start[0] = 0;
  for (int l = 0; l < 4; ++l) {
    a->map[l] = L.map[l]; a->H[l] = L.H[l]; a->W[l] = L.W[l];
    a->start[l + 1] = a->start[l] + 3 * L.H[l] * L.W[l];
  }
}

size_t rpn_scan_workspace_bytes(const RpnLayers& L) {
  ScanArgs a;
  fill_scan_args(L, &a);
  size_t n = (size_t)a.start[4];
  return 256 + n * 40 + ((n + 255) / 256) * 256;
}

int rpn_scan(const RpnLayers& L, const float* anchor_w, const float* anchor_h, double img_w,
             double img_h, double p_threshold, int cap, float* match_p, int* match_idx,
             double* match_rect, float* match_box, int* count, void* ws, size_t ws_bytes,
             hipStream_t s) {
  ScanArgs a;
  fill_scan_args(L, &a);
  for (int l = 0; l < 4; ++l)
    FR_CHECK(L.H[l] <= 200 && L.W[l] <= 200, "rpn_scan: head map %d is %dx%d, anchor tables hold 200 (Anchors.lua:15)",
             l + 1, L.H[l], L.W[l]);
  const size_t n = (size_t)a.start[4];
  FR_CHECK(ws_bytes >= rpn_scan_workspace_bytes(L), "rpn_scan: workspace too small");
  char* base = (char*)(((uintptr_t)ws + 255) / 256 * 256);
  double* drect = (double*)base;
  float* dp = (float*)(base + n * 32);
  unsigned char* flag = (unsigned char*)(base + n * 36 + 64);
  int grid = (int)std::min<size_t>((n + 255) / 256, 1024);
  FR_LAUNCH(KC_RPN, 0, n * 24.0, s, rpn_scan_kernel, dim3(grid), dim3(256), 0, a, anchor_w, anchor_h, img_w,
            img_h, p_threshold, flag, dp, drect);
  FR_LAUNCH(KC_RPN, 0, n * 1.0, s, rpn_compact_kernel, dim3(1), dim3(1024), 0, a,
            (const unsigned char*)flag, (const float*)dp, (const double*)drect, cap, match_p, match_idx,
            match_rect, match_box, count);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------
// sparse RPN loss.  One thread per example.  ex_idx[e] = {layer, aspect, y, x} (1-based).
// Positives first (e < npos), then negatives.
// ------------------------------------------------------------------------------------------
struct LossArgs {
  const float* map[4];
  float* delta[4];
  int H[4], W[4];
};

// exg (deterministic mode): the six gradient values of example e go to exg[e][0..5] instead of being added to the delta maps;
// rpn_apply_kernel then adds them in example order (two sampled examples may name the same anchor).
__global__ void rpn_loss_kernel(LossArgs a, const int* __restrict__ ex_idx, const double* __restrict__ ex_anchor,
                                const double* __restrict__ ex_roi, const int* __restrict__ ex_class,
                                int npos, int nneg, int bgclass, double* __restrict__ ex_loss,
                                float* __restrict__ crtarget, float* __restrict__ cctarget, float* __restrict__ exg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npos + nneg) return;
  const int* ix = ex_idx + 4 * e;
  const int l = ix[0] - 1, asp = ix[1] - 1, y = ix[2] - 1, x = ix[3] - 1;
  const long hw = (long)a.H[l] * a.W[l];
  const long pix = (long)y * a.W[l] + x;
  const float* m = a.map[l] + (size_t)(asp * 6) * hw + pix;
  float* d = a.delta[l] + (size_t)(asp * 6) * hw + pix;
  const bool pos = e < npos;
  // nn.CrossEntropyCriterion on v[1..2], target 1 (fg) / 2 (bg): objective.lua:104,132
  const float v0 = m[0], v1 = m[hw];
  const double mx = v0 > v1 ? (double)v0 : (double)v1;
  const double lse = mx + log(exp((double)v0 - mx) + exp((double)v1 - mx));
  const float l0 = (float)((double)v0 - lse), l1 = (float)((double)v1 - lse);
  const double cls = pos ? -(double)l0 : -(double)l1;
  const float g0 = (float)(exp((double)l0) - (pos ? 1.0 : 0.0)), g1 = (float)(exp((double)l1) - (pos ? 0.0 : 1.0));
  if (exg) {
    exg[6 * (size_t)e] = g0; exg[6 * (size_t)e + 1] = g1;
    exg[6 * (size_t)e + 2] = exg[6 * (size_t)e + 3] = exg[6 * (size_t)e + 4] = exg[6 * (size_t)e + 5] = 0.f;
  } else {
    unsafeAtomicAdd(d, g0);       // :106 / :134
    unsafeAtomicAdd(d + hw, g1);
  }
  double reg = 0.0;
  float* crt = crtarget + 4 * (size_t)e;
  if (pos) {
    const double* an = ex_anchor + 4 * (size_t)e;
    const double* roi = ex_roi + 4 * (size_t)e;
    const double awd = an[2] - an[0], ahd = an[3] - an[1];
    // Anchors.inputToAnchor(anchor, roi.rect) -> FloatTensor (objective.lua:110)
    float tgt[4];
    tgt[0] = (float)((roi[0] - an[0]) / awd);
    tgt[1] = (float)((roi[1] - an[1]) / ahd);
    tgt[2] = (float)log((roi[2] - roi[0]) / awd);
    tgt[3] = (float)log((roi[3] - roi[1]) / ahd);
    float t[4];
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      t[c] = m[(2 + c) * hw];
      const float z = t[c] - tgt[c];
      const float az = fabsf(z);
      s += az < 1.0f ? 0.5 * (double)z * (double)z : (double)az - 0.5;  // SmoothL1, sizeAverage=false
      const float g = az < 1.0f ? z : (z > 0.f ? 1.0f : -1.0f);
      if (exg) exg[6 * (size_t)e + 2 + c] = g * 10.0f;
      else unsafeAtomicAdd(d + (2 + c) * hw, g * 10.0f);                // :113-114
    }
    reg = (double)(float)s * 10.0;                                      // :112
    // reg_proposal = Anchors.anchorToInput(anchor, reg_out) (objective.lua:111), then the cnet
    // regression target relative to the PROPOSAL (objective.lua:156)
    const double px = (double)t[0] * awd + an[0];
    const double py = (double)t[1] * ahd + an[1];
    const double pw = exp((double)t[2]) * awd;
    const double ph = exp((double)t[3]) * ahd;
    crt[0] = (float)((roi[0] - px) / pw);
    crt[1] = (float)((roi[1] - py) / ph);
    crt[2] = (float)log((roi[2] - roi[0]) / pw);
    crt[3] = (float)log((roi[3] - roi[1]) / ph);
    cctarget[e] = (float)ex_class[e];                                   // :155
  } else {
    crt[0] = crt[1] = crt[2] = crt[3] = 0.f;                            // :149 :zero()
    cctarget[e] = (float)bgclass;                                       // :159
  }
  ex_loss[2 * (size_t)e] = cls;
  ex_loss[2 * (size_t)e + 1] = reg;
}

__global__ void rpn_apply_kernel(LossArgs a, const int* __restrict__ ex_idx, const float* __restrict__ exg, int E) {
  const int c = threadIdx.x;   // one wave; lanes 0..5 own the six planes of an anchor, examples in order
  if (c >= 6) return;
  for (int e = 0; e < E; ++e) {
    const int* ix = ex_idx + 4 * e;
    const int l = ix[0] - 1, asp = ix[1] - 1, y = ix[2] - 1, x = ix[3] - 1;
    const long hw = (long)a.H[l] * a.W[l];
    float* d = a.delta[l] + (size_t)(asp * 6 + c) * hw + (long)y * a.W[l] + x;
    *d += exg[6 * (size_t)e + c];
  }
}

int rpn_loss(const RpnLayers& L, float* const* delta, const int* ex_idx, const double* ex_anchor,
             const double* ex_roi, const int* ex_class, int npos, int nneg, int bgclass,
             double* ex_loss, float* crtarget, float* cctarget, hipStream_t s) {
  const int E = npos + nneg;
  if (E <= 0) return FRCNN_OK;
  LossArgs a;
  for (int l = 0; l < 4; ++l) {
    a.map[l] = L.map[l]; a.delta[l] = delta[l]; a.H[l] = L.H[l]; a.W[l] = L.W[l];
  }
  float* exg = nullptr;
  if (deterministic()) FR_TRY(det_workspace(s, (size_t)E * 6, &exg));
  FR_LAUNCH(KC_RPN, 0, E * 200.0, s, rpn_loss_kernel, dim3(cdiv(E, 64)), dim3(64), 0, a, ex_idx, ex_anchor,
            ex_roi, ex_class, npos, nneg, bgclass, ex_loss, crtarget, cctarget, exg);
  if (exg) FR_LAUNCH(KC_RPN, 0, E * 48.0, s, rpn_apply_kernel, dim3(1), dim3(64), 0, a, ex_idx, (const float*)exg, E);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

// acc[0] += sum_e ex_loss[e][0]; acc[1] += sum_e ex_loss[e][1], summed in a fixed order by one
// wave (E is a few hundred): the fp64 Lua accumulators cls_loss / reg_loss of objective.lua:52.
__global__ void loss_accumulate_kernel(const double* __restrict__ ex_loss, int E, double* acc) {
  // one wave: lane-strided partial sums, then a fixed-order butterfly (deterministic)
  double c = 0.0, r = 0.0;
  for (int e = threadIdx.x; e < E; e += 64) { c += ex_loss[2 * e]; r += ex_loss[2 * e + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o, 64); r += __shfl_down(r, o, 64); }
  if (threadIdx.x == 0) { acc[0] += c; acc[1] += r; }
}
int loss_accumulate(const double* ex_loss, int E, double* acc, hipStream_t s) {
  if (E <= 0) return FRCNN_OK;
  FR_LAUNCH(KC_RPN, 0, E * 16.0, s, loss_accumulate_kernel, dim3(1), dim3(64), 0, ex_loss, E, acc);
  FR_LAUNCH_CHECK();
  return FRCNN_OK;
}

}  // namespace frcnn
