// api.cpp -- C ABI plumbing of libfrcnn_hip.so (include/frcnn_hip.h): errors, device buffers,
// the HIP-event profiler, and the per-operator entry points that wrap the kernels.
#include <cmath>
#include <cstring>
#include <vector>
#include <mutex>

#include "kernels.h"

namespace frcnn {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---------------------------------------------------------------- profiler
struct ProfRec { int klass; double flops, bytes; hipEvent_t a, b; hipStream_t s; };
static hipEvent_t g_prof_base = nullptr;
static unsigned g_prof_mask = 0;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_pending;
static std::mutex g_mu;

bool prof_enabled(int klass) { return (g_prof_mask >> klass) & 1u; }
static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void prof_before(int, hipStream_t s) {
  g_pending = get_event();
  (void)hipEventRecord(g_pending, s);
}
void prof_after(int klass, double flops, double bytes, hipStream_t s) {
  ProfRec r;
  r.klass = klass; r.flops = flops; r.bytes = bytes; r.a = g_pending; r.b = get_event(); r.s = s;
  (void)hipEventRecord(r.b, s);
  g_recs.push_back(r);
}

}  // namespace frcnn

using namespace frcnn;

static_assert(frcnn::KC_COUNT == FRCNN_KC_COUNT && frcnn::KC_IMAGE == FRCNN_KC_IMAGE && frcnn::KC_OPTIM == FRCNN_KC_OPTIM &&
              frcnn::KC_CONV_IGEMM_K3 == FRCNN_KC_CONV_IGEMM_K3, "kernel classes of common.h and frcnn_hip.h differ");

extern "C" {

int frcnn_version(void) { return 100; }
const char* frcnn_last_error(void) { return g_err.c_str(); }

int frcnn_device_count(int* n) {
  FR_HIP(hipGetDeviceCount(n));
  return FRCNN_OK;
}
int frcnn_set_device(int device) {
  FR_HIP(hipSetDevice(device));
  return FRCNN_OK;
}
int frcnn_device_name(char* buf, int len) {
  int dev;
  FR_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  FR_HIP(hipGetDeviceProperties(&p, dev));
  snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return FRCNN_OK;
}

int frcnn_malloc(void** ptr, size_t bytes) {
  FR_HIP(hipMalloc(ptr, (bytes ? bytes : 16) + 64));   // (64 bytes of slack, as the model's own buffers: see conv_wgradx's loaders)
  return FRCNN_OK;
}
int frcnn_free(void* ptr) {
  FR_HIP(hipFree(ptr));
  return FRCNN_OK;
}
int frcnn_host_alloc(void** ptr, size_t bytes) {
  FR_CHECK(ptr, "frcnn_host_alloc: NULL result pointer");
  FR_HIP(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
  return FRCNN_OK;
}
int frcnn_host_free(void* ptr) {
  FR_HIP(hipHostFree(ptr));
  return FRCNN_OK;
}
int frcnn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  FR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
  return FRCNN_OK;
}
int frcnn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  FR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
  return FRCNN_OK;
}
int frcnn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  FR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream)));
  return FRCNN_OK;
}
int frcnn_stream_sync(void* stream) {
  FR_HIP(hipStreamSynchronize(S(stream)));
  return FRCNN_OK;
}
int frcnn_zero(void* ptr, size_t bytes, void* stream) { return fill_zero(ptr, bytes, S(stream)); }
int frcnn_scale(float* x, long long n, float s, void* stream) { return scale_inplace(x, n, s, S(stream)); }
int frcnn_add(float* y, const float* x, long long n, void* stream) {
  FR_CHECK(y && x && n >= 0, "frcnn_add: bad arguments");
  return n ? add_inplace(y, x, n, S(stream)) : FRCNN_OK;
}

int frcnn_prof_enable(int class_mask) {
  if (class_mask && !g_prof_mask && getenv("FRCNN_PROF_DUMP")) {   // debug: time base for the per-launch start offsets
    if (!g_prof_base) FR_HIP(hipEventCreate(&g_prof_base));
    FR_HIP(hipDeviceSynchronize());
    FR_HIP(hipEventRecord(g_prof_base, nullptr));
  }
  g_prof_mask = (unsigned)class_mask;
  return FRCNN_OK;
}
int frcnn_prof_collect(long long* launches, double* ms, double* flops, double* bytes) {
  FR_HIP(hipDeviceSynchronize());
  for (int k = 0; k < KC_COUNT; ++k) { launches[k] = 0; ms[k] = 0; flops[k] = 0; bytes[k] = 0; }
  FILE* dump = (getenv("FRCNN_PROF_DUMP") && g_prof_base) ? fopen(getenv("FRCNN_PROF_DUMP"), "a") : nullptr;
  for (auto& r : g_recs) {
    float t = 0.f;
    FR_HIP(hipEventElapsedTime(&t, r.a, r.b));
    if (dump) {   // debug timeline: class, stream, start offset (us) from frcnn_prof_enable, duration (us)
      float t0 = 0.f;
      if (hipEventElapsedTime(&t0, g_prof_base, r.a) == hipSuccess)
        fprintf(dump, "%d %p %.1f %.1f\n", r.klass, (void*)r.s, t0 * 1e3, t * 1e3);
    }
    launches[r.klass] += 1; ms[r.klass] += t; flops[r.klass] += r.flops; bytes[r.klass] += r.bytes;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  if (dump) { fprintf(dump, "-1 0 0 0\n"); fclose(dump); }
  g_recs.clear();
  return FRCNN_OK;
}

// ---------------------------------------------------------------- nms
size_t frcnn_nms_workspace_bytes(int n) { return nms_workspace_bytes(n); }
int frcnn_nms_device(const float* boxes, int n, int ncols, float overlap, int key_mode, int key_col,
                     long long* pick, int* count, void* ws, size_t ws_bytes, void* stream) {
  return nms_device(boxes, n, ncols, overlap, key_mode, key_col, pick, count, ws, ws_bytes, S(stream));
}
int frcnn_nms_device_classes(const float* boxes, int n, int ncols, float overlap, int key_mode, int key_col, const int* cls,
                             long long* pick, int* count, void* ws, size_t ws_bytes, void* stream) {
  return nms_device(boxes, n, ncols, overlap, key_mode, key_col, pick, count, ws, ws_bytes, S(stream), cls);
}
int frcnn_nms_device_n(const float* boxes, int n_cap, const int* n_dev, int ncols, float overlap, int key_mode, int key_col,
                       const int* cls, long long* pick, int* count, void* ws, size_t ws_bytes, void* stream) {
  FR_CHECK(n_dev, "frcnn_nms_device_n: NULL device count");
  return nms_device(boxes, n_cap, ncols, overlap, key_mode, key_col, pick, count, ws, ws_bytes, S(stream), cls, n_dev);
}
int frcnn_roi_windows(const double* rect, const long long* pick, int k, const int* layers_host, int nlayers, int fmH, int fmW,
                      int* wins, void* stream) {
  FR_CHECK(rect && wins && (layers_host || nlayers == 0), "frcnn_roi_windows: NULL argument");
  return roi_windows(rect, pick, k, layers_host, nlayers, fmH, fmW, wins, S(stream));
}
int frcnn_detect_post(const int* cls, const float* conf, const float* bbox, const double* rect, const long long* pick, int R,
                      int bgclass, double min_conf, float* bb, int* kc, int* keep_row, double* r2, int* K_dev, void* stream) {
  return detect_post(cls, conf, bbox, rect, pick, R, bgclass, min_conf, bb, kc, keep_row, r2, K_dev, S(stream));
}
int frcnn_detect_gather(const long long* wpick, const int* nwin_dev, int cap, const int* keep_row, const int* kc, const float* bb,
                        const double* r2, const long long* pick, const float* match_p, const double* match_rect,
                        const int* match_idx, double* rec, void* stream) {
  return detect_gather(wpick, nwin_dev, cap, keep_row, kc, bb, r2, pick, match_p, match_rect, match_idx, rec, S(stream));
}
int frcnn_nms_host(const float* boxes_host, int n, int ncols, float overlap, int key_mode, int key_col,
                   long long* pick_host, int* count_host) {
  if (n <= 0) { *count_host = 0; return FRCNN_OK; }
  size_t wsb = nms_workspace_bytes(n);
  char* buf = nullptr;
  size_t bb = ((size_t)n * ncols * 4 + 255) / 256 * 256, pb = ((size_t)n * 8 + 255) / 256 * 256;
  FR_HIP(hipMalloc((void**)&buf, bb + pb + 256 + wsb));
  float* dboxes = (float*)buf;
  long long* dpick = (long long*)(buf + bb);
  int* dcount = (int*)(buf + bb + pb);
  void* ws = buf + bb + pb + 256;
  int rc = FRCNN_OK;
  hipError_t e = hipMemcpy(dboxes, boxes_host, (size_t)n * ncols * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) rc = nms_device(dboxes, n, ncols, overlap, key_mode, key_col, dpick, dcount, ws, wsb, 0);
  if (e == hipSuccess && rc == FRCNN_OK) e = hipMemcpy(count_host, dcount, 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == FRCNN_OK && *count_host > 0)
    e = hipMemcpy(pick_host, dpick, (size_t)*count_host * 8, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (e != hipSuccess) { set_error("frcnn_nms_host: %s", hipGetErrorString(e)); return FRCNN_ERR_HIP; }
  return rc;
}

// ---------------------------------------------------------------- conv (operator-level, for parity tests
// and hosts that drive single layers; the model runtime keeps packed weights resident instead)
// option x3_f16: the op-level split launches take the two-plane fp16 form; the magnitude records of their two operands (amax.h) and
// the scalar the pack publishes live in a buffer of the CALL (freed with it: two host threads or streams may run these entry points
// side by side)
#define X3_REC_BYTES ((size_t)(2 * AMAX_REC + 4) * 4)
static int x3_f16_records(float** out) {
  *out = nullptr;
  if (!get_x3_f16()) return FRCNN_OK;
  FR_HIP(hipMalloc((void**)out, X3_REC_BYTES));
  return FRCNN_OK;
}

// The two-plane fp16 form leaves ONE binade of headroom for a per-channel input scale (convx.hip: a dropout scale's entries are
// <= 1).  The operator-level entry points accept any caller-supplied scale: entries beyond 1 take the three-plane bf16 form,
// which has fp32's exponent range.  A host read -- these entry points synchronise anyway.
static int scale_fits_f16_form(const float* in_scale, int C, hipStream_t s, bool* ok) {
  *ok = true;
  if (!in_scale) return FRCNN_OK;
  std::vector<float> h((size_t)C);
  FR_HIP(hipMemcpyAsync(h.data(), in_scale, (size_t)C * 4, hipMemcpyDeviceToHost, s));
  FR_HIP(hipStreamSynchronize(s));
  for (float v : h) if (!(std::fabs(v) <= 1.f)) { *ok = false; break; }
  return FRCNN_OK;
}

int frcnn_conv2d_forward(const float* in, int C, int H, int W, const float* in_slope, const float* in_scale,
                         const float* weight, const float* bias, int O, int k, int pad, float* out,
                         void* stream) {
  float* wf = nullptr;
  if (conv_x3_eligible(C, O, k) && (k == 3 || (!in_slope && !in_scale))) {   // split-bf16 operand form (convx.hip)
    FR_HIP(hipMalloc((void**)&wf, conv_x3_pack_bytes(C, O, k)));
    float* am = nullptr;
    bool unit = true;
    int rcx = scale_fits_f16_form(in_scale, C, S(stream), &unit);
    if (rcx == FRCNN_OK && unit) rcx = x3_f16_records(&am);
    float* const amw = am ? am + AMAX_REC : nullptr;          // the weights' record
    float* const aws = am ? am + 2 * AMAX_REC : nullptr;      // their largest magnitude
    if (am && rcx == FRCNN_OK) {
      rcx = tensor_absmax(in, (long)C * H * W, am, S(stream));
      if (rcx == FRCNN_OK) rcx = tensor_absmax(weight, (long)O * C * k * k, amw, S(stream));
    }
    if (rcx == FRCNN_OK) rcx = conv_x3_pack(weight, O, C, k, 0, wf, S(stream), H + 2 * pad - k + 1, W + 2 * pad - k + 1, amw, aws);
    if (rcx == FRCNN_OK) rcx = conv_x3(in, C, H, W, in_slope, in_scale, wf, bias, O, k, pad, out, OUT_STORE, 0, S(stream), 0, nullptr, am, aws);
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(wf); (void)hipFree(am);
    return rcx;
  }
  FR_HIP(hipMalloc((void**)&wf, conv_pack_floats(C, O, k) * 4));
  int rc = conv_pack_weights(weight, O, C, k, wf, nullptr, S(stream));
  if (rc == FRCNN_OK) rc = conv_igemm(in, C, H, W, in_slope, in_scale, wf, bias, O, k, pad, out, OUT_STORE, 0, S(stream));
  (void)hipStreamSynchronize(S(stream));
  (void)hipFree(wf);
  return rc;
}
int frcnn_conv2d_backward_input(const float* gout, int O, int Ho, int Wo, const float* weight, int C, int k,
                                int pad, float* gin, int accumulate, void* stream) {
  float* wd = nullptr;
  if (conv_x3_eligible(O, C, k)) {
    FR_HIP(hipMalloc((void**)&wd, conv_x3_pack_bytes(O, C, k)));
    float* am = nullptr;
    int rcx = x3_f16_records(&am);
    float* const amw = am ? am + AMAX_REC : nullptr;
    float* const aws = am ? am + 2 * AMAX_REC : nullptr;
    if (am && rcx == FRCNN_OK) {
      rcx = tensor_absmax(gout, (long)O * Ho * Wo, am, S(stream));
      if (rcx == FRCNN_OK) rcx = tensor_absmax(weight, (long)O * C * k * k, amw, S(stream));
    }
    if (rcx == FRCNN_OK) rcx = conv_x3_pack(weight, O, C, k, 1, wd, S(stream), Ho + 2 * (k - 1 - pad) - k + 1, Wo + 2 * (k - 1 - pad) - k + 1, amw, aws);
    if (rcx == FRCNN_OK) rcx = conv_x3(gout, O, Ho, Wo, nullptr, nullptr, wd, nullptr, C, k, k - 1 - pad, gin, accumulate ? OUT_ADD : OUT_STORE, 0, S(stream), 0, nullptr, am, aws);
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(wd); (void)hipFree(am);
    return rcx;
  }
  FR_HIP(hipMalloc((void**)&wd, conv_pack_floats(O, C, k) * 4));
  int rc = conv_pack_weights(weight, O, C, k, nullptr, wd, S(stream));
  if (rc == FRCNN_OK)
    rc = conv_igemm(gout, O, Ho, Wo, nullptr, nullptr, wd, nullptr, C, k, k - 1 - pad, gin,
                    accumulate ? OUT_ADD : OUT_STORE, 0, S(stream));
  (void)hipStreamSynchronize(S(stream));
  (void)hipFree(wd);
  return rc;
}
int frcnn_conv2d_backward_weight(const float* in, int C, int H, int W, const float* in_slope,
                                 const float* in_scale, const float* gout, int O, int k, int pad,
                                 float* gweight, float* gbias, void* stream) {
  size_t wsb = conv_wgrad_workspace_bytes(C, H, W, O, k, pad);
  void* ws = nullptr;
  FR_HIP(hipMalloc(&ws, wsb));
  int rc = FRCNN_OK;
  float* am = nullptr;   // (fp16 form: records of both tensors)
  bool unit = true;
  rc = scale_fits_f16_form(in_scale, C, S(stream), &unit);
  if (rc == FRCNN_OK && unit && k == 3 && conv_wgradx_eligible(C, O, k)) rc = x3_f16_records(&am);
  if (am && rc == FRCNN_OK) {
    rc = tensor_absmax(in, (long)C * H * W, am, S(stream));
    if (rc == FRCNN_OK) rc = tensor_absmax(gout, (long)O * (H + 2 * pad - k + 1) * (W + 2 * pad - k + 1), am + AMAX_REC, S(stream));
  }
  if (rc == FRCNN_OK)
    rc = conv_wgrad(in, C, H, W, in_slope, in_scale, gout, O, k, pad, gweight, ws, wsb, S(stream), nullptr, am, am ? am + AMAX_REC : nullptr);
  (void)hipStreamSynchronize(S(stream));
  (void)hipFree(ws); (void)hipFree(am);
  FR_TRY(rc);
  if (gbias) {
    int Ho = H + 2 * pad - k + 1, Wo = W + 2 * pad - k + 1;
    FR_TRY(channel_sum(gout, O, (long)Ho * Wo, gbias, S(stream)));
  }
  return FRCNN_OK;
}

int frcnn_maxpool_act_forward(const float* x, int C, int H, int W, const float* slope, const float* scale,
                              float* out, unsigned char* idx, void* stream) {
  return maxpool_act_forward(x, C, H, W, slope, scale, out, idx, S(stream));
}
int frcnn_maxpool_act_backward(const float* gpool, const unsigned char* idx, const float* x, int C, int H, int W,
                               const float* slope, const float* scale, float* gx, float* gbias, float* gslope,
                               void* stream) {
  return maxpool_act_backward(gpool, idx, x, C, H, W, slope, scale, gx, gbias, gslope, S(stream));
}
int frcnn_act_forward(const float* x, int C, long long hw, const float* slope, const float* scale, float* y,
                      void* stream) {
  return act_forward(x, C, hw, slope, scale, y, S(stream));
}
int frcnn_act_backward(const float* gy, const float* x, int C, long long hw, const float* slope,
                       const float* scale, float* gx, float* gbias, float* gslope, void* stream) {
  return act_backward(gy, x, C, hw, slope, scale, gx, gbias, gslope, S(stream));
}

int frcnn_roi_pool_forward(const float* fmap, int C, int H, int W, const int* wins, int R, int kh, int kw,
                           float* out, int* idx, void* stream) {
  return roi_pool_forward(fmap, C, H, W, wins, R, kh, kw, out, idx, S(stream));
}
int frcnn_roi_pool_backward(float* gmap, int C, int H, int W, const float* gout, const int* idx, int R, int kh,
                            int kw, void* stream) {
  return roi_pool_backward(gmap, C, H, W, gout, idx, R, kh, kw, S(stream));
}

static void to_layers(const float* const* maps, const int* H, const int* W, RpnLayers* L) {
  for (int l = 0; l < 4; ++l) { L->map[l] = maps ? maps[l] : nullptr; L->H[l] = H[l]; L->W[l] = W[l]; }
}
size_t frcnn_rpn_scan_workspace_bytes(const int* H, const int* W) {
  RpnLayers L;
  to_layers(nullptr, H, W, &L);
  return rpn_scan_workspace_bytes(L);
}
int frcnn_rpn_scan(const float* const* maps, const int* H, const int* W, const float* anchor_w,
                   const float* anchor_h, double img_w, double img_h, double p_threshold, int cap,
                   float* match_p, int* match_idx, double* match_rect, float* match_box, int* count,
                   void* ws, size_t ws_bytes, void* stream) {
  RpnLayers L;
  to_layers(maps, H, W, &L);
  return rpn_scan(L, anchor_w, anchor_h, img_w, img_h, p_threshold, cap, match_p, match_idx, match_rect,
                  match_box, count, ws, ws_bytes, S(stream));
}
int frcnn_rpn_loss(const float* const* maps, float* const* deltas, const int* H, const int* W, const int* ex_idx,
                   const double* ex_anchor, const double* ex_roi, const int* ex_class, int npos, int nneg,
                   int bgclass, double* ex_loss, float* crtarget, float* cctarget, void* stream) {
  RpnLayers L;
  to_layers(maps, H, W, &L);
  return rpn_loss(L, deltas, ex_idx, ex_anchor, ex_roi, ex_class, npos, nneg, bgclass, ex_loss, crtarget,
                  cctarget, S(stream));
}

int frcnn_loss_accumulate(const double* ex_loss, int E, double* acc, void* stream) {
  return loss_accumulate(ex_loss, E, acc, S(stream));
}

int frcnn_linear_forward(const float* x, int R, int I, const float* weight, const float* bias, int O, float* y,
                         void* stream) {
  // Y[R][O] = X[R][I] * W[O][I]^T + b
  if (linear_x_eligible(1, R, I, O)) {   // split-bf16 operand form (gemmx.hip); the model runtime keeps the planes resident
    void* xp = nullptr;
    FR_HIP(hipMalloc(&xp, (size_t)3 * R * I * 2));
    float* am = nullptr;   // (two-plane fp16 form: records of x and of the weight matrix)
    int rc = x3_f16_records(&am);
    if (am && rc == FRCNN_OK) {
      rc = tensor_absmax(x, (long)R * I, am, S(stream));
      if (rc == FRCNN_OK) rc = tensor_absmax(weight, (long)O * I, am + AMAX_REC, S(stream));
    }
    if (rc == FRCNN_OK) rc = split_planes(x, R, I, xp, nullptr, S(stream), am);
    if (rc == FRCNN_OK) rc = linear_x_forward(xp, R, I, weight, bias, O, y, S(stream), 0, nullptr, am, am ? am + AMAX_REC : nullptr);
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(xp); (void)hipFree(am);
    return rc;
  }
  return gemm_f32(x, I, 1, weight, 1, I, y, O, R, O, I, OUT_STORE, bias, S(stream));
}
int frcnn_linear_backward(const float* x, const float* gy, int R, int I, const float* weight, int O, float* gx,
                          float* gweight, float* gbias, void* stream) {
  const bool xd = gx && linear_x_eligible(2, R, I, O), xw = gweight && linear_x_eligible(4, R, I, O);
  if (xd || xw) {
    const size_t Rp = (size_t)linear_x_rows_padded(R);
    void *gp = nullptr, *gpT = nullptr, *xpT = nullptr;
    int rc = FRCNN_OK;
    auto alloc = [&](void** p, size_t bytes) {   // (a failed allocation frees what the earlier ones got: no early return)
      if (rc != FRCNN_OK) return;
      hipError_t e = hipMalloc(p, bytes);
      if (e != hipSuccess) { set_error("frcnn_linear_backward: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); rc = FRCNN_ERR_HIP; }
    };
    if (xd) alloc(&gp, (size_t)3 * R * O * 2);
    if (xw) { alloc(&gpT, (size_t)3 * O * Rp * 2); alloc(&xpT, (size_t)3 * I * Rp * 2); }
    float* am = nullptr;   // (two-plane fp16 form of the input-gradient product)
    if (rc == FRCNN_OK && xd) rc = x3_f16_records(&am);
    if (rc == FRCNN_OK && am) {
      rc = tensor_absmax(gy, (long)R * O, am, S(stream));
      if (rc == FRCNN_OK) rc = tensor_absmax(weight, (long)O * I, am + AMAX_REC, S(stream));
      if (rc == FRCNN_OK) rc = split_planes(gy, R, O, gp, nullptr, S(stream), am);
      if (rc == FRCNN_OK && gpT) rc = split_planes(gy, R, O, nullptr, gpT, S(stream));
    } else if (rc == FRCNN_OK) rc = split_planes(gy, R, O, gp, gpT, S(stream));
    if (rc == FRCNN_OK && xw) rc = split_planes(x, R, I, nullptr, xpT, S(stream));
    if (rc == FRCNN_OK && xd) rc = linear_x_dgrad(gp, R, O, weight, I, gx, OUT_STORE, S(stream), 0, nullptr, am, am ? am + AMAX_REC : nullptr);
    if (rc == FRCNN_OK && xw) rc = linear_x_wgrad(gpT, xpT, R, O, I, gweight, S(stream));
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(gp); (void)hipFree(gpT); (void)hipFree(xpT); (void)hipFree(am);
    FR_TRY(rc);
    if (gx && !xd) FR_TRY(gemm_f32(gy, O, 1, weight, I, 1, gx, I, R, I, O, OUT_STORE, nullptr, S(stream)));
    if (gweight && !xw) FR_TRY(gemm_f32(gy, 1, O, x, I, 1, gweight, I, O, I, R, OUT_ADD, nullptr, S(stream)));
    if (gbias) FR_TRY(channel_sum_cols(gy, R, O, gbias, S(stream)));
    return FRCNN_OK;
  }
  if (gx) FR_TRY(gemm_f32(gy, O, 1, weight, I, 1, gx, I, R, I, O, OUT_STORE, nullptr, S(stream)));
  if (gweight) FR_TRY(gemm_f32(gy, 1, O, x, I, 1, gweight, I, O, I, R, OUT_ADD, nullptr, S(stream)));
  if (gbias) FR_TRY(channel_sum_cols(gy, R, O, gbias, S(stream)));
  return FRCNN_OK;
}

int frcnn_rmsprop(float* x, const float* g, float* m, long long n, float lr, float alpha, float eps,
                  void* stream) {
  return rmsprop_step(x, const_cast<float*>(g), m, n, lr, alpha, eps, 1.f, false, S(stream));
}

int frcnn_scale_rmsprop(float* x, float* g, float gscale, float* m, long long n, float lr, float alpha, float eps,
                        void* stream) {
  return rmsprop_step(x, g, m, n, lr, alpha, eps, gscale, true, S(stream));
}
int frcnn_scale_rmsprop_slice(float* x, float* g, float gscale, float* m, long long lo, long long hi, float lr, float alpha,
                              float eps, void* stream) {
  return rmsprop_slice(x, g, m, lo, hi, lr, alpha, eps, gscale, gscale != 1.0f, S(stream));
}
int frcnn_scale_rmsprop_dev(float* x, float* g, const double* gcount_dev, float* m, long long n, float lr, float alpha,
                            float eps, void* stream) {
  FR_CHECK(gcount_dev, "frcnn_scale_rmsprop_dev: NULL divisor");
  return rmsprop_step(x, g, m, n, lr, alpha, eps, 1.f, true, S(stream), gcount_dev);
}

int frcnn_cnet_losses(float* crout, const float* crtarget, const float* ccout, const float* cctarget, int R,
                      int npos, int ncls, float* crdelta, float* ccdelta, double* loss2, void* stream) {
  return cnet_losses(crout, crtarget, ccout, cctarget, R, npos, ncls, crdelta, ccdelta, loss2, S(stream));
}
int frcnn_cnet_decode(const float* cls_out, int R, int ncls, int* cls, float* conf, void* stream) {
  return cnet_decode(cls_out, R, ncls, cls, conf, S(stream));
}

// ---------------------------------------------------------------- image preparation (image.hip)
int frcnn_image_rgb2yuv(const float* rgb, float* yuv, int H, int W, void* stream) {
  FR_CHECK(rgb != yuv, "image_rgb2yuv: in-place conversion is not supported");
  return image_rgb2yuv(rgb, yuv, H, W, S(stream));
}
int frcnn_image_rgb2hsv(const float* rgb, float* hsv, int H, int W, void* stream) {
  FR_CHECK(rgb != hsv, "image_rgb2hsv: in-place conversion is not supported");
  return image_rgb2hsv(rgb, hsv, H, W, S(stream));
}
int frcnn_image_rgb2lab(const float* rgb, float* lab, int H, int W, void* stream) {
  FR_CHECK(rgb != lab, "image_rgb2lab: in-place conversion is not supported");
  return image_rgb2lab(rgb, lab, H, W, S(stream));
}
int frcnn_image_scale(const float* src, int C, int H, int W, float* dst, int dH, int dW, float* tmp, int rgb2yuv,
                      void* stream) {
  return image_scale(src, C, H, W, dst, dH, dW, tmp, rgb2yuv, S(stream));
}
int frcnn_image_scale_u8(const unsigned char* src_hwc, int H, int W, float* dst, int dH, int dW, float* tmp, int rgb2yuv,
                         void* stream) {
  return image_scale_u8(src_hwc, H, W, dst, dH, dW, tmp, rgb2yuv, S(stream));
}
int frcnn_image_crop_flip(const float* src, int C, int H, int W, int x0, int y0, int w, int h, int hflip, int vflip,
                          float* dst, void* stream) {
  return image_crop_flip(src, C, H, W, x0, y0, w, h, hflip, vflip, dst, S(stream));
}
size_t frcnn_image_normalize_workspace_bytes(int C) { return image_normalize_workspace_bytes(C); }
int frcnn_image_normalize(float* img, int C, int H, int W, int centering, int scaling, void* ws, size_t ws_bytes,
                          void* stream) {
  return image_normalize(img, C, H, W, centering, scaling, ws, ws_bytes, S(stream));
}
int frcnn_image_contrastive_norm(const float* in, int H, int W, const float* kernel_host, int K, float threshold,
                                 float* out, float* tmp, void* stream) {
  return image_contrastive_norm(in, H, W, kernel_host, K, threshold, out, tmp, S(stream));
}

}  // extern "C"
