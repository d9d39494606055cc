/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so: the MI355X (gfx950) Faster R-CNN detection/training
 * hot path of andreaskoepf/faster-rcnn.torch.
 *
 * The reference has no FFI of its own: its hot path reaches its arithmetic through the Torch7
 * Lua object protocol (nn.Module:forward/backward, nn.SpatialAdaptiveMaxPooling, criteria,
 * nms(), optim.rmsprop) implemented by the un-vendored cunn/cutorch packages.  This header is
 * the boundary a LuaJIT `ffi.cdef` -- or any other host -- binds in their
 * place.  Each entry point names the reference call site it replaces (file:line in the
 * reference tree).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *  - Every function returns FRCNN_OK (0) or an FRCNN_ERR_* code; frcnn_last_error() returns a
 *    thread-local message that a Lua wrapper turns into error().
 *  - Pointers are DEVICE pointers unless the parameter name ends in _host.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All work is
 *    asynchronous on that stream unless a function says it synchronises.
 *  - Tensors are fp32, CHW (or row-major R x D), contiguous, exactly like the reference's
 *    CudaTensors.  Index values exchanged through this ABI are 1-based like the Lua surface
 *    wherever the reference's own tables are 1-based (anchor indices, NMS ids, ROI windows,
 *    class ids); this is stated per function.
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK 0
#define FRCNN_ERR_ARG 1
#define FRCNN_ERR_HIP 2
#define FRCNN_ERR_STATE 3

/* ---- library / device --------------------------------------------------------------- */
int frcnn_version(void);
const char *frcnn_last_error(void);
int frcnn_device_count(int *n);
int frcnn_set_device(int device);            /* cutorch.setDevice, main.lua:52 (0-based here) */
int frcnn_device_name(char *buf_host, int len);
/* Library options.  "side_stream" (default 1; environment FRCNN_SIDE_STREAM): independent parts of a
 * pass (anchor nets, weight gradients) are issued on a library-owned second HIP stream and joined before the
 * entry point's results are used on the caller's stream.  0 = strictly serial on the caller's stream.
 * "deterministic" (default 0): the places where floating-point partial results meet in fp32 atomics (bias / slope
 * gradient partials of the activation backward passes, the scatter-adds of the ROI-pooling and sparse anchor-net backward
 * passes, the anchor deltas of two examples naming one anchor) switch to order-independent forms -- partials folded in
 * index order, gathers, 64-bit fixed-point accumulation -- so that two runs on the same inputs are bit-identical.
 * "split_bf16" (default 1; environment FRCNN_SPLIT_BF16): the 3x3 convolutions whose shapes fit (forward and input gradient:
 * input channels a multiple of 16, filters a multiple of 64; weight gradient: both multiples of 64) run in the split-operand
 * form: fp32 tensors in and out, fp32 accumulation, every fp32 product formed from six exact bf16 x bf16 partial products of
 * three-way split operands on the bf16 matrix cores (24 significand bits per operand; against an fp64 reference the error
 * equals that of the fp32 matrix-core kernels, tests/test_gpu_convx.py).  0 = fp32 matrix-core kernels only.  Applies to the
 * operator-level entry points at once and to a model from its next (re)shaping on.
 * "gemm_x_roles" (default -1; environment FRCNN_GEMM_X): which products of a large nn.Linear (the cnet's Linear(13824, 1024))
 * take the split-bf16 operand form of csrc/gemmx.hip -- bit 1 forward, 2 input gradient, 4 weight gradient; -1 = the
 * measured rule (the input gradient always; forward and weight gradient from 192 rows on, where they beat the fp32 kernels),
 * 0 = fp32 matrix-core kernels for all three.
 * "x3_f16" (default 1; environment FRCNN_X3_F16): the operand form of the split convolutions (with "split_bf16" on).  1 = two
 * fp16 planes per operand and THREE exact partial products per fp32 product: each tensor is scaled by a power of two taken from
 * its largest magnitude (recorded by the launch that wrote it), h = f16(x 2^e), l = f16(x 2^e - h), 22 significand bits per
 * operand; measured against fp64 the error is that of the fp32 matrix-core kernels (tools/x3_f16_check.py,
 * tests/test_gpu_convx.py run both forms).  0 = three bf16 planes and six partial products: the split is exact (24 bits, no
 * dependence on the tensor's range) at twice the matrix-pipe time.  Takes effect with the next forward pass.
 * "static_weights" (default 0): the host promises that it does not write the weight vector between evaluate-mode forward passes
 * (Detector:detect on a trained model); the library then re-uses the packed / split copies of the convolution weights instead of
 * remaking them per pass.  A training-mode pass, a change of the frame size and every frcnn_set_option("static_weights", v) call
 * drop the copies -- a host that has written the weights itself says so by setting the option again.
 * "drop_compact" (default 1; environment FRCNN_DROP_COMPACT): a training pass leaves out the channels an nn.SpatialDropout
 * (models/model_utilities.lua:10-12) drops.  In a block of two or more convolutions the dropout follows the first one, so for the
 * length of a step the dropped filters of the first convolution and the dropped input channels of the second multiply zeros --
 * forward (objective.lua:71) and backward (:189).  With the keep vector known on the host (drawn with the hash the device kernel
 * uses; an explicit mask is read back -- a synchronous copy, parity runs) the step computes the first convolution for the kept
 * filters only, stores its output and that output's gradient compact, and runs the second convolution, both input gradients and
 * both weight gradients on the kept channels (padded to the kernels' 16 / 64-channel granularity); packs gather, the weight-
 * gradient folds scatter.  Results equal the dense pass's to rounding (sums of exact zeros are left out); the dropped filters'
 * and channels' gradients stay the zeros objective.lua:49 put there.  0 = multiply by the zeros like the reference does. */
int frcnn_set_option(const char *name, int value);
int frcnn_get_option(const char *name, int *value_host);

/* ---- device buffers (torch.CudaTensor storage; main.lua:86-89, objective.lua:66,147-149) */
int frcnn_malloc(void **ptr_out_host, size_t bytes);
int frcnn_free(void *ptr);
/* Page-locked host memory: a frcnn_memcpy_h2d / _d2h whose host side lives in such a buffer is truly asynchronous (the call
 * returns at once and the copy runs in stream order).  From ordinary (pageable) host memory the same call only returns when
 * the stream has reached the copy -- the host loses whatever lead it had over the device (in the training step: the example
 * tables of objective.lua:91-140 are uploaded mid-step, and a blocking upload made the device wait ~0.45 ms per step for
 * the launches that follow it).  The caller keeps the buffer untouched until the copy has run (an event or a later sync). */
int frcnn_host_alloc(void **ptr_out_host, size_t bytes);
int frcnn_host_free(void *ptr_host);
int frcnn_memcpy_h2d(void *dst, const void *src_host, size_t bytes, void *stream);
int frcnn_memcpy_d2h(void *dst_host, const void *src, size_t bytes, void *stream);
int frcnn_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int frcnn_stream_sync(void *stream);
int frcnn_zero(void *ptr, size_t bytes, void *stream);            /* gradient:zero(), objective.lua:49 */
int frcnn_scale(float *x, long long n, float s, void *stream);    /* gradient:div(n), objective.lua:200 */
int frcnn_add(float *y, const float *x, long long n, void *stream);   /* y:add(x), objective.lua:106,114,184 */

/* ---- per-kernel-class HIP-event profile (bench.py roofline leg) ---------------------- */
#define FRCNN_KC_CONV_IGEMM_K3 0
#define FRCNN_KC_CONV_IGEMM_OTHER 1
#define FRCNN_KC_CONV_WGRAD_K3 2
#define FRCNN_KC_CONV_WGRAD_OTHER 3
#define FRCNN_KC_GEMM 4
#define FRCNN_KC_ELEMWISE 5
#define FRCNN_KC_ROI 6
#define FRCNN_KC_RPN 7
#define FRCNN_KC_NMS 8
#define FRCNN_KC_OPTIM 9
#define FRCNN_KC_IMAGE 10
#define FRCNN_KC_CONV_X3 11      /* 3x3 forward + input gradient, split-bf16 operand form */
#define FRCNN_KC_CONV_WGRADX 12  /* 3x3 weight gradient, split-bf16 operand form */
#define FRCNN_KC_COUNT 13
/* class_mask: bit k set -> every launch of kernel class k is bracketed by two hipEvents on its
 * launch stream (0 = profiling off). */
int frcnn_prof_enable(int class_mask);
/* Synchronises the device; per class: launches, total ms, algorithmic flops, algorithmic bytes.
 * Arrays of FRCNN_KC_COUNT entries (host).  Resets the profile. */
int frcnn_prof_collect(long long *launches_host, double *ms_host, double *flops_host,
                       double *bytes_host);

/* ---- nms(boxes, overlap, scores): nms.lua:23-102 ------------------------------------- */
/* key_mode: 0 = y2 (what every reference call site gets, Detector.lua:82,133: a tensor/nil
 * `scores` falls through to nms.lua:42), 1 = 'area' (nms.lua:39-40), 2 = column key_col
 * (1-based; nms.lua:37-38).  pick: int64[n] of 1-based row ids in pick order; *count = number
 * picked.  Bit-exact with the reference's fp32 CPU arithmetic. */
size_t frcnn_nms_workspace_bytes(int n);
int frcnn_nms_device(const float *boxes, int n, int ncols, float overlap, int key_mode,
                     int key_col, long long *pick, int *count, void *workspace,
                     size_t workspace_bytes, void *stream);
/* The per-class loop of Detector.lua:125-136 (`for i,c in pairs(yclass) do ... nms(bb, 0.1, ...)`) in ONE call: row i only
 * suppresses rows of the same class cls[i] (device int[n]; any integers).  Picks come in global pick order (descending key),
 * i.e. the per-class pick lists interleaved: a stable partition of the picks by class gives, per class, exactly the result
 * of nms() on that class's rows (same relative order, same fp32 arithmetic).  cls == NULL: frcnn_nms_device. */
int frcnn_nms_device_classes(const float *boxes, int n, int ncols, float overlap, int key_mode, int key_col,
                             const int *cls, long long *pick, int *count, void *workspace,
                             size_t workspace_bytes, void *stream);
/* The same with the row count in DEVICE memory: n = min(*n_dev, n_cap); launches and workspace are sized for n_cap
 * (frcnn_nms_workspace_bytes(n_cap)).  Lets Detector.lua:39-85 run scan -> NMS, and :115-136 class test -> per-class NMS,
 * without a host round trip for the count in between.  cls may be NULL. */
int frcnn_nms_device_n(const float *boxes, int n_cap, const int *n_dev, int ncols, float overlap, int key_mode, int key_col,
                       const int *cls, long long *pick, int *count, void *workspace, size_t workspace_bytes, void *stream);
/* ---- Detector:detect glue kept on the device (Detector.lua:88-136; csrc/detect.hip) ------------------------------
 * frcnn_roi_windows: extract_roi_pooling_input (objective.lua:5-13) for k ROIs at once -- Localizer:inputToFeatureRect
 * (Localizer.lua:41-67, the reference's double arithmetic incl. its dH/dW mix-ups) over layers_host[nlayers][6] =
 * {kW,kH,dW,dH,padW,padH} (frcnn_model_localizer_layers), clip to the fmH x fmW map, 1-based inclusive window
 * wins[k][4] = {row_lo,row_hi,col_lo,col_hi}.  rect: device double[n][4]; pick: optional device int64[k] of 1-based rows of
 * rect (the NMS candidates), NULL = rows 0..k-1. */
int frcnn_roi_windows(const double *rect, const long long *pick, int k, const int *layers_host, int nlayers, int fmH, int fmW,
                      int *wins, void *stream);
/* frcnn_detect_post: Detector.lua:106-122 for the R candidates (cls / conf from frcnn_cnet_decode, bbox = cnet's R x 4
 * output, rect / pick as above): candidate r survives iff cls[r] != bgclass and exp(conf[r]) > min_conf; survivors, in
 * candidate order, get r2 = Anchors.anchorToInput(anchor rect, bbox row) in double (r2[K][4]), bb[K][5] = {float(r2), conf},
 * kc[K] = class, keep_row[K] = r (0-based); *K_dev = their number (device).  All outputs need room for R rows. */
int frcnn_detect_post(const int *cls, const float *conf, const float *bbox, const double *rect, const long long *pick, int R,
                      int bgclass, double min_conf, float *bb, int *kc, int *keep_row, double *r2, int *K_dev, void *stream);
/* frcnn_detect_gather: one record of 16 doubles per winner q < min(*nwin_dev, cap), in the pick order wpick of the
 * per-class NMS over bb/kc: {class, candidate row (1-based), confidence (log-prob), p (anchor log-prob), anchor rect x4,
 * r2 x4, anchor index {layer,aspect,y,x}} -- everything Detector.lua:116-122 puts into a detection's table. */
int frcnn_detect_gather(const long long *wpick, const int *nwin_dev, int cap, const int *keep_row, const int *kc,
                        const float *bb, const double *r2, const long long *pick, const float *match_p,
                        const double *match_rect, const int *match_idx, double *rec, void *stream);
/* Host-pointer variant (the reference's nms runs on CPU FloatTensors): uploads, runs the same
 * kernels, downloads, synchronises. */
int frcnn_nms_host(const float *boxes_host, int n, int ncols, float overlap, int key_mode,
                   int key_col, long long *pick_host, int *count_host);

/* ---- nn.SpatialConvolution (models/model_utilities.lua:8,31,33) ---------------------- */
/* out[O][Ho][Wo] = bias + W (*) act(in), Ho = H + 2*pad - k + 1.  act(x) = in_scale[c] *
 * prelu(x, *in_slope) is the producing layer's nn.PReLU + nn.SpatialDropout fused into the
 * load (pass NULL for identity).  weight is the canonical [O][C][k][k] tensor.  k in {1,3,5,7},
 * stride 1.  :updateOutput, reached from objective.lua:71 / Detector.lua:33. */
int frcnn_conv2d_forward(const float *in, int C, int H, int W, const float *in_slope,
                         const float *in_scale, const float *weight, const float *bias, int O,
                         int k, int pad, float *out, void *stream);
/* :updateGradInput (objective.lua:189): gin[C][H][W] = W^T (*) gout. accumulate!=0 -> += */
int frcnn_conv2d_backward_input(const float *gout, int O, int Ho, int Wo, const float *weight,
                                int C, int k, int pad, float *gin, int accumulate, void *stream);
/* :accGradParameters (objective.lua:189): gweight += gout (x) act(in); gbias += sum gout. */
int frcnn_conv2d_backward_weight(const float *in, int C, int H, int W, const float *in_slope,
                                 const float *in_scale, const float *gout, int O, int k, int pad,
                                 float *gweight, float *gbias, void *stream);

/* ---- nn.PReLU + nn.SpatialDropout + nn.SpatialMaxPooling(2,2,2,2):ceil()
 *      (models/model_utilities.lua:9-12,23) -------------------------------------------- */
/* out = maxpool2x2_ceil(in_scale[c]*prelu(x)); idx[c][oy][ox] = dy*2+dx of the (first) max. */
int frcnn_maxpool_act_forward(const float *x, int C, int H, int W, const float *slope,
                              const float *scale, float *out, unsigned char *idx, void *stream);
/* gx = unpool(gpool) * scale[c] * prelu'(x); gbias[c] += sum gx; *gslope += sum_{x<=0} x*g. */
int frcnn_maxpool_act_backward(const float *gpool, const unsigned char *idx, const float *x, int C,
                               int H, int W, const float *slope, const float *scale, float *gx,
                               float *gbias, float *gslope, void *stream);
int frcnn_act_forward(const float *x, int C, long long hw, const float *slope, const float *scale,
                      float *y, void *stream);
int frcnn_act_backward(const float *gy, const float *x, int C, long long hw, const float *slope,
                       const float *scale, float *gx, float *gbias, float *gslope, void *stream);

/* ---- extract_roi_pooling_input + nn.SpatialAdaptiveMaxPooling, batched over ROIs
 *      (objective.lua:5-13,117-118,137-138,182-185; Detector.lua:96-97) ----------------- */
/* wins: int[R][4] = {row_lo,row_hi,col_lo,col_hi}, 1-based inclusive (objective.lua:11).
 * out: [R][C*kh*kw] (row r = amp:forward(view):view(kh*kw*C)); idx: int[R][C*kh*kw], flat
 * 0-based y*W+x position of each max inside the full map (amp.indices equivalent). */
int frcnn_roi_pool_forward(const float *fmap, int C, int H, int W, const int *wins, int R, int kh,
                           int kw, float *out, int *idx, void *stream);
/* gmap[C][H][W] += scatter(gout) -- delta_outputs[5][idx]:add(amp:backward(...)) */
int frcnn_roi_pool_backward(float *gmap, int C, int H, int W, const float *gout, const int *idx,
                            int R, int kh, int kw, void *stream);

/* ---- RPN anchor scan: Detector.lua:39-66 --------------------------------------------- */
/* maps_host: 4 device pointers to the [18][H_l][W_l] head outputs (pnet outputs 1..4).
 * anchor_w / anchor_h: the fp32 tables of Anchors.lua:18-19, [4][3][200][2].
 * Outputs in scan order (layer, y, x, aspect), at most cap entries:
 *   match_p[i]    = c[1] (log-prob of foreground)
 *   match_idx[i]  = {layer, aspect, y, x} 1-based (Anchors:get arguments)
 *   match_rect[i] = decoded rect as doubles (Lua numbers), {minX,minY,maxX,maxY}
 *   match_box[i]  = the same as fp32 (the FloatTensor row of Detector.lua:74-79, NMS input)
 *   *count        = number of matches found (may exceed cap; only cap are written) */
size_t frcnn_rpn_scan_workspace_bytes(const int *H_host, const int *W_host);
int frcnn_rpn_scan(const float *const *maps_host, const int *H_host, const int *W_host,
                   const float *anchor_w, const float *anchor_h, double img_w, double img_h,
                   double p_threshold, int cap, float *match_p, int *match_idx,
                   double *match_rect, float *match_box, int *count, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ---- sparse RPN loss: objective.lua:91-140 (+ cnet targets, objective.lua:149-159) ---- */
/* Examples: positives first (npos) then negatives (nneg).  ex_idx int[E][4] {layer,aspect,y,x}
 * 1-based; ex_anchor double[E][4]; ex_roi double[npos][4] (roi.rect of each positive);
 * ex_class int[npos] (roi.class_index).  deltas_host: 4 device pointers to delta_outputs[1..4]
 * (gradients are ADDED).  ex_loss double[E][2] = {cls, reg*10}; crtarget float[E][4];
 * cctarget float[E] (class index, bgclass for negatives). */
int frcnn_rpn_loss(const float *const *maps_host, float *const *deltas_host, const int *H_host,
                   const int *W_host, const int *ex_idx, const double *ex_anchor,
                   const double *ex_roi, const int *ex_class, int npos, int nneg, int bgclass,
                   double *ex_loss, float *crtarget, float *cctarget, void *stream);

/* acc[0] += sum_e ex_loss[e][0]; acc[1] += sum_e ex_loss[e][1] (device fp64; the Lua accumulators
 * cls_loss / reg_loss of objective.lua:52, summed in example order). */
int frcnn_loss_accumulate(const double *ex_loss, int E, double *acc, void *stream);

/* ---- nn.Linear (models/model_utilities.lua:82,99,103) -------------------------------- */
int frcnn_linear_forward(const float *x, int R, int I, const float *weight, const float *bias,
                         int O, float *y, void *stream);
/* gx (may be NULL) = gy W ; gweight += gy^T x ; gbias += colsum(gy) (either may be NULL) */
int frcnn_linear_backward(const float *x, const float *gy, int R, int I, const float *weight,
                          int O, float *gx, float *gweight, float *gbias, void *stream);

/* ---- optim.rmsprop (main.lua:122,133): m = a*m + (1-a)*g^2 ; x -= lr*g/(sqrt(m)+eps) -- */
int frcnn_rmsprop(float *x, const float *g, float *m, long long n, float lr, float alpha,
                  float eps, void *stream);
/* frcnn_scale(g, n, gscale) followed by frcnn_rmsprop in ONE pass over the flat vectors: gradient:div(n)
 * (objective.lua:200) folded into the optimiser step that follows it in main.lua:133.  g holds the scaled
 * gradient afterwards, exactly as after the two separate calls. */
int frcnn_scale_rmsprop(float *x, float *g, float gscale, float *m, long long n, float lr, float alpha,
                        float eps, void *stream);
/* The same with the divisor read from the device: gscale = 1 / *gcount_dev (a count of 0 leaves g unscaled).  For the
 * data-parallel step, where objective.lua:200's cls_count is the all-reduced sum that frcnn_allreduce_f64 left in
 * the accumulator vector: the update is queued behind the exchange without any host read-back. */
int frcnn_scale_rmsprop_dev(float *x, float *g, const double *gcount_dev, float *m, long long n, float lr,
                            float alpha, float eps, void *stream);
/* frcnn_scale_rmsprop on elements [lo, hi) of the vectors only (x, g, m = the 16-byte aligned STARTS of the flat vectors; any
 * bounds; gscale = 1: the gradient is left unscaled): what main.lua:133 does to that slice, bit for bit.  See "the update beside
 * the backward pass" below. */
int frcnn_scale_rmsprop_slice(float *x, float *g, float gscale, float *m, long long lo, long long hi, float lr,
                              float alpha, float eps, void *stream);

/* ---- model runtime: models/model_utilities.lua:3-136 --------------------------------- */
typedef struct {
  int nblocks;                      /* vgg_small.lua:5-10 `layers` */
  int filters[8], ksize[8], pad[8], conv_steps[8];
  float dropout[8];
  int nheads;                       /* vgg_small.lua:12-17 `anchor_nets` */
  int head_k[8], head_n[8], head_input[8]; /* head_input 1-based */
  int ncls;                         /* vgg_small.lua:19-22 `class_layers` */
  int cls_n[8], cls_bn[8];
  float cls_dropout[8];
  int class_count;                  /* cfg.class_count (excluding background) */
  int kh, kw;                       /* cfg.roi_pooling */
} frcnn_model_desc;

typedef struct frcnn_model frcnn_model;

int frcnn_model_create(const frcnn_model_desc *desc_host, frcnn_model **out_host);
int frcnn_model_destroy(frcnn_model *);
/* Flat parameter vector (utilities.lua:136-147): total and pnet-only element counts. */
int frcnn_model_param_count(const frcnn_model *, long long *total_host, long long *pnet_host);
/* Parameter table, one row per tensor in flat order: {offset, count, kind, aux}.
 * kind: 0 conv weight (aux = kW*kH*nOutputPlane, model_utilities.lua:63-64), 1 conv bias,
 * 2 PReLU slope, 3 Linear weight (aux = fan_in), 4 Linear bias (aux = fan_in),
 * 5 BatchNorm weight, 6 BatchNorm bias. */
int frcnn_model_param_table(const frcnn_model *, long long *table_host, int cap, int *n_host);
/* Localizer.lua:6-39 for output node i (1..nheads = anchor heads, nheads+1 = last feature map):
 * int[n][6] = {kW,kH,dW,dH,padW,padH}, input first. */
int frcnn_model_localizer_layers(const frcnn_model *, int output_index, int *layers_host, int cap,
                                 int *n_host);

/* pnet:forward(img) (objective.lua:71, Detector.lua:33).  training!=0: SpatialDropout masks are
 * drop_masks_host[b] (device float[filters_b] of 0/1, for parity runs) or, when NULL, drawn on
 * device from `seed`.  training==0: x(1-p) (pnet:evaluate(), Detector.lua:31). */
int frcnn_pnet_forward(frcnn_model *, const float *weights, const float *img, int H, int W,
                       int training, const float *const *drop_masks_host, unsigned long long seed,
                       void *stream);
/* Training-mode pnet:forward (objective.lua:71) that leaves the anchor nets in flight on the library's side
 * stream: on return only outputs[nheads+1] (the last pooled map) is final in `stream` order, so the
 * fine-tuning stage (objective.lua:146-186) can start beside them.  outputs[1..nheads] are consumed by
 * frcnn_pnet_anchor_loss_begin; frcnn_pnet_backward (or the next forward) joins whatever is still running.
 * With the side stream off this is frcnn_pnet_forward(training=1). */
int frcnn_pnet_forward_async_heads(frcnn_model *, const float *weights, const float *img, int H, int W,
                                   const float *const *drop_masks_host, unsigned long long seed,
                                   void *stream);
/* outputs[i], i = 1..nheads+1; the buffers are owned by the model and reused by the next
 * forward (callers that keep results must copy: objective.lua:119). */
int frcnn_pnet_output(frcnn_model *, int i, float **ptr_host, int *C_host, int *H_host,
                      int *W_host);
/* Inspection of what the last forward pass left in HBM -- for parity tests, which hand the path's DISCRETE decisions
 * (max-pool window winners, PReLU branches) to the CPU restatement so that gradients can be compared at the strict
 * tolerance (oracle/frcnn_oracle.h orc_set_decisions).  The pointer stays owned by the model and valid until the next
 * forward pass of another image size.
 *   kind 0: pre-activation output of backbone convolution `index` (float [O][Ho][Wo]; model_utilities.lua:8).  A training pass
 *           with option "drop_compact" does not compute the output channels the block's SpatialDropout (:10-12) drops: they read 0
 *   kind 1: arg-max of block `index`'s 2x2 ceil-mode max pool (unsigned char [C][Hp][Wp], value dy*2+dx; :23)
 *   kind 2: pre-activation output of anchor net `index`'s k x k convolution (float [n][Ho][Wo]; :31)
 *   kind 3: input of classification layer `index`'s PReLU (float [R][n]; :85-86), after frcnn_cnet_forward
 *   kind 4: the nn.SpatialDropout scale vector of block `index` in the last pass (float [C]; a 0/1 keep vector while training) */
int frcnn_model_debug_buffer(frcnn_model *, int kind, int index, void **ptr_host, long long *bytes_host);
/* delta_outputs[i] (objective.lua:78-84): gradient buffers with the shapes of the outputs. */
int frcnn_pnet_delta(frcnn_model *, int i, float **ptr_host);
int frcnn_pnet_zero_deltas(frcnn_model *, void *stream);
/* Optional one-shot hint for the next frcnn_pnet_backward: delta_outputs[head] (head = 1..nheads) is
 * zero outside `count` positions (device int array of unique flat indices y*W_head + x, 0-based), which is
 * how objective.lua:91-134 fills it (only the sampled anchors).  The head's backward then runs on those
 * positions only; the result is identical.  count < 0 (default) or > 512: dense backward. */
int frcnn_pnet_set_sparse_deltas(frcnn_model *, int head, const int *positions, int count);
/* Optional early start of pnet:backward (objective.lua:189): once delta_outputs[1..nheads] are final (after
 * the anchor loop, objective.lua:91-140 -- the fine-tuning stage only adds into delta_outputs[nheads+1]),
 * the anchor-net part of the backward pass may begin on the library's side stream, beside the
 * classification network's forward/backward on `stream`.  frcnn_pnet_backward then joins it.  Without this
 * call frcnn_pnet_backward does everything itself; the result is the same. */
int frcnn_pnet_backward_heads_begin(frcnn_model *, const float *weights, float *grad, void *stream);
/* The anchor loop of objective.lua:91-140 on the model's own outputs[1..nheads] / delta_outputs[1..nheads]
 * (arguments as frcnn_rpn_loss + frcnn_loss_accumulate), followed by the anchor-net part of pnet:backward
 * (as frcnn_pnet_backward_heads_begin) -- all on the library's side stream, ordered after everything queued
 * on `stream` so far (example tables, zeroed delta buffers).  frcnn_pnet_anchor_loss_wait makes `stream` wait
 * until ex_loss / crtarget / cctarget / acc are final (needed before frcnn_cnet_losses, objective.lua:156);
 * frcnn_pnet_backward joins the rest.  With the side stream off the losses run on `stream` itself. */
int frcnn_pnet_anchor_loss_begin(frcnn_model *, const float *weights, float *grad, const int *ex_idx,
                                 const double *ex_anchor, const double *ex_roi, const int *ex_class,
                                 int npos, int nneg, int bgclass, double *ex_loss, float *crtarget,
                                 float *cctarget, double *acc, void *stream);
int frcnn_pnet_anchor_loss_wait(frcnn_model *, void *stream);
/* Makes `stream` wait for the anchor-net part started by frcnn_pnet_backward_heads_begin.  *joined_host = 1:
 * the anchor nets' slice of `grad` is final in stream order (e.g. for an early all-reduce of that slice beside
 * the remaining backward pass); 0: nothing had been started, frcnn_pnet_backward will compute that part. */
int frcnn_pnet_backward_heads_join(frcnn_model *, void *stream, int *joined_host);
/* pnet:backward(img, delta_outputs) (objective.lua:189): accumulates into the flat gradient.
 * The (unused) input gradient of the first convolution is not computed. */
int frcnn_pnet_backward(frcnn_model *, const float *weights, float *grad, void *stream);

/* After frcnn_pnet_backward has been queued: makes `stream` (any stream) wait until every gradient of backbone
 * block `block` (1-based; its convolutions' weights, biases and PReLU slopes -- a contiguous slice of the flat
 * vector) is final.  The deepest block finishes first, long before the call's own stream reaches the end of the
 * pass: a data-parallel caller starts that slice's all-reduce behind this wait, beside the remaining backward pass. */
int frcnn_pnet_wait_block_gradients(frcnn_model *, int block, void *stream);

/* ---- the update beside the backward pass (round 6): optim.rmsprop's step (main.lua:133) applied slice by slice ----------
 * main.lua:133 updates the whole flat vector after lossAndGradient (objective.lua:45-218) has returned, and the next
 * pnet:forward (objective.lua:71) re-packs every weight tensor before its first convolution: two serial stretches of a step
 * that is otherwise bound by the matrix cores.  Slices of the gradient are final long before the pass ends -- the classification
 * net's (55 % of the vector) after cnet:backward (objective.lua:179), the anchor nets' once their part of pnet:backward has
 * run, a backbone block's when the pass has left the block -- and the divisor of gradient:div (objective.lua:200) is a host
 * number known before the pass.  A host that owns the optimiser may therefore queue
 *     frcnn_scale_rmsprop_slice(x, g, 1/cls_count, m, lo, hi, ...)  [+ frcnn_pnet_refresh_packs(model, x, group, ...)]
 * on the UPDATE STREAM as each slice becomes final, and frcnn_model_update_join before anything reads the weights again.  The
 * result is bit-identical to frcnn_scale_rmsprop on the whole vector (same arithmetic per element, tests/test_gpu_eager.py).
 *   frcnn_model_update_stream : the library-owned stream for this work (the one the classification net's weight gradients
 *                               run on: a slice update queued there is ordered behind them by itself).  Call it BEFORE the pass
 *                               whose slices are to be updated: from then on the passes record the events the waits below
 *                               need (each a marker on the caller's stream, which a host that never asks does not pay for)
 *   frcnn_model_update_fork   : the update stream waits for everything queued on `stream` so far (the last readers of the
 *                               weights about to change, e.g. cnet:backward's input-gradient chain)
 *   frcnn_model_update_join   : `stream` waits for everything queued on the update stream so far
 *   frcnn_pnet_wait_backward_begun : after frcnn_pnet_backward has been queued: `stream` waits until the caller's stream has
 *                               joined the anchor nets and begun the backbone's backward pass (objective.lua:189) -- the
 *                               matrix-core-bound stretch of the step, beside which bandwidth-bound update work costs least
 *   frcnn_pnet_wait_heads_done: after frcnn_pnet_backward has been queued: `stream` waits until the anchor nets' backward pass is
 *                               over, parameter gradients included (they may run on beside the backbone's pass: the caller's
 *                               stream only waits for their contribution to the pooled maps' gradients before it starts)
 *   frcnn_pnet_wait_block_done: after frcnn_pnet_backward has been queued: `stream` waits until block `block` (1-based) may be
 *                               updated -- its gradients are final AND the last launch reading its weights, packs or weight
 *                               magnitudes has run (frcnn_pnet_wait_block_gradients only promises the first)
 *   frcnn_pnet_refresh_packs  : renews on `stream` the packed weight images / weight magnitudes of one owner (group = 0-based
 *                               backbone block, or nblocks for the anchor nets) from `weights`; when EVERY group has been
 *                               renewed from the vector the next training-mode frcnn_pnet_forward is given, that forward skips
 *                               its own re-pack.  The caller promises that between this call and that forward nothing writes
 *                               the weights except updates followed by their refresh; frcnn_pnet_invalidate_packs withdraws
 *                               the promise (a host that wrote the weights some other way calls it). */
int frcnn_model_update_stream(frcnn_model *, void **stream_host);
int frcnn_model_update_fork(frcnn_model *, void *stream);
int frcnn_model_update_join(frcnn_model *, void *stream);
int frcnn_pnet_wait_backward_begun(frcnn_model *, void *stream);
int frcnn_pnet_wait_heads_done(frcnn_model *, void *stream);
int frcnn_pnet_wait_block_done(frcnn_model *, int block, void *stream);
int frcnn_pnet_refresh_packs(frcnn_model *, const float *weights, int group, void *stream);
int frcnn_pnet_invalidate_packs(frcnn_model *);

/* cnet:forward(cinput) (objective.lua:164, Detector.lua:101).  weights/grad point at the START
 * of the flat vectors.  bn_running: float[2*n] {mean,var} per BatchNorm layer (updated when
 * training).  drop_masks_host[l]: device float[R][n_l] keep masks or NULL (seeded RNG). */
int frcnn_cnet_forward(frcnn_model *, const float *weights, const float *x, int R, int training,
                       const float *const *drop_masks_host, unsigned long long seed,
                       float *bn_running, float *bbox_out, float *cls_out, void *stream);
/* cnet:backward(cinput, {crdelta, ccdelta}) (objective.lua:179) -> gx [R][D] */
/* cnet:backward.  The input gradient gx is final on `stream` in stream order; the weight gradients and bias sums it adds to
 * `grad` are queued on a library-owned stream beside that chain (option "cnet_wgrad_async", default 1) and are final on
 * `stream` after frcnn_pnet_backward, the next frcnn_cnet_forward, or frcnn_cnet_backward_join -- or after a device-wide
 * synchronisation.
 * LIFETIME: until one of those joins has been queued, the library-owned stream still READS caller-owned memory -- g_bbox,
 * g_cls, the `x` that was passed to frcnn_cnet_forward (first layer's weight gradient) -- and WRITES `grad`.  The caller
 * must neither modify nor free those buffers, nor let a stream-ordered allocator hand them out again, before the join
 * is queued on the stream that does so (a caching allocator: record the join's stream on the tensors, or hold references
 * until then).  frcnn_set_option("cnet_wgrad_async", 0) keeps everything on `stream` for a caller that cannot promise it. */
int frcnn_cnet_backward_join(frcnn_model *, void *stream);
int frcnn_cnet_backward(frcnn_model *, const float *weights, const float *g_bbox,
                        const float *g_cls, float *gx, float *grad, void *stream);
/* objective.lua:170-177: zero negative rows of crout, SmoothL1*10 and ClassNLL with gradients.
 * loss2 (device double[2]) += {creg_loss, ccls_loss}. */
int frcnn_cnet_losses(float *crout, const float *crtarget, const float *ccout,
                      const float *cctarget, int R, int npos, int ncls, float *crdelta,
                      float *ccdelta, double *loss2, void *stream);
/* Detector.lua:110-113: class (1-based argmax) and confidence (max log-prob) per row */
int frcnn_cnet_decode(const float *cls_out, int R, int ncls, int *cls, float *conf, void *stream);

/* ---- image preparation: BatchIterator:processImage (BatchIterator.lua:101-164) ----
 * The arithmetic the reference delegates to the torch `image` / `nn` packages, on a decoded float frame
 * [C][H][W] resident in device memory.  All calls are asynchronous on `stream`. */
/* image.rgb2yuv (utilities.lua load_image, color_space 'yuv'): rgb, yuv float[3][H][W] (may not alias). */
int frcnn_image_rgb2yuv(const float *rgb, float *yuv, int H, int W, void *stream);
/* image.rgb2hsv / image.rgb2lab (utilities.lua:212-215 load_image, color_space 'hsv' / 'lab'; main.lua:174-177): rgb in [0,1],
 * out float[3][H][W] (may not alias).  hsv: h in [0,1), s, v as image/generic/image.c computes them (h = s = 0 for a grey
 * pixel).  lab: sRGB gamma expansion, XYZ with the D65 white point, CIE L*a*b* (epsilon 216/24389, kappa 24389/27). */
int frcnn_image_rgb2hsv(const float *rgb, float *hsv, int H, int W, void *stream);
int frcnn_image_rgb2lab(const float *rgb, float *lab, int H, int W, void *stream);
/* image.scale(img, dW, dH), 'bilinear' (BatchIterator.lua:51): rows then columns; up-scaling interpolates
 * linearly, down-scaling averages the covered source interval.  tmp: device float[C*H*dW].
 * rgb2yuv != 0 (C == 3): src is the RGB frame and every source sample goes through image.rgb2yuv on the fly --
 * the result of frcnn_image_rgb2yuv followed by the scaling, without the full-resolution round trip. */
int frcnn_image_scale(const float *src, int C, int H, int W, float *dst, int dH, int dW, float *tmp,
                      int rgb2yuv, void *stream);
/* image.load(fn, 3, 'float') + (optionally) image.rgb2yuv + image.scale for a frame that is still the decoder's
 * 8-bit interleaved RGB [H][W][3] in device memory: samples are converted with v * (1/255) on the fly in the row
 * pass (6 MB instead of 25 MB per 1080p frame cross PCIe).  dst float[3][dH][dW]; tmp: device float[3*H*dW]. */
int frcnn_image_scale_u8(const unsigned char *src_hwc, int H, int W, float *dst, int dH, int dW,
                         float *tmp, int rgb2yuv, void *stream);
/* image.crop(img, x0, y0, x0+w, y0+h) followed by image.hflip / image.vflip when the flags are set
 * (BatchIterator.lua:57-80), one pass: dst float[C][h][w]. */
int frcnn_image_crop_flip(const float *src, int C, int H, int W, int x0, int y0, int w, int h, int hflip,
                          int vflip, float *dst, void *stream);
/* img[i]:add(-img[i]:mean()) for every channel (centering != 0), then img[i]:div(img[i]:std()) where
 * std > 1e-8 (scaling != 0), in place (BatchIterator.lua:146-160); mean and the unbiased standard deviation
 * accumulate in fp64 in a fixed order.  workspace: frcnn_image_normalize_workspace_bytes(C) device bytes. */
size_t frcnn_image_normalize_workspace_bytes(int C);
int frcnn_image_normalize(float *img, int C, int H, int W, int centering, int scaling, void *workspace,
                          size_t workspace_bytes, void *stream);
/* nn.SpatialContrastiveNormalization(1, kernel) on one plane (BatchIterator.lua:162, kernel =
 * image.gaussian1D(cfg.normalization.width)): subtractive then divisive normalisation with the 1-D kernel
 * (K odd, <= 15, host pointer) applied along x then y over a zero-padded plane, border-corrected by the
 * response to a plane of ones; threshold = thresval (1e-4 in the reference).  tmp: device float[H*W]. */
int frcnn_image_contrastive_norm(const float *in, int H, int W, const float *kernel_host, int K,
                                 float threshold, float *out, float *tmp, void *stream);

/* ---- example assembly on the host: Anchors.lua:69-235 + BatchIterator.lua:198-225 ----
 * Host-only native code (no device work), for loaders that must keep up with 200+ images/s per GPU.
 * frcnn_anchors_create: w_tab / h_tab = the fp32 tables of Anchors.lua:18-19 ([nscales][3][width][2]), cx / cy = the
 * anchor centres that key the 16-pixel bins of findNearby (double[nscales][3][width]); all host pointers. */
typedef struct frcnn_anchors frcnn_anchors;
int frcnn_anchors_create(const float *w_tab_host, const float *h_tab_host, const double *cx_host,
                         const double *cy_host, int nscales, int width, frcnn_anchors **out_host);
int frcnn_anchors_destroy(frcnn_anchors *);
/* One image of nextTraining: findPositive(rois, image, pos_thr, neg_thr, best_match), sampleNegative(image, rois,
 * neg_thr, negatives) and, when nearby_aversion != 0, the nearby anchors below neg_thr shuffled with shuffle_n
 * (min(#positive, #candidates) of them).  rois_host double[nroi][4].  Random draws come from the MT19937 state
 * mt_state_host[624] / *mt_index_host (torch.random's generator), updated in place.  Output, positives first:
 * ex_host int[cap][5] = {layer, aspect, y, x, roi (1-based; 0 for negatives)} 1-based, ex_rect_host double[cap][4]. */
int frcnn_anchors_assemble(frcnn_anchors *, const double *rois_host, int nroi, double img_w, double img_h,
                           double pos_thr, double neg_thr, int best_match, int nearby_aversion,
                           int negatives, unsigned int *mt_state_host, int *mt_index_host, int *ex_host,
                           double *ex_rect_host, int cap, int *npos_host, int *nneg_host);

/* ---- data-parallel exchange step: communicator + all-reduce (SURVEY 8b last row, 8e) --------------
 * Not in the reference (one process, one device: main.lua:52).  The training step shards over images; the ranks
 * meet once per step, between the last pnet:backward (objective.lua:189) and gradient:div(cls_count)
 * (objective.lua:197-200): sum of the flat gradient (frcnn_allreduce_f32) and of the 8 fp64 accumulators declared
 * at objective.lua:52-58 (frcnn_allreduce_f64), then every rank applies the identical optim.rmsprop step.
 * One process per GPU; the communicator runs RCCL (xGMI inside a node), bound at first use (dlopen "librccl.so.1").
 * Rendezvous: rank 0 creates a 128-byte id (frcnn_comm_get_unique_id) and hands it to the other ranks by any
 * host-side channel; frcnn_comm_init_rank_file does that through a file on a path every rank can see: rank 0 removes
 * whatever a previous job left there and writes {id, job nonce, its pid} atomically; the others poll up to timeout_ms
 * and accept a file only when its nonce equals theirs and (same host) its writer is still alive, so an id left
 * behind by a crashed job is never joined.  The nonce is the environment variable FRCNN_COMM_NONCE (any string
 * common to the ranks of ONE job, e.g. the launcher's pid and start time; default "0").
 * frcnn_comm_init_rank* is a collective call: every rank of the job makes it, after frcnn_set_device.
 * The all-reduces are IN PLACE sums, asynchronous on `stream`, ordered like any other work of that stream. */
#define FRCNN_COMM_ID_BYTES 128
typedef struct frcnn_comm frcnn_comm;
int frcnn_comm_get_unique_id(void *id_host);
int frcnn_comm_exchange_id_file(const char *path, int rank, void *id_host, int timeout_ms);   /* host only, no RCCL */
int frcnn_comm_init_rank(frcnn_comm **out_host, int nranks, int rank, const void *id_host);
/* The same under a watchdog: a peer that died between the rendezvous and this collective call surfaces as an error after
 * timeout_ms instead of a hang (non-blocking ncclCommInitRankConfig + ncclCommAbort where RCCL has them).
 * frcnn_comm_init_rank_file applies its timeout_ms to the file wait and to the initialisation.  Environment:
 * FRCNN_COMM_CHANNELS=n caps the channels (CUs) RCCL's kernels occupy beside the training step. */
int frcnn_comm_init_rank_timeout(frcnn_comm **out_host, int nranks, int rank, const void *id_host, int timeout_ms);
int frcnn_comm_init_rank_file(frcnn_comm **out_host, int nranks, int rank, const char *path, int timeout_ms);
int frcnn_comm_destroy(frcnn_comm *);
int frcnn_comm_info(const frcnn_comm *, int *nranks_host, int *rank_host);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): lets a bench
 * or a test assert that N ranks on N distinct devices really joined.  Any pointer may be NULL. */
int frcnn_comm_query(const frcnn_comm *, int *count_host, int *user_rank_host, int *device_host);
int frcnn_allreduce_f32(frcnn_comm *, float *buf, long long n, void *stream);
int frcnn_allreduce_f64(frcnn_comm *, double *buf, long long n, void *stream);
/* one-time weight broadcast after load_model / restore (main.lua:92-98) so that every replica starts identical */
int frcnn_broadcast_f32(frcnn_comm *, float *buf, long long n, int root, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
