/*
 * frcnn_oracle.h -- CPU restatement (plain C) of the Faster R-CNN hot path of
 * andreaskoepf/faster-rcnn.torch.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (faster-rcnn.torch_amd/ + libfrcnn_hip.so) never links, imports or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, fixtures or golden
 * vectors, is written in Lua on Torch7 (no Lua/Torch7 runtime exists in the build
 * container) and therefore can be neither compiled nor imported.  This restatement
 * follows the Lua sources line by line (each function cites the file:line it follows);
 * it is pinned by (1) the hand-derived known answers of SURVEY.md Appendix B,
 * (2) PyTorch-CPU as an independent implementation of the layer math, and
 * (3) a second, deliberately naive numpy restatement (oracle/naive_np.py).
 *
 * Torch7 semantics that are assumed (un-vendored third-party packages, [ext]):
 * see oracle/ASSUMPTIONS.md.
 *
 * Conventions: rect = double[4] {minX, minY, maxX, maxY}; all *indices* at this API
 * are 1-based exactly like the Lua surface unless a comment says otherwise; tensors are
 * CHW fp32, row-major contiguous.
 */
#ifndef FRCNN_ORACLE_H
#define FRCNN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Rect.lua ------------------------------------------------------------------- */
double orc_rect_iou(const double *a, const double *b);             /* Rect.lua:138-141 */
int orc_rect_overlaps(const double *a, const double *b);           /* Rect.lua:90-93   */
void orc_rect_clip(const double *r, const double *c, double *out); /* Rect.lua:73-80   */
void orc_rect_intersect(const double *a, const double *b, double *out); /* Rect.lua:126-136 */
void orc_rect_snap_to_int(const double *r, double *out);           /* Rect.lua:147-149 */

/* ---- Localizer.lua -------------------------------------------------------------- */
/* layers: int[nlayers][6] = {kW,kH,dW,dH,padW,padH}; layer_index<=0 means "all". */
void orc_loc_input_to_feature(const int *layers, int nlayers, int layer_index,
                              const double *rect, double *out);   /* Localizer.lua:41-67 */
void orc_loc_feature_to_input(const int *layers, int nlayers, int layer_index, double minX,
                              double minY, double maxX, double maxY,
                              double *out);                       /* Localizer.lua:69-79 */

/* ---- torch.random() (MT19937, [ext]) -------------------------------------------- */
typedef struct orc_mt orc_mt;
orc_mt *orc_mt_new(uint32_t seed);
void orc_mt_free(orc_mt *);
uint32_t orc_mt_random(orc_mt *);

/* ---- Anchors.lua ---------------------------------------------------------------- */
typedef struct orc_anchors orc_anchors;
/* layers_concat: the nscales localizer layer lists back to back; nlayers[i] entries each. */
orc_anchors *orc_anchors_new(const int *layers_concat, const int *nlayers, const double *scales,
                             int nscales);                        /* Anchors.lua:7-58 */
void orc_anchors_free(orc_anchors *);
const float *orc_anchors_w(const orc_anchors *); /* [nscales][3][200][2] fp32 (main.lua:51) */
const float *orc_anchors_h(const orc_anchors *);
void orc_anchors_get(const orc_anchors *, int layer, int aspect, int y, int x,
                     double *rect_out);                            /* Anchors.lua:60-67 */
/* ranges_out: int[12][6] = {layer, aspect, lx, ly, ux, uy}; returns count. clip may be NULL. */
int orc_anchors_find_ranges_xy(const orc_anchors *, const double *rect, const double *clip,
                               int *ranges_out);                   /* Anchors.lua:86-145 */
/* out_idx: int[cap][5] = {layer, aspect, y, x, roi(1-based)}; out_rect: double[cap][4].
 * Returns number of matches (may exceed cap; only cap are written). */
int orc_anchors_find_positive(const orc_anchors *, const double *rois, int nroi,
                              const double *clip, double pos_thr, double neg_thr,
                              int include_best, int *out_idx, double *out_rect,
                              int cap);                            /* Anchors.lua:147-195 */
/* out_idx: int[cap][4] = {layer, aspect, y, x}. */
int orc_anchors_sample_negative(const orc_anchors *, const double *image_rect,
                                const double *rois, int nroi, double neg_thr, int count,
                                orc_mt *rng, int *out_idx, double *out_rect,
                                int cap);                          /* Anchors.lua:197-235 */
int orc_anchors_find_nearby(const orc_anchors *, double cx, double cy, int *out_idx,
                            double *out_rect, int cap);            /* Anchors.lua:69-84 */
void orc_input_to_anchor(const double *anchor, const double *rect,
                         float *t4);                               /* Anchors.lua:237-243 */
void orc_anchor_to_input(const double *anchor, const float *t4,
                         double *rect_out);                        /* Anchors.lua:245-252 */

/* ---- nms.lua -------------------------------------------------------------------- */
/* key_mode: 0 = y2 (what every call site of the reference gets: nms.lua:37-43 with a tensor
 * or nil `scores`), 1 = 'area', 2 = column key_col (1-based).  pick_out: int64[n], 1-based
 * row ids in pick order.  Returns the number picked.  Tie rule (TH quicksort is unstable,
 * tie order is unpinned): ascending key, ties by ascending row id. */
int orc_nms(const float *boxes, int n, int ncols, float overlap, int key_mode, int key_col,
            int64_t *pick_out);                                    /* nms.lua:23-102 */

/* ---- ROI pooling (objective.lua:5-13 + nn.SpatialAdaptiveMaxPooling [ext]) ------- */
/* win_out: {row_lo,row_hi,col_lo,col_hi}, 1-based inclusive, as objective.lua:11. */
void orc_extract_roi_window(const int *layers, int nlayers, const double *input_rect, int fmH,
                            int fmW, int *win_out);
/* out: [C][kh][kw]; idx: flat 0-based (y*W+x) position in the full map of each max. */
void orc_adaptive_max_pool_fwd(const float *fmap, int C, int H, int W, const int *win, int kh,
                               int kw, float *out, int32_t *idx);
/* gmap[C][H][W] += scatter(gout[C][kh][kw]) */
void orc_adaptive_max_pool_bwd(float *gmap, int C, int H, int W, int kh, int kw,
                               const float *gout, const int32_t *idx);

/* ---- layer math ([ext] Torch7 nn semantics, see ASSUMPTIONS.md) ------------------ */
void orc_conv2d_fwd(const float *in, int C, int H, int W, const float *wt, const float *bias,
                    int O, int kh, int kw, int pad, float *out);
void orc_conv2d_bwd_input(const float *gout, int O, int Ho, int Wo, const float *wt, int C,
                          int kh, int kw, int pad, int H, int W, float *gin);
void orc_conv2d_bwd_weight(const float *in, int C, int H, int W, const float *gout, int O,
                           int kh, int kw, int pad, float *gw, float *gb); /* accumulates */
void orc_prelu_fwd(const float *x, long n, float a, float *y);
/* gx = gy * (x>0 ? 1 : a); returns sum over x<=0 of x*gy (double) */
double orc_prelu_bwd(const float *x, const float *gy, long n, float a, float *gx);
void orc_maxpool2x2_ceil_fwd(const float *in, int C, int H, int W, float *out, int32_t *idx);
void orc_maxpool2x2_ceil_bwd(const float *gout, const int32_t *idx, int C, int H, int W,
                             float *gin);
void orc_linear_fwd(const float *x, int R, int I, const float *wt, const float *b, int O,
                    float *y);
void orc_linear_bwd(const float *x, const float *gy, int R, int I, const float *wt, int O,
                    float *gx, float *gw, float *gb); /* gw, gb accumulate */
void orc_log_softmax(const float *x, int R, int n, float *y);

/* ---- model description (models/vgg_small.lua:5-22 etc.) ------------------------- */
typedef struct {
  int nblocks;
  int filters[8], ksize[8], pad[8], conv_steps[8];
  double dropout[8];
  int nheads;
  int head_k[8], head_n[8], head_input[8]; /* head_input is 1-based block index */
  int ncls;
  int cls_n[8], cls_bn[8];
  double cls_dropout[8];
  int class_count; /* excluding background */
  int kh, kw;      /* roi pooling */
  double scales[4];
} orc_model;

long orc_model_param_count(const orc_model *, long *pnet_count);
/* Localizer layer list for output node i (1..nheads = heads, nheads+1 = feature map);
 * returns nlayers, writes int[n][6]. (Localizer.lua:6-39 applied to model_utilities.lua:43-58) */
int orc_model_localizer_layers(const orc_model *, int output_index, int *layers_out);

typedef struct orc_pnet_state orc_pnet_state;
orc_pnet_state *orc_pnet_state_new(void);
void orc_pnet_state_free(orc_pnet_state *);
/* drop_masks: nblocks pointers (or NULL array / NULL entries): per-channel 0/1 keep masks for
 * the SpatialDropout after the first conv of each block with dropout>0. training!=0 ->
 * masks applied with no rescale; training==0 -> x(1-p) ([ext] 2015 semantics). */
void orc_pnet_forward(const orc_model *, const float *weights, const float *img, int H, int W,
                      int training, const float *const *drop_masks, orc_pnet_state *st);
const float *orc_pnet_output(const orc_pnet_state *, int i /*1..nheads+1*/, int *C, int *H,
                             int *W);
/* delta_outputs: nheads+1 pointers with the shapes of the outputs. grad accumulates. */
void orc_pnet_backward(const orc_model *, const float *weights, const orc_pnet_state *st,
                       const float *const *delta_outputs, float *grad);

typedef struct orc_cnet_state orc_cnet_state;
orc_cnet_state *orc_cnet_state_new(void);
void orc_cnet_state_free(orc_cnet_state *);
/* weights/grad point at the START of the flat vector (pnet params first). bn_running:
 * float[2*n] {mean, var} per BN layer concatenated, updated in training.  drop_masks[l]:
 * R x n_l 0/1 keep masks (training: y = x*mask/(1-p), Dropout v2 [ext]). */
void orc_cnet_forward(const orc_model *, const float *weights, const float *x, int R,
                      int training, const float *const *drop_masks, float *bn_running,
                      orc_cnet_state *st, float *bbox_out /*R x 4*/,
                      float *cls_out /*R x (classes+1) log-probs*/);
void orc_cnet_backward(const orc_model *, const float *weights, const orc_cnet_state *st,
                       const float *g_bbox, const float *g_cls, float *gx /*R x D*/,
                       float *grad);

/* ---- objective.lua:45-218 for ONE image (the body of the `for i,x in ipairs(batch)` loop).
 * pos_idx: int[np][5] {layer,aspect,y,x,roi(1-based)}; pos_rect double[np][4] (anchor rects);
 * rois double[nroi][4]; roi_class int[nroi] (1-based class index);
 * neg_idx: int[nn][4]; neg_rect double[nn][4].  Examples must already be cleanAnchors()-ed.
 * acc: double[8] {cls_loss, reg_loss, cls_count, reg_count, creg_loss, creg_count,
 * ccls_loss, ccls_count} accumulated (objective.lua:52-58).  grad accumulates, NOT divided. */
/* Decision injection -- TESTS ONLY.  Discrete choices of the path (max-pool window winner, adaptive max-pool cell
 * winner, PReLU branch) that the caller wants taken as given (`inject`: the device's own choices) and / or written out
 * as this restatement takes them (`record`: caller-allocated buffers of the same shapes, written during the next
 * orc_pnet_forward / orc_cnet_forward / orc_train_image).  NULL entries (or a NULL struct) leave that decision to the
 * restatement.  Global state: set, run, reset to (NULL, NULL).
 *   pool_idx[b]   int32  [C][Hp][Wp]  flat y*W+x into the pooled layer's (activated) input plane, block b
 *   conv_pos[i]   uint8  [O][Ho][Wo]  1 where backbone conv i's pre-activation takes the x > 0 branch
 *   head_pos[h]   uint8  [n][Ho][Wo]  the same for anchor net h's k x k convolution
 *   cnet_pos[l]   uint8  [R][n]       the same for classification layer l (PReLU input = BN output / Linear output)
 *   roi_idx       int32  [R][planes*kh*kw] flat y*W+x into the last pooled map's plane (orc_train_image only)
 *   slope_abs     double [48]         (record only) see below */
typedef struct {
  const int32_t *pool_idx[8];
  const uint8_t *conv_pos[32];
  const uint8_t *head_pos[8];
  const uint8_t *cnet_pos[8];
  const int32_t *roi_idx;
  double *slope_abs;  /* record only, 48 doubles, ADDED to: sum |x * gy| over the entries each PReLU slope gradient sums --
                         [0..31] backbone convolutions, [32..39] anchor nets, [40..47] classification layers */
} orc_decisions;
void orc_set_decisions(const orc_decisions *inject, orc_decisions *record);

void orc_train_image(const orc_model *, const float *weights, float *grad, const float *img,
                     int H, int W, const int *pos_idx, const double *pos_rect, int np,
                     const double *rois, const int *roi_class, int nroi, const int *neg_idx,
                     const double *neg_rect, int nn, const float *const *pnet_drop_masks,
                     const float *const *cnet_drop_masks, float *bn_running, double *acc);

/* ---- Detector.lua:17-141.  Outputs (all caller-allocated with capacity cap):
 * match_*: the scan result before NMS (Detector.lua:39-66), in scan order;
 * returns number of matches via *nmatch, candidates (after NMS 0.25) via cand_ids (1-based
 * ids into matches), winners as rows {class, confidence(log-prob), r2[4], cand_id}. */
typedef struct {
  int nmatch, ncand, nwin;
} orc_detect_counts;
void orc_detect(const orc_model *, const float *weights, const float *bn_running,
                const float *img, int H, int W, int cap, float *match_p, int *match_idx /*[cap][4]*/,
                double *match_rect /*[cap][4]*/, int64_t *cand_ids, float *cand_bbox /*[cap][4]*/,
                float *cand_cls /*[cap][classes+1]*/, double *win_rows /*[cap][7]*/,
                orc_detect_counts *counts);

/* ---- optim.rmsprop ([ext]): m = a*m + (1-a)*g*g ; x -= lr * g / (sqrt(m) + eps) ---- */
void orc_rmsprop(float *x, const float *g, float *m, long n, float lr, float alpha, float eps);

void orc_set_threads(int n);
int orc_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
