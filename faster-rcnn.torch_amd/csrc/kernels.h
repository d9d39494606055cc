// kernels.h -- internal host-side launchers (C++), shared by the C-ABI entry points (api.cpp)
// and the native pnet/cnet runtime (net.cpp).  All pointers are DEVICE pointers unless named
// *_host.  Every launcher is asynchronous on `s` and returns FRCNN_OK / error code.
#pragma once
#include <algorithm>

#include "common.h"

namespace frcnn {

// ---------------------------------------------------------------- conv (conv.hip)
// Packed weight matrices for the implicit GEMM: [Kp rows][Mpad cols], M fastest.
//   fwd   : row ((cp*k+ky)*k+kx)*2+h  <-  W[m][c=2cp+h][ky][kx]            (M = O, K-chan = C)
//   dgrad : row ((op*k+ky)*k+kx)*2+h  <-  W[o=2op+h][m][k-1-ky][k-1-kx]    (M = C, K-chan = O)
int conv_cc(int k);                          // channels per K-chunk for kernel size k
int conv_mpad(int M);                        // padded M
size_t conv_pack_floats(int Kchan, int M, int k);
int conv_pack_weights(const float* w, int O, int C, int k, float* wf, float* wd, hipStream_t s);
// table-driven variant: every pack of a model in one launch (jobs live in device memory)
struct PackJob { long w_off; long total; float* dst; int O, C, k, mode, Mpad, blk_begin, nblk; };
PackJob conv_pack_job(long w_off, int O, int C, int k, int mode, float* dst);
int conv_pack_assign_blocks(PackJob* jobs, int njobs, int total_blocks);
int conv_pack_weights_multi(const float* weights, const PackJob* jobs_dev, int njobs, int grid, hipStream_t s);

enum { OUT_STORE = 0, OUT_ADD = 1 };
// out[M][Ho][Wo] (=|+=) conv(act(in)[Cin][H][W], wp) (+ bias).  act(x) = scale[c]*prelu(x, *slope)
// when the pointers are non-null.  Ho = H + 2*pad - k + 1.
// Optional `pool`: also produce maxpool_act_forward(out) (2x2, stride 2, ceil mode, act = scale[m] * prelu(., *slope)) from the
// accumulators; *pool_fused tells whether the launch could do it (3x3, plain store, one K split) -- otherwise the caller
// runs maxpool_act_forward itself.
struct IgemmPool { float* out; unsigned char* idx; const float* slope; const float* scale; float* amax = nullptr; };   // amax: magnitude record of `out` (amax.h)
int conv_igemm(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale,
               const float* wp, const float* bias, int M, int k, int pad, float* out, int out_mode,
               double algo_flops, hipStream_t s, int ws_slot = 0,  // ws_slot: split-K workspace (0 | 1)
               const IgemmPool* pool = nullptr, bool* pool_fused = nullptr);

// ---- split operand forms of the convolutions (convx.hip): fp32 tensors in and out, fp32 accumulation on the 16-bit matrix
// cores; every product formed from three exact fp16 x fp16 partial products of operands scaled by powers of two and split in
// two (option x3_f16, the default; needs the magnitudes of both tensors: amax_* below), or from six exact bf16 x bf16 partial
// products of a three-way split -- the accuracy of the fp32 matrix-core kernel at 3/16 (6/16) of its matrix-pipe time.
// wp = stages packed by conv_x3_pack*.
void set_split_bf16(int on);   // option "split_bf16": 1 (default) eligible 3x3 launches take this form, 0 = fp32 MFMA only
int get_split_bf16();
bool conv_x3_eligible(int Cin, int M, int k);   // k == 3: Cin % 16 == 0, M % 64 == 0; k in {5, 7}: M % 128 == 0 (and the option is on)
size_t conv_x3_pack_bytes(int Kchan, int M, int k);
struct PackXJob {
  long w_off; long total; void* dst; const float* amax; float* amax_w; int O, C, k, mode, bm, blk_begin, nblk;   // mode 0 forward, 1 input gradient; bm = filters per block
  // Optional gather (round 6, the channels a SpatialDropout keeps): the pack describes a convolution of O x C filters whose
  // filter o' / channel c' is the tensor's filter oidx[o'] / channel cidx[c'] (device tables; a negative entry = a filter /
  // channel of zeros; null = the identity), Cs = channels of the SOURCE tensor (its row pitch; 0 = C).  bias_dst (mode 0 with
  // oidx): bias_dst[o'] = weights[bias_off + oidx[o']] (0 for a negative entry), written by the job's first block.
  const int* oidx = nullptr; const int* cidx = nullptr; int Cs = 0; long bias_off = 0; float* bias_dst = nullptr;
};
PackXJob conv_x3_pack_job(long w_off, int O, int C, int k, int mode, void* dst, int Ho, int Wo);   // Ho x Wo: output map of the launch it feeds
int conv_x3_pack_assign_blocks(PackXJob* jobs, int njobs);   // -> grid size
int conv_x3_pack_multi(const float* weights, const PackXJob* jobs_dev, int njobs, int grid, hipStream_t s);
int conv_x3_pack(const float* w, int O, int C, int k, int mode, void* dst, hipStream_t s, int Ho, int Wo,
                 const float* amax_rec_w = nullptr, float* amax_w = nullptr);   // fp16 form: the weights' magnitude record in, their largest magnitude out
// Magnitude records (amax.h): rec = AMAX_REC floats of device memory per tensor.
#define AMAX_REC 16400   // floats per record: the count + one entry per block of the producing launch (up to 16 384) + pad
#define AMAX_MAX_BLOCKS 16384   // a launch that keeps a record may have at most this many blocks (checked by the launchers)
int tensor_absmax(const float* x, long n, float* rec, hipStream_t s);   // a pass of its own over a tensor
struct AmaxJob { long off; long n; float* out; int blk_begin; };   // a segment of the flat parameter vector and its record
int tensor_absmax_assign_blocks(AmaxJob* jobs, int njobs);   // -> grid size
long tensor_absmax_record_floats(long n);                     // floats of a segment's record
int tensor_absmax_multi(const float* w, const AmaxJob* jobs_dev, int njobs, int grid, hipStream_t s);
void set_x3_f16(int on);   // option "x3_f16": the split launches take the two-plane fp16 form (three partial products) instead of three bf16 planes (six)
int get_x3_f16();
int conv_x3(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale, const void* wp,
            const float* bias, int M, int k, int pad, float* out, int out_mode, double algo_flops, hipStream_t s, int ws_slot = 0,
            const struct X3PostAct* post = nullptr, const float* amax_in = nullptr, const float* amax_w = nullptr,
            float* amax_out = nullptr);   // amax_in / amax_w: the two-plane fp16 form; amax_out: magnitude of what is stored (amax.h)
// Backward of the PReLU + SpatialDropout the OUTPUT gradient of an input-gradient launch passes through next, fused into the
// launch's epilogue (or into the fold of its split-K slabs): out = prelu'(x) * scale[m] * (conv result), *gslope += sum over
// x <= 0 of x * scale[m] * (conv result).  What act_backward (elem.hip) does in a pass of its own, minus the bias sums,
// which the weight-gradient launch reading `out` takes along (conv_wgrad's gbias).
struct X3PostAct {
  const float* x;        // [M][Ho][Wo] pre-activation output of the layer whose activation is undone
  const float* slope;    // device scalar
  const float* scale;    // [M] dropout scale or null
  float* gslope;         // device scalar, accumulated
};

// weight gradient in the same split-bf16 form (wgradx.hip): k == 3, Cin % 64 == 0, O % 64 == 0; conv_wgrad routes to it
bool conv_wgradx_eligible(int Cin, int O, int k);
size_t conv_wgradx_workspace_bytes(int Cin, int H, int W, int O, int pad);
// Where a weight gradient computed for O x Cin gathered filters / channels lands in the full tensor (round 6): filter o' is the
// tensor's filter omap[o'], channel c' its channel cmap[c'] (device tables; negative = padding, nothing is written; null = the
// identity), Cfull = channels of the full tensor.  gbias follows omap.
struct WgradMap { const int* omap = nullptr; const int* cmap = nullptr; int Cfull = 0; };
int conv_wgradx(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale, const float* g, int O,
                int pad, float* gw, void* ws, size_t ws_bytes, hipStream_t s, float* gbias = nullptr,
                const float* amax_in = nullptr, const float* amax_g = nullptr,   // both records (amax.h): the two-plane fp16 form
                const WgradMap* map = nullptr);
// gw[o][c][tap] += sum_s slab[s][tap][o][c]   (the fold shared by the weight-gradient kernels)
int wgrad_reduce(const float* slab, int nSplit, int taps, int OC, float* gw, hipStream_t s);
// the same through a WgradMap: slab [s][tap][O][C] of the gathered problem, gw the full tensor
int wgrad_reduce_map(const float* slab, int nSplit, int taps, int O, int C, float* gw, const WgradMap& map, hipStream_t s);
// first layer of a one-convolution block: weight, bias and slope gradients straight from the pooled map's gradient
bool conv_wgrad_first_pooled_eligible(int Cin, int O, int k, int Wo);
int conv_wgrad_first_pooled(const float* in, int Cin, int H, int W, const float* gpool, const unsigned char* pidx, const float* x,
                            const float* slope, int O, int pad, float* gw, float* gbias, float* gslope, void* ws, size_t ws_bytes,
                            hipStream_t s);

// gw[O][Cin][k][k] += sum_pix g[O][Ho][Wo] * act(in)[Cin][H][W]   (split-K slabs in `ws`, folded in a fixed order)
// `ws`: split-K slab workspace of at least conv_wgrad_workspace_bytes(...) bytes.
size_t conv_wgrad_workspace_bytes(int Cin, int H, int W, int O, int k, int pad);
int conv_wgrad(const float* in, int Cin, int H, int W, const float* in_slope, const float* in_scale,
               const float* g, int O, int k, int pad, float* gw, void* ws, size_t ws_bytes, hipStream_t s,
               float* gbias = nullptr,   // gbias: also gbias[o] += sum over pixels of g (in the same launch where the kernel can)
               const float* amax_in = nullptr, const float* amax_g = nullptr,   // magnitude records of in / g: conv_wgradx's fp16 form
               const WgradMap* map = nullptr);   // (conv_wgradx shapes only)

// ---------------------------------------------------------------- deterministic mode (frcnn_set_option("deterministic", 1))
// Default: per-block partial sums of the bias / slope gradients and the scatter-adds of the ROI-pooling and sparse anchor-net
// backward passes meet in fp32 atomics, whose order -- hence the last bits of the result -- varies from run to run.
// Deterministic: block partials go to a scratch array and are folded in index order, the anchor deltas are applied in example
// order, the sparse col2im becomes a gather, and the ROI-pooling backward accumulates in 64-bit fixed point (exact, so the order
// does not matter).  Two runs on the same inputs are then bit-identical.
bool deterministic();
void set_deterministic(bool on);
// stream-ordered scratch of at least `floats` floats owned by the library (one buffer per stream, grown on demand)
int det_workspace(hipStream_t s, size_t floats, float** out);

// ---------------------------------------------------------------- elementwise (elem.hip)
int fill_zero(void* p, size_t bytes, hipStream_t s);
int scale_inplace(float* x, long n, float sc, hipStream_t s);
int add_inplace(float* y, const float* x, long n, hipStream_t s);   // y += x
// y[c][hw] = scale[c] * prelu(x[c][hw])       (materialise an activated tensor; tests only)
int act_forward(const float* x, int C, long hw, const float* slope, const float* scale, float* y,
                hipStream_t s);
// 2x2 stride-2 ceil-mode max pool of act(x); idx = argmax code 0..3 (dy*2+dx), first max wins
int maxpool_act_forward(const float* x, int C, int H, int W, const float* slope, const float* scale,
                        float* out, unsigned char* idx, hipStream_t s, float* amax = nullptr);   // amax: see amax.h
// gx = route(gpool, idx) * scale[c] * prelu'(x);  gbias[c] += sum gx;  *gslope += sum_{x<=0} x*gy*scale
int maxpool_act_backward(const float* gpool, const unsigned char* idx, const float* x, int C, int H,
                         int W, const float* slope, const float* scale, float* gx, float* gbias,
                         float* gslope, hipStream_t s, float* amax = nullptr);
// gx = gy * scale[c] * prelu'(x) (in place allowed); gbias[c] += sum gx; *gslope += ...
int act_backward(const float* gy, const float* x, int C, long hw, const float* slope,
                 const float* scale, float* gx, float* gbias, float* gslope, hipStream_t s, float* amax = nullptr);
// gbias[c] += sum_hw g[c][hw]
int channel_sum(const float* g, int C, long hw, float* gbias, hipStream_t s);
// gb[o] += sum_r g[r][o] for a row-major R x O matrix
int channel_sum_cols(const float* g, int R, int O, float* gb, hipStream_t s);
// sparse anchor-head backward helpers (positions = flat y*Wo+x indices into the head's output map)
int gather_positions(const float* src, int C, long hw, const int* pos, int P, float* dst, const float* slope,
                     float* dst_act, hipStream_t s);
int im2col_positions(const float* X, int C, int H, int W, int k, int Wo, const int* pos, int P, float* col, hipStream_t s);
int col2im_positions_add(const float* col, int C, int H, int W, int k, int Wo, const int* pos, int P, float* gX,
                         hipStream_t s);
// per-channel Bernoulli keep mask (nn.SpatialDropout, model_utilities.lua:10-12): scale[c] in {0,1}
int dropout_channel_mask(float* scale, int C, float p, unsigned long long seed, hipStream_t s);
int fill_value(float* x, long n, float v, hipStream_t s);
int rmsprop_step(float* x, float* g, float* m, long n, float lr, float alpha, float eps, float gscale,
                 bool scale_first, hipStream_t s, const double* gcount_dev = nullptr);
// the same step on elements [lo, hi) only (any bounds; the vectors' bases 16-byte aligned): bit-identical to what rmsprop_step
// leaves in those elements
int rmsprop_slice(float* x, float* g, float* m, long lo, long hi, float lr, float alpha, float eps, float gscale, bool scale_first,
                  hipStream_t s);

// ---------------------------------------------------------------- anchor nets, sampled positions only (heads.hip)
#define FRCNN_HEAD_OUT 18   // 3 * (2 + 4) planes of an anchor net's 1 x 1 convolution (model_utilities.lua:33)
struct HeadJob {
  int Cin, H, W, k, Ho, Wo, n;         // input map, kernel size, output map of the k x k valid convolution, its filters
  int P; const int* pos;               // sampled positions (y * Wo + x), device
  const float* in; float* gin;         // the input map and its gradient
  const float *bias3, *slope, *bias1;  // parameters: k x k bias, PReLU slope, 1 x 1 bias
  float *gbias3, *gslope, *gbias1;     // ... their gradients
  float* out; const float* delta;      // the 1 x 1 convolution's output map [18][Ho Wo] and its gradient map
  float *COL, *HX, *HY, *OUT, *D, *GH, *DX;   // scratch: [P][ckk], [n][P], [n][P], [18][P], [18][P], [n][P], [P][ckk]
  const float* hx_slab; int hx_splits; // partial sums of HX over K splits, [split][n][P]
};
struct HeadJobs { HeadJob j[4]; int n; };
int heads_im2col(const HeadJobs& g, hipStream_t s);
int heads_bias_act(const HeadJobs& g, hipStream_t s);
int heads_scatter(const HeadJobs& g, hipStream_t s);
int heads_gather_delta(const HeadJobs& g, hipStream_t s);
int heads_act_backward(const HeadJobs& g, hipStream_t s);
int heads_col2im(const HeadJobs& g, hipStream_t s);

// ---------------------------------------------------------------- gemm (gemm.hip)
// C[M][N] (=|+=) A[M][K] * B[K][N] with explicit element strides.  defer: see GemmFold below.
struct GemmFold;
// several independent fp32 products in one launch (gemm.hip gemm_group_kernel): C[M][N] (=|+=) A * B with explicit strides
#define GEMM_GROUP_MAX 4
struct GemmJob {
  const float* A; long sAm, sAk; const float* B; long sBk, sBn; float* C; long ldc;
  int M, N, K, out_mode;   // OUT_STORE | OUT_ADD
  int splits = 1;          // > 1: C receives [split][M][N] partial sums over K instead (the caller folds them)
  int kPerSplit = 0;       // (set by the launcher)
};
int gemm_f32_group(GemmJob* jobs, int n, hipStream_t s);
int gemm_f32(const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C,
             long ldc, int M, int N, int K, int out_mode, const float* bias_n, hipStream_t s, int ws_slot = 0,
             GemmFold* defer = nullptr);
int gemm_workspace_get(size_t need, float** out, int slot);
// A product whose split-K fold is left to its CONSUMER (round 4: the classification net's row-wise layers read the partial sums
// themselves instead of waiting for a fold launch on the dependent chain): value(r, j) = bias[j] + sum_s slab[s][r * n + j] in
// split order -- the number gemm_reduce_kernel would have stored -- or, nSplit == 0, what the product stored at its destination.
// The slabs live in the stream's split-K workspace: the consumer must be the next launch that uses it.
struct GemmFold {
  const float* slab = nullptr;
  int nSplit = 0;
  const float* bias = nullptr;
};
int gemm_reduce_slabs(const float* slab, int nSplit, int M, int N, const float* bias, float* C, long ldc, bool accumulate,
                      hipStream_t s);
// ---- split-bf16 operand form of the large Linear (gemmx.hip): activation operands as three bf16 planes written once
// (split_planes), weights split in registers; fp32 results of fp32 accuracy (see convx.hip)
void set_gemm_x_roles(int mask);   // option "gemm_x_roles": -1 the built-in rule, else bit mask of roles in the split form
int get_gemm_x_roles();
bool linear_x_eligible(int role, int R, int I, int O);   // role: 1 forward, 2 input gradient, 4 weight gradient
int linear_x_rows_padded(int R);                       // rows of the transposed planes (R rounded up to 16, zero filled)
int split_planes(const float* src, int R, int C, void* P /* [3][C/8][R][8] bf16 or null */, void* PT /* [3][Rp/8][C][8] or null */,
                 hipStream_t s, const float* amax = nullptr);   // amax (record of src, amax.h): two fp16 planes [2][C/8][R][8] instead (P only)
// amax_x / amax_g + amax_w (records of the activation operand and of the weight matrix): the two-plane fp16 form (planes made by
// split_planes with the same record)
int linear_x_forward(const void* Xp, int R, int I, const float* W, const float* bias, int O, float* y, hipStream_t s, int ws_slot = 0,
                     GemmFold* defer = nullptr, const float* amax_x = nullptr, const float* amax_w = nullptr);
int linear_x_dgrad(const void* Gp, int R, int O, const float* W, int I, float* gx, int out_mode, hipStream_t s, int ws_slot = 0,
                   GemmFold* defer = nullptr, const float* amax_g = nullptr, const float* amax_w = nullptr);
int linear_x_wgrad(const void* GpT, const void* XpT, int R, int O, int I, float* gw, hipStream_t s, int ws_slot = 0,
                   const float* amax_g = nullptr, const float* amax_x = nullptr);   // both records: two fp16 planes per operand (PT of split_planes with amax)


// ---------------------------------------------------------------- roi (roi.hip)
int roi_pool_forward(const float* fmap, int C, int H, int W, const int* wins, int R, int kh, int kw,
                     float* out, int* idx, hipStream_t s);
int roi_pool_backward(float* gmap, int C, int H, int W, const float* gout, const int* idx, int R,
                      int kh, int kw, hipStream_t s);

// ---------------------------------------------------------------- rpn (rpn.hip)
struct RpnLayers {
  const float* map[4];
  int H[4], W[4];
};
int rpn_scan(const RpnLayers& L, const float* anchor_w, const float* anchor_h, double img_w,
             double img_h, double p_threshold, int cap, float* match_p, int* match_idx,
             double* match_rect, float* match_box, int* count, void* ws, size_t ws_bytes,
             hipStream_t s);
size_t rpn_scan_workspace_bytes(const RpnLayers& L);
int rpn_loss(const RpnLayers& L, float* const* delta, const int* ex_idx, const double* ex_anchor,
             const double* ex_roi, const int* ex_class, int npos, int nneg, int bgclass,
             double* ex_loss, float* crtarget, float* cctarget, hipStream_t s);

int loss_accumulate(const double* ex_loss, int E, double* acc, hipStream_t s);

// ---------------------------------------------------------------- nms (nms.hip)
size_t nms_workspace_bytes(int n);
// cls (optional, int[n]): rows only suppress rows of the same class (Detector.lua:125-136 in one pass)
int nms_device(const float* boxes, int n, int ncols, float overlap, int key_mode, int key_col,
               long long* pick, int* count, void* ws, size_t ws_bytes, hipStream_t s, const int* cls = nullptr,
               const int* n_dev = nullptr);   // n_dev: the row count is read from device memory (<= n)
// ---- Detector:detect glue that stays on the device (detect.hip)
int roi_windows(const double* rect, const long long* pick, int k, const int* layers, int nlayers, int fmH, int fmW, int* wins,
                hipStream_t s);
int detect_post(const int* cls, const float* conf, const float* bbox, const double* rect, const long long* pick, int R,
                int bgclass, double min_conf, float* bb, int* kc, int* keep_row, double* r2, int* K_dev, hipStream_t s);
int detect_gather(const long long* wpick, const int* nwin_dev, int cap, const int* keep_row, const int* kc, const float* bb,
                  const double* r2, const long long* pick, const float* mp, const double* rect, const int* midx, double* rec,
                  hipStream_t s);

// ---------------------------------------------------------------- cnet small ops (cnet.hip)
int bn_forward(const float* x, int R, int n, const float* gamma, const float* beta, float* running,
               int training, float* xhat, float* invstd, float* y, hipStream_t s);
int bn_backward(const float* gy, const float* xhat, const float* invstd, const float* gamma, int R,
                int n, int training, float* gx, float* ggamma, float* gbeta, hipStream_t s);
int prelu_dropout_forward(const float* x, long n, const float* slope, const float* mask, float inv_keep,
                          float* y, hipStream_t s);
// same, drawing the keep mask (probability 1-p, counter-based stream `seed`) into mask_out on the fly
int prelu_dropout_forward_gen(const float* x, long n, const float* slope, float* mask_out, float p,
                              unsigned long long seed, float* y, hipStream_t s);
int prelu_dropout_backward(const float* gy, const float* x, long n, const float* slope,
                           const float* mask, float inv_keep, float* gx, float* gslope, hipStream_t s);
// ---- the same layers FUSED, and reading a product's split-K partial sums themselves (GemmFold; src.nSplit == 0: plain x)
// fold (+ bias) -> [BatchNormalization ->] PReLU -> Dropout in ONE launch.  lin: the folded product (kept: the backward pass
// of a layer without batch normalisation reads it); gamma == null: no batch normalisation (xhat / invstd / pre unused).
// mask: given (gen = false; may be null = no dropout) or drawn here into it (gen = true, probability p, stream `seed`).
int cnet_act_forward(const float* x, GemmFold src, int R, int n, const float* gamma, const float* beta, float* running,
                     int training, float* lin, float* xhat, float* invstd, float* pre, const float* slope, float* mask, bool gen,
                     float inv_keep, float p, unsigned long long seed, float* post, hipStream_t s);
// Dropout -> PReLU -> BatchNormalization backward in ONE launch (gy may be a deferred product: the next layer's input gradient)
int cnet_act_bn_backward(const float* gy, GemmFold src, const float* pre, const float* xhat, const float* invstd,
                         const float* gamma, const float* slope, const float* mask, float inv_keep, int R, int n, int training,
                         float* gx, float* ggamma, float* gbeta, float* gslope, hipStream_t s);
// The two output heads of the classification net in ONE launch each way (4 + nc <= 32 outputs; models/model_utilities.lua:117-124):
//   forward : bbox = x Wb^T + bb ; logits = x Wc^T + bc ; lsm = LogSoftMax(logits) (also copied to cls_out)
//   backward: glog = LogSoftMax'(g_cls) ; gfeat = g_bbox Wb + glog Wc   (the weight gradients stay GEMMs on their own stream)
// fp32 multiply-adds along the features (four partial sums per lane half), LogSoftMax in fp64.
bool cnet_heads_fused_eligible(int nf, int nc);
int cnet_heads_forward(const float* x, int R, int nf, const float* Wb, const float* bb, const float* Wc, const float* bc, int nc,
                       float* bbox_out, float* logits, float* lsm, float* cls_out, hipStream_t s);
// post: the Dropout + PReLU backward of the layer below applied to the stored gradient (gfeat is then that layer's gradient; the
// slope sum leaves through one atomic per block -- not in deterministic mode)
struct HeadsPostAct {
  const float* pre = nullptr;    // [R][nf] the layer's pre-activation values
  const float* mask = nullptr;   // [R][nf] keep mask or null
  float inv_keep = 1.f;
  const float* slope = nullptr;  // device scalar
  float* gslope = nullptr;       // device scalar, accumulated
};
int cnet_heads_backward(const float* g_bbox, const float* g_cls, const float* lsm, int R, int nf, const float* Wb, const float* Wc,
                        int nc, float* glog, float* gfeat, hipStream_t s, const HeadsPostAct* post = nullptr);
// nn.LogSoftMax of a (possibly deferred) product, written to one or two destinations
int log_softmax_rows_fold(const float* x, GemmFold src, int R, int n, float* y, float* y2, hipStream_t s);
int dropout_mask(float* mask, long n, float p, unsigned long long seed, hipStream_t s);
struct DropoutJobs { float* ptr[8]; int C[8]; float p[8]; unsigned long long seed[8]; int n; };
int dropout_channel_masks(const DropoutJobs& j, hipStream_t s);
int log_softmax_rows(const float* x, int R, int n, float* y, hipStream_t s);
// losses of objective.lua:170-177 + their gradients, fused
int cnet_losses(float* crout, const float* crtarget, const float* ccout, const float* cctarget, int R,
                int npos, int ncls, float* crdelta, float* ccdelta, double* loss2, hipStream_t s);
int log_softmax_backward(const float* gy, const float* lsm, int R, int n, float* gx, hipStream_t s);
int cnet_decode(const float* cls_lsm, int R, int ncls, int* cls_out, float* conf_out, hipStream_t s);

// ---------------------------------------------------------------- image (image.hip): BatchIterator:processImage
int image_rgb2yuv(const float* rgb, float* yuv, int H, int W, hipStream_t s);
int image_rgb2hsv(const float* rgb, float* hsv, int H, int W, hipStream_t s);
int image_rgb2lab(const float* rgb, float* lab, int H, int W, hipStream_t s);
// image.scale 'bilinear': src [C][H][W] -> dst [C][dH][dW]; tmp holds C*H*dW floats
int image_scale(const float* src, int C, int H, int W, float* dst, int dH, int dW, float* tmp, int rgb2yuv,
                hipStream_t s);
// the same for the decoder's 8-bit interleaved RGB frame [H][W][3] (converted with v/255, optionally to YUV, on the fly)
int image_scale_u8(const unsigned char* src_hwc, int H, int W, float* dst, int dH, int dW, float* tmp, int rgb2yuv,
                   hipStream_t s);
// dst [C][h][w] = flips(crop(src, x0, y0, w, h))
int image_crop_flip(const float* src, int C, int H, int W, int x0, int y0, int w, int h, int hflip, int vflip,
                    float* dst, hipStream_t s);
size_t image_normalize_workspace_bytes(int C);
int image_normalize(float* img, int C, int H, int W, int centering, int scaling, void* ws, size_t ws_bytes,
                    hipStream_t s);
int image_contrastive_norm(const float* in, int H, int W, const float* kernel_host, int K, float threshold, float* out,
                           float* tmp, hipStream_t s);

}  // namespace frcnn
