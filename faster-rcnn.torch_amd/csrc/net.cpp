// net.cpp -- native runtime of the two networks built by models/model_utilities.lua:
//   create_proposal_net (:3-74)       -> frcnn_pnet_forward / frcnn_pnet_backward
//   create_classification_net (:76-124) -> frcnn_cnet_forward / frcnn_cnet_backward
// The Lua modules own their output/gradInput buffers and reuse them on the next call; so does
// this runtime (all activations, gradient buffers and packed weights live in HBM for the life of
// the model and are re-used every step; nothing is allocated inside a step once shapes settle).
//
// What is stored per convolution is only its PRE-activation output x.  nn.PReLU and
// nn.SpatialDropout are applied by whoever consumes x (the next conv's LDS loader, the pool, the
// weight-gradient loader), and their backward is fused with the pooling backward / bias-gradient
// reduction, so each layer costs one HBM write in forward and one in backward.
//
// Flat parameter order (utilities.lua:136-147): see frcnn_model_param_table.
#include <array>
#include <vector>
#include <cmath>
#include <cstring>

#include "kernels.h"

namespace frcnn {

static const int HEAD_OUT = 18;  // 3 * (2 + 4), model_utilities.lua:33
static const int SPARSE_MAX_POS = 512;  // above this the dense head backward is used

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;
  void view(void* ptr, size_t n) { p = ptr; bytes = n; owned = false; }  // slice of an arena
  int ensure(size_t need) {
    if (need <= bytes) return FRCNN_OK;
    if (p && owned) (void)hipFree(p);
    owned = true;
    p = nullptr; bytes = 0;
    FR_HIP(hipMalloc(&p, need + 64));   // (64 bytes of slack: conv_wgradx's unaligned 16-byte segment loads may read 12 bytes past a tensor)
    bytes = need;
    return FRCNN_OK;
  }
  void release() { if (p && owned) (void)hipFree(p); p = nullptr; bytes = 0; owned = true; }
  float* f() const { return (float*)p; }
};

struct Conv {
  int Cin, Cout, k, pad;
  int H, W, Ho, Wo;           // input / output spatial size (set by ensure_shapes)
  long w_off, b_off, a_off;   // flat offsets; a_off = PReLU slope following this conv (-1: none)
  int block, step;            // backbone position (block -1 for head convs)
  DevBuf wf, wd;              // packed weights (forward / input-gradient)
  DevBuf wx, wxd;             // split-bf16 operand stages (convx.hip) for the launches that take that form
  bool x_f = false, x_d = false;
  DevBuf x, gx;               // pre-activation output and its gradient
  int am = -1;                // first of this convolution's three magnitude records in frcnn_model::amax: |x| (its output), |gx|, |w|
};

struct Block {
  int first_conv, nconv;
  bool has_drop;
  float p_drop;
  DevBuf scale;               // per-channel SpatialDropout scale (mask | 1-p)
  DevBuf pooled, gpooled;
  DevBuf pidx;
  int Hp, Wp;
  int am = -1;                // magnitude record of `pooled`
  // Channels a SpatialDropout keeps (round 6, option "drop_compact"): a block of >= 2 convolutions whose dropout sits behind its
  // first one (models/model_utilities.lua:9-12) is, for the length of one training step, a NARROWER network -- the dropped
  // channels of the first convolution's output are multiplied by zero on the way forward and on the way back, so neither they
  // nor anything computed from them is needed.  With the keep vector known on the host, the step runs the first convolution
  // with the kept filters only (its output, and that output's gradient, are stored COMPACT: kept channels first) and the second
  // one with the kept input channels only; the packs gather the filters / channels, the weight gradients are scattered back.
  // Same results as multiplying by the zeros (sums of exact zeros left out), 25-37 % less matrix work in these layers.
  bool dc_on = false;         // this step runs the block compact
  int nk = 0, nkM = 0, nkK = 0;   // channels kept; padded to a multiple of 64 (a filter / tile dimension) and of 16 (a K dimension)
  const int* dc_idx = nullptr;    // device table of this step: entry i < nk = the i-th kept channel, nk <= i < C: -1
  DevBuf dc_bias;             // the first convolution's bias, gathered (nkM entries, padding 0)
};

struct Head {
  Conv c3;                    // k x k valid conv + PReLU
  Conv c1;                    // 1 x 1 conv -> 18 planes
  int input;                  // 0-based block index
  DevBuf delta;               // delta_outputs[h]
  const int* sp_pos = nullptr; // optional one-shot hint: delta is zero outside these positions
  int sp_count = -1;
  DevBuf spD, spHX, spHY, spGH, spCol, spDX;  // scratch of the sparse backward pass (per anchor net: they run concurrently)
  DevBuf spOut, spSlab;       // sparse training path (heads.hip): the 18 output planes at the sampled positions; K-split partial sums of HX
  hipStream_t stream = nullptr;  // this anchor net's own stream (forward and sparse backward beside the other anchor nets)
  hipEvent_t done = nullptr;     // last work queued on `stream`
  hipEvent_t gin_done = nullptr; // sparse backward: this net's contribution to the pooled map's gradient has been added (the
                                 // backbone's backward pass waits for THIS; the parameter gradients behind it are joined at its end)
  bool gin_recorded = false;
};

struct ClsLayer {
  int in, n;
  bool bn;
  float p_drop;
  long w_off, b_off, bnw_off, bnb_off, a_off;
  DevBuf lin, pre, post, xhat, invstd, mask, g;
  DevBuf gin;                 // gradient wrt this layer's input (layers > 0; a buffer of its own: the weight gradient reads the input beside it)
  DevBuf xp, xpT, gp, gpT;    // split-bf16 planes of the layer's input and of its output gradient (gemmx.hip), when it takes that form
  int x_form = 0;             // set per call: which of this layer's products run in the split-bf16 operand form
                              // (bit 1 forward, 2 input gradient, 4 weight gradient: linear_x_eligible)
  DevBuf am;                  // two-plane fp16 form of the forward / input-gradient products: magnitude records (amax.h) of the
                              // layer's input, of its output gradient and of its weight matrix, AMAX_REC floats each
  const float* am_w_of = nullptr;   // the weight vector the third record was taken from ...
  long am_w_gen = -1;               // ... in an evaluate-mode pass of this static_weights generation (-1: a training pass)
};

}  // namespace frcnn

using namespace frcnn;

static int g_drop_compact = -1;   // option "drop_compact" (environment FRCNN_DROP_COMPACT), default on
static int drop_compact_on() {
  if (g_drop_compact < 0) g_drop_compact = getenv("FRCNN_DROP_COMPACT") ? (atoi(getenv("FRCNN_DROP_COMPACT")) != 0) : 1;
  return g_drop_compact;
}
// The library's streams are the PROCESS's, not a model's: the runtime multiplexes every stream onto four hardware queues
// (GPU_MAX_HW_QUEUES; five or more run this step 1.6x slower, tools/r6_hwq.sh), two streams on one queue serialise, and a second
// model with six streams of its own (bench.py's legs, a training and an evaluation model side by side) made both slower.
// Slot 0: the side stream (weight gradients), 1: the classification net's weight gradients / the update stream, 2..6: anchor nets.
static int pool_stream(int slot, hipStream_t* out) {
  static hipStream_t pool[8] = {};
  if (!pool[slot]) FR_HIP(hipStreamCreateWithFlags(&pool[slot], hipStreamNonBlocking));
  *out = pool[slot];
  return FRCNN_OK;
}

// ---- the anchor nets' sparse training path (heads.hip) ---------------------------------------------------------------------
static int g_sparse_heads = -1;   // option "sparse_heads" (environment FRCNN_SPARSE_HEADS), default on
static int sparse_heads_on() {
  if (g_sparse_heads < 0) g_sparse_heads = getenv("FRCNN_SPARSE_HEADS") ? (atoi(getenv("FRCNN_SPARSE_HEADS")) != 0) : 1;
  return g_sparse_heads;
}
static int g_static_weights = 0;   // option "static_weights" (see forward_impl)
static long g_static_gen = 0;      // bumped by every set_option("static_weights", v)

struct frcnn_model {
  frcnn_model_desc d;
  std::vector<Conv> convs;     // backbone convs in order
  std::vector<Block> blocks;
  std::vector<Head> heads;
  std::vector<ClsLayer> cls;
  long pnet_params = 0, total_params = 0;
  long bbox_w_off = 0, bbox_b_off = 0, clsw_off = 0, clsb_off = 0;
  int H = 0, W = 0;            // current image size
  int training = 0;
  DevBuf delta_last;           // delta_outputs[nheads+1]
  DevBuf zero_arena;           // delta_outputs[1..n+1] followed by the pooled-map gradients: zeroed by ONE memset each
  size_t delta_bytes = 0, gpool_bytes = 0;
  DevBuf pack_jobs;            // device table of PackJob (fwd packs first, then dgrad packs)
  DevBuf x3_jobs;              // device table of PackXJob, same arrangement
  // Two-plane fp16 form of the split launches (option x3_f16): every tensor such a launch reads is scaled by a power of two that
  // puts its largest magnitude just below the end of the fp16 range, so the magnitudes travel with the tensors -- one record
  // (amax.h: a maximum per block of the launch that wrote the tensor) per convolution output, output gradient, weight tensor and
  // pooled map; amax_ws[c.am] = the weight tensor's largest magnitude as a scalar (published by the pack).
  DevBuf amax, amax_ws, amax_jobs;   // the records; the weight scalars; AmaxJob table of the weight tensors
  int n_amax = 0, n_amax_jobs = 0, amax_grid = 0;
  float* rec(int id) const { return (float*)amax.p + (size_t)id * AMAX_REC; }
  bool f16_packed = false;     // the current packs are in the fp16 form
  long eval_packs_gen = -1;    // option static_weights: generation (g_static_gen) the evaluate-mode packs were made in, -1 = none
  const float* eval_packs_w = nullptr;
  int n_x3_all = 0, n_x3_fwd = 0, x3_grid_all = 0, x3_grid_fwd = 0;
  int n_pack_fwd = 0, n_pack_all = 0, n_pack_heads = 0, pack_grid_fwd = 0, pack_grid_all = 0, pack_grid_heads = 0;
  bool head_packs_fresh = false;   // head input-gradient packs match the weights of the last forward
  // Packs by owner (round 6, the update that runs beside the backward pass): group b < nblocks = backbone block b, group nblocks =
  // the anchor nets.  Each group has sub-tables of its own behind the model-wide ones in pack_jobs / amax_jobs / x3_jobs, so that
  // frcnn_pnet_refresh_packs can renew one owner's packs as soon as ITS slice of the weight vector has been updated; a training
  // forward whose groups are all fresh for the weight vector it is given skips the three model-wide launches.
  struct PackGroup { int pk_off = 0, pk_n = 0, pk_grid = 0, am_off = 0, am_n = 0, am_grid = 0, x3_off = 0, x3_off16 = 0, x3_n = 0, x3_grid = 0; };
  std::vector<PackGroup> groups;
  unsigned fresh_mask = 0;         // bit g: group g's training packs were renewed by frcnn_pnet_refresh_packs since the last forward
  const float* fresh_w = nullptr;  // ... from this weight vector
  bool fresh_f16 = false;          // ... in this form
  // per-step tables of the compact blocks (see Block::dc_on): a ring of page-locked host slots and device slots, one
  // asynchronous copy per step: [PackXJob table of the step][kept-channel tables of the blocks]
  static const int DC_RING = 4;
  char* dc_pin = nullptr; DevBuf dc_dev; size_t dc_slot_bytes = 0, dc_idx_off = 0; unsigned dc_step = 0;
  hipEvent_t dc_ev[DC_RING] = {};             // slot s's copy to the device has run (a host that queues more than DC_RING passes ahead waits here)
  std::vector<PackXJob> x3_host, x3_host16;   // the model's training pack jobs as built by ensure_shapes (plain | fp16 form), host copies
  std::vector<int> x3_conv;                   // ... the convolution each job belongs to (index into convs, -1: an anchor net)
  DevBuf dbg_expand;                          // frcnn_model_debug_buffer: a compact tensor laid out dense
  std::vector<hipEvent_t> block_rd_ev;   // block b's weights and packs have been read for the last time (caller's stream, frcnn_pnet_backward)
  hipStream_t side = nullptr;      // accGradParameters stream (runs beside the updateGradInput chain)
  hipStream_t cw = nullptr;        // the classification net's weight gradients / bias sums (beside its input-gradient chain)
  hipEvent_t cw_fork = nullptr, cw_done = nullptr;
  bool cw_pending = false;         // work on `cw` that no stream has been made to wait for yet
  std::vector<hipEvent_t> cw_ev;   // one fork point per layer + one for the two heads
  std::vector<hipEvent_t> fork_ev;
  hipEvent_t join_ev = nullptr;
  bool update_armed = false;       // a host has asked for the update stream (frcnn_model_update_stream): the passes record the
                                   // events its waits need (bwd_ev, block_rd_ev) -- each costs the caller's stream a marker packet
  hipEvent_t bwd_ev = nullptr;     // the backbone's backward pass has begun on the caller's stream (anchor nets joined)
  hipEvent_t upd_ev = nullptr, upd_join_ev = nullptr;   // fork / join points of the update stream (frcnn_model_update_*)
  bool head_x3_fresh = true;       // the anchor nets' split-operand packs and weight magnitudes match the last forward pass's weights
  int am_bb_off = 0, am_bb_n = 0, am_bb_grid = 0;   // AmaxJob sub-table: the backbone's weight tensors only
  int pk_bb_off = 0, pk_bb_n = 0, pk_bb_grid = 0;   // PackJob sub-table: the backbone's training packs only
  bool heads_deferred = false;     // training pass: the anchor nets' forward part has not been launched yet (frcnn_pnet_forward_async_heads
                                   // leaves it to the call that knows the sampled positions: heads.hip)
  bool heads_sparse_fwd = false;   // ... and was then computed at the sampled positions only
  const float* last_w = nullptr;   // weight vector of the last forward pass
  hipEvent_t heads_gin_ev = nullptr;   // sparse path: the anchor nets' contributions to the pooled maps' gradients have been added (side stream)
  bool heads_gin = false;
  bool heads_begun = false;        // anchor-net backward already running on the side stream
  bool heads_joined = false;       // ... and the caller's stream already waits for it
  bool side_busy = false;          // work was forked to the side stream and not joined yet
  hipEvent_t loss_ev = nullptr;    // anchor losses of frcnn_pnet_anchor_loss_begin are final (side stream)
  hipEvent_t chain_ev = nullptr;   // ... and the pooled-map gradient buffers are zeroed: the anchor nets' backward may start
  std::vector<hipEvent_t> block_ev;   // block b's parameter gradients are final (recorded by frcnn_pnet_backward)
  bool block_ev_valid = false;
  bool loss_pending = false;
  DevBuf wg_ws;                // split-K slab workspace of the weight-gradient kernels
  DevBuf wg_ws_first;          // ... of the first layer's, which runs on the caller's stream beside the side stream's last launches
  DevBuf img;                  // copy of the input image (needed by the first conv's accGradParameters)
  // cnet state
  int R = 0, D = 0;
  const float* cnet_x = nullptr;
  DevBuf feat_g, logits, lsm, glog, gtmp;
  std::vector<std::array<long long, 4>> table;
};

static int pool_out(int n) { return (n - 2 + 1) / 2 + 1; }
static int refresh_group(frcnn_model* m, const float* w, int group, hipStream_t s);

static void build_layout(frcnn_model* m) {
  const frcnn_model_desc& d = m->d;
  long off = 0;
  int cin = 3;
  for (int b = 0; b < d.nblocks; ++b) {
    Block blk;
    blk.first_conv = (int)m->convs.size();
    blk.nconv = d.conv_steps[b];
    blk.has_drop = d.dropout[b] > 0.f;  // model_utilities.lua:10
    blk.p_drop = d.dropout[b];
    for (int s = 0; s < d.conv_steps[b]; ++s) {
      Conv c;
      c.Cin = cin; c.Cout = d.filters[b]; c.k = d.ksize[b]; c.pad = d.pad[b];
      c.block = b; c.step = s;
      long wsz = (long)c.Cout * c.Cin * c.k * c.k;
      c.w_off = off; c.b_off = off + wsz; c.a_off = off + wsz + c.Cout;
      m->table.push_back({c.w_off, wsz, 0, (long long)c.k * c.k * c.Cout});
      m->table.push_back({c.b_off, c.Cout, 1, 0});
      m->table.push_back({c.a_off, 1, 2, 0});
      off += wsz + c.Cout + 1;
      cin = c.Cout;
      m->convs.push_back(std::move(c));
    }
    m->blocks.push_back(std::move(blk));
  }
  for (int h = 0; h < d.nheads; ++h) {
    Head hd;
    hd.input = d.head_input[h] - 1;
    Conv& a = hd.c3;
    a.Cin = d.filters[hd.input]; a.Cout = d.head_n[h]; a.k = d.head_k[h]; a.pad = 0;
    a.block = -1; a.step = 0;
    long wsz = (long)a.Cout * a.Cin * a.k * a.k;
    a.w_off = off; a.b_off = off + wsz; a.a_off = off + wsz + a.Cout;
    m->table.push_back({a.w_off, wsz, 0, (long long)a.k * a.k * a.Cout});
    m->table.push_back({a.b_off, a.Cout, 1, 0});
    m->table.push_back({a.a_off, 1, 2, 0});
    off += wsz + a.Cout + 1;
    Conv& c = hd.c1;
    c.Cin = a.Cout; c.Cout = HEAD_OUT; c.k = 1; c.pad = 0; c.block = -1; c.step = 1;
    long w1 = (long)HEAD_OUT * c.Cin;
    c.w_off = off; c.b_off = off + w1; c.a_off = -1;
    m->table.push_back({c.w_off, w1, 0, (long long)HEAD_OUT});
    m->table.push_back({c.b_off, HEAD_OUT, 1, 0});
    off += w1 + HEAD_OUT;
    m->heads.push_back(std::move(hd));
  }
  m->pnet_params = off;
  int in = d.kh * d.kw * d.filters[d.nblocks - 1];  // model_utilities.lua:127
  m->D = in;
  for (int l = 0; l < d.ncls; ++l) {
    ClsLayer L;
    L.in = in; L.n = d.cls_n[l]; L.bn = d.cls_bn[l] != 0; L.p_drop = d.cls_dropout[l];
    L.w_off = off; L.b_off = off + (long)in * L.n;
    m->table.push_back({L.w_off, (long long)in * L.n, 3, in});
    m->table.push_back({L.b_off, L.n, 4, in});
    off += (long)in * L.n + L.n;
    L.bnw_off = L.bnb_off = -1;
    if (L.bn) {
      L.bnw_off = off; L.bnb_off = off + L.n;
      m->table.push_back({L.bnw_off, L.n, 5, 0});
      m->table.push_back({L.bnb_off, L.n, 6, 0});
      off += 2L * L.n;
    }
    L.a_off = off;
    m->table.push_back({L.a_off, 1, 2, 0});
    off += 1;
    in = L.n;
    m->cls.push_back(std::move(L));
  }
  m->bbox_w_off = off; m->bbox_b_off = off + (long)in * 4;
  m->table.push_back({m->bbox_w_off, (long long)in * 4, 3, in});
  m->table.push_back({m->bbox_b_off, 4, 4, in});
  off += (long)in * 4 + 4;
  int nc = d.class_count + 1;
  m->clsw_off = off; m->clsb_off = off + (long)in * nc;
  m->table.push_back({m->clsw_off, (long long)in * nc, 3, in});
  m->table.push_back({m->clsb_off, nc, 4, in});
  off += (long)in * nc + nc;
  m->total_params = off;
}

static int ensure_conv(Conv& c, int H, int W, bool need_dgrad) {
  c.H = H; c.W = W;
  c.Ho = H + 2 * c.pad - c.k + 1; c.Wo = W + 2 * c.pad - c.k + 1;
  FR_CHECK(c.Ho > 0 && c.Wo > 0, "image too small: a %dx%d map reaches a %dx%d convolution", H, W, c.k, c.k);
  size_t n = (size_t)c.Cout * c.Ho * c.Wo * 4;
  FR_TRY(c.x.ensure(n));
  FR_TRY(c.gx.ensure(n));
  // packed weights: the padding (channels beyond C, columns beyond M) is zeroed once here and never written again
  if (c.wf.bytes < conv_pack_floats(c.Cin, c.Cout, c.k) * 4) {
    FR_TRY(c.wf.ensure(conv_pack_floats(c.Cin, c.Cout, c.k) * 4));
    FR_HIP(hipMemset(c.wf.p, 0, c.wf.bytes));
  }
  if (need_dgrad && c.wd.bytes < conv_pack_floats(c.Cout, c.Cin, c.k) * 4) {
    FR_TRY(c.wd.ensure(conv_pack_floats(c.Cout, c.Cin, c.k) * 4));
    FR_HIP(hipMemset(c.wd.p, 0, c.wd.bytes));
  }
  // Launches whose shape fits take the split-bf16 operand form (convx.hip): fp32 results at 6/16 of the matrix-pipe time.  The 3x3
  // layers since round 2; the 5x5 / 7x7 anchor nets since round 5: rounds 2 and 3 had measured them slower in that form (8x10-pixel
  // tiles under the 204-position patch limit; then 8x16 tiles but a stage loop that needed three resident blocks to hide its LDS
  // round trips: 3.14 against 3.11 ms/step).  With the stage pipelined inside the wave (convx.hip) two blocks per CU suffice: the 7x7
  // net 118.8 -> 63.8 us alone, the 5x5 net 67.4 -> 56.8, the step 2.918 -> 2.834 ms.  FRCNN_X3_ANCHOR_K=3 restores the fp32 kernels.
  static const int x3_anchor_k = getenv("FRCNN_X3_ANCHOR_K") ? atoi(getenv("FRCNN_X3_ANCHOR_K")) : 7;
  c.x_f = (c.k == 3 || (c.block < 0 && c.k <= x3_anchor_k)) && conv_x3_eligible(c.Cin, c.Cout, c.k);
  c.x_d = c.block >= 0 && need_dgrad && conv_x3_eligible(c.Cout, c.Cin, c.k);
  if (c.x_f) FR_TRY(c.wx.ensure(conv_x3_pack_bytes(c.Cin, c.Cout, c.k)));
  if (c.x_d) FR_TRY(c.wxd.ensure(conv_x3_pack_bytes(c.Cout, c.Cin, c.k)));
  return FRCNN_OK;
}

static int ensure_shapes(frcnn_model* m, int H, int W) {
  int h = H, w = W;
  for (size_t b = 0; b < m->blocks.size(); ++b) {
    Block& blk = m->blocks[b];
    for (int s = 0; s < blk.nconv; ++s) {
      Conv& c = m->convs[blk.first_conv + s];
      FR_TRY(ensure_conv(c, h, w, !(b == 0 && s == 0)));
      h = c.Ho; w = c.Wo;
    }
    FR_CHECK(h >= 2 && w >= 2, "image too small for block %zu pooling", b + 1);
    blk.Hp = pool_out(h); blk.Wp = pool_out(w);
    int C = m->d.filters[b];
    FR_TRY(blk.pooled.ensure((size_t)C * blk.Hp * blk.Wp * 4));
    FR_TRY(blk.pidx.ensure((size_t)C * blk.Hp * blk.Wp));
    if (blk.has_drop) FR_TRY(blk.scale.ensure((size_t)C * 4));
    h = blk.Hp; w = blk.Wp;
  }
  for (auto& hd : m->heads) {
    const Block& in = m->blocks[hd.input];
    FR_TRY(ensure_conv(hd.c3, in.Hp, in.Wp, true));
    FR_TRY(ensure_conv(hd.c1, hd.c3.Ho, hd.c3.Wo, true));
  }
  const Block& last = m->blocks.back();
  {  // one arena: [deltas of the heads | delta of the last map | gpooled of every block]
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t db = 0, gb = 0;
    for (auto& hd : m->heads) db += al((size_t)HEAD_OUT * hd.c1.Ho * hd.c1.Wo * 4);
    db += al((size_t)m->d.filters[m->d.nblocks - 1] * last.Hp * last.Wp * 4);
    for (size_t b = 0; b < m->blocks.size(); ++b) gb += al((size_t)m->d.filters[b] * m->blocks[b].Hp * m->blocks[b].Wp * 4);
    m->zero_arena.release();
    FR_TRY(m->zero_arena.ensure(db + gb));
    char* q = (char*)m->zero_arena.p;
    for (auto& hd : m->heads) { size_t n = (size_t)HEAD_OUT * hd.c1.Ho * hd.c1.Wo * 4; hd.delta.view(q, n); q += al(n); }
    { size_t n = (size_t)m->d.filters[m->d.nblocks - 1] * last.Hp * last.Wp * 4; m->delta_last.view(q, n); q += al(n); }
    for (size_t b = 0; b < m->blocks.size(); ++b) {
      size_t n = (size_t)m->d.filters[b] * m->blocks[b].Hp * m->blocks[b].Wp * 4;
      m->blocks[b].gpooled.view(q, n); q += al(n);
    }
    m->delta_bytes = db; m->gpool_bytes = gb;
  }
  size_t wsb = 0;
  for (auto& c : m->convs) wsb = std::max(wsb, conv_wgrad_workspace_bytes(c.Cin, c.H, c.W, c.Cout, c.k, c.pad));
  for (auto& hd : m->heads) {
    wsb = std::max(wsb, conv_wgrad_workspace_bytes(hd.c3.Cin, hd.c3.H, hd.c3.W, hd.c3.Cout, hd.c3.k, 0));
    wsb = std::max(wsb, conv_wgrad_workspace_bytes(hd.c1.Cin, hd.c1.H, hd.c1.W, hd.c1.Cout, 1, 0));
  }
  FR_TRY(m->wg_ws.ensure(wsb));
  { const Conv& c0 = m->convs[0]; FR_TRY(m->wg_ws_first.ensure(conv_wgrad_workspace_bytes(c0.Cin, c0.H, c0.W, c0.Cout, c0.k, c0.pad))); }
  {  // pack-job table (buffers may have been re-allocated above)
    std::vector<PackJob> jobs;
    for (auto& c : m->convs)
      if (!c.x_f) jobs.push_back(conv_pack_job(c.w_off, c.Cout, c.Cin, c.k, 0, c.wf.f()));
    for (auto& hd : m->heads) {
      if (!hd.c3.x_f) jobs.push_back(conv_pack_job(hd.c3.w_off, hd.c3.Cout, hd.c3.Cin, hd.c3.k, 0, hd.c3.wf.f()));
      jobs.push_back(conv_pack_job(hd.c1.w_off, hd.c1.Cout, hd.c1.Cin, hd.c1.k, 0, hd.c1.wf.f()));
    }
    m->n_pack_fwd = (int)jobs.size();
    for (auto& c : m->convs)
      if (!(c.block == 0 && c.step == 0) && !c.x_d) jobs.push_back(conv_pack_job(c.w_off, c.Cout, c.Cin, c.k, 1, c.wd.f()));
    m->n_pack_all = (int)jobs.size();
    // the heads' input-gradient packs are only needed by the dense head backward (more than SPARSE_MAX_POS
    // examples on a head); they are refreshed lazily by frcnn_pnet_backward
    std::vector<PackJob> hd_jobs;
    for (auto& hd : m->heads) {
      hd_jobs.push_back(conv_pack_job(hd.c3.w_off, hd.c3.Cout, hd.c3.Cin, hd.c3.k, 1, hd.c3.wd.f()));
      hd_jobs.push_back(conv_pack_job(hd.c1.w_off, hd.c1.Cout, hd.c1.Cin, hd.c1.k, 1, hd.c1.wd.f()));
    }
    m->n_pack_heads = (int)hd_jobs.size();
    // three tables: [training jobs, blocks dealt over all] [forward jobs only] [head input-gradient jobs]
    std::vector<PackJob> both = jobs;
    m->pack_grid_all = conv_pack_assign_blocks(both.data(), m->n_pack_all, 2048);
    std::vector<PackJob> fwd(jobs.begin(), jobs.begin() + m->n_pack_fwd);
    m->pack_grid_fwd = conv_pack_assign_blocks(fwd.data(), m->n_pack_fwd, 2048);
    m->pack_grid_heads = hd_jobs.empty() ? 0 : conv_pack_assign_blocks(hd_jobs.data(), m->n_pack_heads, 2048);
    both.insert(both.end(), fwd.begin(), fwd.end());
    both.insert(both.end(), hd_jobs.begin(), hd_jobs.end());
    // per-owner sub-tables (training packs: forward and input-gradient images of the owner's convolutions)
    m->groups.assign(m->blocks.size() + 1, frcnn_model::PackGroup());
    for (size_t g = 0; g <= m->blocks.size(); ++g) {
      std::vector<PackJob> gj;
      if (g < m->blocks.size()) {
        for (auto& c : m->convs) {
          if (c.block != (int)g) continue;
          if (!c.x_f) gj.push_back(conv_pack_job(c.w_off, c.Cout, c.Cin, c.k, 0, c.wf.f()));
          if (!(c.block == 0 && c.step == 0) && !c.x_d) gj.push_back(conv_pack_job(c.w_off, c.Cout, c.Cin, c.k, 1, c.wd.f()));
        }
      } else {
        for (auto& hd : m->heads) {
          if (!hd.c3.x_f) gj.push_back(conv_pack_job(hd.c3.w_off, hd.c3.Cout, hd.c3.Cin, hd.c3.k, 0, hd.c3.wf.f()));
          gj.push_back(conv_pack_job(hd.c1.w_off, hd.c1.Cout, hd.c1.Cin, hd.c1.k, 0, hd.c1.wf.f()));
        }
      }
      auto& G = m->groups[g];
      G.pk_off = (int)both.size(); G.pk_n = (int)gj.size();
      G.pk_grid = gj.empty() ? 0 : conv_pack_assign_blocks(gj.data(), G.pk_n, 512);
      both.insert(both.end(), gj.begin(), gj.end());
    }
    {   // the backbone's training packs only (a pass that runs the anchor nets sparse: frcnn_model::head_x3_fresh)
      std::vector<PackJob> gj;
      for (auto& c : m->convs) {
        if (!c.x_f) gj.push_back(conv_pack_job(c.w_off, c.Cout, c.Cin, c.k, 0, c.wf.f()));
        if (!(c.block == 0 && c.step == 0) && !c.x_d) gj.push_back(conv_pack_job(c.w_off, c.Cout, c.Cin, c.k, 1, c.wd.f()));
      }
      m->pk_bb_off = (int)both.size(); m->pk_bb_n = (int)gj.size();
      m->pk_bb_grid = gj.empty() ? 0 : conv_pack_assign_blocks(gj.data(), m->pk_bb_n, 512);
      both.insert(both.end(), gj.begin(), gj.end());
    }
    FR_TRY(m->pack_jobs.ensure(both.size() * sizeof(PackJob)));
    FR_HIP(hipMemcpy(m->pack_jobs.p, both.data(), both.size() * sizeof(PackJob), hipMemcpyHostToDevice));
  }
  {  // split-bf16 pack jobs
    // magnitude scalars of the two-plane fp16 form (see frcnn_model::amax) and the table of the weight tensors' segments
    {
      int n = 0;
      for (auto& c : m->convs) { c.am = n; n += 3; }
      for (auto& hd : m->heads) { hd.c3.am = n; n += 3; }
      for (auto& b : m->blocks) b.am = n++;
      m->n_amax = n;
      FR_TRY(m->amax.ensure((size_t)n * AMAX_REC * 4));
      FR_TRY(m->amax_ws.ensure((size_t)n * 4));
      std::vector<AmaxJob> aj;
      auto addw = [&](Conv& c) {
        if (c.x_f || c.x_d) aj.push_back(AmaxJob{c.w_off, (long)c.Cout * c.Cin * c.k * c.k, m->rec(c.am + 2), 0});
      };
      for (auto& c : m->convs) addw(c);
      for (auto& hd : m->heads) addw(hd.c3);
      m->n_amax_jobs = (int)aj.size();
      m->amax_grid = tensor_absmax_assign_blocks(aj.data(), m->n_amax_jobs);
      FR_CHECK(m->amax_grid >= 0, "ensure_shapes: a weight tensor is too large for its magnitude record");
      {   // the backbone's tensors only (a training pass that leaves the anchor nets to the sparse path, heads.hip)
        std::vector<AmaxJob> gj;
        for (auto& c : m->convs)
          if (c.x_f || c.x_d) gj.push_back(AmaxJob{c.w_off, (long)c.Cout * c.Cin * c.k * c.k, m->rec(c.am + 2), 0});
        m->am_bb_off = (int)aj.size(); m->am_bb_n = (int)gj.size();
        m->am_bb_grid = gj.empty() ? 0 : tensor_absmax_assign_blocks(gj.data(), m->am_bb_n);
        aj.insert(aj.end(), gj.begin(), gj.end());
      }
      for (size_t g = 0; g <= m->blocks.size(); ++g) {   // per-owner sub-tables (see frcnn_model::groups)
        std::vector<AmaxJob> gj;
        auto addg = [&](Conv& c) {
          if (c.x_f || c.x_d) gj.push_back(AmaxJob{c.w_off, (long)c.Cout * c.Cin * c.k * c.k, m->rec(c.am + 2), 0});
        };
        if (g < m->blocks.size()) { for (auto& c : m->convs) if (c.block == (int)g) addg(c); }
        else for (auto& hd : m->heads) addg(hd.c3);
        auto& G = m->groups[g];
        G.am_off = (int)aj.size(); G.am_n = (int)gj.size();
        G.am_grid = gj.empty() ? 0 : tensor_absmax_assign_blocks(gj.data(), G.am_n);
        FR_CHECK(G.am_grid >= 0, "ensure_shapes: a weight tensor is too large for its magnitude record");
        aj.insert(aj.end(), gj.begin(), gj.end());
      }
      if (!aj.empty()) {
        FR_TRY(m->amax_jobs.ensure(aj.size() * sizeof(AmaxJob)));
        FR_HIP(hipMemcpy(m->amax_jobs.p, aj.data(), aj.size() * sizeof(AmaxJob), hipMemcpyHostToDevice));
      }
    }
    // four tables: [training jobs] [forward jobs], then the same two with the weight magnitude attached (fp16 form)
    std::vector<PackXJob> all, fwd, all16, fwd16;
    auto add = [&](Conv& c) {
      if (c.x_f) {
        all.push_back(conv_x3_pack_job(c.w_off, c.Cout, c.Cin, c.k, 0, c.wx.p, c.Ho, c.Wo)); fwd.push_back(all.back());
        PackXJob j = all.back(); j.amax = m->rec(c.am + 2); j.amax_w = m->amax_ws.f() + c.am;
        all16.push_back(j); fwd16.push_back(j);
      }
      if (c.x_d) {
        all.push_back(conv_x3_pack_job(c.w_off, c.Cout, c.Cin, c.k, 1, c.wxd.p, c.H, c.W));
        PackXJob j = all.back(); j.amax = m->rec(c.am + 2); j.amax_w = m->amax_ws.f() + c.am;
        all16.push_back(j);
      }
    };
    m->x3_conv.clear();
    for (size_t ci = 0; ci < m->convs.size(); ++ci) {
      const size_t n0 = all.size();
      add(m->convs[ci]);
      m->x3_conv.insert(m->x3_conv.end(), all.size() - n0, (int)ci);
    }
    for (auto& hd : m->heads) { const size_t n0 = all.size(); add(hd.c3); m->x3_conv.insert(m->x3_conv.end(), all.size() - n0, -1); }
    m->x3_host = all; m->x3_host16 = all16;   // (host copies for the per-step tables of the compact blocks)
    {  // ring of per-step tables: [PackXJob x jobs][int x channels of every block], see frcnn_model::dc_pin
      size_t chans = 0;
      for (size_t b = 0; b < m->blocks.size(); ++b) chans += (size_t)m->d.filters[b];
      m->dc_idx_off = (all.size() * sizeof(PackXJob) + 255) / 256 * 256;
      const size_t slot = (m->dc_idx_off + chans * 4 + 255) / 256 * 256;
      if (slot != m->dc_slot_bytes) {
        if (m->dc_pin) (void)hipHostFree(m->dc_pin);
  for (auto e : m->dc_ev) if (e) (void)hipEventDestroy(e);
        m->dc_pin = nullptr;
        FR_HIP(hipHostMalloc((void**)&m->dc_pin, slot * frcnn_model::DC_RING, hipHostMallocDefault));
        m->dc_dev.release();
        FR_TRY(m->dc_dev.ensure(slot * frcnn_model::DC_RING));
        m->dc_slot_bytes = slot;
      }
      for (size_t b = 0; b < m->blocks.size(); ++b)
        if (m->blocks[b].has_drop) FR_TRY(m->blocks[b].dc_bias.ensure((size_t)m->d.filters[b] * 4));
    }
    m->n_x3_all = (int)all.size(); m->n_x3_fwd = (int)fwd.size();
    m->x3_grid_all = conv_x3_pack_assign_blocks(all.data(), m->n_x3_all);
    m->x3_grid_fwd = conv_x3_pack_assign_blocks(fwd.data(), m->n_x3_fwd);
    conv_x3_pack_assign_blocks(all16.data(), m->n_x3_all);
    conv_x3_pack_assign_blocks(fwd16.data(), m->n_x3_fwd);
    all.insert(all.end(), fwd.begin(), fwd.end());
    all.insert(all.end(), all16.begin(), all16.end());
    all.insert(all.end(), fwd16.begin(), fwd16.end());
    for (size_t g = 0; g <= m->blocks.size(); ++g) {   // per-owner sub-tables: [plain jobs][the same with the weight magnitude attached]
      std::vector<PackXJob> gp, g16;
      auto addg = [&](Conv& c) {
        if (c.x_f) {
          gp.push_back(conv_x3_pack_job(c.w_off, c.Cout, c.Cin, c.k, 0, c.wx.p, c.Ho, c.Wo));
          PackXJob j = gp.back(); j.amax = m->rec(c.am + 2); j.amax_w = m->amax_ws.f() + c.am; g16.push_back(j);
        }
        if (c.x_d) {
          gp.push_back(conv_x3_pack_job(c.w_off, c.Cout, c.Cin, c.k, 1, c.wxd.p, c.H, c.W));
          PackXJob j = gp.back(); j.amax = m->rec(c.am + 2); j.amax_w = m->amax_ws.f() + c.am; g16.push_back(j);
        }
      };
      if (g < m->blocks.size()) { for (auto& c : m->convs) if (c.block == (int)g) addg(c); }
      else for (auto& hd : m->heads) addg(hd.c3);
      auto& G = m->groups[g];
      G.x3_n = (int)gp.size();
      G.x3_grid = gp.empty() ? 0 : conv_x3_pack_assign_blocks(gp.data(), G.x3_n);
      if (!g16.empty()) conv_x3_pack_assign_blocks(g16.data(), G.x3_n);
      G.x3_off = (int)all.size(); all.insert(all.end(), gp.begin(), gp.end());
      G.x3_off16 = (int)all.size(); all.insert(all.end(), g16.begin(), g16.end());
    }
    if (!all.empty()) {
      FR_TRY(m->x3_jobs.ensure(all.size() * sizeof(PackXJob)));
      FR_HIP(hipMemcpy(m->x3_jobs.p, all.data(), all.size() * sizeof(PackXJob), hipMemcpyHostToDevice));
    }
  }
  m->H = H; m->W = W;
  m->eval_packs_gen = -1;   // (pack buffers and job tables were rebuilt)
  m->fresh_mask = 0;
  return FRCNN_OK;
}

extern "C" {

int frcnn_model_create(const frcnn_model_desc* desc, frcnn_model** out) {
  FR_CHECK(desc && out, "frcnn_model_create: null argument");
  FR_CHECK(desc->nblocks >= 1 && desc->nblocks <= 8 && desc->nheads >= 0 && desc->nheads <= 8 &&
               desc->ncls >= 0 && desc->ncls <= 8,
           "frcnn_model_create: bad layer counts");
  for (int b = 0; b < desc->nblocks; ++b) {
    FR_CHECK(desc->ksize[b] == 1 || desc->ksize[b] == 3 || desc->ksize[b] == 5 || desc->ksize[b] == 7,
             "block %d: kernel size %d unsupported (1,3,5,7)", b + 1, desc->ksize[b]);
    FR_CHECK(desc->conv_steps[b] >= 1, "block %d: conv_steps must be >= 1", b + 1);
  }
  for (int h = 0; h < desc->nheads; ++h) {
    FR_CHECK(desc->head_input[h] >= 1 && desc->head_input[h] <= desc->nblocks, "anchor net %d: bad input", h + 1);
    FR_CHECK(desc->head_k[h] == 1 || desc->head_k[h] == 3 || desc->head_k[h] == 5 || desc->head_k[h] == 7,
             "anchor net %d: kernel size %d unsupported", h + 1, desc->head_k[h]);
  }
  frcnn_model* m = new frcnn_model();
  m->d = *desc;
  build_layout(m);
  *out = m;
  return FRCNN_OK;
}

int frcnn_model_destroy(frcnn_model* m) {
  if (!m) return FRCNN_OK;
  auto rel = [](Conv& c) { c.wf.release(); c.wd.release(); c.wx.release(); c.wxd.release(); c.x.release(); c.gx.release(); };
  for (auto& c : m->convs) rel(c);
  for (auto& b : m->blocks) { b.scale.release(); b.pooled.release(); b.gpooled.release(); b.pidx.release(); b.dc_bias.release(); }
  if (m->dc_pin) (void)hipHostFree(m->dc_pin);
  for (auto e : m->dc_ev) if (e) (void)hipEventDestroy(e);
  m->dc_dev.release(); m->dbg_expand.release();
  for (auto& h : m->heads) {
    rel(h.c3); rel(h.c1); h.delta.release(); h.spOut.release(); h.spSlab.release();
    h.spD.release(); h.spHX.release(); h.spHY.release(); h.spGH.release(); h.spCol.release(); h.spDX.release();
  }
  for (auto& l : m->cls) {
    l.lin.release(); l.pre.release(); l.post.release(); l.xhat.release(); l.invstd.release();
    l.mask.release(); l.g.release(); l.xp.release(); l.xpT.release(); l.gp.release(); l.gpT.release(); l.am.release();
  }
  m->img.release(); m->wg_ws.release(); m->wg_ws_first.release(); m->pack_jobs.release(); m->x3_jobs.release(); m->zero_arena.release(); m->amax.release(); m->amax_ws.release(); m->amax_jobs.release();
  for (auto e : m->fork_ev) (void)hipEventDestroy(e);
  for (auto& h : m->heads) {
    if (h.done) (void)hipEventDestroy(h.done);
    if (h.gin_done) (void)hipEventDestroy(h.gin_done);
  }
  if (m->chain_ev) (void)hipEventDestroy(m->chain_ev);
  if (m->join_ev) (void)hipEventDestroy(m->join_ev);
  if (m->bwd_ev) (void)hipEventDestroy(m->bwd_ev);
  if (m->heads_gin_ev) (void)hipEventDestroy(m->heads_gin_ev);
  if (m->upd_ev) (void)hipEventDestroy(m->upd_ev);
  if (m->upd_join_ev) (void)hipEventDestroy(m->upd_join_ev);
  for (auto e : m->block_rd_ev) (void)hipEventDestroy(e);
  for (auto e : m->cw_ev) (void)hipEventDestroy(e);
  if (m->cw_fork) (void)hipEventDestroy(m->cw_fork);
  if (m->cw_done) (void)hipEventDestroy(m->cw_done);
  if (m->loss_ev) (void)hipEventDestroy(m->loss_ev);
  for (auto e : m->block_ev) (void)hipEventDestroy(e);
  m->delta_last.release(); m->feat_g.release(); m->logits.release(); m->lsm.release(); m->glog.release();
  m->gtmp.release();
  delete m;
  return FRCNN_OK;
}

int frcnn_model_param_count(const frcnn_model* m, long long* total, long long* pnet) {
  if (total) *total = m->total_params;
  if (pnet) *pnet = m->pnet_params;
  return FRCNN_OK;
}

int frcnn_model_param_table(const frcnn_model* m, long long* table, int cap, int* n) {
  *n = (int)m->table.size();
  for (int i = 0; i < *n && i < cap; ++i)
    for (int j = 0; j < 4; ++j) table[4 * i + j] = m->table[i][j];
  return FRCNN_OK;
}

int frcnn_model_localizer_layers(const frcnn_model* m, int output_index, int* layers, int cap, int* n_out) {
  const frcnn_model_desc& d = m->d;
  FR_CHECK(output_index >= 1 && output_index <= d.nheads + 1, "localizer: output index %d out of range", output_index);
  int nb = output_index <= d.nheads ? d.head_input[output_index - 1] : d.nblocks;
  int n = 0;
  auto push = [&](int kW, int kH, int dW, int dH, int pW, int pH) {
    if (n < cap) { int* l = layers + 6 * n; l[0] = kW; l[1] = kH; l[2] = dW; l[3] = dH; l[4] = pW; l[5] = pH; }
    ++n;
  };
  for (int b = 0; b < nb; ++b) {
    for (int s = 0; s < d.conv_steps[b]; ++s) push(d.ksize[b], d.ksize[b], 1, 1, d.pad[b], d.pad[b]);
    push(2, 2, 2, 2, 0, 0);
  }
  if (output_index <= d.nheads) {
    int k = d.head_k[output_index - 1];
    push(k, k, 1, 1, 0, 0);
    push(1, 1, 1, 1, 0, 0);
  }
  *n_out = n;
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------ pnet
static int g_side_stream = -1;   // -1: not decided yet (environment FRCNN_SIDE_STREAM, default on)
static bool side_enabled() {
  if (g_side_stream < 0) g_side_stream = !(getenv("FRCNN_SIDE_STREAM") && atoi(getenv("FRCNN_SIDE_STREAM")) == 0) ? 1 : 0;
  return g_side_stream != 0;
}

// The classification net's weight-gradient products and bias sums are needed by nobody before the optimiser (or the exchange of
// the cnet slice): with the option on (default) frcnn_cnet_backward queues them on a stream of their own, beside the
// input-gradient chain that the ROI-pooling backward and the backbone wait for, and returns with that stream still busy --
// frcnn_pnet_backward, the next frcnn_cnet_forward and frcnn_cnet_backward_join make the caller's stream wait for it.
static int g_cnet_wgrad_async = getenv("FRCNN_CNET_WGRAD_ASYNC") ? (atoi(getenv("FRCNN_CNET_WGRAD_ASYNC")) != 0) : 1;

static int cw_join(frcnn_model* m, hipStream_t s) {
  if (m->cw_pending) {
    FR_HIP(hipStreamWaitEvent(s, m->cw_done, 0));
    m->cw_pending = false;
  }
  return FRCNN_OK;
}

int frcnn_cnet_backward_join(frcnn_model* m, void* stream) {
  FR_CHECK(m != nullptr, "cnet_backward_join: null model");
  return cw_join(m, S(stream));
}

// The update stream: a library-owned stream on which the host queues the optimiser's update of a slice of the flat vectors (and
// the renewal of the packs made from it) while the caller's stream is still busy with the rest of the backward pass.  It is the
// stream the classification net's weight gradients run on -- idle from the end of that stage to the end of the step -- so a
// slice update queued on it is ordered behind those gradients by itself.
static int ensure_update_stream(frcnn_model* m) {
  if (!m->cw) {
    FR_TRY(pool_stream(1, &m->cw));
    FR_HIP(hipEventCreateWithFlags(&m->cw_done, hipEventDisableTiming));
  }
  if (!m->upd_ev) FR_HIP(hipEventCreateWithFlags(&m->upd_ev, hipEventDisableTiming));
  if (!m->upd_join_ev) FR_HIP(hipEventCreateWithFlags(&m->upd_join_ev, hipEventDisableTiming));
  return FRCNN_OK;
}
int frcnn_model_update_stream(frcnn_model* m, void** stream) {
  FR_CHECK(m && stream, "model_update_stream: null argument");
  FR_TRY(ensure_update_stream(m));
  m->update_armed = true;
  *stream = (void*)m->cw;
  return FRCNN_OK;
}
// the update stream waits for everything queued on `stream` so far (the last readers of the weights about to be updated)
int frcnn_model_update_fork(frcnn_model* m, void* stream) {
  FR_CHECK(m != nullptr, "model_update_fork: null model");
  FR_TRY(ensure_update_stream(m));
  FR_HIP(hipEventRecord(m->upd_ev, S(stream)));
  FR_HIP(hipStreamWaitEvent(m->cw, m->upd_ev, 0));
  return FRCNN_OK;
}
// `stream` waits for everything queued on the update stream so far (before the next pass reads the weights and the packs)
int frcnn_model_update_join(frcnn_model* m, void* stream) {
  FR_CHECK(m != nullptr, "model_update_join: null model");
  FR_TRY(ensure_update_stream(m));
  FR_HIP(hipEventRecord(m->upd_join_ev, m->cw));
  FR_HIP(hipStreamWaitEvent(S(stream), m->upd_join_ev, 0));
  return FRCNN_OK;
}

int frcnn_get_option(const char* name, int* value) {
  FR_CHECK(name != nullptr && value != nullptr, "get_option: null argument");
  if (strcmp(name, "side_stream") == 0) { *value = side_enabled() ? 1 : 0; return FRCNN_OK; }
  if (strcmp(name, "gemm_x_roles") == 0) { *value = get_gemm_x_roles(); return FRCNN_OK; }
  if (strcmp(name, "deterministic") == 0) { *value = deterministic() ? 1 : 0; return FRCNN_OK; }
  if (strcmp(name, "static_weights") == 0) { *value = g_static_weights; return FRCNN_OK; }
  if (strcmp(name, "drop_compact") == 0) { *value = drop_compact_on(); return FRCNN_OK; }
  if (strcmp(name, "sparse_heads") == 0) { *value = sparse_heads_on(); return FRCNN_OK; }
  if (strcmp(name, "winograd") == 0) { *value = 0; return FRCNN_OK; }   // removed in round 3; kept as a name that reads 0
  if (strcmp(name, "cnet_wgrad_async") == 0) { *value = g_cnet_wgrad_async; return FRCNN_OK; }
  if (strcmp(name, "split_bf16") == 0) { *value = get_split_bf16(); return FRCNN_OK; }
  if (strcmp(name, "x3_f16") == 0) { *value = get_x3_f16(); return FRCNN_OK; }
  FR_CHECK(false, "get_option: unknown option '%s'", name);
  return FRCNN_OK;
}

int frcnn_set_option(const char* name, int value) {
  FR_CHECK(name != nullptr, "set_option: null name");
  if (strcmp(name, "side_stream") == 0) { g_side_stream = value ? 1 : 0; return FRCNN_OK; }
  if (strcmp(name, "deterministic") == 0) { set_deterministic(value != 0); return FRCNN_OK; }
  if (strcmp(name, "static_weights") == 0) { g_static_weights = value != 0; ++g_static_gen; return FRCNN_OK; }
  if (strcmp(name, "drop_compact") == 0) { g_drop_compact = value != 0; return FRCNN_OK; }
  if (strcmp(name, "sparse_heads") == 0) { g_sparse_heads = value != 0; return FRCNN_OK; }
  if (strcmp(name, "gemm_x_roles") == 0) { set_gemm_x_roles(value); return FRCNN_OK; }   // takes effect at the next cnet pass
  if (strcmp(name, "split_bf16") == 0) { set_split_bf16(value); return FRCNN_OK; }   // takes effect for models shaped afterwards
  if (strcmp(name, "x3_f16") == 0) { set_x3_f16(value); return FRCNN_OK; }           // takes effect with the next forward pass
  if (strcmp(name, "winograd") == 0) return FRCNN_OK;   // deprecated no-op: the Winograd kernels were removed in round 3
  if (strcmp(name, "cnet_wgrad_async") == 0) { g_cnet_wgrad_async = value ? 1 : 0; return FRCNN_OK; }
  FR_CHECK(false, "set_option: unknown option '%s'", name);
  return FRCNN_OK;
}

static int ensure_head_streams(frcnn_model* m);
static int ensure_side(frcnn_model* m) {
  if (!m->side) {
    FR_TRY(pool_stream(0, &m->side));
    FR_HIP(hipEventCreateWithFlags(&m->join_ev, hipEventDisableTiming));
    FR_HIP(hipEventCreateWithFlags(&m->loss_ev, hipEventDisableTiming));
    FR_HIP(hipEventCreateWithFlags(&m->chain_ev, hipEventDisableTiming));
    // FRCNN_HEAD_STREAMS=1: one stream per anchor net (the four nets are independent of each other: forward a k x k and
    // a 1 x 1 convolution on a pooled map; backward on the sampled anchors ~17 small launches each), so that their chains
    // run beside each other instead of one after the other.  Measured on the training step: the anchor nets' backward
    // chains shrink from 670 to 310 us, but the classification net's chain on the caller's stream, which runs beside them,
    // slows down by as much: 223.6 against 225.4 images/s (round 2).  Round 4: the two chains of the middle phase -- the
    // anchor nets' and the classification net's -- are about equally long, so shortening ONE of them changes nothing; with
    // the classification net's weight gradients off its chain as well (g_cnet_wgrad_async) the step goes from 3.11 to
    // 3.02 ms, with either change alone it stays at 3.11.  On by default since then.
    for (auto& h : m->heads) FR_HIP(hipEventCreateWithFlags(&h.done, hipEventDisableTiming));
  }
  return FRCNN_OK;
}

// The anchor nets' own streams, made when the DENSE path first needs them (round 6): a stream that exists takes a share of a
// hardware queue whether it is used or not, and the sparse training path (heads.hip) runs every anchor net on the side stream.
static int ensure_head_streams(frcnn_model* m) {
  static const int head_streams = getenv("FRCNN_HEAD_STREAMS") ? atoi(getenv("FRCNN_HEAD_STREAMS")) : 1;
  size_t hi = 0;
  for (auto& h : m->heads) {
    // (five workspace slots for streams of their own, head_slot(): a sixth anchor net shares the side stream and its slot)
    if (head_streams && hi < 5 && !h.stream) FR_TRY(pool_stream(2 + (int)hi, &h.stream));
    ++hi;
  }
  return FRCNN_OK;
}

static hipStream_t head_stream(frcnn_model* m, size_t i) { return m->heads[i].stream ? m->heads[i].stream : m->side; }
// split-K workspace slot of anchor net i: 2..6 (slot 7 belongs to the classification net's weight-gradient stream, 0 / 1 to the
// caller's and the side stream); build_layout gives streams of their own to at most five anchor nets
static int head_slot(frcnn_model* m, size_t i) { return m->heads[i].stream ? 2 + (int)(i % 5) : 1; }

// the caller's stream waits for everything queued on the side stream so far
static int join_heads(frcnn_model* m, hipStream_t s) {
  for (auto& h : m->heads) {
    if (!h.stream) continue;
    FR_HIP(hipEventRecord(h.done, h.stream));
    FR_HIP(hipStreamWaitEvent(s, h.done, 0));
  }
  return FRCNN_OK;
}
static int join_side(frcnn_model* m, hipStream_t s) {
  if (!m->side || !m->side_busy) return FRCNN_OK;
  FR_TRY(join_heads(m, s));
  FR_HIP(hipEventRecord(m->join_ev, m->side));
  FR_HIP(hipStreamWaitEvent(s, m->join_ev, 0));
  m->side_busy = false;
  return FRCNN_OK;
}

// stream `to` waits for everything enqueued on `from` so far (event ev is re-recorded: its previous waiters are queued already)
static int chain(hipStream_t from, hipStream_t to, hipEvent_t ev) {
  if (from == to) return FRCNN_OK;
  FR_HIP(hipEventRecord(ev, from));
  FR_HIP(hipStreamWaitEvent(to, ev, 0));
  return FRCNN_OK;
}

// side stream waits for everything enqueued on `s` so far
static int fork_side(frcnn_model* m, hipStream_t s, size_t idx) {
  FR_TRY(ensure_side(m));
  while (m->fork_ev.size() <= idx) {
    hipEvent_t e;
    FR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->fork_ev.push_back(e);
  }
  FR_HIP(hipEventRecord(m->fork_ev[idx], s));
  FR_HIP(hipStreamWaitEvent(m->side, m->fork_ev[idx], 0));
  m->side_busy = true;
  return FRCNN_OK;
}

// stream `to` (the side stream or an anchor net's own) waits for everything enqueued on `s` so far
static int fork_to(frcnn_model* m, hipStream_t s, hipStream_t to, size_t idx) {
  FR_TRY(ensure_side(m));
  while (m->fork_ev.size() <= idx) {
    hipEvent_t e;
    FR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->fork_ev.push_back(e);
  }
  FR_HIP(hipEventRecord(m->fork_ev[idx], s));
  FR_HIP(hipStreamWaitEvent(to, m->fork_ev[idx], 0));
  m->side_busy = true;
  return FRCNN_OK;
}

// one anchor net: k x k conv -> PReLU (fused into the 1x1 loader) -> 1x1 conv (models/model_utilities.lua:31-34)
static int head_forward(frcnn_model* m, Head& h, const float* w, hipStream_t s, int ws_slot) {
  const Block& in = m->blocks[h.input];
  if (h.c3.x_f)
    FR_TRY(conv_x3(in.pooled.f(), h.c3.Cin, h.c3.H, h.c3.W, nullptr, nullptr, h.c3.wx.p, w + h.c3.b_off, h.c3.Cout, h.c3.k, 0,
                   h.c3.x.f(), OUT_STORE, 0, s, ws_slot, nullptr, m->f16_packed ? m->rec(in.am) : nullptr,
                   m->f16_packed ? m->amax_ws.f() + h.c3.am : nullptr));
  else
    FR_TRY(conv_igemm(in.pooled.f(), h.c3.Cin, h.c3.H, h.c3.W, nullptr, nullptr, h.c3.wf.f(), w + h.c3.b_off,
                      h.c3.Cout, h.c3.k, 0, h.c3.x.f(), OUT_STORE, 0, s, ws_slot));
  FR_TRY(conv_igemm(h.c3.x.f(), h.c1.Cin, h.c1.H, h.c1.W, w + h.c3.a_off, nullptr, h.c1.wf.f(), w + h.c1.b_off,
                    HEAD_OUT, 1, 0, h.c1.x.f(), OUT_STORE, 0, s, ws_slot));
  return FRCNN_OK;
}

// ---- compact blocks (Block::dc_on) --------------------------------------------------------------------------------------
static bool any_compact(const frcnn_model* m) {
  for (auto& b : m->blocks) if (b.dc_on) return true;
  return false;
}
static bool fuse_act_on() { return !deterministic() && !(getenv("FRCNN_FUSE_ACT") && atoi(getenv("FRCNN_FUSE_ACT")) == 0); }

// Decides, block by block, whether this training pass runs compact, from the keep vectors -- drawn here with the very hash the
// device kernel uses (common.h frcnn_keep_mask: the vectors on the device and on the host are the same), or read back from the
// caller's explicit masks (a synchronous copy: parity runs only) -- and fills the step's slot of the table ring with the kept
// channels.  The job table follows in pack_compact.
static int plan_compact(frcnn_model* m, int training, const float* const* drop_masks, unsigned long long seed) {
  for (auto& b : m->blocks) { b.dc_on = false; b.dc_idx = nullptr; }
  if (!training || !drop_compact_on() || !fuse_act_on() || !m->dc_pin) return FRCNN_OK;
  const unsigned slot = m->dc_step % frcnn_model::DC_RING;
  if (m->dc_ev[slot]) FR_HIP(hipEventSynchronize(m->dc_ev[slot]));   // (the slot's previous copy, DC_RING passes ago: long done unless the host never looks back)
  int* idx_host = (int*)(m->dc_pin + (size_t)slot * m->dc_slot_bytes + m->dc_idx_off);
  const int* idx_dev = (const int*)((char*)m->dc_dev.p + (size_t)slot * m->dc_slot_bytes + m->dc_idx_off);
  size_t at = 0;
  std::vector<float> keep;
  for (size_t b = 0; b < m->blocks.size(); ++b) {
    Block& blk = m->blocks[b];
    const int C = m->d.filters[b];
    const size_t my = at;
    at += (size_t)C;
    if (!blk.has_drop || blk.nconv < 2) continue;
    Conv &c0 = m->convs[blk.first_conv], &c1 = m->convs[blk.first_conv + 1];
    if (!(c0.k == 3 && c1.k == 3 && c0.x_f && c1.x_f && c1.x_d && (b == 0 || c0.x_d))) continue;
    keep.resize(C);
    if (drop_masks && drop_masks[b]) {
      FR_HIP(hipMemcpy(keep.data(), drop_masks[b], (size_t)C * 4, hipMemcpyDeviceToHost));
    } else {
      for (int i = 0; i < C; ++i) keep[i] = frcnn_keep_mask(seed * 131 + b, (unsigned long long)i, blk.p_drop);
    }
    int nk = 0;
    bool binary = true;
    for (int i = 0; i < C; ++i) {
      if (keep[i] == 1.0f) idx_host[my + nk++] = i;
      else if (keep[i] != 0.0f) binary = false;   // (a caller's scale vector that is no keep vector: the dense path multiplies by it)
    }
    if (!binary) continue;
    for (int i = nk; i < C; ++i) idx_host[my + i] = -1;
    const int nkM = std::min(C, (nk + 63) / 64 * 64), nkK = std::min(C, (nk + 15) / 16 * 16);
    if (nk == 0 || nkK >= C) continue;                       // nothing to leave out (or nothing kept: the dense path handles the zeros)
    if (!conv_x3_eligible(c0.Cin, nkM, 3) || !conv_x3_eligible(nkK, c1.Cout, 3) || !conv_x3_eligible(c1.Cout, nkM, 3) ||
        (b > 0 && !conv_x3_eligible(nkK, c0.Cin, 3)) || !conv_wgradx_eligible(nkM, c1.Cout, 3) || !conv_wgradx_eligible(c0.Cin, nkM, 3))
      continue;
    blk.dc_on = true; blk.nk = nk; blk.nkM = nkM; blk.nkK = nkK; blk.dc_idx = idx_dev + my;
  }
  return FRCNN_OK;
}

// The step's pack-job table: the model's jobs, those of the compact blocks re-written for the kept filters / channels; one
// asynchronous copy brings it and the kept-channel tables over, one launch packs.
static int pack_compact(frcnn_model* m, const float* w, bool f16, hipStream_t s) {
  const unsigned slot = m->dc_step % frcnn_model::DC_RING;
  ++m->dc_step;
  char* hslot = m->dc_pin + (size_t)slot * m->dc_slot_bytes;
  char* dslot = (char*)m->dc_dev.p + (size_t)slot * m->dc_slot_bytes;
  const std::vector<PackXJob>& src = f16 ? m->x3_host16 : m->x3_host;
  PackXJob* jobs = (PackXJob*)hslot;
  int n = 0;
  for (int i = 0; i < (int)src.size(); ++i) {
    PackXJob j = src[i];
    const int ci = m->x3_conv[i];
    if (ci < 0 && !m->head_x3_fresh) continue;   // an anchor net's pack in a pass that runs the anchor nets sparse
    if (ci >= 0) {
      const Conv& c = m->convs[ci];
      const Block& blk = m->blocks[c.block];
      if (blk.dc_on && c.step <= 1) {
        // (b, 0): filters gathered -- forward M = nkM, input gradient K = nkK;  (b, 1): channels gathered -- forward K = nkK,
        // input gradient M = nkM.  Entries beyond the kept ones are -1: rows / channels of zeros.
        PackXJob g;
        if (c.step == 0) {
          const int O = j.mode == 0 ? blk.nkM : blk.nkK;
          g = j.mode == 0 ? conv_x3_pack_job(c.w_off, O, c.Cin, c.k, 0, c.wx.p, c.Ho, c.Wo) : conv_x3_pack_job(c.w_off, O, c.Cin, c.k, 1, c.wxd.p, c.H, c.W);
          g.oidx = blk.dc_idx; g.Cs = c.Cin;
          if (j.mode == 0) { g.bias_off = c.b_off; g.bias_dst = blk.dc_bias.f(); }
        } else {
          const int C = j.mode == 0 ? blk.nkK : blk.nkM;
          g = j.mode == 0 ? conv_x3_pack_job(c.w_off, c.Cout, C, c.k, 0, c.wx.p, c.Ho, c.Wo) : conv_x3_pack_job(c.w_off, c.Cout, C, c.k, 1, c.wxd.p, c.H, c.W);
          g.cidx = blk.dc_idx; g.Cs = c.Cin;
        }
        g.amax = j.amax; g.amax_w = j.amax_w;
        j = g;
      }
    }
    jobs[n++] = j;
  }
  const int grid = conv_x3_pack_assign_blocks(jobs, n);
  FR_HIP(hipMemcpyAsync(dslot, hslot, m->dc_slot_bytes, hipMemcpyHostToDevice, s));
  if (!m->dc_ev[slot]) FR_HIP(hipEventCreateWithFlags(&m->dc_ev[slot], hipEventDisableTiming));
  FR_HIP(hipEventRecord(m->dc_ev[slot], s));
  FR_TRY(conv_x3_pack_multi(w, (const PackXJob*)dslot, n, grid, s));
  return FRCNN_OK;
}

static int pnet_forward_impl(frcnn_model* m, const float* w, const float* img, int H, int W, int training,
                             const float* const* drop_masks, unsigned long long seed, void* stream, bool async_heads) {
  hipStream_t s = S(stream);
  FR_TRY(join_side(m, s));   // (anchor nets of an abandoned asynchronous forward still read the model's buffers)
  if (H != m->H || W != m->W) FR_TRY(ensure_shapes(m, H, W));
  m->training = training;
  m->heads_begun = false; m->heads_joined = false; m->loss_pending = false;
  const bool use_side = side_enabled();
  async_heads = async_heads && use_side && training;
  // training through the objective: the anchor nets wait for the call that knows the sampled positions (heads.hip)
  m->heads_deferred = async_heads && sparse_heads_on() && !deterministic() && m->heads.size() <= 4;
  m->heads_sparse_fwd = false; m->heads_gin = false;
  m->last_w = w;
  if (use_side && !m->heads_deferred) { FR_TRY(ensure_side(m)); FR_TRY(ensure_head_streams(m)); }
  // SpatialDropout scales (device-drawn ones: one launch for all blocks)
  DropoutJobs dj;
  dj.n = 0;
  for (size_t b = 0; b < m->blocks.size(); ++b) {
    Block& blk = m->blocks[b];
    if (!blk.has_drop) continue;
    int C = m->d.filters[b];
    if (!training) {
      FR_TRY(fill_value(blk.scale.f(), C, 1.0f - blk.p_drop, s));  // evaluate(): x(1-p) [ext]
    } else if (drop_masks && drop_masks[b]) {
      FR_HIP(hipMemcpyAsync(blk.scale.p, drop_masks[b], (size_t)C * 4, hipMemcpyDeviceToDevice, s));
    } else {
      dj.ptr[dj.n] = blk.scale.f(); dj.C[dj.n] = C; dj.p[dj.n] = blk.p_drop; dj.seed[dj.n] = seed * 131 + b;
      ++dj.n;
    }
  }
  FR_TRY(dropout_channel_masks(dj, s));
  FR_TRY(plan_compact(m, training, drop_masks, seed));
  // weights change every optimiser step: refresh the packed copies (one table-driven launch each).  Option static_weights: an
  // evaluate-mode pass re-uses the packs of the previous evaluate-mode pass with the same weight vector -- the host's promise that
  // it has not written the weights in between (a training-mode pass, a shape change and every frcnn_set_option("static_weights", v)
  // call drop them).
  m->head_packs_fresh = false;
  const bool f16 = get_x3_f16() && m->n_amax_jobs > 0;
  const bool reuse = !training && g_static_weights && m->eval_packs_gen == g_static_gen && m->eval_packs_w == w && m->f16_packed == f16;
  m->f16_packed = f16;
  // ... or every owner's packs were renewed from this weight vector since the last pass (frcnn_pnet_refresh_packs: the update
  // ran beside the previous backward pass).  One-shot: the promise covers the pass that consumes it.
  const bool fresh = training && !m->groups.empty() && m->fresh_mask == (1u << m->groups.size()) - 1 && m->fresh_w == w && m->fresh_f16 == f16 &&
                     !any_compact(m);   // (a compact block's packs depend on the step's keep vector)
  m->fresh_mask = 0;
  if (!reuse && !fresh) {
    // (a training pass that leaves the anchor nets to the sparse path packs nothing of theirs: frcnn_model::head_x3_fresh)
    static const bool skip_on = !(getenv("FRCNN_HEAD_PACK_SKIP") && atoi(getenv("FRCNN_HEAD_PACK_SKIP")) == 0);
    const bool skip_heads = skip_on && training && m->heads_deferred && any_compact(m) && m->am_bb_n > 0;
    m->head_x3_fresh = !skip_heads;
    if (training && skip_heads) {
      if (m->pk_bb_n) FR_TRY(conv_pack_weights_multi(w, (const PackJob*)m->pack_jobs.p + m->pk_bb_off, m->pk_bb_n, m->pk_bb_grid, s));
    } else if (training)
      FR_TRY(conv_pack_weights_multi(w, (const PackJob*)m->pack_jobs.p, m->n_pack_all, m->pack_grid_all, s));
    else
      FR_TRY(conv_pack_weights_multi(w, (const PackJob*)m->pack_jobs.p + m->n_pack_all, m->n_pack_fwd, m->pack_grid_fwd, s));
    const PackXJob* xjobs = (const PackXJob*)m->x3_jobs.p + (f16 ? m->n_x3_all + m->n_x3_fwd : 0);
    if (f16 && skip_heads)
      FR_TRY(tensor_absmax_multi(w, (const AmaxJob*)m->amax_jobs.p + m->am_bb_off, m->am_bb_n, m->am_bb_grid, s));
    else if (f16)   // the weight tensors' magnitudes first: their packs are scaled by them
      FR_TRY(tensor_absmax_multi(w, (const AmaxJob*)m->amax_jobs.p, m->n_amax_jobs, m->amax_grid, s));
    if (training && any_compact(m))
      FR_TRY(pack_compact(m, w, f16, s));   // the step's own table: the compact blocks' jobs gather the kept filters / channels
    else if (training)
      FR_TRY(conv_x3_pack_multi(w, xjobs, m->n_x3_all, m->x3_grid_all, s));
    else
      FR_TRY(conv_x3_pack_multi(w, xjobs + m->n_x3_all, m->n_x3_fwd, m->x3_grid_fwd, s));
  }
  m->eval_packs_gen = training ? -1 : g_static_gen;
  m->eval_packs_w = w;
  FR_TRY(m->img.ensure((size_t)3 * H * W * 4));
  FR_HIP(hipMemcpyAsync(m->img.p, img, (size_t)3 * H * W * 4, hipMemcpyDeviceToDevice, s));
  const float* cur = m->img.f();
  const float* cur_slope = nullptr;
  const float* cur_scale = nullptr;
  const float* cur_am = nullptr;   // magnitude scalar of `cur` (fp16 form)
  for (size_t b = 0; b < m->blocks.size(); ++b) {
    Block& blk = m->blocks[b];
    bool pooled_in_conv = false;
    for (int st = 0; st < blk.nconv; ++st) {
      Conv& c = m->convs[blk.first_conv + st];
      const bool last = st == blk.nconv - 1;
      // the block's max pool rides in the epilogue of its last convolution when that launch is a single K split
      IgemmPool pl = {blk.pooled.f(), (unsigned char*)blk.pidx.p, w + c.a_off,
                      (st == 0 && blk.has_drop) ? blk.scale.f() : nullptr, f16 ? m->rec(blk.am) : nullptr};
      // fp16 form: the next convolution reads c.x scaled by its largest magnitude, which the launch that writes c.x records
      const bool want_am = f16 && !last && m->convs[blk.first_conv + st + 1].x_f;
      if (c.x_f && blk.dc_on && st <= 1) {
        // compact block: the first convolution computes its kept filters only (output stored compact), the second reads the
        // kept channels only -- their dropout scale is 1
        const int cin = st == 0 ? c.Cin : blk.nkK, mo = st == 0 ? blk.nkM : c.Cout;
        FR_TRY(conv_x3(cur, cin, c.H, c.W, cur_slope, nullptr, c.wx.p, st == 0 ? blk.dc_bias.f() : w + c.b_off, mo, c.k, c.pad, c.x.f(),
                       OUT_STORE, 0, s, 0, nullptr, f16 ? cur_am : nullptr, f16 ? m->amax_ws.f() + c.am : nullptr,
                       want_am ? m->rec(c.am) : nullptr));
      } else if (c.x_f)
        FR_TRY(conv_x3(cur, c.Cin, c.H, c.W, cur_slope, cur_scale, c.wx.p, w + c.b_off, c.Cout, c.k, c.pad, c.x.f(), OUT_STORE, 0, s, 0,
                       nullptr, f16 ? cur_am : nullptr, f16 ? m->amax_ws.f() + c.am : nullptr, want_am ? m->rec(c.am) : nullptr));
      else
        FR_TRY(conv_igemm(cur, c.Cin, c.H, c.W, cur_slope, cur_scale, c.wf.f(), w + c.b_off, c.Cout, c.k, c.pad,
                          c.x.f(), OUT_STORE, 0, s, 0, last ? &pl : nullptr, last ? &pooled_in_conv : nullptr));
      if (want_am) {
        if (!c.x_f) FR_TRY(tensor_absmax(c.x.f(), (long)c.Cout * c.Ho * c.Wo, m->rec(c.am), s));   // (the fp32 kernel keeps no record)
        cur_am = m->rec(c.am);
      }
      cur = c.x.f();
      cur_slope = w + c.a_off;
      cur_scale = (st == 0 && blk.has_drop) ? blk.scale.f() : nullptr;  // model_utilities.lua:20
    }
    const Conv& lc = m->convs[blk.first_conv + blk.nconv - 1];
    if (!pooled_in_conv)
      FR_TRY(maxpool_act_forward(cur, lc.Cout, lc.Ho, lc.Wo, cur_slope, cur_scale, blk.pooled.f(),
                                 (unsigned char*)blk.pidx.p, s, f16 ? m->rec(blk.am) : nullptr));
    // (pooled inside the fp32 kernel of the first layer: that launch kept the pooled map's record)
    cur = blk.pooled.f();
    cur_slope = nullptr;
    cur_scale = nullptr;
    cur_am = f16 ? m->rec(blk.am) : nullptr;   // the pooled map feeds the next block and the anchor nets on it
    // anchor nets on an earlier block's map run beside the following blocks, each on its own stream
    if (use_side && b + 1 < m->blocks.size() && !m->heads_deferred) {
      FR_TRY(ensure_side(m));
      for (size_t i = 0; i < m->heads.size(); ++i) {
        Head& h = m->heads[i];
        if (h.input != (int)b) continue;
        FR_TRY(fork_to(m, s, head_stream(m, i), 16 + i));
        FR_TRY(head_forward(m, h, w, head_stream(m, i), head_slot(m, i)));
      }
    }
  }
  // heads on the LAST block's map: the heaviest stays on the caller's stream, the others run beside it
  if (!m->heads_deferred) {
    const int last = (int)m->blocks.size() - 1;
    int heavy = -1;
    for (size_t i = 0; i < m->heads.size(); ++i)
      if ((!use_side || m->heads[i].input == last) && (heavy < 0 || m->heads[i].c3.k > m->heads[heavy].c3.k)) heavy = (int)i;
    if (async_heads) heavy = -1;   // the caller's stream goes on with the last map; every anchor net stays off it
    if (use_side) FR_TRY(ensure_side(m));
    for (size_t i = 0; use_side && i < m->heads.size(); ++i) {
      if (m->heads[i].input != last || (int)i == heavy) continue;
      FR_TRY(fork_to(m, s, head_stream(m, i), 16 + i));
      FR_TRY(head_forward(m, m->heads[i], w, head_stream(m, i), head_slot(m, i)));
    }
    for (size_t i = 0; i < m->heads.size(); ++i)
      if ((int)i == heavy || (!use_side)) FR_TRY(head_forward(m, m->heads[i], w, s, 0));
  }
  if (use_side && !async_heads) FR_TRY(join_side(m, s));   // the caller's stream continues after every head is done
  return FRCNN_OK;
}

int frcnn_pnet_forward(frcnn_model* m, const float* w, const float* img, int H, int W, int training,
                       const float* const* drop_masks, unsigned long long seed, void* stream) {
  return pnet_forward_impl(m, w, img, H, W, training, drop_masks, seed, stream, false);
}

int frcnn_pnet_forward_async_heads(frcnn_model* m, const float* w, const float* img, int H, int W,
                                   const float* const* drop_masks, unsigned long long seed, void* stream) {
  return pnet_forward_impl(m, w, img, H, W, 1, drop_masks, seed, stream, true);
}

int frcnn_pnet_output(frcnn_model* m, int i, float** ptr, int* C, int* H, int* W) {
  FR_CHECK(m->H > 0, "pnet_output: call frcnn_pnet_forward first");
  FR_CHECK(i >= 1 && i <= (int)m->heads.size() + 1, "pnet_output: index %d out of range", i);
  if (i <= (int)m->heads.size()) {
    Head& h = m->heads[i - 1];
    *ptr = h.c1.x.f(); *C = HEAD_OUT; *H = h.c1.Ho; *W = h.c1.Wo;
  } else {
    Block& b = m->blocks.back();
    *ptr = b.pooled.f(); *C = m->d.filters[m->d.nblocks - 1]; *H = b.Hp; *W = b.Wp;
  }
  return FRCNN_OK;
}

int frcnn_model_debug_buffer(frcnn_model* m, int kind, int index, void** ptr, long long* bytes) {
  FR_CHECK(m && ptr && bytes, "frcnn_model_debug_buffer: NULL argument");
  const DevBuf* b = nullptr;
  size_t n = 0;
  switch (kind) {
    case 0: {
      FR_CHECK(index >= 0 && index < (int)m->convs.size() && m->H > 0, "debug_buffer: backbone convolution %d (after a forward pass)", index);
      const Conv& c = m->convs[index]; b = &c.x; n = (size_t)c.Cout * c.Ho * c.Wo * 4;
      const Block& k = m->blocks[c.block];
      if (k.dc_on && c.step == 0) {   // stored compact in this pass (Block::dc_on): laid out dense here, zeros where nothing was computed
        FR_HIP(hipDeviceSynchronize());
        FR_TRY(m->dbg_expand.ensure(n));
        FR_HIP(hipMemset(m->dbg_expand.p, 0, n));
        std::vector<int> idx(c.Cout);
        FR_HIP(hipMemcpy(idx.data(), k.dc_idx, (size_t)c.Cout * 4, hipMemcpyDeviceToHost));
        const size_t row = (size_t)c.Ho * c.Wo * 4;
        for (int i = 0; i < k.nk; ++i)
          FR_HIP(hipMemcpy((char*)m->dbg_expand.p + (size_t)idx[i] * row, (const char*)c.x.p + (size_t)i * row, row, hipMemcpyDeviceToDevice));
        b = &m->dbg_expand;
      }
    } break;
    case 4: {   // the SpatialDropout scale vector of block `index` in the last pass (a keep vector while training)
      FR_CHECK(index >= 0 && index < (int)m->blocks.size() && m->H > 0 && m->blocks[index].has_drop, "debug_buffer: block %d has no dropout", index);
      const Block& k = m->blocks[index]; b = &k.scale; n = (size_t)m->d.filters[index] * 4;
    } break;
    case 1: {
      FR_CHECK(index >= 0 && index < (int)m->blocks.size() && m->H > 0, "debug_buffer: block %d (after a forward pass)", index);
      const Block& k = m->blocks[index]; b = &k.pidx; n = (size_t)m->d.filters[index] * k.Hp * k.Wp;
    } break;
    case 2: {
      FR_CHECK(index >= 0 && index < (int)m->heads.size() && m->H > 0, "debug_buffer: anchor net %d (after a forward pass)", index);
      const Conv& c = m->heads[index].c3; b = &c.x; n = (size_t)c.Cout * c.Ho * c.Wo * 4;
      if ((m->heads_sparse_fwd || m->heads_deferred) && m->last_w) {
        // the training pass computed this map at the sampled positions only (heads.hip), or not yet: the dense convolution, now
        FR_HIP(hipDeviceSynchronize());
        if (!m->head_x3_fresh) { FR_TRY(refresh_group(m, m->last_w, (int)m->blocks.size(), nullptr)); m->head_x3_fresh = true; }
        FR_TRY(head_forward(m, m->heads[index], m->last_w, nullptr, 0));
        FR_HIP(hipDeviceSynchronize());
      }
    } break;
    case 3: {
      FR_CHECK(index >= 0 && index < (int)m->cls.size() && m->R > 0, "debug_buffer: classification layer %d (after frcnn_cnet_forward)", index);
      const ClsLayer& L = m->cls[index]; b = L.bn ? &L.pre : &L.lin; n = (size_t)m->R * L.n * 4;
    } break;
    default: FR_CHECK(false, "debug_buffer: unknown kind %d", kind);
  }
  FR_CHECK(b->p && b->bytes >= n, "debug_buffer: buffer not allocated yet");
  *ptr = b->p; *bytes = (long long)n;
  return FRCNN_OK;
}

int frcnn_pnet_delta(frcnn_model* m, int i, float** ptr) {
  FR_CHECK(m->H > 0, "pnet_delta: call frcnn_pnet_forward first");
  FR_CHECK(i >= 1 && i <= (int)m->heads.size() + 1, "pnet_delta: index %d out of range", i);
  *ptr = i <= (int)m->heads.size() ? m->heads[i - 1].delta.f() : m->delta_last.f();
  return FRCNN_OK;
}

int frcnn_pnet_set_sparse_deltas(frcnn_model* m, int head, const int* positions, int count) {
  FR_CHECK(head >= 1 && head <= (int)m->heads.size(), "pnet_set_sparse_deltas: head %d out of range", head);
  m->heads[head - 1].sp_pos = positions;
  m->heads[head - 1].sp_count = count;
  return FRCNN_OK;
}

int frcnn_pnet_zero_deltas(frcnn_model* m, void* stream) {
  FR_CHECK(m->H > 0, "pnet_zero_deltas: call frcnn_pnet_forward first");
  FR_TRY(fill_zero(m->zero_arena.p, m->delta_bytes, S(stream)));
  return FRCNN_OK;
}

// One anchor net's part of pnet:backward: gradients of its parameters, input gradient added into the pooled map's gradient
// buffer (nngraph fan-out; atomic adds -- several anchor nets feed the same map).  Runs on stream s with split-K workspace
// slot ws_slot.  *dense is set when the dense fallback was taken (no usable sparse hint).
static bool head_is_sparse(const Head& h) { return h.sp_count >= 0 && h.sp_count <= SPARSE_MAX_POS; }

static int backward_head(frcnn_model* m, Head& h, const float* w, float* grad, hipStream_t s, int ws_slot) {
  Block& in = m->blocks[h.input];
  Conv &a = h.c3, &c = h.c1;
  const long hw1 = (long)c.Ho * c.Wo;
  if (head_is_sparse(h)) {
    // ---- sparse path: the same arithmetic restricted to the P positions where delta is non-zero
    const int P = h.sp_count;
    const int* pos = h.sp_pos;
    h.sp_count = -1; h.sp_pos = nullptr;   // one-shot hint
    if (P == 0) return FRCNN_OK;           // no example on this head: every gradient term is zero
    const int n = a.Cout, ckk = a.Cin * a.k * a.k;
    FR_TRY(h.spD.ensure((size_t)HEAD_OUT * SPARSE_MAX_POS * 4));
    FR_TRY(h.spHX.ensure((size_t)n * SPARSE_MAX_POS * 4));
    FR_TRY(h.spHY.ensure((size_t)n * SPARSE_MAX_POS * 4));
    FR_TRY(h.spGH.ensure((size_t)n * SPARSE_MAX_POS * 4));
    FR_TRY(h.spCol.ensure((size_t)ckk * SPARSE_MAX_POS * 4));
    FR_TRY(h.spDX.ensure((size_t)ckk * SPARSE_MAX_POS * 4));
    float *D = h.spD.f(), *HX = h.spHX.f(), *HY = h.spHY.f(), *GH = h.spGH.f(), *COL = h.spCol.f(), *DX = h.spDX.f();
    FR_TRY(gather_positions(h.delta.f(), HEAD_OUT, hw1, pos, P, D, nullptr, nullptr, s));
    FR_TRY(gather_positions(a.x.f(), n, hw1, pos, P, HX, w + a.a_off, HY, s));
    // the input-gradient path first (the backbone's backward pass waits for it), the parameter gradients after it:
    // GH[n][P] = W1^T[n][18] * D[18][P], then PReLU backward (+ bias / slope gradients of the k x k conv)
    FR_TRY(gemm_f32(w + c.w_off, 1, n, D, P, 1, GH, P, n, P, HEAD_OUT, OUT_STORE, nullptr, s, ws_slot));
    FR_TRY(act_backward(GH, HX, n, P, w + a.a_off, nullptr, GH, grad + a.b_off, grad + a.a_off, s));
    // DX[P][ckk] = GH^T[P][n] * W[n][ckk], scattered back into the pooled-map gradient
    FR_TRY(gemm_f32(GH, 1, P, w + a.w_off, ckk, 1, DX, ckk, P, ckk, n, OUT_STORE, nullptr, s, ws_slot));
    FR_TRY(col2im_positions_add(DX, a.Cin, a.H, a.W, a.k, a.Wo, pos, P, in.gpooled.f(), s));
    if (h.stream && s == h.stream) {   // (on its own stream: see Head::gin_done)
      if (!h.gin_done) FR_HIP(hipEventCreateWithFlags(&h.gin_done, hipEventDisableTiming));
      FR_HIP(hipEventRecord(h.gin_done, s));
      h.gin_recorded = true;
    }
    // 1x1 conv: gW1[18][n] += D[18][P] * HY[n][P]^T ; gb1 += rowsum(D)
    FR_TRY(gemm_f32(D, P, 1, HY, 1, P, grad + c.w_off, n, HEAD_OUT, n, P, OUT_ADD, nullptr, s, ws_slot));
    FR_TRY(channel_sum(D, HEAD_OUT, P, grad + c.b_off, s));
    // k x k conv: gW[n][ckk] += GH[n][P] * COL[P][ckk]
    FR_TRY(im2col_positions(in.pooled.f(), a.Cin, a.H, a.W, a.k, a.Wo, pos, P, COL, s));
    FR_TRY(gemm_f32(GH, P, 1, COL, ckk, 1, grad + a.w_off, ckk, n, ckk, P, OUT_ADD, nullptr, s, ws_slot));
    return FRCNN_OK;
  }
  h.sp_count = -1; h.sp_pos = nullptr;
  if (!m->head_packs_fresh) {   // dense fallback: bring the heads' input-gradient packs up to date
    FR_TRY(conv_pack_weights_multi(w, (const PackJob*)m->pack_jobs.p + m->n_pack_all + m->n_pack_fwd, m->n_pack_heads,
                                   m->pack_grid_heads, s));
    m->head_packs_fresh = true;
  }
  // 1x1 conv: accGradParameters + updateGradInput
  FR_TRY(conv_wgrad(a.x.f(), c.Cin, c.H, c.W, w + a.a_off, nullptr, h.delta.f(), HEAD_OUT, 1, 0, grad + c.w_off, m->wg_ws.p, m->wg_ws.bytes, s));
  FR_TRY(channel_sum(h.delta.f(), HEAD_OUT, hw1, grad + c.b_off, s));
  double f1 = 2.0 * HEAD_OUT * c.Cin * (double)hw1;
  FR_TRY(conv_igemm(h.delta.f(), HEAD_OUT, c.Ho, c.Wo, nullptr, nullptr, c.wd.f(), nullptr, c.Cin, 1, 0, a.gx.f(),
                    OUT_STORE, f1, s, ws_slot));
  // PReLU backward of the head (+ bias gradient of the k x k conv)
  FR_TRY(act_backward(a.gx.f(), a.x.f(), a.Cout, (long)a.Ho * a.Wo, w + a.a_off, nullptr, a.gx.f(),
                      grad + a.b_off, grad + a.a_off, s));
  FR_TRY(conv_wgrad(in.pooled.f(), a.Cin, a.H, a.W, nullptr, nullptr, a.gx.f(), a.Cout, a.k, 0, grad + a.w_off, m->wg_ws.p, m->wg_ws.bytes, s));
  double f3 = 2.0 * a.Cout * a.Cin * a.k * a.k * (double)a.Ho * a.Wo;
  FR_TRY(conv_igemm(a.gx.f(), a.Cout, a.Ho, a.Wo, nullptr, nullptr, a.wd.f(), nullptr, a.Cin, a.k, a.k - 1,
                    in.gpooled.f(), OUT_ADD, f3, s, ws_slot));  // nngraph fan-out: gradients add up
  return FRCNN_OK;
}

// every anchor net on ONE stream, one after the other (serial mode, and the fallback of frcnn_pnet_backward)
static int backward_heads(frcnn_model* m, const float* w, float* grad, hipStream_t s, int ws_slot) {
  for (auto& h : m->heads) FR_TRY(backward_head(m, h, w, grad, s, ws_slot));
  return FRCNN_OK;
}

// The anchor nets' backward pass off the caller's stream: the side stream holds whatever must precede it (its last
// event = chain_ev); every anchor net with a sparse hint then runs on its own stream beside the others, the dense
// fallback (shared weight-gradient workspace) stays on the side stream.
static int backward_heads_fanout(frcnn_model* m, const float* w, float* grad) {
  FR_TRY(ensure_head_streams(m));
  FR_TRY(fill_zero((char*)m->zero_arena.p + m->delta_bytes, m->gpool_bytes, m->side));
  FR_HIP(hipEventRecord(m->chain_ev, m->side));
  for (auto& h : m->heads) h.gin_recorded = false;
  std::vector<char> own(m->heads.size(), 0);   // (the hint is consumed by backward_head: decide before calling it)
  for (size_t i = 0; i < m->heads.size(); ++i) own[i] = m->heads[i].stream && head_is_sparse(m->heads[i]);
  // A dense-fallback net adds into the pooled-map gradient with plain read-modify-writes and uses the shared weight-gradient
  // workspace, while the sparse nets add into the same map with atomics from their own streams: nothing orders the two
  // groups, so as soon as ONE net is dense every net runs on the side stream, one after the other.
  bool any_dense = false;
  for (size_t i = 0; i < m->heads.size(); ++i) any_dense = any_dense || !head_is_sparse(m->heads[i]);
  // Deterministic mode: the order-independent forms (the gathering col2im, in-order folds) assume ONE writer of the shared
  // pooled-map gradient at a time.
  if (any_dense || deterministic()) std::fill(own.begin(), own.end(), 0);
  for (size_t i = 0; i < m->heads.size(); ++i) {
    Head& h = m->heads[i];
    if (!own[i]) continue;
    FR_HIP(hipStreamWaitEvent(h.stream, m->chain_ev, 0));
    FR_TRY(backward_head(m, h, w, grad, h.stream, head_slot(m, i)));
  }
  for (size_t i = 0; i < m->heads.size(); ++i)
    if (!own[i]) FR_TRY(backward_head(m, m->heads[i], w, grad, m->side, 1));
  return FRCNN_OK;
}

// The forward part a training pass deferred (frcnn_model::heads_deferred), as the dense convolutions after all: every anchor net
// on its own stream behind everything queued on `s` (the pooled maps are final there).
static int heads_forward_dense(frcnn_model* m, const float* w, hipStream_t s) {
  FR_TRY(ensure_side(m));
  FR_TRY(ensure_head_streams(m));
  if (!m->head_x3_fresh) {   // the pass packed nothing of the anchor nets (it meant to run them sparse)
    FR_TRY(refresh_group(m, w, (int)m->blocks.size(), s));
    m->head_x3_fresh = true;
  }
  for (size_t i = 0; i < m->heads.size(); ++i) {
    FR_TRY(fork_to(m, s, head_stream(m, i), 16 + i));
    FR_TRY(head_forward(m, m->heads[i], w, head_stream(m, i), head_slot(m, i)));
  }
  m->heads_deferred = false;
  return FRCNN_OK;
}

static bool heads_all_sparse(const frcnn_model* m) {
  for (auto& h : m->heads) if (!head_is_sparse(h)) return false;
  return !m->heads.empty() && m->heads.size() <= 4;
}

// The job table of the sparse path: the anchor nets that have sampled positions in this pass
static int heads_jobs(frcnn_model* m, const float* w, float* grad, HeadJobs& g) {
  g.n = 0;
  for (auto& h : m->heads) {
    const int P = h.sp_count;
    if (P <= 0) continue;
    Block& in = m->blocks[h.input];
    Conv &a = h.c3, &c = h.c1;
    const size_t n = (size_t)a.Cout, ckk = (size_t)a.Cin * a.k * a.k, cap = SPARSE_MAX_POS;
    FR_TRY(h.spD.ensure((size_t)HEAD_OUT * cap * 4)); FR_TRY(h.spOut.ensure((size_t)HEAD_OUT * cap * 4));
    FR_TRY(h.spHX.ensure(n * cap * 4)); FR_TRY(h.spHY.ensure(n * cap * 4)); FR_TRY(h.spGH.ensure(n * cap * 4));
    FR_TRY(h.spCol.ensure(ckk * cap * 4)); FR_TRY(h.spDX.ensure(ckk * cap * 4));
    HeadJob& j = g.j[g.n++];
    j.Cin = a.Cin; j.H = a.H; j.W = a.W; j.k = a.k; j.Ho = a.Ho; j.Wo = a.Wo; j.n = a.Cout;
    j.P = P; j.pos = h.sp_pos;
    j.in = in.pooled.f(); j.gin = in.gpooled.f();
    j.bias3 = w + a.b_off; j.slope = w + a.a_off; j.bias1 = w + c.b_off;
    j.gbias3 = grad ? grad + a.b_off : nullptr; j.gslope = grad ? grad + a.a_off : nullptr; j.gbias1 = grad ? grad + c.b_off : nullptr;
    j.out = c.x.f(); j.delta = h.delta.f();
    j.COL = h.spCol.f(); j.HX = h.spHX.f(); j.HY = h.spHY.f(); j.OUT = h.spOut.f(); j.D = h.spD.f(); j.GH = h.spGH.f(); j.DX = h.spDX.f();
    // K splits of HX = W COL^T: K = ckk is long, the tile grid (n / 64) x (P / 64) small -- about a thousand blocks in all
    const long tiles = (long)cdiv((int)n, 64) * cdiv(P, 64);
    int splits = (int)std::max<long>(1, std::min<long>(std::min<long>((long)ckk / 256, 64), 256 / tiles));
    const int per = cdiv(cdiv((int)ckk, splits), 32) * 32;
    splits = cdiv((int)ckk, per);
    FR_TRY(h.spSlab.ensure(n * 4096 * 4));   // (splits * P <= 64 * 64 whatever P: allocated once)
    FR_CHECK((size_t)splits * P <= 4096, "heads_jobs: %d K splits of %d positions", splits, P);
    j.hx_slab = h.spSlab.f(); j.hx_splits = splits;
  }
  return FRCNN_OK;
}

// forward at the sampled positions: the output maps are written THERE only (objective.lua:91-140 reads nothing else)
static int heads_sparse_forward(frcnn_model* m, const float* w, const HeadJobs& g, hipStream_t s) {
  if (g.n == 0) return FRCNN_OK;
  FR_TRY(heads_im2col(g, s));
  GemmJob q[4];
  for (int i = 0; i < g.n; ++i) {   // HX[n][P] = W[n][ckk] COL[P][ckk]^T, K-split partial sums
    const HeadJob& j = g.j[i];
    const Head* hd = nullptr;
    for (auto& h : m->heads) if (h.spCol.f() == j.COL) hd = &h;
    const int ckk = j.Cin * j.k * j.k;
    q[i] = GemmJob{w + hd->c3.w_off, (long)ckk, 1, j.COL, 1, (long)ckk, const_cast<float*>(j.hx_slab), (long)j.P, j.n, j.P, ckk, OUT_STORE, j.hx_splits};
    if (j.hx_splits == 1) q[i].splits = 1;
  }
  // (a single split still goes through the slab: heads_bias_act folds `hx_splits` of them)
  for (int i = 0; i < g.n; ++i) if (q[i].splits == 1) { q[i].out_mode = OUT_STORE; }
  FR_TRY(gemm_f32_group(q, g.n, s));
  FR_TRY(heads_bias_act(g, s));
  for (int i = 0; i < g.n; ++i) {   // OUT[18][P] = W1[18][n] HY[n][P]
    const HeadJob& j = g.j[i];
    const Head* hd = nullptr;
    for (auto& h : m->heads) if (h.spCol.f() == j.COL) hd = &h;
    q[i] = GemmJob{w + hd->c1.w_off, (long)j.n, 1, j.HY, (long)j.P, 1, j.OUT, (long)j.P, HEAD_OUT, j.P, j.n, OUT_STORE, 1};
  }
  FR_TRY(gemm_f32_group(q, g.n, s));
  FR_TRY(heads_scatter(g, s));
  return FRCNN_OK;
}

// backward at the sampled positions (the arithmetic of backward_head's sparse branch; HX, HY and COL are the forward pass's)
static int heads_sparse_backward(frcnn_model* m, const float* w, float* grad, const HeadJobs& g, hipStream_t s) {
  if (g.n == 0) return FRCNN_OK;
  auto head_of = [&](const HeadJob& j) -> const Head* {
    for (auto& h : m->heads) if (h.spCol.f() == j.COL) return &h;
    return nullptr;
  };
  FR_TRY(heads_gather_delta(g, s));                    // D, gb1
  GemmJob q[4];
  for (int i = 0; i < g.n; ++i) {                      // GH[n][P] = W1^T[n][18] D[18][P]
    const HeadJob& j = g.j[i]; const Head* hd = head_of(j);
    q[i] = GemmJob{w + hd->c1.w_off, 1, (long)j.n, j.D, (long)j.P, 1, j.GH, (long)j.P, j.n, j.P, HEAD_OUT, OUT_STORE, 1};
  }
  FR_TRY(gemm_f32_group(q, g.n, s));
  FR_TRY(heads_act_backward(g, s));                    // PReLU backward, gb3, gslope
  for (int i = 0; i < g.n; ++i) {                      // DX[P][ckk] = GH^T[P][n] W[n][ckk]
    const HeadJob& j = g.j[i]; const Head* hd = head_of(j);
    const int ckk = j.Cin * j.k * j.k;
    q[i] = GemmJob{j.GH, 1, (long)j.P, w + hd->c3.w_off, (long)ckk, 1, j.DX, (long)ckk, j.P, ckk, j.n, OUT_STORE, 1};
  }
  FR_TRY(gemm_f32_group(q, g.n, s));
  FR_TRY(heads_col2im(g, s));                          // the pooled maps' gradients
  if (!m->heads_gin_ev) FR_HIP(hipEventCreateWithFlags(&m->heads_gin_ev, hipEventDisableTiming));
  FR_HIP(hipEventRecord(m->heads_gin_ev, s));          // what the backbone's backward pass waits for; the parameter gradients follow
  m->heads_gin = true;
  for (int i = 0; i < g.n; ++i) {                      // gW1[18][n] += D[18][P] HY[n][P]^T
    const HeadJob& j = g.j[i]; const Head* hd = head_of(j);
    q[i] = GemmJob{j.D, (long)j.P, 1, j.HY, 1, (long)j.P, grad + hd->c1.w_off, (long)j.n, HEAD_OUT, j.n, j.P, OUT_ADD, 1};
  }
  FR_TRY(gemm_f32_group(q, g.n, s));
  for (int i = 0; i < g.n; ++i) {                      // gW[n][ckk] += GH[n][P] COL[P][ckk]
    const HeadJob& j = g.j[i]; const Head* hd = head_of(j);
    const int ckk = j.Cin * j.k * j.k;
    q[i] = GemmJob{j.GH, (long)j.P, 1, j.COL, (long)ckk, 1, grad + hd->c3.w_off, (long)ckk, j.n, ckk, j.P, OUT_ADD, 1};
  }
  FR_TRY(gemm_f32_group(q, g.n, s));
  for (auto& h : m->heads) { h.sp_count = -1; h.sp_pos = nullptr; }   // one-shot hints
  return FRCNN_OK;
}

int frcnn_pnet_backward_heads_begin(frcnn_model* m, const float* w, float* grad, void* stream) {
  hipStream_t s = S(stream);
  FR_CHECK(m->H > 0 && m->training, "pnet_backward_heads_begin: needs a preceding training-mode forward");
  if (!side_enabled() || m->heads_begun) return FRCNN_OK;   // frcnn_pnet_backward does everything
  if (m->heads_deferred) FR_TRY(heads_forward_dense(m, w, s));   // (this entry point has no sparse forward: the outputs were the caller's to read)
  FR_TRY(fork_side(m, s, m->blocks.size() + 1));             // delta_outputs[1..nheads] are final on s
  FR_TRY(join_heads(m, m->side));                            // (anchor nets of an asynchronous forward still in flight)
  FR_TRY(backward_heads_fanout(m, w, grad));
  m->heads_begun = true;
  return FRCNN_OK;
}

int frcnn_pnet_anchor_loss_begin(frcnn_model* m, const float* w, float* grad, const int* ex_idx, const double* ex_anchor,
                                 const double* ex_roi, const int* ex_class, int npos, int nneg, int bgclass,
                                 double* ex_loss, float* crtarget, float* cctarget, double* acc, void* stream) {
  hipStream_t s = S(stream);
  FR_CHECK(m->H > 0 && m->training, "pnet_anchor_loss_begin: needs a preceding training-mode forward");
  FR_CHECK(m->heads.size() <= 4, "pnet_anchor_loss_begin: at most 4 anchor nets");
  FR_CHECK(!m->heads_begun, "pnet_anchor_loss_begin: the anchor nets' backward pass has already been started");
  RpnLayers L;
  float* deltas[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int l = 0; l < 4; ++l) {
    const bool have = l < (int)m->heads.size();
    L.map[l] = have ? m->heads[l].c1.x.f() : nullptr;
    L.H[l] = have ? m->heads[l].c1.Ho : 0; L.W[l] = have ? m->heads[l].c1.Wo : 0;
    deltas[l] = have ? m->heads[l].delta.f() : nullptr;
  }
  const int E = npos + nneg;
  if (!side_enabled()) {   // serial: the losses on the caller's stream, the backward part left to frcnn_pnet_backward
    FR_TRY(rpn_loss(L, deltas, ex_idx, ex_anchor, ex_roi, ex_class, npos, nneg, bgclass, ex_loss, crtarget, cctarget, s));
    FR_TRY(loss_accumulate(ex_loss, E, acc, s));
    return FRCNN_OK;
  }
  if (m->heads_deferred && heads_all_sparse(m)) {
    // The sparse path (heads.hip): forward at the sampled positions, the losses, the backward part -- one chain of grouped
    // launches on the side stream, behind everything queued on `s` (pooled maps, example tables, zeroed delta buffers)
    HeadJobs g;
    FR_TRY(heads_jobs(m, w, grad, g));
    FR_TRY(fork_side(m, s, m->blocks.size() + 1));
    FR_TRY(heads_sparse_forward(m, w, g, m->side));
    m->heads_deferred = false; m->heads_sparse_fwd = true;
    FR_TRY(rpn_loss(L, deltas, ex_idx, ex_anchor, ex_roi, ex_class, npos, nneg, bgclass, ex_loss, crtarget, cctarget, m->side));
    FR_TRY(loss_accumulate(ex_loss, E, acc, m->side));
    FR_HIP(hipEventRecord(m->loss_ev, m->side));
    m->loss_pending = true;
    FR_TRY(fill_zero((char*)m->zero_arena.p + m->delta_bytes, m->gpool_bytes, m->side));
    FR_TRY(heads_sparse_backward(m, w, grad, g, m->side));
    m->heads_begun = true;
    return FRCNN_OK;
  }
  if (m->heads_deferred) FR_TRY(heads_forward_dense(m, w, s));   // more positions than the sparse path takes: the dense convolutions after all
  FR_TRY(fork_side(m, s, m->blocks.size() + 1));   // example tables and zeroed delta buffers are final on s
  FR_TRY(join_heads(m, m->side));                  // the anchor nets' outputs (each on its own stream) are final
  FR_TRY(rpn_loss(L, deltas, ex_idx, ex_anchor, ex_roi, ex_class, npos, nneg, bgclass, ex_loss, crtarget, cctarget, m->side));
  FR_TRY(loss_accumulate(ex_loss, E, acc, m->side));
  FR_HIP(hipEventRecord(m->loss_ev, m->side));
  m->loss_pending = true;
  FR_TRY(backward_heads_fanout(m, w, grad));
  m->heads_begun = true;
  return FRCNN_OK;
}

int frcnn_pnet_anchor_loss_wait(frcnn_model* m, void* stream) {
  if (m->loss_pending) {
    FR_HIP(hipStreamWaitEvent(S(stream), m->loss_ev, 0));
    m->loss_pending = false;
  }
  return FRCNN_OK;
}

int frcnn_pnet_backward_heads_join(frcnn_model* m, void* stream, int* joined) {
  hipStream_t s = S(stream);
  if (joined) *joined = 0;
  if (!m->heads_begun) return FRCNN_OK;   // nothing was started: frcnn_pnet_backward computes the anchor nets' part
  if (!m->heads_joined) {
    FR_TRY(join_heads(m, s));
    FR_HIP(hipEventRecord(m->join_ev, m->side));
    FR_HIP(hipStreamWaitEvent(s, m->join_ev, 0));
    m->heads_joined = true;   // frcnn_pnet_backward will not wait again
  }
  if (joined) *joined = 1;
  return FRCNN_OK;
}

// After frcnn_pnet_backward has been queued: `stream` waits until the caller's stream has joined the anchor nets and begun the
// backbone's backward pass -- the stretch of the step that is bound by the matrix cores, beside which bandwidth-bound work
// (the update of slices that are final already) costs least.  Everything the caller's stream ran before it is final too.
int frcnn_pnet_wait_backward_begun(frcnn_model* m, void* stream) {
  FR_CHECK(m->update_armed && m->bwd_ev && m->block_ev_valid,
           "pnet_wait_backward_begun: call frcnn_model_update_stream before the pass and frcnn_pnet_backward first");
  FR_HIP(hipStreamWaitEvent(S(stream), m->bwd_ev, 0));
  return FRCNN_OK;
}

// After frcnn_pnet_backward has been queued: `stream` waits until the anchor nets' whole backward pass -- their parameter
// gradients included, which may still be running beside the backbone's pass -- is over: their slice may then be updated.
int frcnn_pnet_wait_heads_done(frcnn_model* m, void* stream) {
  FR_CHECK(m->block_ev_valid, "pnet_wait_heads_done: call frcnn_pnet_backward first");
  for (auto& h : m->heads)
    if (h.stream && h.done) FR_HIP(hipStreamWaitEvent(S(stream), h.done, 0));
  if (m->side) {   // (the sparse training path runs the anchor nets on the side stream: everything queued there so far)
    FR_TRY(ensure_update_stream(m));
    FR_TRY(chain(m->side, S(stream), m->upd_ev));
  }
  return FRCNN_OK;
}

// Stream `stream` waits until block `block`'s slice of the flat vectors may be UPDATED: its parameter gradients are final (block_ev)
// and the caller's stream has queued -- hence, in stream order, finished -- the last launch that reads the block's weights, its
// packs or its weight-magnitude scalars (the input-gradient launch of the block's first convolution).
int frcnn_pnet_wait_block_done(frcnn_model* m, int block, void* stream) {
  FR_CHECK(block >= 1 && block <= (int)m->blocks.size(), "pnet_wait_block_done: block %d out of range", block);
  FR_CHECK(m->update_armed && m->block_ev_valid && (size_t)block <= m->block_ev.size() && (size_t)block <= m->block_rd_ev.size(),
           "pnet_wait_block_done: call frcnn_model_update_stream before the pass and frcnn_pnet_backward first");
  FR_HIP(hipStreamWaitEvent(S(stream), m->block_ev[block - 1], 0));
  FR_HIP(hipStreamWaitEvent(S(stream), m->block_rd_ev[block - 1], 0));
  return FRCNN_OK;
}

// Renews the training packs of one owner (group = 0-based backbone block, or the number of blocks for the anchor nets) from the
// weight vector w on `stream`: what the prologue of frcnn_pnet_forward does for the whole model, for the slice whose update has
// just been queued on the same stream.  When every group has been renewed from the vector the next training-mode forward is
// given, that forward skips its prologue.  The caller promises that nothing writes w between this call and that forward except
// updates followed by their own refresh; frcnn_pnet_invalidate_packs withdraws the promise.
static int refresh_group(frcnn_model* m, const float* w, int group, hipStream_t s) {
  const bool f16 = get_x3_f16() && m->n_amax_jobs > 0;
  const frcnn_model::PackGroup& G = m->groups[group];
  if (G.pk_n) FR_TRY(conv_pack_weights_multi(w, (const PackJob*)m->pack_jobs.p + G.pk_off, G.pk_n, G.pk_grid, s));
  if (f16 && G.am_n) FR_TRY(tensor_absmax_multi(w, (const AmaxJob*)m->amax_jobs.p + G.am_off, G.am_n, G.am_grid, s));
  if (G.x3_n) FR_TRY(conv_x3_pack_multi(w, (const PackXJob*)m->x3_jobs.p + (f16 ? G.x3_off16 : G.x3_off), G.x3_n, G.x3_grid, s));
  return FRCNN_OK;
}
int frcnn_pnet_refresh_packs(frcnn_model* m, const float* w, int group, void* stream) {
  FR_CHECK(m->H > 0, "pnet_refresh_packs: call frcnn_pnet_forward first");
  FR_CHECK(group >= 0 && group < (int)m->groups.size(), "pnet_refresh_packs: group %d out of range", group);
  const bool f16 = get_x3_f16() && m->n_amax_jobs > 0;
  if (m->fresh_mask && (m->fresh_w != w || m->fresh_f16 != f16)) m->fresh_mask = 0;
  FR_TRY(refresh_group(m, w, group, S(stream)));
  if (group == (int)m->blocks.size()) m->head_x3_fresh = true;
  m->fresh_mask |= 1u << group; m->fresh_w = w; m->fresh_f16 = f16;
  return FRCNN_OK;
}

int frcnn_pnet_invalidate_packs(frcnn_model* m) {
  m->fresh_mask = 0;
  return FRCNN_OK;
}

int frcnn_pnet_wait_block_gradients(frcnn_model* m, int block, void* stream) {
  FR_CHECK(block >= 1 && block <= (int)m->blocks.size(), "pnet_wait_block_gradients: block %d out of range", block);
  FR_CHECK(m->block_ev_valid && (size_t)block <= m->block_ev.size(), "pnet_wait_block_gradients: call frcnn_pnet_backward first");
  FR_HIP(hipStreamWaitEvent(S(stream), m->block_ev[block - 1], 0));
  return FRCNN_OK;
}

static int record_block_read(frcnn_model* m, int b, hipStream_t s) {
  if (!m->update_armed) return FRCNN_OK;
  while (m->block_rd_ev.size() < m->blocks.size()) {
    hipEvent_t e;
    FR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->block_rd_ev.push_back(e);
  }
  FR_HIP(hipEventRecord(m->block_rd_ev[b], s));
  return FRCNN_OK;
}

int frcnn_pnet_backward(frcnn_model* m, const float* w, float* grad, void* stream) {
  hipStream_t s = S(stream);
  FR_CHECK(m->H > 0 && m->training, "pnet_backward: needs a preceding training-mode forward "
                                    "(nn.SpatialDropout: backprop only defined while training)");
  const int nb = (int)m->blocks.size();
  bool heads_tail = false;   // anchor nets still computing their parameter gradients on their own streams: joined at the end
  if (m->heads_begun) {   // started by frcnn_pnet_backward_heads_begin: wait for the side stream
    if (!m->heads_joined) {
      // What the backbone's pass needs from an anchor net is its contribution to the pooled map's gradient; a net on a stream
      // of its own marks that point (Head::gin_done) and goes on with its parameter gradients beside the backbone's pass.
      static const bool partial = !(getenv("FRCNN_HEADS_PARTIAL_JOIN") && atoi(getenv("FRCNN_HEADS_PARTIAL_JOIN")) == 0);
      if (partial && m->heads_gin) {   // the sparse path's one chain on the side stream: its input-gradient part, the rest at the end
        FR_HIP(hipStreamWaitEvent(s, m->heads_gin_ev, 0));
        heads_tail = true;
      }
      for (auto& h : m->heads) {
        if (!h.stream || m->heads_gin) continue;
        if (partial && h.gin_recorded) {
          FR_HIP(hipStreamWaitEvent(s, h.gin_done, 0));
          heads_tail = true;
        } else {
          FR_HIP(hipEventRecord(h.done, h.stream));
          FR_HIP(hipStreamWaitEvent(s, h.done, 0));
        }
      }
      if (!(partial && m->heads_gin)) {
        FR_HIP(hipEventRecord(m->join_ev, m->side));
        FR_HIP(hipStreamWaitEvent(s, m->join_ev, 0));
      }
    }
    m->heads_begun = false; m->heads_joined = false; m->side_busy = false;
  } else {
    if (m->heads_deferred) {   // nobody asked for the anchor nets' outputs; their backward part below needs the forward part unless no net has a position
      bool any = false;
      for (auto& h : m->heads) any = any || h.sp_count != 0;
      if (any) FR_TRY(heads_forward_dense(m, w, s));
      m->heads_deferred = false;
    }
    FR_TRY(join_side(m, s));   // anchor nets of frcnn_pnet_forward_async_heads still in flight (image without examples)
    FR_TRY(fill_zero((char*)m->zero_arena.p + m->delta_bytes, m->gpool_bytes, s));
    FR_TRY(backward_heads(m, w, grad, s, 0));
  }
  {  // output nheads+1 is the last pooled map itself (model_utilities.lua:55)
    Block& last = m->blocks.back();
    FR_TRY(add_inplace(last.gpooled.f(), m->delta_last.f(), (long)m->d.filters[nb - 1] * last.Hp * last.Wp, s));
  }
  // from here on the caller's stream is busy with matrix-core work for the rest of the pass (frcnn_pnet_wait_backward_begun)
  if (m->update_armed) {
    if (!m->bwd_ev) FR_HIP(hipEventCreateWithFlags(&m->bwd_ev, hipEventDisableTiming));
    FR_HIP(hipEventRecord(m->bwd_ev, s));
  }
  // The weight gradient of a layer and the input gradient that feeds the next act_backward are independent:
  // accGradParameters goes to a side stream, so its blocks fill the CUs that the tail of the updateGradInput
  // kernel (one wave of blocks, retiring unevenly) leaves idle, and the ~4 us dispatch gaps of one chain
  // are covered by the other.
  const bool use_side = side_enabled();
  if (use_side) FR_TRY(ensure_side(m));
  hipStream_t ws = use_side ? m->side : s;
  size_t n_fork = 0;
  // Fused activation backward (round 4): the input-gradient launch of convolution st writes its result THROUGH the backward of
  // convolution st - 1's PReLU / SpatialDropout (conv_x3's X3PostAct: epilogue, or the fold of its split-K slabs) and adds the
  // slope gradient; the bias gradient of st - 1 rides on its weight-gradient launch (conv_wgrad's gbias).  act_backward --
  // a read-read-write pass over the tensor on the dependent chain -- is then not launched for st - 1.  (Not in deterministic
  // mode: the sums leave through atomics.)
  const bool fuse_act = !deterministic() && !(getenv("FRCNN_FUSE_ACT") && atoi(getenv("FRCNN_FUSE_ACT")) == 0);
  bool act_done = false;   // the gradient tensor of the convolution being visited already went through its activation
  for (int b = nb - 1; b >= 0; --b) {
    Block& blk = m->blocks[b];
    for (int st = blk.nconv - 1; st >= 0; --st) {
      Conv& c = m->convs[blk.first_conv + st];
      const float* scale = (st == 0 && blk.has_drop) ? blk.scale.f() : nullptr;
      const bool fused_here = act_done;
      act_done = false;
      // the very first convolution when it is its block's only one: its input gradient is not needed, so the pooling + PReLU
      // backward is computed inside its weight-gradient launch and the full-resolution gradient never exists
      static const bool first_pooled_on = !(getenv("FRCNN_FIRST_POOLED") && atoi(getenv("FRCNN_FIRST_POOLED")) == 0);
      const bool first_pooled = fuse_act && first_pooled_on && b == 0 && st == 0 && blk.nconv == 1 && scale == nullptr &&
                                conv_wgrad_first_pooled_eligible(c.Cin, c.Cout, c.k, c.Wo);
      if (first_pooled) {
        // (on the caller's stream with the slab workspace of its own, like the unfused first layer: see on_caller below)
        FR_TRY(conv_wgrad_first_pooled(m->img.f(), c.Cin, c.H, c.W, blk.gpooled.f(), (const unsigned char*)blk.pidx.p, c.x.f(),
                                       w + c.a_off, c.Cout, c.pad, grad + c.w_off, grad + c.b_off, grad + c.a_off,
                                       m->wg_ws_first.p, m->wg_ws_first.bytes, s));
        while (m->block_ev.size() < (size_t)nb) {
          hipEvent_t e;
          FR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          m->block_ev.push_back(e);
        }
        FR_HIP(hipEventRecord(m->block_ev[b], s));
        FR_TRY(record_block_read(m, b, s));
        break;
      }
      if (fused_here) {
        // (nothing: c.gx is final; its bias gradient comes with the weight gradient below)
      } else if (st == blk.nconv - 1) {
        FR_TRY(maxpool_act_backward(blk.gpooled.f(), (const unsigned char*)blk.pidx.p, c.x.f(), c.Cout, c.Ho, c.Wo,
                                    w + c.a_off, scale, c.gx.f(), grad + c.b_off, grad + c.a_off, s,
                                    (c.x_d && m->f16_packed) ? m->rec(c.am + 1) : nullptr));
      } else {
        FR_TRY(act_backward(c.gx.f(), c.x.f(), c.Cout, (long)c.Ho * c.Wo, w + c.a_off, scale, c.gx.f(),
                            grad + c.b_off, grad + c.a_off, s, (c.x_d && m->f16_packed) ? m->rec(c.am + 1) : nullptr));
      }
      // accGradParameters: the input is the previous conv's x (activation fused) or a pooled map / image
      const float* in; const float* in_slope = nullptr; const float* in_scale = nullptr;
      if (st > 0) {
        Conv& pc = m->convs[blk.first_conv + st - 1];
        in = pc.x.f(); in_slope = w + pc.a_off;
        in_scale = (st - 1 == 0 && blk.has_drop) ? blk.scale.f() : nullptr;
      } else {
        in = b == 0 ? m->img.f() : m->blocks[b - 1].pooled.f();
      }
      // compact block (Block::dc_on): the second convolution's weight gradient is computed for the kept input channels (its
      // input IS the compact tensor; their dropout scale is 1), the first one's for the kept filters (its output gradient is
      // compact); the fold scatters both into the full tensors, the bias gradient follows the filter map
      const bool dc = blk.dc_on && st <= 1;
      int wg_cin = c.Cin, wg_o = c.Cout;
      WgradMap wmap;
      if (dc) {
        wmap.Cfull = c.Cin;
        if (st == 1) { wg_cin = blk.nkM; wmap.cmap = blk.dc_idx; in_scale = nullptr; }
        else { wg_o = blk.nkM; wmap.omap = blk.dc_idx; }
      }
      // fp16 form of the weight-gradient launch: the records of both tensors exist when the forward launch that wrote `in` and
      // the backward launch that wrote c.gx kept them (the same two tensors feed c's forward and c's input gradient)
      const float* wa_in = nullptr; const float* wa_g = nullptr;
      if (m->f16_packed && c.x_f && c.x_d && (st > 0 || b > 0)) {
        wa_in = st > 0 ? m->rec(m->convs[blk.first_conv + st - 1].am) : m->rec(m->blocks[b - 1].am);
        wa_g = m->rec(c.am + 1);
      }
      // The first layer's weight gradient ends the pass and the caller's stream has nothing left to do (no input gradient for
      // the image): it runs THERE, with a slab workspace of its own, beside the side stream's last launches instead of
      // behind them (the optimiser waited ~80 us for the side stream's tail).
      const bool on_caller = use_side && b == 0 && st == 0;
      if (use_side && !on_caller) FR_TRY(fork_side(m, s, n_fork++));   // c.gx is final here
      if (on_caller) {
        FR_TRY(conv_wgrad(in, wg_cin, c.H, c.W, in_slope, in_scale, c.gx.f(), wg_o, c.k, c.pad, grad + c.w_off, m->wg_ws_first.p,
                          m->wg_ws_first.bytes, s, fused_here ? grad + c.b_off : nullptr, wa_in, wa_g, dc ? &wmap : nullptr));
        FR_HIP(hipEventRecord(m->join_ev, ws));          // (block 0's other convolutions, if any, are on the side stream)
        FR_HIP(hipStreamWaitEvent(s, m->join_ev, 0));
      } else {
        FR_TRY(conv_wgrad(in, wg_cin, c.H, c.W, in_slope, in_scale, c.gx.f(), wg_o, c.k, c.pad, grad + c.w_off, m->wg_ws.p, m->wg_ws.bytes, ws,
                          fused_here ? grad + c.b_off : nullptr, wa_in, wa_g, dc ? &wmap : nullptr));
      }
      if (st == 0) {   // every gradient of block b's parameters is final once this launch has run (its fork also
                       // covers the bias / slope sums that act_backward accumulates on the caller's stream)
        while (m->block_ev.size() < (size_t)nb) {
          hipEvent_t e;
          FR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          m->block_ev.push_back(e);
        }
        FR_HIP(hipEventRecord(m->block_ev[b], on_caller ? s : ws));
      }
      if (b == 0 && st == 0) {   // gradInput of the first conv is unused (objective.lua:189)
        FR_TRY(record_block_read(m, b, s));
        break;
      }
      float* gin = st > 0 ? m->convs[blk.first_conv + st - 1].gx.f() : m->blocks[b - 1].gpooled.f();
      const int gmode = st > 0 ? OUT_STORE : OUT_ADD;
      // fp16 form: the magnitude of c.gx was recorded by the launch that finished it (the pooling / activation backward above,
      // or the previous input-gradient launch through its fused activation backward)
      const float* ag = nullptr;
      const float* aw = nullptr;
      if (c.x_d && m->f16_packed) { ag = m->rec(c.am + 1); aw = m->amax_ws.f() + c.am; }
      // compact block: the second convolution's input gradient is computed for the kept channels only and stored compact (it
      // passes through the first convolution's PReLU; the kept channels' dropout scale is 1); the first convolution's reads
      // that compact gradient: K = the kept filters
      const int dg_k = (dc && st == 0) ? blk.nkK : c.Cout, dg_m = (dc && st == 1) ? blk.nkM : c.Cin;
      const double fl = 2.0 * dg_k * dg_m * c.k * c.k * (double)c.Ho * c.Wo;
      if (c.x_d && st > 0 && fuse_act && c.k == 3) {
        Conv& pc = m->convs[blk.first_conv + st - 1];
        X3PostAct post{pc.x.f(), w + pc.a_off, (st - 1 == 0 && blk.has_drop && !dc) ? blk.scale.f() : nullptr, grad + pc.a_off};
        FR_TRY(conv_x3(c.gx.f(), dg_k, c.Ho, c.Wo, nullptr, nullptr, c.wxd.p, nullptr, dg_m, c.k, c.k - 1 - c.pad, gin, gmode, fl, s, 0, &post, ag, aw,
                       (pc.x_d && m->f16_packed) ? m->rec(pc.am + 1) : nullptr));
        act_done = true;
      } else if (c.x_d)
        FR_TRY(conv_x3(c.gx.f(), dg_k, c.Ho, c.Wo, nullptr, nullptr, c.wxd.p, nullptr, dg_m, c.k, c.k - 1 - c.pad, gin, gmode, fl, s, 0, nullptr, ag, aw));
      else
        FR_TRY(conv_igemm(c.gx.f(), c.Cout, c.Ho, c.Wo, nullptr, nullptr, c.wd.f(), nullptr, c.Cin, c.k,
                          c.k - 1 - c.pad, gin, gmode, fl, s));
      if (st == 0) FR_TRY(record_block_read(m, b, s));   // the block's weights, packs and weight magnitudes have had their last reader
    }
  }
  m->block_ev_valid = true;
  if (heads_tail && !m->heads_gin) FR_TRY(join_heads(m, s));   // the anchor nets' parameter gradients (see above; the sparse path's are on the side stream, joined below)
  FR_TRY(cw_join(m, s));   // the classification net's weight gradients (frcnn_cnet_backward) belong to the same gradient vector
  if (use_side) {   // the caller's stream continues after every weight gradient has landed
    FR_HIP(hipEventRecord(m->join_ev, ws));
    FR_HIP(hipStreamWaitEvent(s, m->join_ev, 0));
    m->side_busy = false;
  }
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------ cnet
// the row-wise layers fused with the folds of the products around them (cnet.hip: "fused forms"); FRCNN_CNET_FUSE=0: one launch each
static bool cnet_fuse() {
  const char* e = getenv("FRCNN_CNET_FUSE");   // (read per call: the tests switch it)
  return !(e && atoi(e) == 0);
}
// the classification net's large products in the two-plane fp16 form: with option x3_f16 (FRCNN_GEMM_F16=0: these stay three-plane)
static bool gemm_f16_on() {
  static const bool env_on = !(getenv("FRCNN_GEMM_F16") && atoi(getenv("FRCNN_GEMM_F16")) == 0);
  return env_on && get_x3_f16();
}
static int ensure_cnet(frcnn_model* m, int R) {
  for (auto& L : m->cls) {
    size_t n = (size_t)R * L.n * 4;
    FR_TRY(L.lin.ensure(n));
    if (L.bn) { FR_TRY(L.pre.ensure(n)); FR_TRY(L.xhat.ensure(n)); FR_TRY(L.invstd.ensure((size_t)L.n * 4)); }
    FR_TRY(L.post.ensure(n));
    FR_TRY(L.g.ensure(n));
    FR_TRY(L.gin.ensure((size_t)R * L.in * 4));
    if (L.p_drop > 0.f) FR_TRY(L.mask.ensure(n));
    L.x_form = 0;
    for (int role : {1, 2, 4}) if (linear_x_eligible(role, R, L.in, L.n)) L.x_form |= role;
    const size_t Rp = (size_t)linear_x_rows_padded(R);
    if (L.x_form & 1) FR_TRY(L.xp.ensure((size_t)3 * R * L.in * 2));
    if (L.x_form & 2) FR_TRY(L.gp.ensure((size_t)3 * R * L.n * 2));
    if (L.x_form & 4) { FR_TRY(L.xpT.ensure((size_t)3 * L.in * Rp * 2)); FR_TRY(L.gpT.ensure((size_t)3 * L.n * Rp * 2)); }
    if (L.x_form & 7) FR_TRY(L.am.ensure((size_t)5 * AMAX_REC * 4));   // records: input, output gradient, weights; the weight-gradient stream's own two
  }
  int nc = m->d.class_count + 1;
  int nf = m->cls.empty() ? m->D : m->cls.back().n;
  FR_TRY(m->feat_g.ensure((size_t)R * nf * 4));
  FR_TRY(m->logits.ensure((size_t)R * nc * 4));
  FR_TRY(m->lsm.ensure((size_t)R * nc * 4));
  FR_TRY(m->glog.ensure((size_t)R * nc * 4));
  return FRCNN_OK;
}

int frcnn_cnet_forward(frcnn_model* m, const float* weights, const float* x, int R, int training,
                       const float* const* drop_masks, unsigned long long seed, float* bn_running,
                       float* bbox_out, float* cls_out, void* stream) {
  hipStream_t s = S(stream);
  FR_CHECK(R > 0, "cnet_forward: empty batch");
  FR_TRY(cw_join(m, s));   // (weight gradients of a previous backward pass still read the buffers written below)
  FR_TRY(ensure_cnet(m, R));
  m->R = R; m->cnet_x = x; m->training = training;
  const float* w = weights;
  const float* cur = x;
  float* bnr = bn_running;
  for (size_t l = 0; l < m->cls.size(); ++l) {
    ClsLayer& L = m->cls[l];
    GemmFold fold;   // (fused: the layer's row-wise kernel folds the product's split-K slabs itself)
    GemmFold* defer = cnet_fuse() ? &fold : nullptr;
    if ((L.x_form & 3) && gemm_f16_on()) {   // two-plane fp16 form: the weight matrix's magnitude (forward and input gradient share it)
      float* rw = L.am.f() + 2 * AMAX_REC;
      const bool have = !training && g_static_weights && L.am_w_gen == g_static_gen && L.am_w_of == w;
      if (!have) FR_TRY(tensor_absmax(w + L.w_off, (long)L.n * L.in, rw, s));
      L.am_w_of = w; L.am_w_gen = training ? -1 : g_static_gen;
    } else {
      L.am_w_of = nullptr;   // no record from THIS forward pass: the input-gradient product must not scale by an older step's
    }
    if ((L.x_form & 1) && gemm_f16_on()) {   // two fp16 planes of the input, scaled by its largest magnitude
      float* rx = L.am.f();
      FR_TRY(tensor_absmax(cur, (long)R * L.in, rx, s));
      FR_TRY(split_planes(cur, R, L.in, L.xp.p, nullptr, s, rx));
      FR_TRY(linear_x_forward(L.xp.p, R, L.in, w + L.w_off, w + L.b_off, L.n, L.lin.f(), s, 0, defer, rx, rx + 2 * AMAX_REC));
    } else if (L.x_form & 1) {   // split-bf16 operand form: the input's planes once
      FR_TRY(split_planes(cur, R, L.in, L.xp.p, nullptr, s));
      FR_TRY(linear_x_forward(L.xp.p, R, L.in, w + L.w_off, w + L.b_off, L.n, L.lin.f(), s, 0, defer));
    } else {
      FR_TRY(gemm_f32(cur, L.in, 1, w + L.w_off, 1, L.in, L.lin.f(), L.n, R, L.n, L.in, OUT_STORE, w + L.b_off, s, 0, defer));
    }
    if (L.bn) FR_CHECK(training || bnr, "cnet_forward: evaluate mode needs bn_running");
    const float* mask = nullptr;
    float inv_keep = 1.f;
    bool draw = false;
    if (training && L.p_drop > 0.f) {  // nn.Dropout v2: train = mask/(1-p), evaluate = identity [ext]
      inv_keep = 1.0f / (1.0f - L.p_drop);
      if (drop_masks && drop_masks[l])
        FR_HIP(hipMemcpyAsync(L.mask.p, drop_masks[l], (size_t)R * L.n * 4, hipMemcpyDeviceToDevice, s));
      else
        draw = true;   // mask drawn inside the activation kernel
      mask = L.mask.f();
    }
    if (defer) {   // fold -> [BatchNormalization ->] PReLU -> Dropout: one launch
      FR_TRY(cnet_act_forward(L.lin.f(), fold, R, L.n, L.bn ? w + L.bnw_off : nullptr, L.bn ? w + L.bnb_off : nullptr, L.bn ? bnr : nullptr,
                              training, L.lin.f(), L.xhat.f(), L.invstd.f(), L.pre.f(), w + L.a_off, const_cast<float*>(mask), draw,
                              inv_keep, L.p_drop, seed * 977 + l + 17, L.post.f(), s));
      if (L.bn && bnr) bnr += 2 * L.n;
      cur = L.post.f();
      continue;
    }
    const float* pre = L.lin.f();
    if (L.bn) {
      FR_TRY(bn_forward(L.lin.f(), R, L.n, w + L.bnw_off, w + L.bnb_off, bnr, training, L.xhat.f(), L.invstd.f(),
                        L.pre.f(), s));
      if (bnr) bnr += 2 * L.n;
      pre = L.pre.f();
    }
    if (draw)
      FR_TRY(prelu_dropout_forward_gen(pre, (long)R * L.n, w + L.a_off, L.mask.f(), L.p_drop, seed * 977 + l + 17,
                                       L.post.f(), s));
    else
      FR_TRY(prelu_dropout_forward(pre, (long)R * L.n, w + L.a_off, mask, inv_keep, L.post.f(), s));
    cur = L.post.f();
  }
  const int nf = m->cls.empty() ? m->D : m->cls.back().n;
  const int nc = m->d.class_count + 1;
  if (cnet_fuse() && cnet_heads_fused_eligible(nf, nc)) {   // both heads, nn.LogSoftMax and the copy to the caller's tensor: one launch
    return cnet_heads_forward(cur, R, nf, w + m->bbox_w_off, w + m->bbox_b_off, w + m->clsw_off, w + m->clsb_off, nc, bbox_out,
                              m->logits.f(), m->lsm.f(), cls_out, s);
  }
  FR_TRY(gemm_f32(cur, nf, 1, w + m->bbox_w_off, 1, nf, bbox_out, 4, R, 4, nf, OUT_STORE, w + m->bbox_b_off, s));
  if (cnet_fuse()) {   // the class head's fold, nn.LogSoftMax and the copy to the caller's tensor: one launch
    GemmFold fold;
    FR_TRY(gemm_f32(cur, nf, 1, w + m->clsw_off, 1, nf, m->logits.f(), nc, R, nc, nf, OUT_STORE, w + m->clsb_off, s, 0, &fold));
    FR_TRY(log_softmax_rows_fold(m->logits.f(), fold, R, nc, m->lsm.f(), cls_out, s));
    return FRCNN_OK;
  }
  FR_TRY(gemm_f32(cur, nf, 1, w + m->clsw_off, 1, nf, m->logits.f(), nc, R, nc, nf, OUT_STORE, w + m->clsb_off, s));
  FR_TRY(log_softmax_rows(m->logits.f(), R, nc, m->lsm.f(), s));
  FR_HIP(hipMemcpyAsync(cls_out, m->lsm.p, (size_t)R * nc * 4, hipMemcpyDeviceToDevice, s));
  return FRCNN_OK;
}

int frcnn_cnet_backward(frcnn_model* m, const float* weights, const float* g_bbox, const float* g_cls, float* gx,
                        float* grad, void* stream) {
  hipStream_t s = S(stream);
  FR_CHECK(m->R > 0, "cnet_backward: call frcnn_cnet_forward first");
  const int R = m->R;
  const float* w = weights;
  const int nf = m->cls.empty() ? m->D : m->cls.back().n;
  const int nc = m->d.class_count + 1;
  const float* feat = m->cls.empty() ? m->cnet_x : m->cls.back().post.f();
  // `s` carries the input-gradient chain (what roi_pool_backward and the backbone wait for); the weight-gradient products and
  // bias sums go to `ws`: a stream of their own (see g_cnet_wgrad_async), or `s` itself
  FR_TRY(cw_join(m, s));
  const bool async = g_cnet_wgrad_async && !deterministic();
  if (async && !m->cw) {
    FR_TRY(pool_stream(1, &m->cw));
    FR_HIP(hipEventCreateWithFlags(&m->cw_done, hipEventDisableTiming));
  }
  while (async && m->cw_ev.size() < m->cls.size() + 1) {
    hipEvent_t e;
    FR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->cw_ev.push_back(e);
  }
  hipStream_t ws = async ? m->cw : s;
  const int wslot = async ? 7 : 0;     // split-K workspace of the products on `ws`
  size_t nfork = 0;
  auto fork = [&]() -> int {           // everything queued on `s` so far is final for `ws`
    if (!async) return FRCNN_OK;
    FR_HIP(hipEventRecord(m->cw_ev[nfork], s));
    FR_HIP(hipStreamWaitEvent(ws, m->cw_ev[nfork], 0));
    ++nfork;
    return FRCNN_OK;
  };
  // heads: LogSoftMax backward first, then the two input gradients on the chain, the two weight gradients beside it
  bool top_act_done = false;   // the last hidden layer's Dropout + PReLU backward was applied by the heads' launch (L.g is final)
  if (cnet_fuse() && cnet_heads_fused_eligible(nf, nc)) {   // LogSoftMax backward + both input gradients: one launch
    HeadsPostAct post;
    float* dst = m->feat_g.f();
    if (!m->cls.empty() && !m->cls.back().bn && !deterministic()) {
      ClsLayer& T = m->cls.back();
      const bool drop = m->training && T.p_drop > 0.f;
      post.pre = T.lin.f(); post.mask = drop ? T.mask.f() : nullptr; post.inv_keep = drop ? 1.0f / (1.0f - T.p_drop) : 1.f;
      post.slope = w + T.a_off; post.gslope = grad + T.a_off;
      dst = T.g.f();
      top_act_done = true;
    }
    FR_TRY(cnet_heads_backward(g_bbox, g_cls, m->lsm.f(), R, nf, w + m->bbox_w_off, w + m->clsw_off, nc, m->glog.f(), dst, s,
                               top_act_done ? &post : nullptr));
    FR_TRY(fork());
  } else {
    FR_TRY(log_softmax_backward(g_cls, m->lsm.f(), R, nc, m->glog.f(), s));
    FR_TRY(fork());
    FR_TRY(gemm_f32(g_bbox, 4, 1, w + m->bbox_w_off, nf, 1, m->feat_g.f(), nf, R, nf, 4, OUT_STORE, nullptr, s));
    FR_TRY(gemm_f32(m->glog.f(), nc, 1, w + m->clsw_off, nf, 1, m->feat_g.f(), nf, R, nf, nc, OUT_ADD, nullptr, s));
  }
  FR_TRY(gemm_f32(g_bbox, 1, 4, feat, nf, 1, grad + m->bbox_w_off, nf, 4, nf, R, OUT_ADD, nullptr, ws, wslot));
  FR_TRY(channel_sum_cols(g_bbox, R, 4, grad + m->bbox_b_off, ws));
  FR_TRY(gemm_f32(m->glog.f(), 1, nc, feat, nf, 1, grad + m->clsw_off, nf, nc, nf, R, OUT_ADD, nullptr, ws, wslot));
  FR_TRY(channel_sum_cols(m->glog.f(), R, nc, grad + m->clsb_off, ws));
  const float* g = m->feat_g.f();
  GemmFold gfold;   // (fused: the gradient arriving from the layer above is a product whose slabs this layer's kernel folds)
  for (int l = (int)m->cls.size() - 1; l >= 0; --l) {
    ClsLayer& L = m->cls[l];
    const float* pre = L.bn ? L.pre.f() : L.lin.f();
    const bool drop = m->training && L.p_drop > 0.f;
    if (top_act_done && l == (int)m->cls.size() - 1) {
      // (nothing: the heads' launch stored L.g)
    } else if (L.bn && cnet_fuse()) {   // Dropout -> PReLU -> BatchNormalization backward: one launch
      FR_TRY(cnet_act_bn_backward(g, gfold, pre, L.xhat.f(), L.invstd.f(), w + L.bnw_off, w + L.a_off, drop ? L.mask.f() : nullptr,
                                  drop ? 1.0f / (1.0f - L.p_drop) : 1.f, R, L.n, m->training, L.g.f(), grad + L.bnw_off,
                                  grad + L.bnb_off, grad + L.a_off, s));
    } else {
      FR_TRY(prelu_dropout_backward(g, pre, (long)R * L.n, w + L.a_off, drop ? L.mask.f() : nullptr,
                                    drop ? 1.0f / (1.0f - L.p_drop) : 1.f, L.g.f(), grad + L.a_off, s));
      if (L.bn)
        FR_TRY(bn_backward(L.g.f(), L.xhat.f(), L.invstd.f(), w + L.bnw_off, R, L.n, m->training, L.g.f(),
                           grad + L.bnw_off, grad + L.bnb_off, s));
    }
    gfold = GemmFold{};
    // the input gradient of this layer is folded by the layer below when that one runs the fused kernel and nothing else
    // touches this stream's split-K workspace in between (the weight-gradient products are on their own stream and slot)
    GemmFold* gdefer = (cnet_fuse() && async && l > 0 && m->cls[l - 1].bn) ? &gfold : nullptr;
    FR_TRY(fork());   // L.g is final
    const float* in = l == 0 ? m->cnet_x : m->cls[l - 1].post.f();
    float* gin = l == 0 ? gx : L.gin.f();
    const bool xw = (L.x_form & 4) != 0, xd = (L.x_form & 2) != 0 && gin;
    // chain: the gradient's planes (row-major orientation) and the input-gradient product
    if (xd && gemm_f16_on() && L.am_w_of == w) {   // (the weight record of this step's forward pass)
      float* rg = L.am.f() + AMAX_REC;
      FR_TRY(tensor_absmax(L.g.f(), (long)R * L.n, rg, s));
      FR_TRY(split_planes(L.g.f(), R, L.n, L.gp.p, nullptr, s, rg));
      FR_TRY(linear_x_dgrad(L.gp.p, R, L.n, w + L.w_off, L.in, gin, OUT_STORE, s, 0, gdefer, rg, L.am.f() + 2 * AMAX_REC));
    } else if (xd) {
      FR_TRY(split_planes(L.g.f(), R, L.n, L.gp.p, nullptr, s));
      FR_TRY(linear_x_dgrad(L.gp.p, R, L.n, w + L.w_off, L.in, gin, OUT_STORE, s, 0, gdefer));
    } else if (gin) {
      FR_TRY(gemm_f32(L.g.f(), L.n, 1, w + L.w_off, L.in, 1, gin, L.in, R, L.in, L.n, OUT_STORE, nullptr, s, 0, gdefer));
    }
    // beside it: the weight gradient (planes in the transposed orientation) and the bias sums
    static const bool wgrad_f16 = !(getenv("FRCNN_GEMM_WGRAD_F16") && atoi(getenv("FRCNN_GEMM_WGRAD_F16")) == 0);
    if (xw && wgrad_f16 && gemm_f16_on() && L.am.bytes >= (size_t)5 * AMAX_REC * 4) {
      // two fp16 planes per operand here too (round 6; three products instead of six): both tensors' magnitudes are taken on this
      // stream (records 3 and 4 of the layer: the weight-gradient stream forked before the chain took its own)
      float *rgT = L.am.f() + 3 * AMAX_REC, *rxT = L.am.f() + 4 * AMAX_REC;
      FR_TRY(tensor_absmax(L.g.f(), (long)R * L.n, rgT, ws));
      if ((L.x_form & 1) && L.am_w_of == w) rxT = L.am.f();   // (this step's forward pass took the input's magnitude for its own planes)
      else FR_TRY(tensor_absmax(in, (long)R * L.in, rxT, ws));
      FR_TRY(split_planes(L.g.f(), R, L.n, nullptr, L.gpT.p, ws, rgT));
      FR_TRY(split_planes(in, R, L.in, nullptr, L.xpT.p, ws, rxT));
      FR_TRY(linear_x_wgrad(L.gpT.p, L.xpT.p, R, L.n, L.in, grad + L.w_off, ws, wslot, rgT, rxT));
    } else if (xw) {
      FR_TRY(split_planes(L.g.f(), R, L.n, nullptr, L.gpT.p, ws));
      FR_TRY(split_planes(in, R, L.in, nullptr, L.xpT.p, ws));
      FR_TRY(linear_x_wgrad(L.gpT.p, L.xpT.p, R, L.n, L.in, grad + L.w_off, ws, wslot));
    } else {
      FR_TRY(gemm_f32(L.g.f(), 1, L.n, in, L.in, 1, grad + L.w_off, L.in, L.n, L.in, R, OUT_ADD, nullptr, ws, wslot));
    }
    FR_TRY(channel_sum_cols(L.g.f(), R, L.n, grad + L.b_off, ws));
    g = gin;
  }
  if (m->cls.empty() && gx) FR_HIP(hipMemcpyAsync(gx, g, (size_t)R * m->D * 4, hipMemcpyDeviceToDevice, s));
  if (async) {
    FR_HIP(hipEventRecord(m->cw_done, ws));
    m->cw_pending = true;
  }
  return FRCNN_OK;
}

}  // extern "C"
