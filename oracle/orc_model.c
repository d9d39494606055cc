/*
 * orc_model.c -- CPU restatement of models/model_utilities.lua (pnet / cnet construction and
 * their forward/backward through Torch7 nn [ext]), objective.lua:45-218 (one image of
 * lossAndGradient) and Detector.lua:17-141.  TEST INFRASTRUCTURE ONLY (see frcnn_oracle.h).
 *
 * Flat parameter order (utilities.lua:136-147 = pnet:parameters() then cnet:parameters();
 * nngraph's traversal order is [ext]-defined, the order chosen here is documented in
 * ASSUMPTIONS.md and is the one the product uses):
 *   pnet: for each block, for each conv step: W[O][C][k][k], b[O], prelu[1];
 *         then for each anchor head: W[n][C][k][k], b[n], prelu[1], W1[18][n], b1[18]
 *   cnet: for each class layer: W[n][in], b[n], (bn_w[n], bn_b[n])?, prelu[1];
 *         then bbox head W[4][n], b[4]; class head W[classes+1][n], b[classes+1]
 */
#include "frcnn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HEAD_OUT 18 /* 3 * (2 + 4), model_utilities.lua:33 */
#define BN_EPS 1e-5
#define BN_MOM 0.1

static int pool_out(int n) { return (int)ceil((n - 2) / 2.0) + 1; }

long orc_model_param_count(const orc_model *m, long *pnet_count) {
  long n = 0;
  int cin = 3;
  for (int b = 0; b < m->nblocks; ++b) {
    for (int s = 0; s < m->conv_steps[b]; ++s) {
      n += (long)m->filters[b] * cin * m->ksize[b] * m->ksize[b] + m->filters[b] + 1;
      cin = m->filters[b];
    }
  }
  for (int h = 0; h < m->nheads; ++h) {
    int c = m->filters[m->head_input[h] - 1];
    n += (long)m->head_n[h] * c * m->head_k[h] * m->head_k[h] + m->head_n[h] + 1;
    n += (long)HEAD_OUT * m->head_n[h] + HEAD_OUT;
  }
  if (pnet_count) *pnet_count = n;
  long in = (long)m->kh * m->kw * m->filters[m->nblocks - 1]; /* model_utilities.lua:127 */
  for (int l = 0; l < m->ncls; ++l) {
    n += in * m->cls_n[l] + m->cls_n[l];
    if (m->cls_bn[l]) n += 2L * m->cls_n[l];
    n += 1;
    in = m->cls_n[l];
  }
  n += in * 4 + 4;
  n += in * (m->class_count + 1) + (m->class_count + 1);
  return n;
}

/* Localizer.lua:6-39 over the graph built by model_utilities.lua:43-58: every conv and pool
 * on the path input -> output i, input first. */
int orc_model_localizer_layers(const orc_model *m, int output_index, int *out) {
  int nb = output_index <= m->nheads ? m->head_input[output_index - 1] : m->nblocks;
  int n = 0;
  for (int b = 0; b < nb; ++b) {
    for (int s = 0; s < m->conv_steps[b]; ++s) {
      int *l = out + 6 * n++;
      l[0] = m->ksize[b]; l[1] = m->ksize[b]; l[2] = 1; l[3] = 1; l[4] = m->pad[b]; l[5] = m->pad[b];
    }
    int *l = out + 6 * n++;
    l[0] = 2; l[1] = 2; l[2] = 2; l[3] = 2; l[4] = 0; l[5] = 0;
  }
  if (output_index <= m->nheads) {
    int k = m->head_k[output_index - 1];
    int *l = out + 6 * n++;
    l[0] = k; l[1] = k; l[2] = 1; l[3] = 1; l[4] = 0; l[5] = 0;
    l = out + 6 * n++;
    l[0] = 1; l[1] = 1; l[2] = 1; l[3] = 1; l[4] = 0; l[5] = 0;
  }
  return n;
}

/* ------------------------------------------------------------------ pnet */

typedef struct { float *d; int C, H, W; } tens;
static tens tnew(int C, int H, int W) {
  tens t = {(float *)malloc(sizeof(float) * (size_t)C * H * W), C, H, W};
  return t;
}
static long tnum(const tens *t) { return (long)t->C * t->H * t->W; }

/* ---- decision injection (tests only; frcnn_oracle.h) -------------------------------------------------
 * The path has three kinds of DISCRETE decisions -- the arg-max of a max-pooling window, the arg-max of an adaptive
 * max-pooling cell, the branch of a PReLU -- that an fp32 evaluation and this fp64-accumulating one may take
 * differently when two candidates (or an activation and zero) agree to rounding.  One differing decision re-routes a
 * gradient path, which is not an arithmetic error of either side.  With a decision set injected, every such choice is
 * TAKEN FROM THE CALLER (the device's own choices) and everything else is computed here as always, so that the
 * comparison of the gradients tests the arithmetic at the strict tolerance; a second run without injection, recording
 * its own choices, lets the test count how many differ. */
static const orc_decisions *g_inject = NULL;
static orc_decisions *g_record = NULL;
void orc_set_decisions(const orc_decisions *inject, orc_decisions *record) { g_inject = inject; g_record = record; }

static void prelu_fwd_d(const float *x, long n, float a, const uint8_t *pos, uint8_t *rec, float *y) {
#pragma omp parallel for
  for (long i = 0; i < n; ++i) {
    int p = pos ? pos[i] != 0 : x[i] > 0.0f;
    if (rec) rec[i] = (uint8_t)(x[i] > 0.0f);
    y[i] = p ? x[i] : a * x[i];
  }
}
/* slope_abs (optional): receives sum |x * gy| over the same entries -- the size of the terms the slope gradient is the
 * (cancelling) sum of, i.e. the scale its achievable accuracy has to be measured against */
static double prelu_bwd_d(const float *x, const float *gy, long n, float a, const uint8_t *pos, float *gx, double *slope_abs) {
  double ga = 0, gabs = 0;
#pragma omp parallel for reduction(+ : ga, gabs)
  for (long i = 0; i < n; ++i) {
    int p = pos ? pos[i] != 0 : x[i] > 0.0f;
    if (p) {
      gx[i] = gy[i];
    } else {
      gx[i] = a * gy[i];
      ga += (double)x[i] * (double)gy[i];
      gabs += fabs((double)x[i] * (double)gy[i]);
    }
  }
  if (slope_abs) *slope_abs += gabs;
  return ga;
}

#define MAXC 32
struct orc_pnet_state {
  int nconv;                 /* backbone convs */
  tens cin[MAXC], cx[MAXC];  /* conv input (activated) and pre-activation output */
  long woff[MAXC];           /* flat offset of W (b and prelu follow) */
  int ck[MAXC], cpad[MAXC], cblock[MAXC], cstep[MAXC];
  float cscale_eval[MAXC];   /* (1-p) or 1 */
  const float *cmask[MAXC];  /* training keep mask per channel or NULL */
  const uint8_t *cpos[MAXC], *hpos[8]; /* injected PReLU branches (caller-owned) or NULL */
  int training;
  tens cy[MAXC];             /* activated (PReLU + dropout) output of each conv, owned */
  int last_conv[8];          /* index of the last conv of each block (pool input = cy[..]) */
  tens pooled[8];
  int32_t *pidx[8];
  int nheads;
  tens hx[8], hy[8], hout[8];
  long hwoff[8];
  tens img;
};

orc_pnet_state *orc_pnet_state_new(void) { return (orc_pnet_state *)calloc(1, sizeof(orc_pnet_state)); }
static void pnet_state_clear(orc_pnet_state *s) {
  /* cin[i] aliases img / pooled / cy tensors, which are freed here */
  for (int i = 0; i < MAXC; ++i) { free(s->cx[i].d); free(s->cy[i].d); }
  for (int b = 0; b < 8; ++b) { free(s->pooled[b].d); free(s->pidx[b]); }
  for (int h = 0; h < 8; ++h) { free(s->hx[h].d); free(s->hy[h].d); free(s->hout[h].d); }
  free(s->img.d);
  memset(s, 0, sizeof(*s));
}
void orc_pnet_state_free(orc_pnet_state *s) {
  if (!s) return;
  pnet_state_clear(s);
  free(s);
}

/* y = scale_c * prelu(x)  (nn.PReLU then nn.SpatialDropout, model_utilities.lua:9-12) */
static void act_apply(const tens *x, float a, const float *mask, float eval_scale, int training,
                      const uint8_t *pos, uint8_t *rec, tens *y) {
  long hw = (long)x->H * x->W;
#pragma omp parallel for
  for (int c = 0; c < x->C; ++c) {
    float sc = 1.0f;
    if (mask && training) sc = mask[c];
    else if (!training) sc = eval_scale;
    const float *xp = x->d + c * hw;
    float *yp = y->d + c * hw;
    for (long t = 0; t < hw; ++t) {
      int p = pos ? pos[c * hw + t] != 0 : xp[t] > 0.0f;
      if (rec) rec[c * hw + t] = (uint8_t)(xp[t] > 0.0f);
      float v = p ? xp[t] : a * xp[t];
      yp[t] = sc == 1.0f ? v : v * sc;
    }
  }
}

void orc_pnet_forward(const orc_model *m, const float *w, const float *img, int H, int W,
                      int training, const float *const *drop_masks, orc_pnet_state *s) {
  pnet_state_clear(s);
  s->training = training;
  s->img = tnew(3, H, W);
  memcpy(s->img.d, img, sizeof(float) * 3 * (size_t)H * W);
  tens cur = s->img;
  long off = 0;
  int nc = 0;
  for (int b = 0; b < m->nblocks; ++b) {
    for (int st = 0; st < m->conv_steps[b]; ++st) {
      int k = m->ksize[b], p = m->pad[b], O = m->filters[b];
      int Ho = cur.H + 2 * p - k + 1, Wo = cur.W + 2 * p - k + 1;
      s->cin[nc] = cur;
      s->cx[nc] = tnew(O, Ho, Wo);
      s->woff[nc] = off;
      s->ck[nc] = k; s->cpad[nc] = p; s->cblock[nc] = b; s->cstep[nc] = st;
      long wsz = (long)O * cur.C * k * k;
      orc_conv2d_fwd(cur.d, cur.C, cur.H, cur.W, w + off, w + off + wsz, O, k, k, p, s->cx[nc].d);
      float a = w[off + wsz + O];
      off += wsz + O + 1;
      int has_drop = (st == 0 && m->dropout[b] > 0); /* model_utilities.lua:10-12,20 */
      s->cmask[nc] = has_drop && drop_masks ? drop_masks[b] : NULL;
      s->cscale_eval[nc] = has_drop ? (float)(1.0 - m->dropout[b]) : 1.0f;
      tens y = tnew(O, Ho, Wo);
      s->cpos[nc] = g_inject ? g_inject->conv_pos[nc] : NULL;
      act_apply(&s->cx[nc], a, s->cmask[nc], s->cscale_eval[nc], training, s->cpos[nc],
                g_record ? (uint8_t *)g_record->conv_pos[nc] : NULL, &y);
      s->cy[nc] = y;
      if (st == m->conv_steps[b] - 1) s->last_conv[b] = nc;
      cur = y;
      ++nc;
    }
    int Hp = pool_out(cur.H), Wp = pool_out(cur.W);
    s->pooled[b] = tnew(cur.C, Hp, Wp);
    s->pidx[b] = (int32_t *)malloc(sizeof(int32_t) * (size_t)cur.C * Hp * Wp);
    orc_maxpool2x2_ceil_fwd(cur.d, cur.C, cur.H, cur.W, s->pooled[b].d, s->pidx[b]);
    {
      long np_ = (long)cur.C * Hp * Wp, plane = (long)cur.H * cur.W, pp = (long)Hp * Wp;
      if (g_record && g_record->pool_idx[b]) memcpy((int32_t *)g_record->pool_idx[b], s->pidx[b], sizeof(int32_t) * np_);
      if (g_inject && g_inject->pool_idx[b]) { /* the caller's window winners: value and route follow them */
        const int32_t *inj = g_inject->pool_idx[b];
        for (long i = 0; i < np_; ++i) {
          s->pidx[b][i] = inj[i];
          s->pooled[b].d[i] = cur.d[(i / pp) * plane + inj[i]];
        }
      }
    }
    cur = s->pooled[b];
  }
  s->nconv = nc;
  s->nheads = m->nheads;
  for (int h = 0; h < m->nheads; ++h) {
    tens in = s->pooled[m->head_input[h] - 1];
    int k = m->head_k[h], n = m->head_n[h];
    int Ho = in.H - k + 1, Wo = in.W - k + 1;
    s->hwoff[h] = off;
    long wsz = (long)n * in.C * k * k;
    s->hx[h] = tnew(n, Ho, Wo);
    orc_conv2d_fwd(in.d, in.C, in.H, in.W, w + off, w + off + wsz, n, k, k, 0, s->hx[h].d);
    float a = w[off + wsz + n];
    off += wsz + n + 1;
    s->hy[h] = tnew(n, Ho, Wo);
    s->hpos[h] = g_inject ? g_inject->head_pos[h] : NULL;
    prelu_fwd_d(s->hx[h].d, tnum(&s->hx[h]), a, s->hpos[h], g_record ? (uint8_t *)g_record->head_pos[h] : NULL, s->hy[h].d);
    s->hout[h] = tnew(HEAD_OUT, Ho, Wo);
    orc_conv2d_fwd(s->hy[h].d, n, Ho, Wo, w + off, w + off + (long)HEAD_OUT * n, HEAD_OUT, 1, 1, 0,
                   s->hout[h].d);
    off += (long)HEAD_OUT * n + HEAD_OUT;
  }
}

const float *orc_pnet_output(const orc_pnet_state *s, int i, int *C, int *H, int *W) {
  const tens *t = i <= s->nheads ? &s->hout[i - 1] : NULL;
  if (!t) {
    int b = 0;
    while (b < 8 && s->pooled[b].d) ++b;
    t = &s->pooled[b - 1];
  }
  if (C) *C = t->C;
  if (H) *H = t->H;
  if (W) *W = t->W;
  return t->d;
}

void orc_pnet_backward(const orc_model *m, const float *w, const orc_pnet_state *s,
                       const float *const *delta, float *grad) {
  /* gradient wrt each block's pooled output */
  float *gpool[8] = {0};
  for (int b = 0; b < m->nblocks; ++b)
    gpool[b] = (float *)calloc((size_t)tnum(&s->pooled[b]), sizeof(float));
  /* output nheads+1 = last pooled map (model_utilities.lua:55) */
  {
    const tens *t = &s->pooled[m->nblocks - 1];
    const float *d = delta[m->nheads];
    for (long i = 0; i < tnum(t); ++i) gpool[m->nblocks - 1][i] += d[i];
  }
  for (int h = 0; h < m->nheads; ++h) {
    const tens *in = &s->pooled[m->head_input[h] - 1];
    int k = m->head_k[h], n = m->head_n[h];
    const tens *hx = &s->hx[h], *hy = &s->hy[h];
    long off = s->hwoff[h];
    long wsz = (long)n * in->C * k * k;
    long off1 = off + wsz + n + 1;
    /* 1x1 conv backward */
    orc_conv2d_bwd_weight(hy->d, n, hy->H, hy->W, delta[h], HEAD_OUT, 1, 1, 0, grad + off1,
                          grad + off1 + (long)HEAD_OUT * n);
    tens ghy = tnew(n, hy->H, hy->W);
    orc_conv2d_bwd_input(delta[h], HEAD_OUT, hy->H, hy->W, w + off1, n, 1, 1, 0, hy->H, hy->W, ghy.d);
    tens ghx = tnew(n, hy->H, hy->W);
    double ga = prelu_bwd_d(hx->d, ghy.d, tnum(hx), w[off + wsz + n], s->hpos[h], ghx.d, g_record && g_record->slope_abs ? g_record->slope_abs + 32 + h : NULL);
    grad[off + wsz + n] = (float)((double)grad[off + wsz + n] + ga);
    orc_conv2d_bwd_weight(in->d, in->C, in->H, in->W, ghx.d, n, k, k, 0, grad + off, grad + off + wsz);
    tens gin = tnew(in->C, in->H, in->W);
    orc_conv2d_bwd_input(ghx.d, n, hx->H, hx->W, w + off, in->C, k, k, 0, in->H, in->W, gin.d);
    float *gp = gpool[m->head_input[h] - 1];
    for (long i = 0; i < tnum(&gin); ++i) gp[i] += gin.d[i]; /* nngraph fan-out sum */
    free(ghy.d); free(ghx.d); free(gin.d);
  }
  int nc = s->nconv;
  for (int b = m->nblocks - 1; b >= 0; --b) {
    const tens *al = &s->cy[s->last_conv[b]];
    tens g = tnew(al->C, al->H, al->W);
    orc_maxpool2x2_ceil_bwd(gpool[b], s->pidx[b], al->C, al->H, al->W, g.d);
    for (int st = m->conv_steps[b] - 1; st >= 0; --st) {
      --nc;
      const tens *x = &s->cx[nc], *in = &s->cin[nc];
      int k = s->ck[nc], p = s->cpad[nc], O = x->C;
      long off = s->woff[nc], wsz = (long)O * in->C * k * k;
      /* SpatialDropout backward: same per-channel mask (training only) */
      if (s->cmask[nc]) {
        long hw = (long)x->H * x->W;
        for (int c = 0; c < O; ++c)
          if (s->cmask[nc][c] != 1.0f)
            for (long t = 0; t < hw; ++t) g.d[c * hw + t] *= s->cmask[nc][c];
      }
      tens gx = tnew(O, x->H, x->W);
      double ga = prelu_bwd_d(x->d, g.d, tnum(x), w[off + wsz + O], s->cpos[nc], gx.d, g_record && g_record->slope_abs ? g_record->slope_abs + nc : NULL);
      grad[off + wsz + O] = (float)((double)grad[off + wsz + O] + ga);
      orc_conv2d_bwd_weight(in->d, in->C, in->H, in->W, gx.d, O, k, k, p, grad + off,
                            grad + off + wsz);
      free(g.d);
      if (nc == 0) { /* input gradient of the first conv is unused (objective.lua:189) */
        g.d = NULL;
        free(gx.d);
        break;
      }
      g = tnew(in->C, in->H, in->W);
      orc_conv2d_bwd_input(gx.d, O, x->H, x->W, w + off, in->C, k, k, p, in->H, in->W, g.d);
      free(gx.d);
    }
    if (b > 0) {
      /* g is the gradient wrt pooled[b-1] from the conv path; add to fan-out sum */
      float *gp = gpool[b - 1];
      for (long i = 0; i < tnum(&s->pooled[b - 1]); ++i) gp[i] += g.d[i];
      free(g.d);
    }
  }
  for (int b = 0; b < m->nblocks; ++b) free(gpool[b]);
}

/* ------------------------------------------------------------------ cnet */

struct orc_cnet_state {
  int R, D;
  int ncls;
  float *in[8];    /* input of Linear l (R x in_l) */
  float *lin[8];   /* Linear output (pre-BN) */
  float *xhat[8];  /* BN normalised (if bn) */
  float *invstd[8];
  float *pre[8];   /* PReLU input */
  float *post[8];  /* after dropout = next input (aliases in[l+1]) */
  const float *mask[8];
  const uint8_t *pos[8];   /* injected PReLU branches (caller-owned) or NULL */
  int training;
  float *cls_logits;
  float *cls_lsm;
};
orc_cnet_state *orc_cnet_state_new(void) { return (orc_cnet_state *)calloc(1, sizeof(orc_cnet_state)); }
static void cnet_state_clear(orc_cnet_state *s) {
  free(s->in[0]);
  for (int l = 0; l < 8; ++l) {
    free(s->lin[l]); free(s->xhat[l]); free(s->invstd[l]); free(s->post[l]);
    if (s->pre[l] != s->lin[l]) free(s->pre[l]);
  }
  free(s->cls_logits); free(s->cls_lsm);
  memset(s, 0, sizeof(*s));
}
void orc_cnet_state_free(orc_cnet_state *s) {
  if (!s) return;
  cnet_state_clear(s);
  free(s);
}

void orc_cnet_forward(const orc_model *m, const float *weights, const float *x, int R,
                      int training, const float *const *drop_masks, float *bn_running,
                      orc_cnet_state *s, float *bbox_out, float *cls_out) {
  cnet_state_clear(s);
  long pn;
  orc_model_param_count(m, &pn);
  const float *w = weights + pn;
  int D = m->kh * m->kw * m->filters[m->nblocks - 1];
  s->R = R; s->D = D; s->ncls = m->ncls; s->training = training;
  s->in[0] = (float *)malloc(sizeof(float) * (size_t)R * D);
  memcpy(s->in[0], x, sizeof(float) * (size_t)R * D);
  int in = D;
  long off = 0;
  float *bnr = bn_running;
  for (int l = 0; l < m->ncls; ++l) {
    int n = m->cls_n[l];
    s->lin[l] = (float *)malloc(sizeof(float) * (size_t)R * n);
    orc_linear_fwd(s->in[l], R, in, w + off, w + off + (long)in * n, n, s->lin[l]);
    off += (long)in * n + n;
    float *cur = s->lin[l];
    if (m->cls_bn[l]) { /* nn.BatchNormalization(n): eps 1e-5, momentum 0.1, affine [ext] */
      const float *gw = w + off, *gb = w + off + n;
      off += 2L * n;
      s->xhat[l] = (float *)malloc(sizeof(float) * (size_t)R * n);
      s->invstd[l] = (float *)malloc(sizeof(float) * n);
      float *y = (float *)malloc(sizeof(float) * (size_t)R * n);
      for (int j = 0; j < n; ++j) {
        double mean, var;
        if (training) {
          mean = 0;
          for (int r = 0; r < R; ++r) mean += cur[(size_t)r * n + j];
          mean /= R;
          var = 0;
          for (int r = 0; r < R; ++r) { double d = cur[(size_t)r * n + j] - mean; var += d * d; }
          double unb = R > 1 ? var / (R - 1) : var / R;
          var /= R;
          if (bnr) {
            bnr[j] = (float)((1 - BN_MOM) * bnr[j] + BN_MOM * mean);
            bnr[n + j] = (float)((1 - BN_MOM) * bnr[n + j] + BN_MOM * unb);
          }
        } else {
          mean = bnr[j]; var = bnr[n + j];
        }
        double is = 1.0 / sqrt(var + BN_EPS);
        s->invstd[l][j] = (float)is;
        for (int r = 0; r < R; ++r) {
          double xh = (cur[(size_t)r * n + j] - mean) * is;
          s->xhat[l][(size_t)r * n + j] = (float)xh;
          y[(size_t)r * n + j] = (float)(xh * gw[j] + gb[j]);
        }
      }
      if (bnr) bnr += 2 * n;
      s->pre[l] = y;
    } else {
      s->pre[l] = s->lin[l];
    }
    float a = w[off];
    off += 1;
    s->post[l] = (float *)malloc(sizeof(float) * (size_t)R * n);
    s->pos[l] = g_inject ? g_inject->cnet_pos[l] : NULL;
    prelu_fwd_d(s->pre[l], (long)R * n, a, s->pos[l], g_record ? (uint8_t *)g_record->cnet_pos[l] : NULL, s->post[l]);
    s->mask[l] = NULL;
    if (m->cls_dropout[l] > 0 && training) { /* nn.Dropout v2: y = x*mask/(1-p) [ext] */
      const float *mk = drop_masks ? drop_masks[l] : NULL;
      s->mask[l] = mk;
      if (mk) {
        float inv = (float)(1.0 / (1.0 - m->cls_dropout[l]));
        for (long i = 0; i < (long)R * n; ++i) s->post[l][i] = s->post[l][i] * (mk[i] * inv);
      }
    }
    s->in[l + 1] = s->post[l];
    in = n;
  }
  const float *feat = s->in[m->ncls];
  orc_linear_fwd(feat, R, in, w + off, w + off + (long)in * 4, 4, bbox_out); /* :99 */
  off += (long)in * 4 + 4;
  int nc = m->class_count + 1;
  s->cls_logits = (float *)malloc(sizeof(float) * (size_t)R * nc);
  s->cls_lsm = (float *)malloc(sizeof(float) * (size_t)R * nc);
  orc_linear_fwd(feat, R, in, w + off, w + off + (long)in * nc, nc, s->cls_logits); /* :103 */
  orc_log_softmax(s->cls_logits, R, nc, s->cls_lsm);                                /* :104 */
  memcpy(cls_out, s->cls_lsm, sizeof(float) * (size_t)R * nc);
}

void orc_cnet_backward(const orc_model *m, const float *weights, const orc_cnet_state *s,
                       const float *g_bbox, const float *g_cls, float *gx_out, float *grad_all) {
  long pn;
  orc_model_param_count(m, &pn);
  const float *w = weights + pn;
  float *grad = grad_all + pn;
  int R = s->R;
  /* offsets */
  long offs[8], off = 0;
  int in = s->D;
  for (int l = 0; l < m->ncls; ++l) {
    offs[l] = off;
    off += (long)in * m->cls_n[l] + m->cls_n[l] + (m->cls_bn[l] ? 2L * m->cls_n[l] : 0) + 1;
    in = m->cls_n[l];
  }
  long off_bbox = off, off_cls = off + (long)in * 4 + 4;
  int nc = m->class_count + 1;
  const float *feat = s->in[m->ncls];
  float *gfeat = (float *)calloc((size_t)R * in, sizeof(float));
  float *tmp = (float *)malloc(sizeof(float) * (size_t)R * in);
  orc_linear_bwd(feat, g_bbox, R, in, w + off_bbox, 4, tmp, grad + off_bbox, grad + off_bbox + (long)in * 4);
  for (long i = 0; i < (long)R * in; ++i) gfeat[i] += tmp[i];
  /* LogSoftMax backward: g_i = gy_i - exp(lsm_i) * sum_j gy_j */
  float *glog = (float *)malloc(sizeof(float) * (size_t)R * nc);
  for (int r = 0; r < R; ++r) {
    double sum = 0;
    for (int j = 0; j < nc; ++j) sum += g_cls[(size_t)r * nc + j];
    for (int j = 0; j < nc; ++j)
      glog[(size_t)r * nc + j] =
          (float)((double)g_cls[(size_t)r * nc + j] - exp((double)s->cls_lsm[(size_t)r * nc + j]) * sum);
  }
  orc_linear_bwd(feat, glog, R, in, w + off_cls, nc, tmp, grad + off_cls, grad + off_cls + (long)in * nc);
  for (long i = 0; i < (long)R * in; ++i) gfeat[i] += tmp[i];
  free(tmp); free(glog);
  float *g = gfeat;
  for (int l = m->ncls - 1; l >= 0; --l) {
    int n = m->cls_n[l];
    int inl = l == 0 ? s->D : m->cls_n[l - 1];
    long o = offs[l];
    long o_bn = o + (long)inl * n + n;
    long o_pr = o_bn + (m->cls_bn[l] ? 2L * n : 0);
    if (s->mask[l]) {
      float inv = (float)(1.0 / (1.0 - m->cls_dropout[l]));
      for (long i = 0; i < (long)R * n; ++i) g[i] = g[i] * (s->mask[l][i] * inv);
    }
    float *gpre = (float *)malloc(sizeof(float) * (size_t)R * n);
    double ga = prelu_bwd_d(s->pre[l], g, (long)R * n, w[o_pr], s->pos[l], gpre, g_record && g_record->slope_abs ? g_record->slope_abs + 40 + l : NULL);
    grad[o_pr] = (float)((double)grad[o_pr] + ga);
    free(g);
    float *glin = gpre;
    if (m->cls_bn[l]) {
      const float *gw = w + o_bn;
      glin = (float *)malloc(sizeof(float) * (size_t)R * n);
      for (int j = 0; j < n; ++j) {
        double sg = 0, sgx = 0;
        for (int r = 0; r < R; ++r) {
          double gy = gpre[(size_t)r * n + j];
          sg += gy;
          sgx += gy * s->xhat[l][(size_t)r * n + j];
        }
        grad[o_bn + j] = (float)((double)grad[o_bn + j] + sgx);       /* d gamma */
        grad[o_bn + n + j] = (float)((double)grad[o_bn + n + j] + sg); /* d beta */
        double is = s->invstd[l][j];
        for (int r = 0; r < R; ++r) {
          double gy = gpre[(size_t)r * n + j];
          double xh = s->xhat[l][(size_t)r * n + j];
          double v = s->training ? (gy - sg / R - xh * sgx / R) * gw[j] * is : gy * gw[j] * is;
          glin[(size_t)r * n + j] = (float)v;
        }
      }
      free(gpre);
    }
    float *gin = (float *)malloc(sizeof(float) * (size_t)R * inl);
    orc_linear_bwd(s->in[l], glin, R, inl, w + o, n, gin, grad + o, grad + o + (long)inl * n);
    free(glin);
    g = gin;
  }
  memcpy(gx_out, g, sizeof(float) * (size_t)R * s->D);
  free(g);
}

/* ------------------------------------------------------------------ objective.lua */

/* nn.CrossEntropyCriterion on a 2-vector (objective.lua:104-106,132-134): fp32 */
static double ce2(const float *v, int target /*0 or 1*/, float *dc) {
  float lsm[2];
  orc_log_softmax(v, 1, 2, lsm);
  for (int j = 0; j < 2; ++j) dc[j] = (float)(exp((double)lsm[j]) - (j == target ? 1.0 : 0.0));
  return -(double)lsm[target];
}
/* nn.SmoothL1Criterion, sizeAverage=false (objective.lua:26-27): sum, grad clamp(z,-1,1) */
static double smooth_l1(const float *x, const float *y, long n, float *g) {
  double s = 0;
  for (long i = 0; i < n; ++i) {
    float z = x[i] - y[i];
    float az = fabsf(z);
    s += az < 1.0f ? 0.5 * (double)z * (double)z : (double)az - 0.5;
    if (g) g[i] = az < 1.0f ? z : (z > 0 ? 1.0f : -1.0f);
  }
  return (double)(float)s;
}

void orc_train_image(const orc_model *m, const float *weights, float *grad, const float *img,
                     int H, int W, const int *pos_idx, const double *pos_rect, int np,
                     const double *rois, const int *roi_class, int nroi, const int *neg_idx,
                     const double *neg_rect, int nn, const float *const *pnet_drop_masks,
                     const float *const *cnet_drop_masks, float *bn_running, double *acc) {
  (void)nroi;
  orc_pnet_state *ps = orc_pnet_state_new();
  orc_pnet_forward(m, weights, img, H, W, 1, pnet_drop_masks, ps); /* :71 */
  int nout = m->nheads + 1;
  float *delta[16];
  int oc[16], oh[16], ow[16];
  const float *outp[16];
  for (int i = 0; i < nout; ++i) { /* :78-84 */
    outp[i] = orc_pnet_output(ps, i + 1, &oc[i], &oh[i], &ow[i]);
    delta[i] = (float *)calloc((size_t)oc[i] * oh[i] * ow[i], sizeof(float));
  }
  int loc_layers[64 * 6];
  int nloc = orc_model_localizer_layers(m, nout, loc_layers); /* objective.lua:22 */
  int kh = m->kh, kw = m->kw, planes = m->filters[m->nblocks - 1];
  int D = kh * kw * planes;
  int R = np + nn;
  float *cinput = (float *)malloc(sizeof(float) * (size_t)(R > 0 ? R : 1) * D);
  int32_t *pidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1) * D);
  float *cctarget = (float *)calloc(R > 0 ? R : 1, sizeof(float));
  float *crtarget = (float *)calloc((size_t)(R > 0 ? R : 1) * 4, sizeof(float));
  const float *fm = outp[nout - 1];
  int fH = oh[nout - 1], fW = ow[nout - 1];
  double cls_loss = 0, reg_loss = 0;
  for (int e = 0; e < np; ++e) { /* :91-120 */
    const int *ix = pos_idx + 5 * e;
    int l = ix[0] - 1, a = ix[1], y = ix[2] - 1, x = ix[3] - 1;
    const double *anchor = pos_rect + 4 * e;
    const double *roi = rois + 4 * (ix[4] - 1);
    long hw = (long)oh[l] * ow[l];
    float v[6];
    for (int c = 0; c < 6; ++c) v[c] = outp[l][(6 * (a - 1) + c) * hw + (long)y * ow[l] + x];
    float dc[2];
    cls_loss += ce2(v, 0, dc); /* :104 target 1 = foreground */
    for (int c = 0; c < 2; ++c) delta[l][(6 * (a - 1) + c) * hw + (long)y * ow[l] + x] += dc[c];
    float reg_target[4], dr[4];
    orc_input_to_anchor(anchor, roi, reg_target); /* :110 */
    double prop[4];
    orc_anchor_to_input(anchor, v + 2, prop);     /* :111 */
    reg_loss += smooth_l1(v + 2, reg_target, 4, dr) * 10; /* :112 */
    for (int c = 0; c < 4; ++c)
      delta[l][(6 * (a - 1) + 2 + c) * hw + (long)y * ow[l] + x] += dr[c] * 10.0f; /* :113-114 */
    int win[4];
    orc_extract_roi_window(loc_layers, nloc, roi, fH, fW, win); /* :117 pools the GT rect */
    orc_adaptive_max_pool_fwd(fm, planes, fH, fW, win, kh, kw, cinput + (size_t)e * D,
                              pidx + (size_t)e * D);
    cctarget[e] = (float)roi_class[ix[4] - 1];            /* :155 */
    orc_input_to_anchor(prop, roi, crtarget + 4 * e);     /* :156 */
  }
  for (int e = 0; e < nn; ++e) { /* :123-140 */
    const int *ix = neg_idx + 4 * e;
    int l = ix[0] - 1, a = ix[1], y = ix[2] - 1, x = ix[3] - 1;
    long hw = (long)oh[l] * ow[l];
    float v[2], dc[2];
    for (int c = 0; c < 2; ++c) v[c] = outp[l][(6 * (a - 1) + c) * hw + (long)y * ow[l] + x];
    cls_loss += ce2(v, 1, dc); /* :132 target 2 = background */
    for (int c = 0; c < 2; ++c) delta[l][(6 * (a - 1) + c) * hw + (long)y * ow[l] + x] += dc[c];
    int win[4];
    orc_extract_roi_window(loc_layers, nloc, neg_rect + 4 * e, fH, fW, win); /* :137 anchor rect */
    orc_adaptive_max_pool_fwd(fm, planes, fH, fW, win, kh, kw, cinput + (size_t)(np + e) * D,
                              pidx + (size_t)(np + e) * D);
    cctarget[np + e] = (float)(m->class_count + 1); /* :159 bgclass */
  }
  if (g_record && g_record->roi_idx && R > 0) memcpy((int32_t *)g_record->roi_idx, pidx, sizeof(int32_t) * (size_t)R * D);
  if (g_inject && g_inject->roi_idx && R > 0) { /* the caller's cell winners (see orc_set_decisions) */
    const int32_t *inj = g_inject->roi_idx;
    long plane = (long)fH * fW;
    int cell = kh * kw;
    for (long i = 0; i < (long)R * D; ++i) {
      pidx[i] = inj[i];
      cinput[i] = fm[((i % D) / cell) * plane + inj[i]];
    }
  }
  if (R > 0) { /* :146-186 */
    int nc = m->class_count + 1;
    orc_cnet_state *cs = orc_cnet_state_new();
    float *crout = (float *)malloc(sizeof(float) * (size_t)R * 4);
    float *ccout = (float *)malloc(sizeof(float) * (size_t)R * nc);
    orc_cnet_forward(m, weights, cinput, R, 1, cnet_drop_masks, bn_running, cs, crout, ccout);
    for (long i = (long)np * 4; i < (long)R * 4; ++i) crout[i] = 0.0f; /* :170 */
    float *crdelta = (float *)malloc(sizeof(float) * (size_t)R * 4);
    acc[4] += smooth_l1(crout, crtarget, (long)R * 4, crdelta) * 10; /* :171 */
    for (long i = 0; i < (long)R * 4; ++i) crdelta[i] *= 10.0f;      /* :172 */
    float *ccdelta = (float *)calloc((size_t)R * nc, sizeof(float));
    double nll = 0; /* nn.ClassNLLCriterion, sizeAverage=true [ext] */
    for (int r = 0; r < R; ++r) {
      int t = (int)cctarget[r] - 1;
      nll -= ccout[(size_t)r * nc + t];
      ccdelta[(size_t)r * nc + t] = (float)(-1.0 / R);
    }
    acc[6] += (double)(float)(nll / R); /* :175-176 */
    float *post = (float *)malloc(sizeof(float) * (size_t)R * D);
    orc_cnet_backward(m, weights, cs, crdelta, ccdelta, post, grad); /* :179 */
    for (int r = 0; r < R; ++r)                                     /* :182-185 */
      orc_adaptive_max_pool_bwd(delta[nout - 1], planes, fH, fW, kh, kw, post + (size_t)r * D,
                                pidx + (size_t)r * D);
    free(crout); free(ccout); free(crdelta); free(ccdelta); free(post);
    orc_cnet_state_free(cs);
  }
  orc_pnet_backward(m, weights, ps, (const float *const *)delta, grad); /* :189 */
  acc[0] += cls_loss;
  acc[1] += reg_loss;
  acc[2] += np + nn; /* :195 cls_count */
  acc[3] += np;      /* :194 reg_count */
  acc[5] += np;      /* :197 creg_count */
  acc[7] += 1;       /* :198 ccls_count */
  for (int i = 0; i < nout; ++i) free(delta[i]);
  free(cinput); free(pidx); free(cctarget); free(crtarget);
  orc_pnet_state_free(ps);
}

/* ------------------------------------------------------------------ Detector.lua */

void orc_detect(const orc_model *m, const float *weights, const float *bn_running,
                const float *img, int H, int W, int cap, float *match_p, int *match_idx,
                double *match_rect, int64_t *cand_ids, float *cand_bbox, float *cand_cls,
                double *win_rows, orc_detect_counts *counts) {
  orc_pnet_state *ps = orc_pnet_state_new();
  orc_pnet_forward(m, weights, img, H, W, 0, NULL, ps); /* Detector.lua:31-33 */
  double input_rect[4] = {0, 0, (double)W, (double)H};   /* :28 */
  int nout = m->nheads + 1;
  /* Anchors.new(model.pnet, cfg.scales), Detector.lua:11 */
  int layers_concat[4 * 64 * 6], nl[4];
  int o = 0;
  for (int i = 0; i < 4; ++i) {
    nl[i] = orc_model_localizer_layers(m, i + 1, layers_concat + o);
    o += nl[i] * 6;
  }
  orc_anchors *A = orc_anchors_new(layers_concat, nl, m->scales, 4);
  int nm = 0;
  for (int i = 1; i <= 4; ++i) { /* :39 */
    int C, Hl, Wl;
    const float *layer = orc_pnet_output(ps, i, &C, &Hl, &Wl);
    long hw = (long)Hl * Wl;
    for (int y = 1; y <= Hl; ++y)
      for (int x = 1; x <= Wl; ++x)
        for (int a = 1; a <= 3; ++a) { /* :45 */
          int ofs = (a - 1) * 6;
          float v[6], lsm[2];
          for (int c = 0; c < 6; ++c) v[c] = layer[(ofs + c) * hw + (long)(y - 1) * Wl + (x - 1)];
          orc_log_softmax(v, 1, 2, lsm);     /* :52 */
          if (exp((double)lsm[0]) > 0.95) {  /* :54 */
            double ar[4], r[4];
            orc_anchors_get(A, i, a, y, x, ar); /* :56 */
            orc_anchor_to_input(ar, v + 2, r);  /* :57 */
            if (orc_rect_overlaps(r, input_rect)) { /* :58 */
              if (nm < cap) {
                match_p[nm] = lsm[0];
                int *mi = match_idx + 4 * nm;
                mi[0] = i; mi[1] = a; mi[2] = y; mi[3] = x;
                memcpy(match_rect + 4 * nm, r, sizeof(r));
              }
              ++nm;
            }
          }
        }
  }
  counts->nmatch = nm;
  counts->ncand = 0;
  counts->nwin = 0;
  if (nm > cap) nm = cap;
  if (nm > 0) { /* :71 */
    float *bb = (float *)malloc(sizeof(float) * (size_t)nm * 4);
    for (int i = 0; i < nm * 4; ++i) bb[i] = (float)match_rect[i]; /* :74-79 FloatTensor */
    int nc = orc_nms(bb, nm, 4, 0.25f, 0, 0, cand_ids);           /* :81-82 key = y2 */
    free(bb);
    counts->ncand = nc;
    int kh = m->kh, kw = m->kw, planes = m->filters[m->nblocks - 1], D = kh * kw * planes;
    int loc_layers[64 * 6];
    int nloc = orc_model_localizer_layers(m, nout, loc_layers); /* :12 */
    int fC, fH, fW;
    const float *fm = orc_pnet_output(ps, nout, &fC, &fH, &fW);
    float *cinput = (float *)malloc(sizeof(float) * (size_t)nc * D);
    int32_t *tmpidx = (int32_t *)malloc(sizeof(int32_t) * D);
    for (int i = 0; i < nc; ++i) { /* :94-98 */
      int win[4];
      orc_extract_roi_window(loc_layers, nloc, match_rect + 4 * (cand_ids[i] - 1), fH, fW, win);
      orc_adaptive_max_pool_fwd(fm, planes, fH, fW, win, kh, kw, cinput + (size_t)i * D, tmpidx);
    }
    free(tmpidx);
    int ncl = m->class_count + 1;
    orc_cnet_state *cs = orc_cnet_state_new();
    float *bn = NULL;
    if (bn_running) { /* eval: running stats are read-only, but keep caller's buffer const */
      long nbn = 0;
      for (int l = 0; l < m->ncls; ++l) if (m->cls_bn[l]) nbn += 2 * m->cls_n[l];
      bn = (float *)malloc(sizeof(float) * (nbn ? nbn : 1));
      memcpy(bn, bn_running, sizeof(float) * nbn);
    }
    orc_cnet_forward(m, weights, cinput, nc, 0, NULL, bn, cs, cand_bbox, cand_cls); /* :101 */
    free(bn); free(cinput);
    orc_cnet_state_free(cs);
    /* :106-122 classify, then per-class NMS :125-136 in ascending class order (the reference
     * iterates with pairs(): order unspecified) */
    int *cls_of = (int *)malloc(sizeof(int) * nc);
    float *conf_of = (float *)malloc(sizeof(float) * nc);
    double *r2 = (double *)malloc(sizeof(double) * (size_t)nc * 4);
    for (int i = 0; i < nc; ++i) {
      orc_anchor_to_input(match_rect + 4 * (cand_ids[i] - 1), cand_bbox + 4 * i, r2 + 4 * i); /* :107 */
      int best = 0;
      for (int j = 1; j < ncl; ++j)
        if (cand_cls[(size_t)i * ncl + j] > cand_cls[(size_t)i * ncl + best]) best = j; /* :110 */
      cls_of[i] = best + 1;
      conf_of[i] = cand_cls[(size_t)i * ncl + best];
    }
    int nw = 0;
    for (int c = 1; c <= ncl; ++c) {
      if (c == m->class_count + 1) continue; /* :115 bgclass */
      int cnt = 0;
      for (int i = 0; i < nc; ++i)
        if (cls_of[i] == c && exp((double)conf_of[i]) > 0.2) ++cnt;
      if (!cnt) continue;
      float *bb5 = (float *)malloc(sizeof(float) * cnt * 5);
      int *ids = (int *)malloc(sizeof(int) * cnt);
      int q = 0;
      for (int i = 0; i < nc; ++i)
        if (cls_of[i] == c && exp((double)conf_of[i]) > 0.2) {
          for (int t = 0; t < 4; ++t) bb5[q * 5 + t] = (float)r2[4 * i + t]; /* :129 */
          bb5[q * 5 + 4] = conf_of[i];                                        /* :130 */
          ids[q++] = i;
        }
      int64_t *pk = (int64_t *)malloc(sizeof(int64_t) * cnt);
      int npk = orc_nms(bb5, cnt, 5, 0.1f, 0, 0, pk); /* :133 tensor scores -> key y2 */
      for (int t = 0; t < npk; ++t) {
        int i = ids[pk[t] - 1];
        if (nw < cap) {
          double *row = win_rows + 7 * nw;
          row[0] = c; row[1] = conf_of[i];
          memcpy(row + 2, r2 + 4 * i, 4 * sizeof(double));
          row[6] = i + 1;
        }
        ++nw;
      }
      free(bb5); free(ids); free(pk);
    }
    counts->nwin = nw;
    free(cls_of); free(conf_of); free(r2);
  }
  orc_anchors_free(A);
  orc_pnet_state_free(ps);
}
