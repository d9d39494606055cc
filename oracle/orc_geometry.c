/*
 * orc_geometry.c -- CPU restatement of Rect.lua, Localizer.lua, Anchors.lua, nms.lua and
 * extract_roi_pooling_input (objective.lua:5-13).  TEST INFRASTRUCTURE ONLY (see header).
 * Every function cites the reference file:line it follows.  All scalar math is double
 * ("Lua number"); anchor tables are stored in fp32 because main.lua:51 sets the default
 * tensor type to Float before Anchors.__init allocates them (Anchors.lua:18-19).
 */
#include "frcnn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Rect.lua */

static double dmin(double a, double b) { return a < b ? a : b; } /* math.min */
static double dmax(double a, double b) { return a > b ? a : b; } /* math.max */

/* Rect.lua:126-136 -- empty() = (0,0,0,0) when disjoint */
void orc_rect_intersect(const double *a, const double *b, double *out) {
  double minx = dmax(a[0], b[0]);
  double miny = dmax(a[1], b[1]);
  double maxx = dmin(a[2], b[2]);
  double maxy = dmin(a[3], b[3]);
  if (maxx >= minx && maxy >= miny) {
    out[0] = minx; out[1] = miny; out[2] = maxx; out[3] = maxy;
  } else {
    out[0] = out[1] = out[2] = out[3] = 0.0;
  }
}

static double rect_area(const double *r) { return (r[2] - r[0]) * (r[3] - r[1]); } /* Rect.lua:61-63 */

/* Rect.lua:138-141 -- no +1 convention here (unlike nms.lua:35) */
double orc_rect_iou(const double *a, const double *b) {
  double t[4];
  orc_rect_intersect(a, b, t);
  double i = rect_area(t);
  return i / (rect_area(a) + rect_area(b) - i);
}

/* Rect.lua:90-93 -- strict */
int orc_rect_overlaps(const double *a, const double *b) {
  return a[0] < b[2] && a[2] > b[0] && a[1] < b[3] && a[3] > b[1];
}

/* Rect.lua:73-80 */
void orc_rect_clip(const double *r, const double *c, double *out) {
  double o0 = dmin(dmax(r[0], c[0]), c[2]);
  double o1 = dmin(dmax(r[1], c[1]), c[3]);
  double o2 = dmax(dmin(r[2], c[2]), c[0]);
  double o3 = dmax(dmin(r[3], c[3]), c[1]);
  out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3;
}

/* Rect.lua:147-149 */
void orc_rect_snap_to_int(const double *r, double *out) {
  out[0] = floor(r[0]); out[1] = floor(r[1]); out[2] = ceil(r[2]); out[3] = ceil(r[3]);
}

/* ------------------------------------------------------------------ Localizer.lua */

/* Lua 5.1 / LuaJIT `a % b` on numbers: a - floor(a/b)*b */
static double lua_mod(double a, double b) { return a - floor(a / b) * b; }

/* Localizer.lua:41-67.  Axis mix-ups of the reference (dH for X, dW in the maxY exact branch)
 * are kept verbatim. layers[i] = {kW,kH,dW,dH,padW,padH}. */
void orc_loc_input_to_feature(const int *layers, int nlayers, int layer_index,
                              const double *rect, double *out) {
  double minX = rect[0], minY = rect[1], maxX = rect[2], maxY = rect[3];
  if (layer_index <= 0) layer_index = nlayers;
  for (int i = 0; i < layer_index; ++i) {
    const int *l = layers + 6 * i;
    double kW = l[0], kH = l[1], dW = l[2], dH = l[3], padW = l[4], padH = l[5];
    if (dW < kW) { /* :45-47 inflate(kW-dW, kH-dH) */
      minX -= (kW - dW); minY -= (kH - dH); maxX += (kW - dW); maxY += (kH - dH);
    }
    minX += padW; minY += padH; maxX += padW; maxY += padH; /* :49 offset */
    minX = minX / dH;                                       /* :52 (dH, sic) */
    minY = minY / dH;                                       /* :53 */
    if (lua_mod(maxX - kW, dW) == 0.0)                      /* :54 */
      maxX = dmax((maxX - kW) / dW + 1.0, minX + 1.0);
    else
      maxX = dmax(ceil((maxX - kW) / dW) + 1.0, minX + 1.0);
    if (lua_mod(maxY - kH, dH) == 0.0)                      /* :59 */
      maxY = dmax((maxY - kH) / dW + 1.0, minY + 1.0);      /* :60 (dW, sic) */
    else
      maxY = dmax(ceil((maxY - kH) / dH) + 1.0, minY + 1.0);
  }
  double r[4] = {minX, minY, maxX, maxY};
  orc_rect_snap_to_int(r, out); /* :66 */
}

/* Localizer.lua:69-79 (padW for minY and padH for maxX, sic) */
void orc_loc_feature_to_input(const int *layers, int nlayers, int layer_index, double minX,
                              double minY, double maxX, double maxY, double *out) {
  if (layer_index <= 0) layer_index = nlayers;
  for (int i = layer_index - 1; i >= 0; --i) {
    const int *l = layers + 6 * i;
    double kW = l[0], kH = l[1], dW = l[2], dH = l[3], padW = l[4], padH = l[5];
    minX = minX * dW - padW;
    minY = minY * dH - padW;
    maxX = maxX * dW - padH + kW - dW;
    maxY = maxY * dH - padH + kH - dH;
  }
  out[0] = minX; out[1] = minY; out[2] = maxX; out[3] = maxY;
}

/* ------------------------------------------------------------------ MT19937 ([ext]) */
/* torch.random() = THRandom_random = the reference mt19937ar genrand_int32. */
struct orc_mt {
  uint32_t mt[624];
  int idx;
};

orc_mt *orc_mt_new(uint32_t seed) {
  orc_mt *g = (orc_mt *)malloc(sizeof(orc_mt));
  g->mt[0] = seed;
  for (int j = 1; j < 624; ++j)
    g->mt[j] = 1812433253u * (g->mt[j - 1] ^ (g->mt[j - 1] >> 30)) + (uint32_t)j;
  g->idx = 624;
  return g;
}
void orc_mt_free(orc_mt *g) { free(g); }

uint32_t orc_mt_random(orc_mt *g) {
  if (g->idx >= 624) {
    for (int k = 0; k < 624; ++k) {
      uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(k + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      g->mt[k] = v;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* ------------------------------------------------------------------ Anchors.lua */

#define ANCH_N 200 /* Anchors.lua:15 */
#define BIN_SIZE 16.0 /* Anchors.lua:5 */

typedef struct { int i, j, v; long key; } nearby_entry;

struct orc_anchors {
  int nscales;
  float *w; /* [nscales][3][200][2] */
  float *h;
  nearby_entry *cx, *cy; /* insertion order (Anchors.lua:24-30) */
  int ncx, ncy;
};

#define WIDX(s, a, i, m) ((((size_t)(s) * 3 + (a)) * ANCH_N + (i)) * 2 + (m))

orc_anchors *orc_anchors_new(const int *layers_concat, const int *nlayers, const double *scales,
                             int nscales) {
  orc_anchors *A = (orc_anchors *)calloc(1, sizeof(orc_anchors));
  A->nscales = nscales;
  A->w = (float *)calloc((size_t)nscales * 3 * ANCH_N * 2, sizeof(float));
  A->h = (float *)calloc((size_t)nscales * 3 * ANCH_N * 2, sizeof(float));
  A->cx = (nearby_entry *)malloc(sizeof(nearby_entry) * nscales * 3 * ANCH_N);
  A->cy = (nearby_entry *)malloc(sizeof(nearby_entry) * nscales * 3 * ANCH_N);
  const int *L = layers_concat;
  for (int i = 0; i < nscales; ++i) { /* Anchors.lua:32 */
    double s = scales[i];
    double a = s / sqrt(2.0); /* :34 */
    double aspects[3][2] = {{s, s}, {2 * a, a}, {a, 2 * a}}; /* :35 */
    for (int j = 0; j < 3; ++j) {
      double bw = aspects[j][0], bh = aspects[j][1];
      for (int y = 1; y <= ANCH_N; ++y) { /* :39-46 */
        double r[4];
        orc_loc_feature_to_input(L, nlayers[i], 0, 0, y - 1, 0, y, r);
        double centerY = (r[1] + r[3]) / 2; /* Rect.lua:65-67 */
        double minY = centerY - bh * 0.5;   /* Rect.lua:34-36, :30-32 */
        double maxY = minY + bh;
        A->h[WIDX(i, j, y - 1, 0)] = (float)minY; /* fp32 store */
        A->h[WIDX(i, j, y - 1, 1)] = (float)maxY;
        nearby_entry e = {i + 1, j + 1, y, (long)floor(centerY / BIN_SIZE)};
        A->cy[A->ncy++] = e;
      }
      for (int x = 1; x <= ANCH_N; ++x) { /* :48-55 */
        double r[4];
        orc_loc_feature_to_input(L, nlayers[i], 0, x - 1, 0, x, 0, r);
        double centerX = (r[0] + r[2]) / 2;
        double minX = centerX - bw * 0.5;
        double maxX = minX + bw;
        A->w[WIDX(i, j, x - 1, 0)] = (float)minX;
        A->w[WIDX(i, j, x - 1, 1)] = (float)maxX;
        nearby_entry e = {i + 1, j + 1, x, (long)floor(centerX / BIN_SIZE)};
        A->cx[A->ncx++] = e;
      }
    }
    L += 6 * nlayers[i];
  }
  return A;
}

void orc_anchors_free(orc_anchors *A) {
  if (!A) return;
  free(A->w); free(A->h); free(A->cx); free(A->cy); free(A);
}
const float *orc_anchors_w(const orc_anchors *A) { return A->w; }
const float *orc_anchors_h(const orc_anchors *A) { return A->h; }

/* Anchors.lua:60-67 (1-based layer/aspect/y/x) */
void orc_anchors_get(const orc_anchors *A, int layer, int aspect, int y, int x, double *r) {
  r[0] = A->w[WIDX(layer - 1, aspect - 1, x - 1, 0)];
  r[1] = A->h[WIDX(layer - 1, aspect - 1, y - 1, 0)];
  r[2] = A->w[WIDX(layer - 1, aspect - 1, x - 1, 1)];
  r[3] = A->h[WIDX(layer - 1, aspect - 1, y - 1, 1)];
}

/* Anchors.lua:87-95 over column m of table t[(s,a)][200][2]; returns 1-based index */
static int lower_bound(const float *t, int m, double value) {
  int low = 1, high = ANCH_N;
  while (low <= high) {
    int mid = (int)floor((low + high) / 2.0);
    double tv = t[(mid - 1) * 2 + m];
    if (tv >= value) high = mid - 1;
    else if (tv < value) low = mid + 1;
  }
  return low;
}
/* Anchors.lua:96-104 */
static int upper_bound(const float *t, int m, double value) {
  int low = 1, high = ANCH_N;
  while (low <= high) {
    int mid = (int)floor((low + high) / 2.0);
    double tv = t[(mid - 1) * 2 + m];
    if (tv > value) high = mid - 1;
    else if (tv <= value) low = mid + 1;
  }
  return low;
}

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* Anchors.lua:86-145.  Hard-codes 4 scales x 3 aspects like the reference (:108-109). */
int orc_anchors_find_ranges_xy(const orc_anchors *A, const double *rect, const double *clip,
                               int *ranges_out) {
  int n = 0;
  for (int i = 1; i <= 4; ++i) {
    for (int j = 1; j <= 3; ++j) {
      const float *wt = A->w + WIDX(i - 1, j - 1, 0, 0);
      const float *ht = A->h + WIDX(i - 1, j - 1, 0, 0);
      int clx = 0, cly = 0, cux = 0, cuy = 0;
      if (clip) { /* :112-118 */
        clx = lower_bound(wt, 0, clip[0]);
        cly = lower_bound(ht, 0, clip[1]);
        cux = upper_bound(wt, 1, clip[2]);
        cuy = upper_bound(ht, 1, clip[3]);
      }
      int lx = upper_bound(wt, 1, rect[0]); /* :123-126 */
      int ly = upper_bound(ht, 1, rect[1]);
      int ux = lower_bound(wt, 0, rect[2]);
      int uy = lower_bound(ht, 0, rect[3]);
      if (clip) { /* :128-133 */
        lx = imax(lx, clx); ly = imax(ly, cly); ux = imin(ux, cux); uy = imin(uy, cuy);
      }
      if (ux > lx && uy > ly) { /* :135 */
        int *o = ranges_out + 6 * n;
        o[0] = i; o[1] = j; o[2] = lx; o[3] = ly; o[4] = ux; o[5] = uy;
        ++n;
      }
    }
  }
  return n;
}

typedef struct { int idx[4]; double rect[4]; } anchor_rec;

/* Anchors.lua:147-195 */
int orc_anchors_find_positive(const orc_anchors *A, const double *rois, int nroi,
                              const double *clip, double pos_thr, double neg_thr,
                              int include_best, int *out_idx, double *out_rect, int cap) {
  int nm = 0;
  anchor_rec *best = NULL;
  int nbest = 0, capbest = 0, have_best = 0;
  double best_iou = 0;
#define EMIT(IDX, RECT, ROI)                                                        \
  do {                                                                              \
    if (nm < cap) {                                                                 \
      int *o_ = out_idx + 5 * nm;                                                   \
      o_[0] = (IDX)[0]; o_[1] = (IDX)[1]; o_[2] = (IDX)[2]; o_[3] = (IDX)[3];       \
      o_[4] = (ROI);                                                                \
      memcpy(out_rect + 4 * nm, (RECT), 4 * sizeof(double));                        \
    }                                                                               \
    ++nm;                                                                           \
  } while (0)

  for (int ri = 0; ri < nroi; ++ri) {
    const double *roi = rois + 4 * ri;
    if (include_best) { /* :153-156 */
      have_best = 1; nbest = 0; best_iou = -1;
    }
    int ranges[12 * 6];
    int nr = orc_anchors_find_ranges_xy(A, roi, clip, ranges);
    for (int j = 0; j < nr; ++j) {
      const int *r = ranges + 6 * j;
      int layer = r[0], aspect = r[1], lx = r[2], ly = r[3], ux = r[4], uy = r[5];
      const float *wt = A->w + WIDX(layer - 1, aspect - 1, 0, 0);
      const float *ht = A->h + WIDX(layer - 1, aspect - 1, 0, 0);
      for (int y = 1; y <= uy - ly; ++y) { /* :162 ys = h[ly..uy-1] */
        double minY = ht[(ly + y - 2) * 2 + 0], maxY = ht[(ly + y - 2) * 2 + 1];
        for (int x = 1; x <= ux - lx; ++x) {
          double ar[4] = {wt[(lx + x - 2) * 2 + 0], minY, wt[(lx + x - 2) * 2 + 1], maxY};
          int idx[4] = {layer, aspect, ly + y - 1, lx + x - 1}; /* :169 */
          double v = orc_rect_iou(roi, ar);                     /* :171 */
          if (v > pos_thr) {                                    /* :172-174 */
            EMIT(idx, ar, ri + 1);
            have_best = 0;
          } else if (v > neg_thr && have_best && v >= best_iou) { /* :175 */
            if (v - 0.025 > best_iou) nbest = 0;                  /* :176-178 */
            if (nbest == capbest) {
              capbest = capbest ? capbest * 2 : 64;
              best = (anchor_rec *)realloc(best, sizeof(anchor_rec) * capbest);
            }
            memcpy(best[nbest].idx, idx, sizeof(idx));
            memcpy(best[nbest].rect, ar, sizeof(ar));
            ++nbest;
            best_iou = v; /* :180 */
          }
        }
      }
    }
    if (have_best && best_iou > 0) { /* :186-190 */
      for (int k = 0; k < nbest; ++k) EMIT(best[k].idx, best[k].rect, ri + 1);
    }
  }
#undef EMIT
  free(best);
  return nm;
}

/* Anchors.lua:197-235 */
int orc_anchors_sample_negative(const orc_anchors *A, const double *image_rect,
                                const double *rois, int nroi, double neg_thr, int count,
                                orc_mt *rng, int *out_idx, double *out_rect, int cap) {
  int ranges[12 * 6];
  int nr = orc_anchors_find_ranges_xy(A, image_rect, image_rect, ranges); /* :199 */
  int nneg = 0, retry = 0;
  if (nr == 0) return 0; /* reference would raise (index nil); documented divergence */
  while (nneg < count && retry < 500) { /* :204 */
    const int *r = ranges + 6 * (orc_mt_random(rng) % (uint32_t)nr); /* :207 */
    int layer = r[0], aspect = r[1], lx = r[2], ly = r[3], ux = r[4], uy = r[5];
    int x = (int)(orc_mt_random(rng) % (uint32_t)(ux - lx)) + 1; /* :208 */
    int y = (int)(orc_mt_random(rng) % (uint32_t)(uy - ly)) + 1; /* :209 */
    const float *wt = A->w + WIDX(layer - 1, aspect - 1, 0, 0);
    const float *ht = A->h + WIDX(layer - 1, aspect - 1, 0, 0);
    double ar[4] = {wt[(lx + x - 2) * 2 + 0], ht[(ly + y - 2) * 2 + 0],
                    wt[(lx + x - 2) * 2 + 1], ht[(ly + y - 2) * 2 + 1]}; /* :211 */
    int match = 0;
    for (int j = 0; j < nroi; ++j) { /* :218-223 */
      if (orc_rect_iou(rois + 4 * j, ar) > neg_thr) { match = 1; break; }
    }
    if (!match) { /* :225-230 */
      retry = 0;
      if (nneg < cap) {
        int *o = out_idx + 4 * nneg;
        o[0] = layer; o[1] = aspect; o[2] = ly + y - 1; o[3] = lx + x - 1; /* :214 */
        memcpy(out_rect + 4 * nneg, ar, sizeof(ar));
      }
      ++nneg;
    } else {
      ++retry;
    }
  }
  return nneg;
}

/* Anchors.lua:69-84 */
int orc_anchors_find_nearby(const orc_anchors *A, double centerX, double centerY, int *out_idx,
                            double *out_rect, int cap) {
  long kx = (long)floor(centerX / BIN_SIZE), ky = (long)floor(centerY / BIN_SIZE);
  int n = 0;
  for (int i = 0; i < A->ncy; ++i) {
    const nearby_entry *y = &A->cy[i];
    if (y->key != ky) continue;
    for (int j = 0; j < A->ncx; ++j) {
      const nearby_entry *x = &A->cx[j];
      if (x->key != kx) continue;
      if (y->i == x->i && y->j == x->j) { /* :77 */
        if (n < cap) {
          int *o = out_idx + 4 * n;
          o[0] = y->i; o[1] = y->j; o[2] = y->v; o[3] = x->v;
          orc_anchors_get(A, y->i, y->j, y->v, x->v, out_rect + 4 * n);
        }
        ++n;
      }
    }
  }
  return n;
}

/* Anchors.lua:237-243 -- top-left relative; result is a FloatTensor */
void orc_input_to_anchor(const double *anchor, const double *rect, float *t4) {
  double aw = anchor[2] - anchor[0], ah = anchor[3] - anchor[1];
  t4[0] = (float)((rect[0] - anchor[0]) / aw);
  t4[1] = (float)((rect[1] - anchor[1]) / ah);
  t4[2] = (float)log((rect[2] - rect[0]) / aw);
  t4[3] = (float)log((rect[3] - rect[1]) / ah);
}

/* Anchors.lua:245-252 + Rect.lua:30-32 */
void orc_anchor_to_input(const double *anchor, const float *t4, double *r) {
  double aw = anchor[2] - anchor[0], ah = anchor[3] - anchor[1];
  double x = (double)t4[0] * aw + anchor[0];
  double y = (double)t4[1] * ah + anchor[1];
  double w = exp((double)t4[2]) * aw;
  double h = exp((double)t4[3]) * ah;
  r[0] = x; r[1] = y; r[2] = x + w; r[3] = y + h;
}

/* ------------------------------------------------------------------ nms.lua */

typedef struct { float key; int id; } sort_rec;
static int cmp_sort_rec(const void *pa, const void *pb) {
  const sort_rec *a = (const sort_rec *)pa, *b = (const sort_rec *)pb;
  if (a->key < b->key) return -1;
  if (a->key > b->key) return 1;
  return (a->id > b->id) - (a->id < b->id); /* documented tie rule */
}

/* nms.lua:23-102.  All tensor arithmetic is fp32, one rounding per TH op; this file must be
 * compiled with -ffp-contract=off (Makefile) so that nothing is fused. */
int orc_nms(const float *boxes, int n, int ncols, float overlap, int key_mode, int key_col,
            int64_t *pick_out) {
  if (n == 0) return 0; /* :26-28 */
  float *area = (float *)malloc(sizeof(float) * n);
  sort_rec *rec = (sort_rec *)malloc(sizeof(sort_rec) * n);
  int *I = (int *)malloc(sizeof(int) * n);
  for (int k = 0; k < n; ++k) {
    const float *b = boxes + (size_t)k * ncols;
    volatile float dx = b[2] - b[0]; /* x2 - x1 */
    volatile float dy = b[3] - b[1];
    volatile float dx1 = dx + 1.0f;  /* + 1 */
    volatile float dy1 = dy + 1.0f;
    area[k] = dx1 * dy1;             /* :35 cmul */
  }
  for (int k = 0; k < n; ++k) { /* :37-43 */
    const float *b = boxes + (size_t)k * ncols;
    rec[k].id = k;
    rec[k].key = key_mode == 2 ? b[key_col - 1] : (key_mode == 1 ? area[k] : b[3]);
  }
  qsort(rec, n, sizeof(sort_rec), cmp_sort_rec); /* :45 ascending */
  for (int k = 0; k < n; ++k) I[k] = rec[k].id;
  int m = n, count = 0;
  while (m > 0) { /* :58 */
    int i = I[m - 1]; /* :59-60 */
    pick_out[count++] = (int64_t)i + 1; /* :62 */
    if (m == 1) break; /* :65-67 */
    --m;               /* :69 */
    const float *bi = boxes + (size_t)i * ncols;
    int keep = 0;
    for (int q = 0; q < m; ++q) {
      int j = I[q];
      const float *bj = boxes + (size_t)j * ncols;
      float xx1 = bj[0] > bi[0] ? bj[0] : bi[0]; /* :78 cmax */
      float yy1 = bj[1] > bi[1] ? bj[1] : bi[1];
      float xx2 = bj[2] < bi[2] ? bj[2] : bi[2]; /* :80 cmin */
      float yy2 = bj[3] < bi[3] ? bj[3] : bi[3];
      volatile float w = xx2 + (-1.0f) * xx1; /* :85 torch.add(w, xx2, -1, xx1) */
      w = w + 1.0f;                           /* :add(1) */
      float w0 = w > 0.0f ? w : 0.0f;         /* :cmax(0) */
      volatile float h = yy2 + (-1.0f) * yy1; /* :86 */
      h = h + 1.0f;
      float h0 = h > 0.0f ? h : 0.0f;
      volatile float inter = w0 * h0;         /* :89 */
      volatile float denom = area[j] + area[i]; /* :94 xx1 + area[i] */
      denom = denom - inter;                    /*      - inter */
      float iou = inter / denom;                /* :94 cdiv */
      if (iou <= overlap) I[keep++] = j;        /* :96 (order preserved) */
    }
    m = keep;
  }
  free(area); free(rec); free(I);
  return count;
}

/* ------------------------------------------------------------------ objective.lua:5-13 */
void orc_extract_roi_window(const int *layers, int nlayers, const double *input_rect, int fmH,
                            int fmW, int *win) {
  double r[4], c[4] = {0, 0, (double)fmW, (double)fmH};
  orc_loc_input_to_feature(layers, nlayers, 0, input_rect, r); /* :6 */
  orc_rect_clip(r, c, r);                                     /* :10 */
  win[0] = (int)dmin(r[1] + 1, r[3]); /* :11 rows { min(minY+1,maxY), maxY } */
  win[1] = (int)r[3];
  win[2] = (int)dmin(r[0] + 1, r[2]); /*     cols { min(minX+1,maxX), maxX } */
  win[3] = (int)r[2];
}
