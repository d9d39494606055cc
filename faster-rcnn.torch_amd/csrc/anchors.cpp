// anchors.cpp -- host-side example assembly of BatchIterator:nextTraining (BatchIterator.lua:198-225) as native
// code: Anchors:findRangesXY / findPositive / sampleNegative / findNearby (Anchors.lua:69-235), the nearby-aversion
// filter and shuffle_n (utilities.lua:31-42).  No device work: at 200+ images/s per GPU this per-image geometry (a few
// thousand IoU evaluations and a few hundred random draws) is what keeps a host core busy, and the Python mirror
// (Anchors.py / synthetic.assemble_examples) spends 2.5 ms per image on it.  Same arithmetic (fp64 on the fp32 anchor
// tables), same iteration order and the same MT19937 draw sequence as the mirror, so both produce identical
// example lists (tests/test_host_mirror.py).
#include <cmath>
#include <cstdint>
#include <vector>

#include "common.h"

namespace {

struct Bin { int layer, aspect, idx; };

struct Range {
  int layer, aspect, lx, ly, ux, uy;   // 1-based, [lx, ux) x [ly, uy)
};

struct Example { int layer, aspect, y, x, roi; double r[4]; };

struct Mt {   // torch.random(): raw 32-bit Mersenne-Twister draws
  uint32_t* mt;
  int* idx;
  uint32_t next() {
    if (*idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7FFFFFFFu);
        uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908B0DFu;
        mt[k] = v;
      }
      *idx = 0;
    }
    uint32_t y = mt[(*idx)++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
  }
};

inline double rect_iou(const double* a, const double* b) {   // Rect.IoU (Rect.lua:126-141)
  const double minx = std::max(a[0], b[0]), miny = std::max(a[1], b[1]);
  const double maxx = std::min(a[2], b[2]), maxy = std::min(a[3], b[3]);
  double i = 0.0;
  if (maxx >= minx && maxy >= miny) i = (maxx - minx) * (maxy - miny);
  return i / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i);
}

}  // namespace

struct frcnn_anchors {
  int n = 0, width = 0;                 // scales, table length (200)
  std::vector<double> w, h;             // [n][3][width][2] (fp32 values widened, as the Lua code reads them)
  std::vector<std::vector<Bin>> cx, cy; // bins of 16 px, entries in the construction order of Anchors.lua:26-57
  int bin0x = 0, bin0y = 0;

  const double* wrow(int i, int j) const { return w.data() + ((size_t)(i * 3 + j) * width) * 2; }
  const double* hrow(int i, int j) const { return h.data() + ((size_t)(i * 3 + j) * width) * 2; }

  // first 1-based index with t[k][col] >= v / > v  (t ascending)
  int lower(const double* t, int col, double v) const {
    int lo = 0, hi = width;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (t[2 * mid + col] < v) lo = mid + 1; else hi = mid; }
    return lo + 1;
  }
  int upper(const double* t, int col, double v) const {
    int lo = 0, hi = width;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (t[2 * mid + col] <= v) lo = mid + 1; else hi = mid; }
    return lo + 1;
  }

  void ranges(const double* rect, const double* clip, std::vector<Range>* out) const {   // Anchors.lua:86-145
    out->clear();
    for (int i = 0; i < 4 && i < n; ++i)       // (4 scales hard-coded in the reference, Q6)
      for (int j = 0; j < 3; ++j) {
        const double *W = wrow(i, j), *H = hrow(i, j);
        int lx = upper(W, 1, rect[0]), ly = upper(H, 1, rect[1]);
        int ux = lower(W, 0, rect[2]), uy = lower(H, 0, rect[3]);
        if (clip) {
          lx = std::max(lx, lower(W, 0, clip[0])); ly = std::max(ly, lower(H, 0, clip[1]));
          ux = std::min(ux, upper(W, 1, clip[2])); uy = std::min(uy, upper(H, 1, clip[3]));
        }
        if (ux > lx && uy > ly) out->push_back({i + 1, j + 1, lx, ly, ux, uy});
      }
  }

  void rect_of(int layer, int aspect, int y, int x, double* r) const {   // Anchors:get (1-based)
    const double *W = wrow(layer - 1, aspect - 1), *H = hrow(layer - 1, aspect - 1);
    r[0] = W[2 * (x - 1)]; r[1] = H[2 * (y - 1)]; r[2] = W[2 * (x - 1) + 1]; r[3] = H[2 * (y - 1) + 1];
  }
};

extern "C" {

int frcnn_anchors_create(const float* w_tab, const float* h_tab, const double* cx, const double* cy, int nscales,
                         int width, frcnn_anchors** out) {
  FR_CHECK(w_tab && h_tab && cx && cy && out && nscales >= 1 && width >= 1, "anchors_create: bad arguments");
  frcnn_anchors* a = new frcnn_anchors();
  a->n = nscales; a->width = width;
  const size_t cnt = (size_t)nscales * 3 * width * 2;
  a->w.resize(cnt); a->h.resize(cnt);
  for (size_t i = 0; i < cnt; ++i) { a->w[i] = (double)w_tab[i]; a->h[i] = (double)h_tab[i]; }
  auto bin_of = [](double v) { return (int)std::floor(v / 16.0); };   // BIN_SIZE, Anchors.lua:5
  int lo_x = 1 << 30, hi_x = -(1 << 30), lo_y = lo_x, hi_y = hi_x;
  for (size_t i = 0; i < (size_t)nscales * 3 * width; ++i) {
    lo_x = std::min(lo_x, bin_of(cx[i])); hi_x = std::max(hi_x, bin_of(cx[i]));
    lo_y = std::min(lo_y, bin_of(cy[i])); hi_y = std::max(hi_y, bin_of(cy[i]));
  }
  a->bin0x = lo_x; a->bin0y = lo_y;
  a->cx.assign(hi_x - lo_x + 1, {}); a->cy.assign(hi_y - lo_y + 1, {});
  for (int i = 0; i < nscales; ++i)
    for (int j = 0; j < 3; ++j) {
      const size_t base = ((size_t)i * 3 + j) * width;
      for (int y = 1; y <= width; ++y) a->cy[bin_of(cy[base + y - 1]) - lo_y].push_back({i + 1, j + 1, y});
      for (int x = 1; x <= width; ++x) a->cx[bin_of(cx[base + x - 1]) - lo_x].push_back({i + 1, j + 1, x});
    }
  *out = a;
  return FRCNN_OK;
}

int frcnn_anchors_destroy(frcnn_anchors* a) {
  delete a;
  return FRCNN_OK;
}

int frcnn_anchors_assemble(frcnn_anchors* a, const double* rois, int nroi, double img_w, double img_h, double pos_thr,
                           double neg_thr, int best_match, int nearby_aversion, int negatives, unsigned int* mt_state,
                           int* mt_index, int* ex, double* ex_rect, int cap, int* npos_out, int* nneg_out) {
  FR_CHECK(a && mt_state && mt_index && ex && ex_rect && npos_out && nneg_out && (rois || nroi == 0), "anchors_assemble: null argument");
  Mt rng{mt_state, mt_index};
  const double img[4] = {0.0, 0.0, img_w, img_h};
  std::vector<Example> pos, neg;
  std::vector<Range> rg;
  // ---- Anchors:findPositive (Anchors.lua:147-195)
  const double lowthr = std::min(pos_thr, neg_thr);
  for (int ri = 0; ri < nroi; ++ri) {
    const double* g = rois + 4 * ri;
    const double garea = (g[2] - g[0]) * (g[3] - g[1]);
    bool have_best = best_match != 0;
    std::vector<Example> best_set;
    double best_iou = -1.0;
    a->ranges(g, img, &rg);
    for (const Range& r : rg) {
      const double *W = a->wrow(r.layer - 1, r.aspect - 1), *H = a->hrow(r.layer - 1, r.aspect - 1);
      for (int y = r.ly; y < r.uy; ++y) {
        const double y0 = H[2 * (y - 1)], y1 = H[2 * (y - 1) + 1];
        const double iy = std::min(g[3], y1) - std::max(g[1], y0);
        for (int x = r.lx; x < r.ux; ++x) {
          const double x0 = W[2 * (x - 1)], x1 = W[2 * (x - 1) + 1];
          const double ix = std::min(g[2], x1) - std::max(g[0], x0);
          const double inter = (ix >= 0 && iy >= 0) ? ix * iy : 0.0;
          const double aarea = (x1 - x0) * (y1 - y0);
          const double v = inter / (garea + aarea - inter);
          if (!(v > lowthr)) continue;
          Example e{r.layer, r.aspect, y, x, ri + 1, {x0, y0, x1, y1}};
          if (v > pos_thr) {
            pos.push_back(e);
            have_best = false;
          } else if (v > neg_thr && have_best && v >= best_iou) {   // hysteresis 0.025, `>=` ties (Q7)
            if (v - 0.025 > best_iou) best_set.clear();
            best_set.push_back(e);
            best_iou = v;
          }
        }
      }
    }
    if (have_best && best_iou > 0)
      for (const Example& e : best_set) pos.push_back(e);
  }
  // ---- Anchors:sampleNegative (Anchors.lua:197-235)
  a->ranges(img, img, &rg);
  FR_CHECK(negatives <= 0 || !rg.empty(), "anchors_assemble: no anchor lies inside a %gx%g image", img_w, img_h);
  int retry = 0;
  while ((int)neg.size() < negatives && retry < 500) {
    const Range& r = rg[rng.next() % rg.size()];
    const int x = (int)(rng.next() % (uint32_t)(r.ux - r.lx)) + 1;
    const int y = (int)(rng.next() % (uint32_t)(r.uy - r.ly)) + 1;
    Example e{r.layer, r.aspect, r.ly + y - 1, r.lx + x - 1, 0, {0, 0, 0, 0}};
    a->rect_of(e.layer, e.aspect, e.y, e.x, e.r);
    bool match = false;
    for (int ri = 0; ri < nroi && !match; ++ri) match = rect_iou(rois + 4 * ri, e.r) > neg_thr;
    if (!match) { retry = 0; neg.push_back(e); } else { ++retry; }
  }
  // ---- nearby aversion (BatchIterator.lua:204-224) + shuffle_n (utilities.lua:31-42)
  if (nearby_aversion) {
    const int count = (int)(pos.size() + neg.size());
    std::vector<Example> nearby;
    for (const Example& p : pos) {
      const double cxp = (p.r[0] + p.r[2]) / 2, cyp = (p.r[1] + p.r[3]) / 2;
      const int bx = (int)std::floor(cxp / 16.0) - a->bin0x, by = (int)std::floor(cyp / 16.0) - a->bin0y;
      if (bx < 0 || by < 0 || bx >= (int)a->cx.size() || by >= (int)a->cy.size()) continue;
      for (const Bin& yb : a->cy[by])
        for (const Bin& xb : a->cx[bx])
          if (yb.layer == xb.layer && yb.aspect == xb.aspect) {
            Example e{yb.layer, yb.aspect, yb.idx, xb.idx, 0, {0, 0, 0, 0}};
            a->rect_of(e.layer, e.aspect, e.y, e.x, e.r);
            if (rect_iou(p.r, e.r) < neg_thr) nearby.push_back(e);
          }
    }
    int c = std::min((int)pos.size(), count);
    c = std::min(c, (int)nearby.size());
    size_t rem = nearby.size();
    for (int i = 0; i < c; ++i) {
      const size_t j = rng.next() % rem + i;
      std::swap(nearby[i], nearby[j]);
      --rem;
    }
    neg.insert(neg.end(), nearby.begin(), nearby.begin() + c);
  }
  *npos_out = (int)pos.size(); *nneg_out = (int)neg.size();
  FR_CHECK((int)(pos.size() + neg.size()) <= cap, "anchors_assemble: %zu examples exceed the capacity %d", pos.size() + neg.size(), cap);
  size_t k = 0;
  for (const std::vector<Example>* v : {&pos, &neg})
    for (const Example& e : *v) {
      int* o = ex + 5 * k;
      o[0] = e.layer; o[1] = e.aspect; o[2] = e.y; o[3] = e.x; o[4] = e.roi;
      for (int t = 0; t < 4; ++t) ex_rect[4 * k + t] = e.r[t];
      ++k;
    }
  return FRCNN_OK;
}

}  // extern "C"
