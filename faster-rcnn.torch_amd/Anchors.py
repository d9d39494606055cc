"""Anchors -- anchor geometry tables (4 scales x 3 aspects x 200 positions), anchor lookup,
ground-truth -> anchor labelling, negative sampling and the bbox parameterisation.  Host-side
mirror of Anchors.lua with the same method names.  Tables are fp32 (`torch.Tensor` under
main.lua:51) and are consumed as doubles, exactly like the reference.  The tables are also what the
device-side RPN scan (frcnn_rpn_scan) reads."""
import ctypes as C
import math

import numpy as np

from .Localizer import Localizer
from .Rect import Rect

BIN_SIZE = 16  # Anchors.lua:5


class MT19937(object):
    """torch.random(): raw 32-bit Mersenne-Twister draws (TH's THRandom_random, [ext])."""

    def __init__(self, seed=5489):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for j in range(1, 624):
            mt[j] = (1812433253 * (mt[j - 1] ^ (mt[j - 1] >> 30)) + j) & 0xFFFFFFFF
        # the state lives in a uint32 array + a C int so that the native example assembly (frcnn_anchors_assemble)
        # draws from the very same stream in place
        self.state = np.array(mt, dtype=np.uint32)
        self.cidx = C.c_int(624)

    @property
    def idx(self):
        return self.cidx.value

    def random(self):
        if self.cidx.value >= 624:
            mt = self.state.tolist()
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                v = mt[(k + 397) % 624] ^ (y >> 1)
                if y & 1:
                    v ^= 0x9908B0DF
                mt[k] = v
            self.state[:] = mt
            self.cidx.value = 0
        y = int(self.state[self.cidx.value]); self.cidx.value += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def uniform(self):
        """[0, 1) double from one draw (TH's THRandom_uniform); stands in for LuaJIT's math.random()."""
        return self.random() * (1.0 / 4294967296.0)

    def randperm(self, n):
        """torch.randperm(n) (TH: Fisher-Yates from the front with random() % (n - i)), 1-based."""
        r = list(range(n))
        for i in range(n - 1):
            z = self.random() % (n - i)
            r[i], r[z + i] = r[z + i], r[i]
        return [v + 1 for v in r]


_default_rng = MT19937()


def manualSeed(seed):
    global _default_rng
    _default_rng = MT19937(seed)


class Anchors(object):
    def __init__(self, proposal_net, scales):  # Anchors.lua:7-58
        self.localizers = [Localizer(proposal_net.outnode.children[i]) for i in range(len(scales))]
        width, height = 200, 200  # :15
        n = len(scales)
        self.w = np.zeros((n, 3, width, 2), dtype=np.float32)
        self.h = np.zeros((n, 3, height, 2), dtype=np.float32)
        self.cx, self.cy = {}, {}
        self._cxx = np.zeros((n, 3, width), dtype=np.float64)   # anchor centres (the keys of the 16-px bins)
        self._cyy = np.zeros((n, 3, height), dtype=np.float64)
        self._native = None

        def add(m, i, j, v, x):
            m.setdefault(math.floor(x / BIN_SIZE), []).append((i, j, v))

        for i, s in enumerate(scales):
            a = s / math.sqrt(2)
            aspects = ((s, s), (2 * a, a), (a, 2 * a))  # :35
            loc = self.localizers[i]
            for j, b in enumerate(aspects):
                for y in range(1, height + 1):
                    r = loc.featureToInputRect(0, y - 1, 0, y)
                    cxx, cyy = r.center()
                    r = Rect.fromCenterWidthHeight(cxx, cyy, b[0], b[1])
                    self.h[i, j, y - 1, 0] = r.minY
                    self.h[i, j, y - 1, 1] = r.maxY
                    add(self.cy, i + 1, j + 1, y, cyy)
                    self._cyy[i, j, y - 1] = cyy
                for x in range(1, width + 1):
                    r = loc.featureToInputRect(x - 1, 0, x, 0)
                    cxx, cyy = r.center()
                    r = Rect.fromCenterWidthHeight(cxx, cyy, b[0], b[1])
                    self.w[i, j, x - 1, 0] = r.minX
                    self.w[i, j, x - 1, 1] = r.maxX
                    add(self.cx, i + 1, j + 1, x, cxx)
                    self._cxx[i, j, x - 1] = cxx
        self._w64 = self.w.astype(np.float64)
        self._h64 = self.h.astype(np.float64)

    new = None

    @staticmethod
    def _tag(r, layer, aspect, y, x):
        r.layer = layer; r.aspect = aspect
        r.index = ((aspect * 6 - 5, aspect * 6), y, x)  # 1-based like the Lua tables
        return r

    def get(self, layer, aspect, y, x):  # Anchors.lua:60-67 (1-based arguments)
        w, h = self._w64, self._h64
        r = Rect(w[layer - 1, aspect - 1, x - 1, 0], h[layer - 1, aspect - 1, y - 1, 0],
                 w[layer - 1, aspect - 1, x - 1, 1], h[layer - 1, aspect - 1, y - 1, 1])
        return Anchors._tag(r, layer, aspect, y, x)

    def findNearby(self, centerX, centerY):  # Anchors.lua:69-84
        found = []
        xl = self.cx.get(math.floor(centerX / BIN_SIZE)); yl = self.cy.get(math.floor(centerY / BIN_SIZE))
        if xl and yl:
            for y in yl:
                for x in xl:
                    if y[0] == x[0] and y[1] == x[1]:
                        found.append(self.get(y[0], y[1], y[2], x[2]))
        return found

    def native(self):
        """Handle of the native twin of these tables (frcnn_anchors_create) for frcnn_anchors_assemble."""
        if self._native is None:
            from . import _lib
            h = C.c_void_p()
            _lib.call("frcnn_anchors_create", self.w.ctypes.data_as(C.c_void_p), self.h.ctypes.data_as(C.c_void_p),
                      self._cxx.ctypes.data_as(C.c_void_p), self._cyy.ctypes.data_as(C.c_void_p), self.w.shape[0], self.w.shape[2],
                      C.byref(h))
            self._native = h
        return self._native

    def __del__(self):
        if getattr(self, "_native", None) is not None:
            try:
                from . import _lib
                _lib.load().frcnn_anchors_destroy(self._native)
            except Exception:
                pass
            self._native = None

    def findNearbyArrays(self, centerX, centerY):
        """findNearby as arrays, in the same order: meta int[n][4] = (layer, aspect, y, x) 1-based and rects float64[n][4]
        = (minX, minY, maxX, maxY).  Cached per pair of bins (the anchors of a bin pair never change)."""
        key = (math.floor(centerX / BIN_SIZE), math.floor(centerY / BIN_SIZE))
        cache = self.__dict__.setdefault("_nearby_cache", {})
        hit = cache.get(key)
        if hit is None:
            xl = self.cx.get(key[0]); yl = self.cy.get(key[1])
            meta = []
            if xl and yl:
                for y in yl:
                    for x in xl:
                        if y[0] == x[0] and y[1] == x[1]:
                            meta.append((y[0], y[1], y[2], x[2]))
            meta = np.array(meta, dtype=np.int64).reshape(-1, 4)
            w, h = self._w64, self._h64
            li, ai, yi, xi = meta[:, 0] - 1, meta[:, 1] - 1, meta[:, 2] - 1, meta[:, 3] - 1
            rects = np.stack([w[li, ai, xi, 0], h[li, ai, yi, 0], w[li, ai, xi, 1], h[li, ai, yi, 1]], 1) if len(meta) else np.zeros((0, 4))
            hit = cache[key] = (meta, rects)
        return hit

    def findRangesXY(self, rect, clip_rect=None):  # Anchors.lua:86-145
        def lower_bound(t, value):  # first index (1-based) with t >= value
            return int(np.searchsorted(t, value, side="left")) + 1

        def upper_bound(t, value):  # first index with t > value
            return int(np.searchsorted(t, value, side="right")) + 1

        ranges = []
        w, h = self._w64, self._h64
        for i in range(4):      # :108 (4 scales hard-coded in the reference)
            for j in range(3):  # :109
                if clip_rect is not None:
                    clx = lower_bound(w[i, j, :, 0], clip_rect.minX); cly = lower_bound(h[i, j, :, 0], clip_rect.minY)
                    cux = upper_bound(w[i, j, :, 1], clip_rect.maxX); cuy = upper_bound(h[i, j, :, 1], clip_rect.maxY)
                lx = upper_bound(w[i, j, :, 1], rect.minX); ly = upper_bound(h[i, j, :, 1], rect.minY)
                ux = lower_bound(w[i, j, :, 0], rect.maxX); uy = lower_bound(h[i, j, :, 0], rect.maxY)
                if clip_rect is not None:
                    lx = max(lx, clx); ly = max(ly, cly); ux = min(ux, cux); uy = min(uy, cuy)
                if ux > lx and uy > ly:
                    ranges.append(dict(layer=i + 1, aspect=j + 1, lx=lx, ly=ly, ux=ux, uy=uy,
                                       xs=w[i, j, lx - 1:ux - 1, :], ys=h[i, j, ly - 1:uy - 1, :]))
        return ranges

    def findPositive(self, roi_list, clip_rect, pos_threshold, neg_threshold, include_best):  # :147-195
        matches = []
        best_set, best_iou = None, None
        for roi in roi_list:
            if include_best:
                best_set = []; best_iou = -1
            g = roi.rect
            garea = g.area()
            for r in self.findRangesXY(g, clip_rect):
                xs, ys = r["xs"], r["ys"]
                # every anchor of a (layer, aspect) range has the same size, and IoU <= min(area) / max(area): ranges whose
                # anchors are too small or too large to pass the lower threshold cannot contribute (exact pruning)
                a0 = (xs[0, 1] - xs[0, 0]) * (ys[0, 1] - ys[0, 0])
                if min(a0, garea) < max(a0, garea) * min(pos_threshold, neg_threshold) * (1.0 - 1e-9):
                    continue
                # Rect.IoU(roi.rect, anchor) for the whole candidate grid, same operation order
                ix = np.minimum(g.maxX, xs[None, :, 1]) - np.maximum(g.minX, xs[None, :, 0])
                iy = np.minimum(g.maxY, ys[:, None, 1]) - np.maximum(g.minY, ys[:, None, 0])
                inter = np.where((ix >= 0) & (iy >= 0), ix * iy, 0.0)
                aarea = (xs[None, :, 1] - xs[None, :, 0]) * (ys[:, None, 1] - ys[:, None, 0])
                iou = inter / (garea + aarea - inter)
                cand = np.argwhere(iou > min(pos_threshold, neg_threshold))  # row-major = (y, x) scan order
                for y0, x0 in cand:
                    v = float(iou[y0, x0])
                    if v > pos_threshold:
                        a = Rect(xs[x0, 0], ys[y0, 0], xs[x0, 1], ys[y0, 1])
                        matches.append((Anchors._tag(a, r["layer"], r["aspect"], r["ly"] + int(y0), r["lx"] + int(x0)), roi))
                        best_set = None
                    elif v > neg_threshold and best_set is not None and v >= best_iou:
                        if v - 0.025 > best_iou:
                            best_set = []
                        a = Rect(xs[x0, 0], ys[y0, 0], xs[x0, 1], ys[y0, 1])
                        best_set.append(Anchors._tag(a, r["layer"], r["aspect"], r["ly"] + int(y0), r["lx"] + int(x0)))
                        best_iou = v
            if best_set is not None and best_iou > 0:
                for a in best_set:
                    matches.append((a, roi))
        return matches

    def sampleNegative(self, image_rect, roi_list, neg_threshold, count, rng=None):  # :197-235
        rng = rng or _default_rng
        ranges = self.findRangesXY(image_rect, image_rect)
        neg = []
        retry = 0
        while len(neg) < count and retry < 500:
            r = ranges[rng.random() % len(ranges)]
            x = rng.random() % r["xs"].shape[0] + 1
            y = rng.random() % r["ys"].shape[0] + 1
            a = Rect(r["xs"][x - 1, 0], r["ys"][y - 1, 0], r["xs"][x - 1, 1], r["ys"][y - 1, 1])
            Anchors._tag(a, r["layer"], r["aspect"], r["ly"] + y - 1, r["lx"] + x - 1)
            match = False
            for roi in roi_list:
                if Rect.IoU(roi.rect, a) > neg_threshold:
                    match = True
                    break
            if not match:
                retry = 0
                neg.append((a,))
            else:
                retry += 1
        return neg

    @staticmethod
    def inputToAnchor(anchor, rect):  # Anchors.lua:237-243 -> FloatTensor
        x = (rect.minX - anchor.minX) / anchor.width()
        y = (rect.minY - anchor.minY) / anchor.height()
        w = math.log(rect.width() / anchor.width())
        h = math.log(rect.height() / anchor.height())
        return np.array([x, y, w, h], dtype=np.float32)

    @staticmethod
    def anchorToInput(anchor, t):  # Anchors.lua:245-252
        return Rect.fromXYWidthHeight(float(t[0]) * anchor.width() + anchor.minX,
                                      float(t[1]) * anchor.height() + anchor.minY,
                                      math.exp(float(t[2])) * anchor.width(),
                                      math.exp(float(t[3])) * anchor.height())


Anchors.new = Anchors
