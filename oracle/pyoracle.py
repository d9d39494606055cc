"""ctypes binding of the CPU oracle (oracle/libfrcnn_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (faster-rcnn.torch_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfrcnn_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orc_geometry.c", "orc_layers.c", "orc_model.c", "frcnn_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfrcnn_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Model(C.Structure):
    _fields_ = [
        ("nblocks", C.c_int),
        ("filters", C.c_int * 8), ("ksize", C.c_int * 8), ("pad", C.c_int * 8), ("conv_steps", C.c_int * 8),
        ("dropout", C.c_double * 8),
        ("nheads", C.c_int),
        ("head_k", C.c_int * 8), ("head_n", C.c_int * 8), ("head_input", C.c_int * 8),
        ("ncls", C.c_int),
        ("cls_n", C.c_int * 8), ("cls_bn", C.c_int * 8),
        ("cls_dropout", C.c_double * 8),
        ("class_count", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int),
        ("scales", C.c_double * 4),
    ]


class DetectCounts(C.Structure):
    _fields_ = [("nmatch", C.c_int), ("ncand", C.c_int), ("nwin", C.c_int)]


def make_model(layers, anchor_nets, class_layers, cfg):
    """layers/anchor_nets/class_layers/cfg: the python dicts of vgg_small.py / config."""
    m = Model()
    m.nblocks = len(layers)
    for i, l in enumerate(layers):
        m.filters[i] = l["filters"]; m.ksize[i] = l["kW"]; m.pad[i] = l["padW"]
        m.conv_steps[i] = l["conv_steps"]; m.dropout[i] = l.get("dropout", 0.0) or 0.0
    m.nheads = len(anchor_nets)
    for i, a in enumerate(anchor_nets):
        m.head_k[i] = a["kW"]; m.head_n[i] = a["n"]; m.head_input[i] = a["input"]
    m.ncls = len(class_layers)
    for i, l in enumerate(class_layers):
        m.cls_n[i] = l["n"]; m.cls_bn[i] = 1 if l.get("batch_norm") else 0
        m.cls_dropout[i] = l.get("dropout", 0.0) or 0.0
    m.class_count = cfg["class_count"]
    m.kh = cfg["roi_pooling"]["kh"]; m.kw = cfg["roi_pooling"]["kw"]
    for i, s in enumerate(cfg["scales"]):
        m.scales[i] = s
    return m


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp = C.POINTER(C.c_double); fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
    L.orc_rect_iou.restype = C.c_double
    L.orc_rect_iou.argtypes = [dp, dp]
    L.orc_mt_new.restype = C.c_void_p
    L.orc_mt_new.argtypes = [C.c_uint32]
    L.orc_mt_free.argtypes = [C.c_void_p]
    L.orc_mt_random.restype = C.c_uint32
    L.orc_mt_random.argtypes = [C.c_void_p]
    L.orc_anchors_new.restype = C.c_void_p
    L.orc_anchors_new.argtypes = [ip, ip, dp, C.c_int]
    L.orc_anchors_free.argtypes = [C.c_void_p]
    L.orc_anchors_w.restype = fp
    L.orc_anchors_w.argtypes = [C.c_void_p]
    L.orc_anchors_h.restype = fp
    L.orc_anchors_h.argtypes = [C.c_void_p]
    L.orc_anchors_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    L.orc_anchors_find_ranges_xy.argtypes = [C.c_void_p, dp, dp, ip]
    L.orc_anchors_find_positive.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_double, C.c_double, C.c_int, ip, dp, C.c_int]
    L.orc_anchors_sample_negative.argtypes = [C.c_void_p, dp, dp, C.c_int, C.c_double, C.c_int, C.c_void_p, ip, dp, C.c_int]
    L.orc_anchors_find_nearby.argtypes = [C.c_void_p, C.c_double, C.c_double, ip, dp, C.c_int]
    L.orc_loc_feature_to_input.argtypes = [ip, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp]
    L.orc_model_param_count.restype = C.c_long
    L.orc_model_param_count.argtypes = [C.POINTER(Model), C.POINTER(C.c_long)]
    L.orc_pnet_state_new.restype = C.c_void_p
    L.orc_pnet_state_free.argtypes = [C.c_void_p]
    L.orc_pnet_forward.argtypes = [C.POINTER(Model), fp, fp, C.c_int, C.c_int, C.c_int, C.POINTER(fp), C.c_void_p]
    L.orc_pnet_output.restype = fp
    L.orc_pnet_output.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
    L.orc_pnet_backward.argtypes = [C.POINTER(Model), fp, C.c_void_p, C.POINTER(fp), fp]
    L.orc_cnet_state_new.restype = C.c_void_p
    L.orc_cnet_state_free.argtypes = [C.c_void_p]
    L.orc_cnet_forward.argtypes = [C.POINTER(Model), fp, fp, C.c_int, C.c_int, C.POINTER(fp), fp, C.c_void_p, fp, fp]
    L.orc_cnet_backward.argtypes = [C.POINTER(Model), fp, C.c_void_p, fp, fp, fp, fp]
    L.orc_prelu_bwd.restype = C.c_double
    L.orc_rmsprop.argtypes = [fp, fp, fp, C.c_long, C.c_float, C.c_float, C.c_float]
    L.orc_set_decisions.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_set_decisions.restype = None
    L.orc_prelu_fwd.argtypes = [fp, C.c_long, C.c_float, fp]
    L.orc_prelu_bwd.argtypes = [fp, fp, C.c_long, C.c_float, fp]
    _lib = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------- geometry
def rect_iou(a, b):
    return lib().orc_rect_iou(_d(f64(a)), _d(f64(b)))


def loc_input_to_feature(layers, rect, layer_index=0):
    layers = i32(layers); out = np.zeros(4)
    lib().orc_loc_input_to_feature(_i(layers), len(layers), layer_index, _d(f64(rect)), _d(out))
    return out


def loc_feature_to_input(layers, minX, minY, maxX, maxY, layer_index=0):
    layers = i32(layers); out = np.zeros(4)
    lib().orc_loc_feature_to_input(_i(layers), len(layers), layer_index, minX, minY, maxX, maxY, _d(out))
    return out


def model_localizer_layers(m, output_index):
    buf = np.zeros((64, 6), dtype=np.int32)
    n = lib().orc_model_localizer_layers(C.byref(m), output_index, _i(buf))
    return buf[:n].copy()


class MT:
    def __init__(self, seed):
        self.h = lib().orc_mt_new(seed)

    def random(self):
        return lib().orc_mt_random(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mt_free(self.h); self.h = None


class Anchors:
    def __init__(self, m):
        lay = [model_localizer_layers(m, i + 1) for i in range(4)]
        cat = i32(np.concatenate(lay, axis=0)); nl = i32([len(x) for x in lay])
        sc = f64(list(m.scales))
        self.h = lib().orc_anchors_new(_i(cat), _i(nl), _d(sc), 4)
        self.w_table = np.ctypeslib.as_array(lib().orc_anchors_w(self.h), shape=(4, 3, 200, 2)).copy()
        self.h_table = np.ctypeslib.as_array(lib().orc_anchors_h(self.h), shape=(4, 3, 200, 2)).copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_anchors_free(self.h); self.h = None

    def get(self, layer, aspect, y, x):
        r = np.zeros(4); lib().orc_anchors_get(self.h, layer, aspect, y, x, _d(r)); return r

    def find_ranges_xy(self, rect, clip=None):
        out = np.zeros((12, 6), dtype=np.int32)
        n = lib().orc_anchors_find_ranges_xy(self.h, _d(f64(rect)), _d(f64(clip)) if clip is not None else None, _i(out))
        return out[:n].copy()

    def find_positive(self, rois, clip, pos_thr, neg_thr, include_best, cap=65536):
        rois = f64(rois).reshape(-1, 4)
        idx = np.zeros((cap, 5), dtype=np.int32); rc = np.zeros((cap, 4))
        n = lib().orc_anchors_find_positive(self.h, _d(rois), len(rois), _d(f64(clip)), pos_thr, neg_thr,
                                            1 if include_best else 0, _i(idx), _d(rc), cap)
        assert n <= cap
        return idx[:n].copy(), rc[:n].copy()

    def sample_negative(self, image_rect, rois, neg_thr, count, rng, cap=4096):
        rois = f64(rois).reshape(-1, 4)
        idx = np.zeros((cap, 4), dtype=np.int32); rc = np.zeros((cap, 4))
        n = lib().orc_anchors_sample_negative(self.h, _d(f64(image_rect)), _d(rois), len(rois), neg_thr, count,
                                              rng.h, _i(idx), _d(rc), cap)
        return idx[:n].copy(), rc[:n].copy()

    def find_nearby(self, cx, cy, cap=4096):
        idx = np.zeros((cap, 4), dtype=np.int32); rc = np.zeros((cap, 4))
        n = lib().orc_anchors_find_nearby(self.h, cx, cy, _i(idx), _d(rc), cap)
        assert n <= cap
        return idx[:n].copy(), rc[:n].copy()


def input_to_anchor(anchor, rect):
    t = np.zeros(4, dtype=np.float32)
    lib().orc_input_to_anchor(_d(f64(anchor)), _d(f64(rect)), _f(t)); return t


def anchor_to_input(anchor, t):
    r = np.zeros(4)
    lib().orc_anchor_to_input(_d(f64(anchor)), _f(f32(t)), _d(r)); return r


def nms(boxes, overlap, key_mode=0, key_col=0):
    boxes = f32(boxes)
    n = boxes.shape[0] if boxes.ndim == 2 else 0
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    pick = np.zeros(n, dtype=np.int64)
    k = lib().orc_nms(_f(boxes), n, boxes.shape[1], C.c_float(overlap), key_mode, key_col,
                      pick.ctypes.data_as(C.POINTER(C.c_int64)))
    return pick[:k].copy()


def extract_roi_window(layers, rect, fmH, fmW):
    layers = i32(layers); win = np.zeros(4, dtype=np.int32)
    lib().orc_extract_roi_window(_i(layers), len(layers), _d(f64(rect)), fmH, fmW, _i(win)); return win


def adaptive_max_pool_fwd(fmap, win, kh, kw):
    fmap = f32(fmap); Cn, H, W = fmap.shape
    out = np.zeros((Cn, kh, kw), dtype=np.float32); idx = np.zeros((Cn, kh, kw), dtype=np.int32)
    lib().orc_adaptive_max_pool_fwd(_f(fmap), Cn, H, W, _i(i32(win)), kh, kw, _f(out), _i(idx))
    return out, idx


def adaptive_max_pool_bwd(gmap, gout, idx):
    Cn, H, W = gmap.shape; kh, kw = gout.shape[1:]
    lib().orc_adaptive_max_pool_bwd(_f(gmap), Cn, H, W, kh, kw, _f(f32(gout)), _i(i32(idx)))


# ---------------------------------------------------------------- layers
def conv2d_fwd(x, w, b, pad):
    x = f32(x); w = f32(w); Cn, H, W = x.shape; O, _, kh, kw = w.shape
    out = np.zeros((O, H + 2 * pad - kh + 1, W + 2 * pad - kw + 1), dtype=np.float32)
    lib().orc_conv2d_fwd(_f(x), Cn, H, W, _f(w), _f(f32(b)) if b is not None else None, O, kh, kw, pad, _f(out))
    return out


def conv2d_bwd_input(gout, w, pad, H, W):
    gout = f32(gout); w = f32(w); O, Ho, Wo = gout.shape; _, Cn, kh, kw = w.shape
    gin = np.zeros((Cn, H, W), dtype=np.float32)
    lib().orc_conv2d_bwd_input(_f(gout), O, Ho, Wo, _f(w), Cn, kh, kw, pad, H, W, _f(gin)); return gin


def conv2d_bwd_weight(x, gout, kh, kw, pad):
    x = f32(x); gout = f32(gout); Cn, H, W = x.shape; O = gout.shape[0]
    gw = np.zeros((O, Cn, kh, kw), dtype=np.float32); gb = np.zeros(O, dtype=np.float32)
    lib().orc_conv2d_bwd_weight(_f(x), Cn, H, W, _f(gout), O, kh, kw, pad, _f(gw), _f(gb)); return gw, gb


def maxpool_fwd(x):
    x = f32(x); Cn, H, W = x.shape
    Ho = int(np.ceil((H - 2) / 2.0)) + 1; Wo = int(np.ceil((W - 2) / 2.0)) + 1
    out = np.zeros((Cn, Ho, Wo), dtype=np.float32); idx = np.zeros((Cn, Ho, Wo), dtype=np.int32)
    lib().orc_maxpool2x2_ceil_fwd(_f(x), Cn, H, W, _f(out), _i(idx)); return out, idx


def maxpool_bwd(gout, idx, H, W):
    gout = f32(gout); Cn = gout.shape[0]
    gin = np.zeros((Cn, H, W), dtype=np.float32)
    lib().orc_maxpool2x2_ceil_bwd(_f(gout), _i(i32(idx)), Cn, H, W, _f(gin)); return gin


def linear_fwd(x, w, b):
    x = f32(x); w = f32(w); R, I = x.shape; O = w.shape[0]
    y = np.zeros((R, O), dtype=np.float32)
    lib().orc_linear_fwd(_f(x), R, I, _f(w), _f(f32(b)) if b is not None else None, O, _f(y)); return y


def rmsprop(x, g, m, lr, alpha, eps=1e-8):
    lib().orc_rmsprop(_f(x), _f(g), _f(m), x.size, lr, alpha, eps)


# ---------------------------------------------------------------- model level
def param_count(m):
    pn = C.c_long(0)
    n = lib().orc_model_param_count(C.byref(m), C.byref(pn))
    return n, pn.value


def _ptr_array(arrs, n):
    PA = C.POINTER(C.c_float) * n
    pa = PA()
    for i in range(n):
        a = arrs[i] if arrs is not None and i < len(arrs) else None
        pa[i] = _f(a) if a is not None else None
    return pa


class PnetState:
    def __init__(self):
        self.h = lib().orc_pnet_state_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pnet_state_free(self.h); self.h = None


def pnet_forward(m, weights, img, training, drop_masks=None, state=None):
    st = state or PnetState()
    img = f32(img); _, H, W = img.shape
    masks = [f32(x) if x is not None else None for x in drop_masks] if drop_masks else None
    pa = _ptr_array(masks, 8)
    lib().orc_pnet_forward(C.byref(m), _f(weights), _f(img), H, W, 1 if training else 0,
                           pa if masks is not None else None, st.h)
    st._keep = (masks, weights)
    outs = []
    for i in range(1, m.nheads + 2):
        c = C.c_int(); h = C.c_int(); w = C.c_int()
        p = lib().orc_pnet_output(st.h, i, C.byref(c), C.byref(h), C.byref(w))
        outs.append(np.ctypeslib.as_array(p, shape=(c.value, h.value, w.value)).copy())
    return outs, st


def pnet_backward(m, weights, st, deltas, grad):
    deltas = [f32(d) for d in deltas]
    pa = _ptr_array(deltas, len(deltas))
    lib().orc_pnet_backward(C.byref(m), _f(weights), st.h, pa, _f(grad))


class CnetState:
    def __init__(self):
        self.h = lib().orc_cnet_state_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cnet_state_free(self.h); self.h = None


def cnet_forward(m, weights, x, training, drop_masks=None, bn_running=None, state=None):
    st = state or CnetState()
    x = f32(x); R = x.shape[0]
    masks = [f32(k) if k is not None else None for k in drop_masks] if drop_masks else None
    pa = _ptr_array(masks, 8)
    bbox = np.zeros((R, 4), dtype=np.float32); cls = np.zeros((R, m.class_count + 1), dtype=np.float32)
    lib().orc_cnet_forward(C.byref(m), _f(weights), _f(x), R, 1 if training else 0,
                           pa if masks is not None else None,
                           _f(bn_running) if bn_running is not None else None, st.h, _f(bbox), _f(cls))
    st._keep = (masks, x)
    return bbox, cls, st


def cnet_backward(m, weights, st, g_bbox, g_cls, grad, D):
    g_bbox = f32(g_bbox); g_cls = f32(g_cls); R = g_bbox.shape[0]
    gx = np.zeros((R, D), dtype=np.float32)
    lib().orc_cnet_backward(C.byref(m), _f(weights), st.h, _f(g_bbox), _f(g_cls), _f(gx), _f(grad))
    return gx


def train_image(m, weights, grad, img, pos_idx, pos_rect, rois, roi_class, neg_idx, neg_rect,
                pnet_masks=None, cnet_masks=None, bn_running=None, acc=None):
    img = f32(img); _, H, W = img.shape
    pos_idx = i32(pos_idx).reshape(-1, 5); pos_rect = f64(pos_rect).reshape(-1, 4)
    neg_idx = i32(neg_idx).reshape(-1, 4); neg_rect = f64(neg_rect).reshape(-1, 4)
    rois = f64(rois).reshape(-1, 4); roi_class = i32(roi_class)
    if acc is None:
        acc = np.zeros(8)
    pm = [f32(x) if x is not None else None for x in pnet_masks] if pnet_masks else None
    cm = [f32(x) if x is not None else None for x in cnet_masks] if cnet_masks else None
    lib().orc_train_image(C.byref(m), _f(weights), _f(grad), _f(img), H, W, _i(pos_idx), _d(pos_rect),
                          len(pos_idx), _d(rois), _i(roi_class), len(rois), _i(neg_idx), _d(neg_rect),
                          len(neg_idx), _ptr_array(pm, 8) if pm is not None else None,
                          _ptr_array(cm, 8) if cm is not None else None,
                          _f(bn_running) if bn_running is not None else None, _d(acc))
    return acc


class _Decisions(C.Structure):
    _fields_ = [("pool_idx", C.c_void_p * 8), ("conv_pos", C.c_void_p * 32), ("head_pos", C.c_void_p * 8),
                ("cnet_pos", C.c_void_p * 8), ("roi_idx", C.c_void_p), ("slope_abs", C.c_void_p)]


def _decisions_struct(d):
    """dict(pool_idx=[int32 arrays], conv_pos=[uint8], head_pos=[uint8], cnet_pos=[uint8], roi_idx=int32) -> C struct + keep-alive."""
    st = _Decisions()
    keep = []
    for name, dt in (("pool_idx", np.int32), ("conv_pos", np.uint8), ("head_pos", np.uint8), ("cnet_pos", np.uint8)):
        for i, a in enumerate(d.get(name) or []):
            if a is None:
                continue
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"], (name, i, a.dtype)
            keep.append(a)
            getattr(st, name)[i] = a.ctypes.data
    a = d.get("roi_idx")
    if a is not None:
        assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
        keep.append(a)
        st.roi_idx = a.ctypes.data
    a = d.get("slope_abs")
    if a is not None:
        assert a.dtype == np.float64 and a.size == 48 and a.flags["C_CONTIGUOUS"]
        keep.append(a)
        st.slope_abs = a.ctypes.data
    return st, keep


class decisions(object):
    """with O.decisions(inject=..., record=...): the oracle calls inside take the discrete choices of `inject` as given
    and / or write their own into the (pre-allocated) arrays of `record` (frcnn_oracle.h orc_set_decisions)."""

    def __init__(self, inject=None, record=None):
        self.inject, self.record = inject, record

    def __enter__(self):
        self._i = _decisions_struct(self.inject) if self.inject is not None else None
        self._r = _decisions_struct(self.record) if self.record is not None else None
        lib().orc_set_decisions(C.byref(self._i[0]) if self._i else None, C.byref(self._r[0]) if self._r else None)
        return self

    def __exit__(self, *a):
        lib().orc_set_decisions(None, None)
        return False


def detect(m, weights, bn_running, img, cap=65536):
    img = f32(img); _, H, W = img.shape
    nc = m.class_count + 1
    match_p = np.zeros(cap, dtype=np.float32); match_idx = np.zeros((cap, 4), dtype=np.int32)
    match_rect = np.zeros((cap, 4)); cand_ids = np.zeros(cap, dtype=np.int64)
    cand_bbox = np.zeros((cap, 4), dtype=np.float32); cand_cls = np.zeros((cap, nc), dtype=np.float32)
    win = np.zeros((cap, 7)); cnt = DetectCounts()
    lib().orc_detect(C.byref(m), _f(weights), _f(bn_running) if bn_running is not None else None, _f(img), H, W,
                     cap, _f(match_p), _i(match_idx), _d(match_rect),
                     cand_ids.ctypes.data_as(C.POINTER(C.c_int64)), _f(cand_bbox), _f(cand_cls), _d(win),
                     C.byref(cnt))
    nm, ncd, nw = min(cnt.nmatch, cap), cnt.ncand, min(cnt.nwin, cap)
    return dict(match_p=match_p[:nm], match_idx=match_idx[:nm], match_rect=match_rect[:nm],
                cand_ids=cand_ids[:ncd], cand_bbox=cand_bbox[:ncd], cand_cls=cand_cls[:ncd], winners=win[:nw])


def set_threads(n):
    lib().orc_set_threads(n)


def get_threads():
    return lib().orc_get_threads()
