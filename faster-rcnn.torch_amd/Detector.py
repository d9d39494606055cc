"""Detector -- host-side mirror of Detector.lua.  detect(input) keeps the reference's pipeline and
thresholds (p > 0.95, NMS 0.25, class != background and p > 0.2, per-class NMS 0.1) but the 26 544
iteration Lua loop of Detector.lua:39-66 is one scan+compaction kernel (frcnn_rpn_scan), the per-ROI
pooling loop (:94-98) one batched kernel, and both NMS passes run on the device.  Note that both NMS
calls of the reference pass a tensor as `scores`, which nms.lua:37-43 ignores: boxes are processed
by descending max-y.  That behaviour is reproduced.

The frame stays on the device between its big steps: scan -> NMS (the match count is read by the NMS kernels from
device memory), ONE read-back of two counts (the cnet's row count sizes its launches), then ROI windows -> ROI pooling
-> cnet -> class test + rect decode + ordered compaction -> per-class NMS -> one record per winner, and ONE read-back
of the winner table.  The list detect() returns builds its {p, a, r, l, r2, class, confidence} tables on access."""
import ctypes as C
import math

import os

import numpy as np

from . import _lib
from .Anchors import Anchors
from .Localizer import Localizer
from .Rect import Rect
from .nms import nms
from .objective import roi_window, roi_windows
from .tensor import DeviceTensor, ptr, stream_ptr, to_device

ASPECTS = 3   # anchors per map position (Anchors.lua:108-109)


class _Detections(object):
    """The list Detector:detect returns (Detector.lua:138-140): one table {p, a, r, l, r2, class, confidence} per winner,
    classes ascending (pairs() order is unspecified in Lua), pick order within a class.  Backed by the winner records the
    device wrote; a table is built when it is looked at."""

    def __init__(self, rec, anchors):
        self._rec, self._anchors = rec, anchors
        self._items = [None] * len(rec)

    def __len__(self):
        return len(self._items)

    def _make(self, q):
        x = self._items[q]
        if x is None:
            v = self._rec[q]
            idx = [int(t) for t in v[12:16]]
            x = dict(p=float(np.float32(v[3])), r=Rect(*v[4:8]), l=idx[0], a=self._anchors.get(*idx), r2=Rect(*v[8:12]),
                     confidence=float(np.float32(v[2])), candidate=int(v[1]), **{"class": int(v[0])})
            self._items[q] = x
        return x

    def __getitem__(self, q):
        if isinstance(q, slice):
            return [self._make(t) for t in range(*q.indices(len(self)))]
        return self._make(q if q >= 0 else q + len(self))

    def __iter__(self):
        return (self._make(q) for q in range(len(self)))

    def __bool__(self):
        return len(self) > 0


class Detector(object):
    def __init__(self, model, static_weights=False):  # Detector.lua:8-15
        """static_weights=True: the caller promises not to write the weight vector between detect() calls; the library then packs
        the convolution weights once instead of once per frame (option static_weights of the C ABI; a training-mode pass or
        another Detector(..., static_weights=...) drops the packs)."""
        self.model = model
        if static_weights or os.environ.get("FRCNN_STATIC_WEIGHTS"):
            _lib.call("frcnn_set_option", b"static_weights", 1)
        cfg = model["cfg"]
        self.anchors = Anchors(model["pnet"], cfg["scales"])
        self.localizer = Localizer(model["pnet"].outnode.children[-1])
        self._loc_layers = np.array([[l["kW"], l["kH"], l["dW"], l["dH"], l["padW"], l["padH"]] for l in self.localizer.layers],
                                    dtype=np.int32).reshape(-1, 6)
        self._aw = DeviceTensor.from_numpy(self.anchors.w)
        self._ah = DeviceTensor.from_numpy(self.anchors.h)
        self._bufs = {}
        self._host = None            # page-locked landing buffer of the two read-backs
        self._host_bytes = 0
        self.verbose = False
        self.last_scan = None
        self._last = {}

    def __del__(self):
        if getattr(self, "_host", None):
            try:
                _lib.load().frcnn_host_free(C.c_void_p(self._host))
            except Exception:
                pass

    def _buf(self, name, shape, dtype=np.float32):
        need = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        b = self._bufs.get(name)
        if b is None or b.nbytes < need:
            b = DeviceTensor.empty((max(need, 256),), np.uint8)
            self._bufs[name] = b
        return DeviceTensor(b.ptr, shape, dtype, owner=b)

    def _read(self, dev_ptr, nbytes, dtype):
        """One asynchronous copy into page-locked host memory + one wait: the frame's read-back."""
        if self._host_bytes < nbytes:
            if self._host:
                _lib.call("frcnn_host_free", C.c_void_p(self._host))
            p = C.c_void_p()
            self._host_bytes = max(int(nbytes), 1 << 16)
            _lib.call("frcnn_host_alloc", C.byref(p), self._host_bytes)
            self._host = p.value
        s = stream_ptr()
        _lib.call("frcnn_memcpy_d2h", C.c_void_p(self._host), C.c_void_p(dev_ptr), int(nbytes), s)
        _lib.call("frcnn_stream_sync", s)
        raw = (C.c_char * int(nbytes)).from_address(self._host)
        return np.frombuffer(raw, dtype=dtype).copy()

    def scan(self, outputs, img_w, img_h, threshold=0.95, counts=None):
        """Detector.lua:39-66 on device -> dict(cap, p, idx, rect, box (device arrays of `cap` rows), cnt (device count)).
        Nothing is read back here."""
        Hs = (C.c_int * 4)(*[outputs[i].shape[1] for i in range(4)])
        Ws = (C.c_int * 4)(*[outputs[i].shape[2] for i in range(4)])
        maps = (C.c_void_p * 4)(*[outputs[i].ptr for i in range(4)])
        wsb = _lib.load().frcnn_rpn_scan_workspace_bytes(Hs, Ws)
        ws = self._buf("scan_ws", (wsb,), np.uint8)
        # every anchor of the four maps may pass (vgg_large 1000x600 scans 45 015): the buffers hold them all, nothing is
        # ever truncated
        cap = ASPECTS * sum(outputs[i].shape[1] * outputs[i].shape[2] for i in range(4))
        mp = self._buf("match_p", (cap,)); mi = self._buf("match_idx", (cap, 4), np.int32)
        mr = self._buf("match_rect", (cap, 4), np.float64); mb = self._buf("match_box", (cap, 4))
        cnt = counts if counts is not None else self._buf("count", (4,), np.int32)
        _lib.call("frcnn_rpn_scan", maps, Hs, Ws, ptr(self._aw), ptr(self._ah), float(img_w), float(img_h),
                  float(threshold), cap, ptr(mp), ptr(mi), ptr(mr), ptr(mb), ptr(cnt), ptr(ws), wsb, stream_ptr())
        return dict(cap=cap, p=mp, idx=mi, rect=mr, box=mb, cnt=cnt, threshold=threshold)

    NMS_FIRST_CAP = 16384   # rows the first NMS launch is sized for (see detect)

    @property
    def last_pick(self):
        """1-based rows of the match arrays that survived the first NMS, in pick order (Detector.lua:82)."""
        L = self._last
        if "pick_host" not in L and "pick" in L:
            L["pick_host"] = L["pick"].numpy()[:L["R"]].copy()
        return L.get("pick_host")

    @property
    def last_cnet(self):
        """cnet outputs of the last frame's candidates: dict(bbox R x 4, cls R x (classes + 1) log-probabilities)."""
        L = self._last
        if "cnet_host" not in L and "bbox" in L:
            L["cnet_host"] = dict(bbox=L["bbox"].numpy(), cls=L["cls"].numpy())
        return L.get("cnet_host")

    def detect(self, input):  # Detector.lua:17-141
        model = self.model
        cfg = model["cfg"]
        pnet, cnet = model["pnet"], model["cnet"]
        kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
        bgclass = cfg["class_count"] + 1
        ncls = cfg["class_count"] + 1
        planes = model["layers"][-1]["filters"]
        s = stream_ptr()
        L = _lib.load()

        inp = to_device(input)
        _, H, W = inp.shape
        pnet.evaluate()  # :31
        outputs = pnet.forward(inp)  # :33
        # counts (device int32[4]): matches, NMS candidates, candidates that pass the class test, winners
        counts = self._buf("counts", (4,), np.int32)
        m = self.scan(outputs, W, H, counts=counts)  # :39-66
        cap = m["cap"]
        # NON-MAXIMUM SUPPRESSION (:74-85) on the device, the match count read from device memory; the score tensor is
        # ignored by nms.lua -> key = max-y
        # The launch and its workspace are sized for a BOUND on the matches, not for every anchor of the maps (vgg_large:
        # 45 015 anchors -> 253 MB of masks and a 704 x 704 tile grid per frame for a few hundred matches); a frame with more
        # matches than the bound repeats the pass sized by the count just read.
        ncap = min(cap, self.NMS_FIRST_CAP)
        wsb = L.frcnn_nms_workspace_bytes(ncap)
        ws = self._buf("nms_ws", (wsb,), np.uint8)
        pick = self._buf("nms_pick", (cap,), np.int64)
        _lib.call("frcnn_nms_device_n", ptr(m["box"]), ncap, ptr(counts), 4, C.c_float(0.25), 0, 0, None, ptr(pick),
                  C.c_void_p(counts.ptr + 4), ptr(ws), wsb, s)
        n, R = [int(v) for v in self._read(counts.ptr, 8, np.int32)]          # ---- read-back 1 of 2: two counts
        if n > cap:
            raise _lib.FrcnnError("Detector: %d anchors pass p > %g, more than the %d the maps hold" % (n, m["threshold"], cap))
        if n > ncap:
            wsb = L.frcnn_nms_workspace_bytes(n)
            ws = self._buf("nms_ws_full", (wsb,), np.uint8)
            _lib.call("frcnn_nms_device", ptr(m["box"]), n, 4, C.c_float(0.25), 0, 0, ptr(pick), C.c_void_p(counts.ptr + 4),
                      ptr(ws), wsb, s)
            R = int(self._read(counts.ptr + 4, 4, np.int32)[0])
        self.last_scan = dict(n=n, p=DeviceTensor(m["p"].ptr, (n,), np.float32, owner=m["p"]),
                              idx=DeviceTensor(m["idx"].ptr, (n, 4), np.int32, owner=m["idx"]),
                              rect=DeviceTensor(m["rect"].ptr, (n, 4), np.float64, owner=m["rect"]),
                              box=DeviceTensor(m["box"].ptr, (n, 4), np.float32, owner=m["box"]))
        self._last = dict(pick=pick, R=R)
        if n == 0:  # :71
            self._last["pick_host"] = np.zeros(0, np.int64)
            return []
        if self.verbose:
            print("candidates: %d" % R)
        # REGION CLASSIFICATION (:90-101)
        cnet.evaluate()
        fm = outputs[-1]
        fmC, fmH, fmW = fm.shape
        dwins = self._buf("wins", (R, 4), np.int32)
        _lib.call("frcnn_roi_windows", ptr(m["rect"]), ptr(pick), R, self._loc_layers.ctypes.data_as(C.c_void_p),
                  len(self._loc_layers), fmH, fmW, ptr(dwins), s)               # objective.lua:5-13 for every candidate
        cinput = self._buf("cinput", (R, kh * kw * planes))
        _lib.call("frcnn_roi_pool_forward", ptr(fm), fmC, fmH, fmW, ptr(dwins), R, kh, kw, ptr(cinput), None, s)
        bbox_out, cls_out = cnet.forward(cinput)  # :101
        self._last.update(bbox=bbox_out, cls=cls_out)
        dcls = self._buf("cls", (R,), np.int32); dconf = self._buf("conf", (R,))
        _lib.call("frcnn_cnet_decode", ptr(cls_out), R, ncls, ptr(dcls), ptr(dconf), s)  # :110-113
        # :106-122 on the device: class test, r2 = Anchors.anchorToInput(r, bbox) in double, survivors compacted in order
        bb = self._buf("bb", (R, 5)); kc = self._buf("kc", (R,), np.int32); keep_row = self._buf("keep_row", (R,), np.int32)
        r2 = self._buf("r2", (R, 4), np.float64)
        _lib.call("frcnn_detect_post", ptr(dcls), ptr(dconf), ptr(bbox_out), ptr(m["rect"]), ptr(pick), R, bgclass, 0.2,
                  ptr(bb), ptr(kc), ptr(keep_row), ptr(r2), C.c_void_p(counts.ptr + 8), s)
        # Per-class NMS (:125-136), all classes in ONE device pass (rows only suppress rows of their own class; a stable
        # partition of the picks by class is, per class, exactly nms(bb_class, 0.1, scores) -- key = max-y), the survivor
        # count read from device memory
        wsb2 = L.frcnn_nms_workspace_bytes(R)
        ws2 = self._buf("nms_ws2", (wsb2,), np.uint8)
        wpick = self._buf("wpick", (R,), np.int64)
        _lib.call("frcnn_nms_device_n", ptr(bb), R, C.c_void_p(counts.ptr + 8), 5, C.c_float(0.1), 0, 0, ptr(kc), ptr(wpick),
                  C.c_void_p(counts.ptr + 12), ptr(ws2), wsb2, s)
        # one record per winner, behind a 128-byte header that carries the four counts
        out = self._buf("winners", (R + 1, 16), np.float64)
        _lib.call("frcnn_memcpy_d2d", ptr(out), ptr(counts), 16, s)
        _lib.call("frcnn_detect_gather", ptr(wpick), C.c_void_p(counts.ptr + 12), R, ptr(keep_row), ptr(kc), ptr(bb), ptr(r2),
                  ptr(pick), ptr(m["p"]), ptr(m["rect"]), ptr(m["idx"]), C.c_void_p(out.ptr + 128), s)
        raw = self._read(out.ptr, (R + 1) * 128, np.float64).reshape(R + 1, 16)   # ---- read-back 2 of 2: the winner table
        hdr = raw[0].view(np.int32)
        nwin = int(hdr[3])
        self._last.update(kept=int(hdr[2]))
        rec = raw[1:1 + nwin]
        # classes in ascending order (pairs() order is unspecified in Lua), pick order within a class
        order = np.argsort(rec[:, 0], kind="stable")
        return _Detections(rec[order], self.anchors)
