"""Detector -- host-side mirror of Detector.lua.  detect(input) keeps the reference's pipeline and
thresholds (p > 0.95, NMS 0.25, class != background and p > 0.2, per-class NMS 0.1) but the 26 544
iteration Lua loop of Detector.lua:39-66 is one scan+compaction kernel (frcnn_rpn_scan), the per-ROI
pooling loop (:94-98) one batched kernel, and both NMS passes run on the device.  Note that both NMS
calls of the reference pass a tensor as `scores`, which nms.lua:37-43 ignores: boxes are processed
by descending max-y.  That behaviour is reproduced."""
import ctypes as C
import math

import numpy as np

from . import _lib
from .Anchors import Anchors
from .Localizer import Localizer
from .Rect import Rect
from .nms import nms
from .objective import roi_window, roi_windows
from .tensor import DeviceTensor, ptr, stream_ptr, to_device

ASPECTS = 3   # anchors per map position (Anchors.lua:108-109)


class Detector(object):
    def __init__(self, model):  # Detector.lua:8-15
        self.model = model
        cfg = model["cfg"]
        self.anchors = Anchors(model["pnet"], cfg["scales"])
        self.localizer = Localizer(model["pnet"].outnode.children[-1])
        self._aw = DeviceTensor.from_numpy(self.anchors.w)
        self._ah = DeviceTensor.from_numpy(self.anchors.h)
        self._bufs = {}
        self.verbose = False
        self.keep_cnet_outputs = True   # last_cnet["cls"]: the R x (classes+1) log-probabilities, read back for inspection

    def _buf(self, name, shape, dtype=np.float32):
        need = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        b = self._bufs.get(name)
        if b is None or b.nbytes < need:
            b = DeviceTensor.empty((max(need, 256),), np.uint8)
            self._bufs[name] = b
        return DeviceTensor(b.ptr, shape, dtype, owner=b)

    def scan(self, outputs, img_w, img_h, threshold=0.95):
        """Detector.lua:39-66 on device -> dict(p, idx, rect, box(dev), n)."""
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
        cnt = self._buf("count", (1,), np.int32)
        _lib.call("frcnn_rpn_scan", maps, Hs, Ws, ptr(self._aw), ptr(self._ah), float(img_w), float(img_h),
                  float(threshold), cap, ptr(mp), ptr(mi), ptr(mr), ptr(mb), ptr(cnt), ptr(ws), wsb, stream_ptr())
        n = int(cnt.numpy()[0])
        if n > cap:
            raise _lib.FrcnnError("Detector: %d anchors pass p > %g, more than the %d the maps hold" % (n, threshold, cap))
        return dict(n=n, p=DeviceTensor(mp.ptr, (n,), np.float32, owner=mp),
                    idx=DeviceTensor(mi.ptr, (n, 4), np.int32, owner=mi),
                    rect=DeviceTensor(mr.ptr, (n, 4), np.float64, owner=mr),
                    box=DeviceTensor(mb.ptr, (n, 4), np.float32, owner=mb))

    def _nms_device(self, boxes, n, ncols, overlap, cls=None):
        """nms(bb, overlap, scores) with the reference's key (max-y) on boxes resident in HBM, workspace and result buffers
        owned by the detector (no allocation per frame).  cls: optional device int32[n] -- rows only suppress rows of the
        same class (frcnn_nms_device_classes).  Returns the 1-based row ids in pick order (one read-back)."""
        wsb = _lib.load().frcnn_nms_workspace_bytes(n)
        ws = self._buf("nms_ws", (wsb,), np.uint8)
        pick = self._buf("nms_pick", (n,), np.int64)
        cnt = self._buf("nms_count", (1,), np.int32)
        if cls is None:
            _lib.call("frcnn_nms_device", ptr(boxes), n, ncols, C.c_float(overlap), 0, 0, ptr(pick), ptr(cnt), ptr(ws), wsb,
                      stream_ptr())
        else:
            _lib.call("frcnn_nms_device_classes", ptr(boxes), n, ncols, C.c_float(overlap), 0, 0, ptr(cls), ptr(pick),
                      ptr(cnt), ptr(ws), wsb, stream_ptr())
        k = int(cnt.numpy()[0])
        return pick.numpy()[:k].copy()

    def detect(self, input):  # Detector.lua:17-141
        model = self.model
        cfg = model["cfg"]
        pnet, cnet = model["pnet"], model["cnet"]
        kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
        bgclass = cfg["class_count"] + 1
        ncls = cfg["class_count"] + 1
        planes = model["layers"][-1]["filters"]
        s = stream_ptr()

        inp = to_device(input)
        _, H, W = inp.shape
        pnet.evaluate()  # :31
        outputs = pnet.forward(inp)  # :33
        m = self.scan(outputs, W, H)  # :39-66
        self.last_scan = m
        winners = []
        if m["n"] == 0:  # :71
            return winners
        # NON-MAXIMUM SUPPRESSION (:74-85) on the device; the score tensor is ignored by nms.lua -> key = max-y
        pick = self._nms_device(m["box"], m["n"], 4, 0.25)
        rect_all = m["rect"].numpy(); p_all = m["p"].numpy(); idx_all = m["idx"].numpy()
        cand = pick - 1
        self.last_pick = pick
        if self.verbose:
            print("candidates: %d" % len(cand))
        # REGION CLASSIFICATION (:90-101)
        cnet.evaluate()
        fm = outputs[-1]
        fmC, fmH, fmW = fm.shape
        R = len(cand)
        wins = roi_windows(rect_all[cand], self.localizer, fmH, fmW)   # all candidates at once (objective.lua:5-13)
        dwins = self._buf("wins", wins.shape, np.int32)
        dwins.copy_from_numpy(wins)
        cinput = self._buf("cinput", (R, kh * kw * planes))
        pidx = self._buf("pidx", (R, kh * kw * planes), np.int32)
        _lib.call("frcnn_roi_pool_forward", ptr(fm), fmC, fmH, fmW, ptr(dwins), R, kh, kw, ptr(cinput), ptr(pidx), s)
        bbox_out, cls_out = cnet.forward(cinput)  # :101
        dcls = self._buf("cls", (R,), np.int32); dconf = self._buf("conf", (R,))
        _lib.call("frcnn_cnet_decode", ptr(cls_out), R, ncls, ptr(dcls), ptr(dconf), s)  # :110-113
        bbox_h = bbox_out.numpy(); cls_h = dcls.numpy(); conf_h = dconf.numpy()
        self.last_cnet = dict(bbox=bbox_h, cls=cls_out.numpy() if self.keep_cnet_outputs else None)
        # the class test of :115 first (vectorised); the per-candidate tables are only built for survivors
        keep = np.nonzero((cls_h != bgclass) & (np.exp(conf_h.astype(np.float64)) > 0.2))[0]
        if len(keep) == 0:
            return winners
        # :106-122 for every surviving candidate at once: r2 = Anchors.anchorToInput(r, bbox_out[i]) in double arithmetic
        # (the products and sums as separately rounded operations, exp through libm like the Lua number path); the per-
        # detection tables {p, a, r, l, r2, class, confidence} are only built for the winners of the per-class NMS
        ci = cand[keep]
        ra = rect_all[ci]
        aw, ah = ra[:, 2] - ra[:, 0], ra[:, 3] - ra[:, 1]
        t = bbox_h[keep].astype(np.float64)
        x0 = t[:, 0] * aw + ra[:, 0]; y0 = t[:, 1] * ah + ra[:, 1]
        ew = np.array([math.exp(v) for v in t[:, 2].tolist()], dtype=np.float64) * aw
        eh = np.array([math.exp(v) for v in t[:, 3].tolist()], dtype=np.float64) * ah
        r2 = np.stack([x0, y0, x0 + ew, y0 + eh], 1)      # Rect.fromXYWidthHeight
        # Per-class NMS (:125-136), all classes in ONE device pass: rows only suppress rows of their own class, and a stable
        # partition of the picks by class is, per class, exactly nms(bb_class, 0.1, scores) -- the score tensor is ignored by
        # nms.lua:42, the key is max-y.  With class_count = 200 (config/imagenet.lua) that is one launch sequence and one
        # read-back instead of up to 200.
        K = len(keep)
        bb = np.empty((K, 5), dtype=np.float32)
        bb[:, 0:4] = r2          # r.r2:totensor() (FloatTensor)
        bb[:, 4] = conf_h[keep]
        kc = cls_h[keep].astype(np.int32)
        blob = np.concatenate([bb.view(np.uint8).ravel(), kc.view(np.uint8).ravel()])
        dblob = self._buf("bbblob", (blob.size,), np.uint8)
        dblob.copy_from_numpy(blob)
        dbb = DeviceTensor(dblob.ptr, (K, 5), np.float32, owner=dblob)
        dkc = DeviceTensor(dblob.ptr + bb.nbytes, (K,), np.int32, owner=dblob)
        pk = self._nms_device(dbb, K, 5, 0.1, cls=dkc)
        # classes in ascending order (pairs() order is unspecified in Lua), pick order within a class
        order = sorted(range(len(pk)), key=lambda q: kc[int(pk[q]) - 1])   # sorted() is stable
        for q in order:
            j = int(pk[q]) - 1
            k = int(keep[j]); i = int(ci[j])
            winners.append(dict(p=float(p_all[i]), r=Rect(*rect_all[i]), l=int(idx_all[i][0]),
                                a=self.anchors.get(*[int(v) for v in idx_all[i]]), r2=Rect(*r2[j]),
                                confidence=float(conf_h[k]), candidate=k + 1,   # (1-based row among the NMS candidates)
                                **{"class": int(cls_h[k])}))
        return winners
