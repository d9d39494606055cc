"""The pieces of Detector:detect that stay on the device between its big steps (csrc/detect.hip, frcnn_nms_device_n)
against the host mirror / the oracle: NMS with the row count in device memory, ROI windows of a batch of rects
(objective.lua:5-13 via Localizer.lua:41-67), the class test + rect decode + ordered compaction (Detector.lua:106-122)
and the winner records."""
import ctypes as C
import math

import numpy as np
import pytest

from util import random_boxes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,cap", [(1, 64), (63, 64), (700, 2000), (2000, 2000), (0, 100)])
def test_nms_with_device_side_count_equals_nms(F, O, n, cap):
    rng = np.random.RandomState(n + cap)
    b = random_boxes(rng, cap)
    cls = rng.randint(1, 5, cap).astype(np.int32)
    db, dc = F.DeviceTensor.from_numpy(b), F.DeviceTensor.from_numpy(cls)
    ndev = F.DeviceTensor.from_numpy(np.array([n], np.int32))
    wsb = F._lib.load().frcnn_nms_workspace_bytes(cap)
    ws = F.DeviceTensor.empty((wsb,), np.uint8)
    pick = F.DeviceTensor.empty((cap,), np.int64); cnt = F.DeviceTensor.zeros((1,), np.int32)
    for with_cls in (False, True):
        F._lib.call("frcnn_nms_device_n", F.ptr(db), cap, F.ptr(ndev), 4, C.c_float(0.25), 0, 0, F.ptr(dc) if with_cls else None,
                    F.ptr(pick), F.ptr(cnt), F.ptr(ws), wsb, F.stream_ptr())
        k = int(cnt.numpy()[0])
        got = pick.numpy()[:k].tolist()
        if not with_cls:
            want = O.nms(b[:n], 0.25).tolist() if n else []
        else:   # one nms per class, merged back into global pick order (descending key = max-y, ties by row)
            want = []
            for c in np.unique(cls[:n]):
                rows = np.nonzero(cls[:n] == c)[0]
                want += [int(rows[i - 1]) + 1 for i in O.nms(b[:n][rows], 0.25).tolist()]
            assert sorted(got) == sorted(want)
            by_class = lambda ids: {c: [i for i in ids if cls[i - 1] == c] for c in np.unique(cls[:n])}
            assert by_class(got) == by_class(want)
            continue
        assert got == want


def test_roi_windows_equal_the_host_mirror(F, small_cfg):
    model = F.vgg_small(dict(small_cfg))
    loc = F.Localizer(model["pnet"].outnode.children[-1])
    rng = np.random.RandomState(3)
    n = 3000
    x0 = rng.uniform(-40, 780, n); y0 = rng.uniform(-40, 430, n)
    rect = np.stack([x0, y0, x0 + rng.uniform(0.5, 300, n), y0 + rng.uniform(0.5, 300, n)], 1)
    rect[:50] = np.round(rect[:50])             # integer corners hit the exact branch of Localizer.lua:54-63
    rect[50:60, 2:] = rect[50:60, :2] + 1e-9    # degenerate
    fmH, fmW = 29, 50
    from frcnn_amd.objective import roi_windows
    want = roi_windows(rect, loc, fmH, fmW)
    layers = np.array([[l["kW"], l["kH"], l["dW"], l["dH"], l["padW"], l["padH"]] for l in loc.layers], dtype=np.int32)
    drect = F.DeviceTensor.from_numpy(rect)
    wins = F.DeviceTensor.empty((n, 4), np.int32)
    F._lib.call("frcnn_roi_windows", F.ptr(drect), None, n, layers.ctypes.data_as(C.c_void_p), len(layers), fmH, fmW, F.ptr(wins),
                F.stream_ptr())
    assert np.array_equal(wins.numpy(), want)
    # through a pick list (1-based rows, arbitrary order)
    pick = (rng.permutation(n)[:777] + 1).astype(np.int64)
    dpick = F.DeviceTensor.from_numpy(pick)
    w2 = F.DeviceTensor.empty((len(pick), 4), np.int32)
    F._lib.call("frcnn_roi_windows", F.ptr(drect), F.ptr(dpick), len(pick), layers.ctypes.data_as(C.c_void_p), len(layers), fmH, fmW,
                F.ptr(w2), F.stream_ptr())
    assert np.array_equal(w2.numpy(), want[pick - 1])


def test_detect_post_and_gather(F):
    rng = np.random.RandomState(5)
    nm, R, ncls_bg = 5000, 2300, 17
    rect = np.cumsum(rng.uniform(1, 9, (nm, 4)), 1)
    pick = (rng.permutation(nm)[:R] + 1).astype(np.int64)
    cls = rng.randint(1, ncls_bg + 1, R).astype(np.int32)
    conf = np.log(rng.uniform(0.05, 1.0, R)).astype(np.float32)
    bbox = (rng.randn(R, 4) * 0.3).astype(np.float32)
    mp = np.log(rng.uniform(0.95, 1.0, nm)).astype(np.float32)
    midx = rng.randint(1, 50, (nm, 4)).astype(np.int32)
    d = {k: F.DeviceTensor.from_numpy(v) for k, v in dict(rect=rect, pick=pick, cls=cls, conf=conf, bbox=bbox, mp=mp, midx=midx).items()}
    bb = F.DeviceTensor.empty((R, 5)); kc = F.DeviceTensor.empty((R,), np.int32); kr = F.DeviceTensor.empty((R,), np.int32)
    r2 = F.DeviceTensor.empty((R, 4), np.float64); K = F.DeviceTensor.zeros((1,), np.int32)
    F._lib.call("frcnn_detect_post", F.ptr(d["cls"]), F.ptr(d["conf"]), F.ptr(d["bbox"]), F.ptr(d["rect"]), F.ptr(d["pick"]), R, ncls_bg,
                0.2, F.ptr(bb), F.ptr(kc), F.ptr(kr), F.ptr(r2), F.ptr(K), F.stream_ptr())
    # the host mirror of Detector.lua:106-122 (what Detector.py computed in numpy before)
    keep = np.nonzero((cls != ncls_bg) & (np.exp(conf.astype(np.float64)) > 0.2))[0]
    k = int(K.numpy()[0])
    assert k == len(keep) and np.array_equal(kr.numpy()[:k], keep) and np.array_equal(kc.numpy()[:k], cls[keep])
    ra = rect[pick[keep] - 1]
    aw, ah = ra[:, 2] - ra[:, 0], ra[:, 3] - ra[:, 1]
    t = bbox[keep].astype(np.float64)
    x0 = t[:, 0] * aw + ra[:, 0]; y0 = t[:, 1] * ah + ra[:, 1]
    ew = np.array([math.exp(v) for v in t[:, 2].tolist()]) * aw; eh = np.array([math.exp(v) for v in t[:, 3].tolist()]) * ah
    want_r2 = np.stack([x0, y0, x0 + ew, y0 + eh], 1)
    got_r2 = r2.numpy()[:k]
    assert np.allclose(got_r2, want_r2, rtol=2e-15, atol=0)            # (device exp vs libm: last bits at most)
    assert np.array_equal(got_r2[:, :2], want_r2[:, :2])               # no exp involved: bit for bit
    gb = bb.numpy()[:k]
    assert np.allclose(gb[:, :4], want_r2.astype(np.float32), rtol=2e-7, atol=0) and np.array_equal(gb[:, 4], conf[keep])
    # winner records for an arbitrary pick list over the survivors
    wp = (rng.permutation(k)[:min(k, 300)] + 1).astype(np.int64)
    dwp = F.DeviceTensor.from_numpy(wp); nw = F.DeviceTensor.from_numpy(np.array([len(wp)], np.int32))
    rec = F.DeviceTensor.empty((R, 16), np.float64)
    F._lib.call("frcnn_detect_gather", F.ptr(dwp), F.ptr(nw), R, F.ptr(kr), F.ptr(kc), F.ptr(bb), F.ptr(r2), F.ptr(d["pick"]),
                F.ptr(d["mp"]), F.ptr(d["rect"]), F.ptr(d["midx"]), F.ptr(rec), F.stream_ptr())
    g = rec.numpy()[:len(wp)]
    j = wp - 1
    i = pick[keep[j]] - 1
    assert np.array_equal(g[:, 0], cls[keep[j]]) and np.array_equal(g[:, 1], keep[j] + 1)
    assert np.array_equal(g[:, 2], conf[keep[j]].astype(np.float64)) and np.array_equal(g[:, 3], mp[i].astype(np.float64))
    assert np.array_equal(g[:, 4:8], rect[i]) and np.array_equal(g[:, 8:12], got_r2[j]) and np.array_equal(g[:, 12:16], midx[i])
