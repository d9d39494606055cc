"""ROI pooling alone on the benchmarked shapes (384 x 29 x 50 map, 560 / 1398 windows, 6 x 6 cells): HIP-event time of the
forward and backward launches, effective bytes/s of the output they write.  python tools/bench_roi.py [R ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import frcnn_amd as F


def run(R, Cn=384, H=29, W=50, kh=6, kw=6, reps=10):
    rng = np.random.RandomState(R)
    fm = F.DeviceTensor.from_numpy(rng.randn(Cn, H, W).astype(np.float32))
    wins = np.zeros((R, 4), np.int32)
    for r in range(R):   # windows like the training examples' : 6..29 rows, 6..50 columns
        h, w = rng.randint(6, H + 1), rng.randint(6, W + 1)
        y0, x0 = rng.randint(0, H - h + 1), rng.randint(0, W - w + 1)
        wins[r] = (y0 + 1, y0 + h, x0 + 1, x0 + w)
    dw = F.DeviceTensor.from_numpy(wins)
    out = F.DeviceTensor.empty((R, Cn * kh * kw)); idx = F.DeviceTensor.empty((R, Cn * kh * kw), np.int32)
    g = F.DeviceTensor.from_numpy(rng.randn(R, Cn * kh * kw).astype(np.float32)); gm = F.DeviceTensor.zeros((Cn, H, W))
    s = F.stream_ptr()
    k = F._lib.KC_NAMES.index("roi")
    nk = len(F._lib.KC_NAMES)
    for name, fn in (("forward", lambda: F._lib.call("frcnn_roi_pool_forward", F.ptr(fm), Cn, H, W, F.ptr(dw), R, kh, kw, F.ptr(out), F.ptr(idx), s)),
                     ("forward (no indices)", lambda: F._lib.call("frcnn_roi_pool_forward", F.ptr(fm), Cn, H, W, F.ptr(dw), R, kh, kw, F.ptr(out), None, s)),
                     ("backward", lambda: F._lib.call("frcnn_roi_pool_backward", F.ptr(gm), Cn, H, W, F.ptr(g), F.ptr(idx), R, kh, kw, s))):
        fn()
        la = (C.c_longlong * nk)(); ms = (C.c_double * nk)(); fl = (C.c_double * nk)(); by = (C.c_double * nk)()
        F._lib.call("frcnn_prof_collect", la, ms, fl, by)
        F._lib.call("frcnn_prof_enable", 1 << k)
        for _ in range(reps):
            fn()
        F._lib.call("frcnn_prof_enable", 0)
        F._lib.call("frcnn_prof_collect", la, ms, fl, by)
        t = ms[k] / reps
        print("R=%-5d %-22s %7.1f us   %6.2f TB/s of algorithmic bytes (%.1f MB)" % (R, name, t * 1e3, by[k] / reps / 1e12 / (t * 1e-3), by[k] / reps / 1e6), flush=True)


if __name__ == "__main__":
    for R in [int(a) for a in sys.argv[1:]] or [560, 1398]:
        run(R)
