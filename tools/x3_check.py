"""conv_x3 against the fp32 matrix-core kernel on the model's own layer shapes (full size), forward with and without the
fused input activation and input gradient: python tools/x3_check.py [layer ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import frcnn_amd as F
from bench_conv import LAYERS


def run(name):
    Cin, H, W, O, k, pad = LAYERS[name]
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    rng = np.random.RandomState(1)
    x = F.DeviceTensor.from_numpy(rng.randn(Cin, H, W).astype(np.float32))
    g = F.DeviceTensor.from_numpy(rng.randn(O, Ho, Wo).astype(np.float32))
    w = F.DeviceTensor.from_numpy((rng.randn(O, Cin, k, k) * 0.05).astype(np.float32))
    b = F.DeviceTensor.from_numpy(rng.randn(O).astype(np.float32))
    ident = os.environ.get("IDENT")   # identity activation: the act = 1 result must equal the act = 0 one
    slope = F.DeviceTensor.from_numpy(np.array([1.0 if ident else 0.25], np.float32))
    scale = F.DeviceTensor.from_numpy(np.ones(Cin, np.float32) if ident else (rng.rand(Cin) > 0.4).astype(np.float32))
    which = os.environ.get("ACT", "both")
    s = F.stream_ptr()
    res = {}
    for split in (1, 0):
        F._lib.call("frcnn_set_option", b"split_bf16", split)
        for act in (0, 1):
            out = F.DeviceTensor.empty((O, Ho, Wo))
            F._lib.call("frcnn_conv2d_forward", F.ptr(x), Cin, H, W, F.ptr(slope) if act and which in ("both", "slope") else None, F.ptr(scale) if act and which in ("both", "scale") else None,
                        F.ptr(w), F.ptr(b), O, k, pad, F.ptr(out), s)
            res[("fwd", act, split)] = out.numpy()
        gin = F.DeviceTensor.empty((Cin, H, W))
        F._lib.call("frcnn_conv2d_backward_input", F.ptr(g), O, Ho, Wo, F.ptr(w), Cin, k, pad, F.ptr(gin), 0, s)
        res[("dgrad", 0, split)] = gin.numpy()
    F._lib.call("frcnn_set_option", b"split_bf16", 1)
    for key in (("fwd", 0), ("fwd", 1), ("dgrad", 0)):
        a, r = res[key + (1,)], res[key + (0,)]
        err = np.abs(a - r) / np.maximum(1.0, np.abs(r))
        bad = np.argwhere(err > 1e-3)
        print("%-5s %-5s act=%d  worst %.2e  bad elements %d%s" % (name, key[0], key[1], err.max(), len(bad),
              ("  first at " + str(bad[0].tolist()) + " last at " + str(bad[-1].tolist())) if len(bad) else ""), flush=True)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["b2c1", "b2c2", "b3c1", "b3c2", "b4c1", "b4c2", "a1", "a2"]:
        run(n)
