"""Micro-benchmark of the conv kernels on the vgg_small 800x450 layer shapes (HIP-event timing through the
library's per-class profiler).  usage: python tools/bench_conv.py [wgrad|fwd|dgrad] [layer ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import frcnn_amd as F

LAYERS = {  # name: (Cin, H, W, Cout, k, pad)
    "b1c1": (3, 450, 800, 64, 3, 1), "b2c1": (64, 225, 400, 128, 3, 1), "b2c2": (128, 225, 400, 128, 3, 1),
    "b3c1": (128, 113, 200, 256, 3, 1), "b3c2": (256, 113, 200, 256, 3, 1), "b4c1": (256, 57, 100, 384, 3, 1),
    "b4c2": (384, 57, 100, 384, 3, 1), "a1": (256, 57, 100, 256, 3, 0), "a2": (384, 29, 50, 256, 3, 0),
    # vgg_large 1000x600 (models/vgg_large.lua: 64/128/256/512, 2-2-3-3)
    "L1c2": (64, 600, 1000, 64, 3, 1), "L2c1": (64, 300, 500, 128, 3, 1), "L2c2": (128, 300, 500, 128, 3, 1),
    "L3c1": (128, 150, 250, 256, 3, 1), "L3c2": (256, 150, 250, 256, 3, 1), "L4c1": (256, 75, 125, 512, 3, 1),
    "L4c2": (512, 75, 125, 512, 3, 1),
    # a step's shapes behind SpatialDropout(0.4) with the dropped channels left out (k: as the K dimension, multiples of 16; m: as
    # filters / a 64-channel tile dimension, multiples of 64)
    "b2c2k": (80, 225, 400, 128, 3, 1), "b3c2k": (160, 113, 200, 256, 3, 1), "b4c2k": (240, 57, 100, 384, 3, 1),
    "b2c1k": (64, 225, 400, 80, 3, 1), "b3c1k": (128, 113, 200, 160, 3, 1), "b4c1k": (256, 57, 100, 240, 3, 1),
    "b3c1m": (128, 113, 200, 192, 3, 1), "b4c1m": (256, 57, 100, 256, 3, 1),
    "b3c2m": (192, 113, 200, 256, 3, 1), "b4c2m": (256, 57, 100, 384, 3, 1),
    "bigk": (2048, 57, 100, 384, 3, 1), "a3": (384, 29, 50, 256, 5, 0), "a4": (384, 29, 50, 256, 7, 0), "a1x": (256, 55, 98, 18, 1, 0),
}


def run(kind, name, reps=5):
    Cin, H, W, O, k, pad = LAYERS[name]
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    rng = np.random.RandomState(0)
    # DATA=zero | bf16 (values with 8 significant bits: the m and l planes of the split are zero) | default: normal deviates
    mode = os.environ.get("DATA", "randn")
    def data(*shape, s=1.0):
        v = (rng.randn(*shape) * s).astype(np.float32)
        if mode == "zero": v[:] = 0
        if mode == "bf16": v = (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        if mode == "sparse": v = np.maximum(v, 0) * (rng.rand(shape[0], 1, 1) > 0.4)   # PReLU-like outputs behind a SpatialDropout
        return np.ascontiguousarray(v, dtype=np.float32)
    x = F.DeviceTensor.from_numpy(data(Cin, H, W))
    g = F.DeviceTensor.from_numpy(data(O, Ho, Wo))
    w = F.DeviceTensor.from_numpy(data(O, Cin, k, k, s=0.05))
    gw = F.DeviceTensor.zeros((O, Cin, k, k)); out = F.DeviceTensor.empty((O, Ho, Wo)); gin = F.DeviceTensor.empty((Cin, H, W))
    act = bool(os.environ.get("WITH_ACT"))   # fused PReLU + dropout scale of the producing layer on the input
    slope = F.DeviceTensor.from_numpy(np.array([0.25], np.float32)); scale = F.DeviceTensor.from_numpy((rng.rand(Cin) > 0.4).astype(np.float32))
    s = F.stream_ptr()
    nk = len(F._lib.KC_NAMES)
    la = (C.c_longlong * nk)(); ms = (C.c_double * nk)(); fl = (C.c_double * nk)(); by = (C.c_double * nk)()

    def once():
        if kind == "wgrad":
            F._lib.call("frcnn_conv2d_backward_weight", F.ptr(x), Cin, H, W, F.ptr(slope) if os.environ.get("WITH_SLOPE") else None, None,
                        F.ptr(g), O, k, pad, F.ptr(gw), None, s)
        elif kind == "fwd":
            F._lib.call("frcnn_conv2d_forward", F.ptr(x), Cin, H, W, F.ptr(slope) if (act or os.environ.get("WITH_SLOPE")) else None, F.ptr(scale) if act else None,
                        F.ptr(w), None, O, k, pad, F.ptr(out), s)
        else:
            F._lib.call("frcnn_conv2d_backward_input", F.ptr(g), O, Ho, Wo, F.ptr(w), Cin, k, pad, F.ptr(gin), 0, s)
    once()
    conv = [i for i, n in enumerate(F._lib.KC_NAMES) if n.startswith("conv_") or (n == "elemwise" and os.environ.get("WITH_FOLD"))]
    F._lib.call("frcnn_prof_enable", sum(1 << i for i in conv))
    for _ in range(reps):
        once()
    F._lib.call("frcnn_prof_enable", 0)
    F._lib.call("frcnn_prof_collect", la, ms, fl, by)
    t = sum(ms[i] for i in conv) / reps
    flops = 2.0 * O * Cin * k * k * Ho * Wo
    print("%-6s %-5s %8.1f us  %6.1f TFLOP/s  (%.2f GFLOP)" % (kind, name, t * 1e3, flops / t / 1e9, flops / 1e9), flush=True)


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
    names = sys.argv[2:] or [n for n in LAYERS if not n.startswith("L") and not n[-1] in "km"]
    for n in names:
        run(kind, n)
