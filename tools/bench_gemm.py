"""Micro-benchmark of the cnet Linear kernels (HIP-event timing via the library profiler)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import frcnn_amd as F

def run(R, I, O, reps=5):
    rng = np.random.RandomState(0)
    x = F.DeviceTensor.from_numpy(rng.randn(R, I).astype(np.float32)); w = F.DeviceTensor.from_numpy(rng.randn(O, I).astype(np.float32))
    b = F.DeviceTensor.from_numpy(rng.randn(O).astype(np.float32)); gy = F.DeviceTensor.from_numpy(rng.randn(R, O).astype(np.float32))
    y = F.DeviceTensor.empty((R, O)); gx = F.DeviceTensor.empty((R, I)); gw = F.DeviceTensor.zeros((O, I)); gb = F.DeviceTensor.zeros((O,))
    s = F.stream_ptr(); nk = len(F._lib.KC_NAMES)
    la = (C.c_longlong * nk)(); ms = (C.c_double * nk)(); fl = (C.c_double * nk)(); by = (C.c_double * nk)()
    ops = {"fwd": lambda: F._lib.call("frcnn_linear_forward", F.ptr(x), R, I, F.ptr(w), F.ptr(b), O, F.ptr(y), s),
           "dgrad": lambda: F._lib.call("frcnn_linear_backward", F.ptr(x), F.ptr(gy), R, I, F.ptr(w), O, F.ptr(gx), None, None, s),
           "wgrad": lambda: F._lib.call("frcnn_linear_backward", F.ptr(x), F.ptr(gy), R, I, F.ptr(w), O, None, F.ptr(gw), None, s)}
    for name, op in ops.items():
        op()
        F._lib.call("frcnn_prof_enable", 0x3FF)
        for _ in range(reps): op()
        F._lib.call("frcnn_prof_enable", 0)
        F._lib.call("frcnn_prof_collect", la, ms, fl, by)
        t = sum(ms) / reps
        kg = F._lib.KC_NAMES.index("gemm"); ke = F._lib.KC_NAMES.index("elemwise")
        print("R=%d I=%d O=%d %-6s %8.1f us  %6.2f TFLOP/s  (product kernel %.1f us x %d, split planes / slab fold %.1f us x %d)"
              % (R, I, O, name, t * 1e3, 2.0 * R * I * O / t / 1e9, ms[kg] / reps * 1e3, la[kg] // reps, ms[ke] / reps * 1e3, la[ke] // reps), flush=True)

if __name__ == "__main__":
    if os.environ.get("ONLY_R"):
        run(int(os.environ["ONLY_R"]), 13824, 1024, reps=3)
        sys.exit(0)
    for R in (138, 320, 560):
        run(R, 13824, 1024)
    run(560, 1024, 512)
