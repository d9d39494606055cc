"""Error of the 3x3 convolution forms against an fp64 reference at sampled outputs, full-size layer shapes:
python tools/x3_f16_check.py bf16x6|f16x3|fp32 [layer ...]   (DYN=1: per-channel magnitudes spread over four decades)"""
import os
import sys

mode = sys.argv[1]
os.environ["FRCNN_X3_F16"] = "1" if mode == "f16x3" else "0"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import frcnn_amd as F
from bench_conv import LAYERS


def run(name, nsamp=4000):
    Cin, H, W, O, k, pad = LAYERS[name]
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    rng = np.random.RandomState(5)
    x = rng.randn(Cin, H, W).astype(np.float32)
    g = (rng.randn(O, Ho, Wo) * 1e-4).astype(np.float32)      # gradients are small numbers
    if os.environ.get("DYN"):
        x *= (10.0 ** rng.uniform(-3, 1, size=(Cin, 1, 1))).astype(np.float32)
        g *= (10.0 ** rng.uniform(-3, 1, size=(O, 1, 1))).astype(np.float32)
    w = (rng.randn(O, Cin, k, k) * 0.05).astype(np.float32)
    F._lib.call("frcnn_set_option", b"split_bf16", 0 if mode == "fp32" else 1)
    s = F.stream_ptr()
    dx, dg, dw = (F.DeviceTensor.from_numpy(a) for a in (x, g, w))
    out = F.DeviceTensor.empty((O, Ho, Wo)); gin = F.DeviceTensor.empty((Cin, H, W))
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx), Cin, H, W, None, None, F.ptr(dw), None, O, k, pad, F.ptr(out), s)
    F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O, Ho, Wo, F.ptr(dw), Cin, k, pad, F.ptr(gin), 0, s)
    y, gi = out.numpy(), gin.numpy()
    xp = np.pad(x.astype(np.float64), ((0, 0), (pad, pad), (pad, pad)))
    w64 = w.astype(np.float64)
    # forward samples
    fe, fs = [], []
    for _ in range(nsamp):
        o, yy, xx = rng.randint(O), rng.randint(Ho), rng.randint(Wo)
        patch = xp[:, yy:yy + k, xx:xx + k]
        ref = float((patch * w64[o]).sum()); mag = float(np.abs(patch * w64[o]).sum())
        fe.append(abs(float(y[o, yy, xx]) - ref)); fs.append(mag)
    # input-gradient samples: gin[c][y][x] = sum_o sum_tap g[o][y + pad - ky][x + pad - kx] w[o][c][ky][kx]
    gp = np.pad(g.astype(np.float64), ((0, 0), (k - 1 - pad, k - 1 - pad), (k - 1 - pad, k - 1 - pad)))
    wf = w64[:, :, ::-1, ::-1]
    de, ds = [], []
    for _ in range(nsamp):
        c, yy, xx = rng.randint(Cin), rng.randint(H), rng.randint(W)
        patch = gp[:, yy:yy + k, xx:xx + k]
        ref = float((patch * wf[:, c]).sum()); mag = float(np.abs(patch * wf[:, c]).sum())
        de.append(abs(float(gi[c, yy, xx]) - ref)); ds.append(mag)
    fe, fs, de, ds = map(np.array, (fe, fs, de, ds))
    # error relative to the sum of the magnitudes of the terms (the scale on which fp32 accumulation itself rounds)
    print("%-6s %-5s fwd   max err/|terms| %.2e  mean %.2e | dgrad max %.2e  mean %.2e | nan %d" %
          (mode, name, (fe / fs).max(), (fe / fs).mean(), (de / ds).max(), (de / ds).mean(),
           int(np.isnan(y).sum() + np.isnan(gi).sum())), flush=True)


if __name__ == "__main__":
    for n in sys.argv[2:] or ["b2c2", "b3c2", "b4c2"]:
        run(n)
