"""Per-block phase timing of conv_igemm (debug build with -DIG_TRACE=1 copied over libfrcnn_hip.so).
usage: python tools/ig_trace.py [layer]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import frcnn_amd as F
from bench_conv import LAYERS

name = sys.argv[1] if len(sys.argv) > 1 else "b2c2"
Cin, H, W, O, k, pad = LAYERS[name]
Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
rng = np.random.RandomState(0)
x = F.DeviceTensor.from_numpy(rng.randn(Cin, H, W).astype(np.float32))
w = F.DeviceTensor.from_numpy((rng.randn(O, Cin, k, k) * 0.05).astype(np.float32))
out = F.DeviceTensor.empty((O, Ho, Wo))
s = F.stream_ptr()
for _ in range(3):
    F._lib.call("frcnn_conv2d_forward", F.ptr(x), Cin, H, W, None, None, F.ptr(w), None, O, k, pad, F.ptr(out), s)
F._lib.call("frcnn_stream_sync", s)
lib = F._lib.load()
buf = np.zeros((4096, 16), dtype=np.uint64)
rc = lib.frcnn_debug_ig_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
t = buf[buf[:, 2] > 0].astype(np.int64)
n = len(t)
t0, tE, t2 = t[:, 0], t[:, 1], t[:, 2]
span = t2.max() - t0.min()
print("blocks traced:", n, " span (ticks):", span)
tot = t2 - t0
print("block total   mean %.0f  min %d  max %d" % (tot.mean(), tot.min(), tot.max()))
print("start offset  mean %.0f  max %d" % ((t0 - t0.min()).mean(), (t0 - t0.min()).max()))
print("end   offset  mean %.0f  min %d (before last end)" % ((t2.max() - t2).mean(), (t2.max() - t2).min()))
for nm, col in (("stage", 3), ("barrier1", 4), ("compute", 5), ("barrier2", 6)):
    print("%-9s sum/block mean %.0f  (%.1f%% of block)" % (nm, t[:, col].mean(), 100.0 * t[:, col].mean() / tot.mean()))
print("epilogue  mean %.0f (%.1f%%)" % ((t2 - tE).mean(), 100.0 * (t2 - tE).mean() / tot.mean()))
pro = tot - (t2 - tE) - t[:, 3:7].sum(1)
print("prologue  mean %.0f (%.1f%%)" % (pro.mean(), 100.0 * pro.mean() / tot.mean()))
hw = t[:, 7]
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
xcc = t[:, 8] & 0xF
key = xcc * 1000 + se * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs:", len(u), " blocks/CU histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
full = np.isin(key, u[cnt == 3])
for nm, sel in (("3-block CUs", full), ("2-block CUs", ~full)):
    tt = t[sel]; to = tot[sel]
    print("%s: n=%d total mean %.0f max %d | stage %.0f b1 %.0f compute %.0f b2 %.0f epi %.0f pro %.0f" % (
        nm, len(tt), to.mean(), to.max(), tt[:, 3].mean(), tt[:, 4].mean(), tt[:, 5].mean(), tt[:, 6].mean(),
        (tt[:, 2] - tt[:, 1]).mean(), pro[sel].mean()))
print("3-block CUs: load issue %.0f  load wait %.0f  lds store %.0f (ticks per block)" % (t[full, 11].mean(), t[full, 12].mean(), t[full, 3].mean()))
for x in np.unique(xcc)[:0]:
    sel = full & (xcc == x)
    t0x = t0[xcc == x].min()
    print("xcc %d: n3=%d total mean %.0f min %d max %d | start-offset mean %.0f max %d | end max %d  compute %.0f stage %.0f" % (
        x, sel.sum(), tot[sel].mean(), tot[sel].min(), tot[sel].max(), (t0[sel] - t0x).mean(), (t0[sel] - t0x).max(),
        (t2[xcc == x] - t0x).max(), t[sel, 5].mean(), t[sel, 3].mean()))
shown = 0
ends = []
for kk in u[cnt == 3]:
    rows = t[key == kk]
    base = rows[:, 0].min()
    o = np.argsort(rows[:, 0])
    rows = rows[o]
    ends.append(((rows[:, 1] - base), (rows[:, 2] - base), rows[:, 0] - base))
    if shown < 6:
        print("CU %d: start %s  loop-end %s  end %s" % (kk, (rows[:, 0] - base).tolist(), (rows[:, 1] - base).tolist(), (rows[:, 2] - base).tolist()))
        shown += 1
E = np.array([np.sort(e[1]) for e in ends]); L = np.array([np.sort(e[0]) for e in ends]); S = np.array([np.sort(e[2]) for e in ends])
print("per-CU sorted starts mean", S.mean(0), " loop-ends mean", L.mean(0), " ends mean", E.mean(0), " last end max", E[:, 2].max())
rt = t[:, 9]
d = (t[:, 9] - t[:, 10]).astype(np.float64)
print("tick frequency: %.3f GHz (mean over blocks), kernel span %.1f us" % ((tot / d).mean() * 0.1, (t[:, 9].max() - t[:, 10].min()) / 100.0))
print("realtime span (100MHz ticks):", rt.max() - rt.min())
