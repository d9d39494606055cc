"""BatchIterator.processImage on the device: time per 1080p frame -> 800x450 prepared frame,
per-kernel HIP-event time of the image class, algorithmic HBM bytes, and the numpy restatement on the host CPU
beside it.  usage: python tests/perf_image.py [H W]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import frcnn_amd as F

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
rng = np.random.RandomState(0)
rgb = rng.rand(3, H, W).astype(np.float32)
data = dict(ground_truth={}, training_set=["a"], validation_set=[], background_files=[])
it = F.BatchIterator(model, data, load_image=lambda fn: d_rgb, seed=1)
d_rgb = F.DeviceTensor.from_numpy(rgb)
for _ in range(3):
    img, _ = it.processImage(it.load_image("a"))
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    img, _ = it.processImage(it.load_image("a"))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
_, h, w = img.shape
# algorithmic traffic: row pass (rgb2yuv fused) r + w(tmp), column pass r(tmp) + w, flip gather r+w (counted although only
# drawn for some frames), centring/scaling: 4 passes = 2 reads + 2 read-modify-writes, contrastive: 2 x (r + w) of one plane
src = 3 * H * W * 4; tmpb = 3 * H * w * 4; dst = 3 * h * w * 4
alg = src + tmpb + tmpb + dst + 2 * dst + 6 * dst + 4 * (dst // 3)
print("processImage %dx%d -> %dx%d: %.1f us/frame wall (%.0f frames/s), algorithmic %.1f MB -> %.2f TB/s" %
      (W, H, w, h, dt * 1e6, 1 / dt, alg / 1e6, alg / dt / 1e12))
nk = len(F._lib.KC_NAMES)
la = (C.c_longlong * nk)(); ms = (C.c_double * nk)(); fl = (C.c_double * nk)(); by = (C.c_double * nk)()
F._lib.call("frcnn_prof_enable", 1 << F._lib.KC_NAMES.index("image"))
for _ in range(10):
    it.processImage(it.load_image("a"))
F._lib.call("frcnn_prof_enable", 0)
F._lib.call("frcnn_prof_collect", la, ms, fl, by)
i = F._lib.KC_NAMES.index("image")
print("image kernels: %d launches/frame, %.1f us of kernel time/frame, %.2f TB/s on the bytes the launches declare" %
      (la[i] // 10, ms[i] / 10 * 1e3, by[i] / (ms[i] * 1e-3) / 1e12 if ms[i] else 0))
import orc_image as OI
t0 = time.perf_counter()
want = OI.process_image(OI.rgb2yuv(rgb), cfg, False, False)
cpu = time.perf_counter() - t0
print("numpy restatement on the host: %.1f ms/frame (%.0fx)" % (cpu * 1e3, cpu / dt))
