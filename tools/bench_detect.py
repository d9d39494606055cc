"""Detector:detect (Detector.lua:17-141) latency on synthetic 3x450x800 frames.
Head logits are amplified so that a realistic number of anchors passes the p > 0.95 test."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import frcnn_amd as F

cfg = dict(F.duplo_cfg)
if os.environ.get("CLASSES"):      # e.g. CLASSES=200: the per-class NMS stage with config/imagenet.lua's class count
    cfg["class_count"] = int(os.environ["CLASSES"])
model = F.vgg_small(cfg)
weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
nat = model["native"]
w = weights.cpu().numpy().copy()
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
for off, cnt, kind, aux in nat.param_table:
    if kind == 0 and aux == 18:
        v = w[off:off + cnt].reshape(18, -1)
        for a in range(3):
            v[a * 6:a * 6 + 2] *= amp
    if kind == 3 and cnt == 512 * (cfg["class_count"] + 1):
        w[off:off + cnt] *= float(os.environ.get("CLS_GAIN", "200"))
weights.copy_(torch.from_numpy(w))
d = F.Detector(model)   # FRCNN_STATIC_WEIGHTS=1: the packed weight copies are made once, not per frame
imgs = [F.to_device(F.synthetic_image(450, 800, i)) if hasattr(F, "to_device") else F.synthetic_image(450, 800, i) for i in range(4)]
for i in range(3):
    r = d.detect(imgs[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 20
for i in range(n):
    r = d.detect(imgs[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("classes %d, winners in %d classes" % (cfg["class_count"], len(set(x["class"] for x in r))))
print("detect: %.2f ms/image (%.1f images/s); matches %d, candidates after NMS %d, winners %d" % (
    dt * 1e3, 1.0 / dt, len(d.last_scan["idx"].numpy()) if d.last_scan else -1, len(d.last_pick) if d.last_pick is not None else -1, len(r)))
if len(sys.argv) > 2:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(10):
        r = d.detect(imgs[i % 4])
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
