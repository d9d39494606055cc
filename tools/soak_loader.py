"""Soak run of the file-fed training loop (decode-ahead pool + native example assembly + async streams): 2000 steps over JPEG\nframes of four different sizes; prints throughput, first/last loss, finiteness.  usage: python tools/soak_loader.py"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from PIL import Image
import frcnn_amd as F
d = tempfile.mkdtemp(prefix="frcnn_soak_")
rng = np.random.RandomState(0)
gt = {}
for i in range(64):
    W, H = [(1920, 1080), (1280, 720), (1000, 1000), (800, 1200)][i % 4]
    px = rng.randint(0, 255, size=(H // 8, W // 8, 3)).astype(np.uint8).repeat(8, 0).repeat(8, 1)
    fn = "f%03d.jpg" % i
    Image.fromarray(px).save(os.path.join(d, fn), quality=80)
    gt[fn] = dict(rois=[F.Roi(F.Rect(0.1 * W + 20 * j, 0.1 * H + 15 * j, 0.5 * W + 30 * j, 0.6 * H + 20 * j), 1 + j) for j in range(1 + i % 3)])
cfg = dict(F.duplo_cfg); cfg["examples_base_path"] = d
model = F.vgg_small(cfg)
data = dict(ground_truth=gt, training_set=sorted(gt), validation_set=[], background_files=[])
w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
it = F.BatchIterator(model, data, workers=12, prefetch=24, seed=1)
class OneImage(object):
    def nextTraining(self, count=None):
        return it.nextTraining(1)
stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
f = F.create_objective(model, w, g, OneImage(), stats)
st = dict(learningRate=1e-4, alpha=0.9)
t0 = time.perf_counter()
for i in range(2000):
    F.rmsprop(f, w, st)
torch.cuda.synchronize()
print("2000 file-fed steps over mixed frame sizes: %.1f s, %.1f images/s; loss first/last %.3f / %.3f; weights finite %s" % (
    time.perf_counter() - t0, 2000 / (time.perf_counter() - t0), stats["pcls"][0] + stats["preg"][0], stats["pcls"][-1] + stats["preg"][-1],
    bool(torch.isfinite(w).all())))
