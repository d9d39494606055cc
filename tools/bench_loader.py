"""End-to-end loader throughput ("at >= 200 img/s/GPU the host loader becomes the bottleneck"): JPEG files
on disk -> BatchIterator (decode-ahead pool, 8-bit upload, processImage + example assembly on the way) -> training step.
usage: python tools/bench_loader.py [n_files] [workers]"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from PIL import Image
import frcnn_amd as F

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 192   # (an epoch longer than the timed loops: the order of the next
# epoch is only drawn when the current one ends, so decode-ahead stalls for one decode latency at every epoch boundary)
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = tempfile.mkdtemp(prefix="frcnn_loader_")
rng = np.random.RandomState(0)
yy, xx = np.mgrid[0:1080, 0:1920]
gt = {}
for i in range(n_files):   # smooth content + noise: JPEGs of a realistic size (a few hundred KB)
    base = np.stack([np.sin(xx / (40.0 + i) + c) * np.cos(yy / (55.0 + 2 * i) - c) for c in range(3)], -1)
    px = np.clip(127 + 90 * base + rng.randn(1080, 1920, 3) * 12, 0, 255).astype(np.uint8)
    fn = "f%03d.jpg" % i
    Image.fromarray(px).save(os.path.join(d, fn), quality=90)
    gt[fn] = dict(rois=[F.Roi(F.Rect(200 + 30 * j + 5 * i, 150 + 40 * j, 700 + 60 * j, 640 + 50 * j), 1 + j) for j in range(3)])
sz = sum(os.path.getsize(os.path.join(d, f)) for f in gt) / n_files / 1e3
cfg = dict(F.duplo_cfg); cfg["examples_base_path"] = d
model = F.vgg_small(cfg)
data = dict(ground_truth=gt, training_set=sorted(gt), validation_set=[], background_files=[])
print("%d JPEG files 1920x1080, %.0f KB each; host cores %d" % (n_files, sz, os.cpu_count()))

t0 = time.perf_counter()
for f in sorted(gt)[:8]:
    F.decode_image(os.path.join(d, f))
print("single-thread decode to float: %.1f ms per frame" % ((time.perf_counter() - t0) / 8 * 1e3))

def loader_rate(workers, n=96):
    it = F.BatchIterator(model, data, workers=workers, prefetch=2 * max(workers, 1), seed=1)
    for _ in range(8):
        it.nextTraining(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        it.nextTraining(1)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)

for w in (0, 4, workers):
    print("BatchIterator.nextTraining alone, workers=%2d: %.0f images/s" % (w, loader_rate(w, 48 if w == 0 else 192)))

w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
for wk in (0, workers):
    it = F.BatchIterator(model, data, workers=wk, prefetch=2 * max(wk, 1), seed=1)
    class OneImage(object):
        def nextTraining(self, count=None):
            return it.nextTraining(1)
    f = F.create_objective(model, w, g, OneImage(), dict(pcls=[], preg=[], dcls=[], dreg=[]))
    st = dict(learningRate=1e-4, alpha=0.9)
    for _ in range(6):
        F.rmsprop(f, w, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30 if wk == 0 else min(150, n_files - 16)
    for _ in range(n):
        F.rmsprop(f, w, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("training step fed from the JPEG files, workers=%2d: %.2f ms/step = %.1f images/s" % (wk, dt * 1e3, 1 / dt))

# the same kind of batches, prepared once and replayed from HBM (what bench.py does with its synthetic pool): the
# device-bound step time of THIS workload, for comparison with the file-fed loop above
it = F.BatchIterator(model, data, workers=workers, seed=1)
pool = []
for _ in range(8):
    b = it.nextTraining(1)
    for x in b:
        x["img"] = x["img"].clone()
    pool.append(b)
class Replay(object):
    i = 0
    def nextTraining(self, count=None):
        self.i += 1
        return pool[self.i % len(pool)]
f = F.create_objective(model, w, g, Replay(), dict(pcls=[], preg=[], dcls=[], dreg=[]))
st = dict(learningRate=1e-4, alpha=0.9)
for _ in range(10):
    F.rmsprop(f, w, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    F.rmsprop(f, w, st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print("the same batches replayed from HBM: %.2f ms/step = %.1f images/s; examples per image %s" % (
    dt * 1e3, 1 / dt, [len(b[0]["positive"]) + len(b[0]["negative"]) for b in pool]))
