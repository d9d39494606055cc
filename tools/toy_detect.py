"""Behavioural check with a quality metric (the reference has no evaluation loop): train vgg_small on generated
frames that contain 1-3 flat-coloured rectangles (class = colour), then run Detector:detect on held-out frames and score the
detections against the ground truth (a detection counts if its class matches and IoU >= 0.5).  Not a benchmark: it shows
that loader -> objective -> optimiser -> detector work together as a system.  usage: python tools/toy_detect.py [steps] [learning rate]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import frcnn_amd as F

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4   # main.lua:122
COLORS = [(0.9, 0.15, 0.15), (0.15, 0.85, 0.2), (0.2, 0.3, 0.95)]
cfg = dict(F.duplo_cfg); cfg["class_count"] = 3
cfg["augmentation"] = dict(vflip=0.5, hflip=0.5, random_scaling=0.0, aspect_jitter=0.0)


def make_frame(rng):
    H, W = 450, 800
    img = (0.35 + 0.08 * rng.randn(3, H, W)).astype(np.float32)
    rois = []
    for _ in range(rng.randint(1, 4)):
        s = [70, 110, 160][rng.randint(3)]
        bw, bh = [(s, s), (1.4 * s, 0.7 * s), (0.7 * s, 1.4 * s)][rng.randint(3)]
        x0 = rng.uniform(5, W - bw - 5); y0 = rng.uniform(5, H - bh - 5)
        c = rng.randint(3)
        r = F.Rect(int(x0), int(y0), int(x0 + bw), int(y0 + bh))
        if any(F.Rect.IoU(r, o.rect) > 0.05 for o in rois):
            continue
        for ch in range(3):
            img[ch, r.minY:r.maxY, r.minX:r.maxX] = COLORS[c][ch] + 0.03 * rng.randn(r.maxY - r.minY, r.maxX - r.minX)
        rois.append(F.Roi(r, c + 1))
    return np.clip(img, 0, 1), rois


rng = np.random.RandomState(0)
frames, gt = {}, {}
for i in range(260):
    frames["t%03d" % i], rois = make_frame(rng)
    gt["t%03d" % i] = dict(rois=rois)
names = sorted(frames)
data = dict(ground_truth=gt, training_set=names[:240], validation_set=names[240:], background_files=[])
model = F.vgg_small(cfg)
w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
it = F.BatchIterator(model, data, load_image=lambda fn: frames[fn], seed=1)
class OneImage(object):
    def nextTraining(self, count=None):
        return it.nextTraining(1)
stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
f = F.create_objective(model, w, g, OneImage(), stats)
st = dict(learningRate=lr, alpha=0.9)
t0 = time.perf_counter()
for i in range(steps):
    F.rmsprop(f, w, st)
    if (i + 1) % 500 == 0:
        k = slice(-200, None)
        print("step %5d  pcls %.3f preg %.3f dcls %.3f dreg %.3f  (%.0f img/s)" % (
            i + 1, np.mean(stats["pcls"][k]), np.mean(stats["preg"][k]), np.mean(stats["dcls"][k]), np.mean(stats["dreg"][k]),
            (i + 1) / (time.perf_counter() - t0)), flush=True)
torch.cuda.synchronize()

# diagnostic: eval-mode classification of the ground-truth boxes themselves (pooled exactly like training positives)
from frcnn_amd.objective import roi_windows
import ctypes as C
pnet, cnet = model["pnet"], model["cnet"]
loc = F.Localizer(pnet.outnode.children[4])
br = model["native"].bn_running.cpu().numpy()
print("bn_running: mean |.| %.4f max %.4f ; var mean %.4f min %.4f max %.4f" % (np.abs(br[:1024]).mean(), np.abs(br[:1024]).max(), br[1024:].mean(), br[1024:].min(), br[1024:].max()))
for mode in ("evaluate", "training"):
  getattr(pnet, mode)(); getattr(cnet, mode)()
  hit = tot = 0
  if True:
    for fn in data["validation_set"]:
        img, rois = it.processImage(it.load_image(fn), [F.Roi(r.rect.clone(), r.class_index) for r in gt[fn]["rois"]])
        outs = pnet.forward(img)
        fm = outs[-1]; fmC, fmH, fmW = fm.shape
        rects = np.array([(r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY) for r in rois], dtype=np.float64)
        wins = roi_windows(rects, loc, fmH, fmW)
        dw = F.DeviceTensor.from_numpy(wins.astype(np.int32))
        R = len(rois); D = fmC * 36
        cin = F.DeviceTensor.empty((R, D)); pidx = F.DeviceTensor.empty((R, D), np.int32)
        F._lib.call("frcnn_roi_pool_forward", F.ptr(fm), fmC, fmH, fmW, F.ptr(dw), R, 6, 6, F.ptr(cin), F.ptr(pidx), F.stream_ptr())
        bbox, cls = cnet.forward(cin)
        pred = cls.numpy().argmax(1) + 1
        hit += int((pred == np.array([r.class_index for r in rois])).sum()); tot += R
  print("%s-mode cnet on the ground-truth boxes of the held-out frames: %d / %d classified correctly" % (mode, hit, tot))
# a training-shaped batch (positives pool the ground-truth box, negatives the anchor box), training-mode BN (batch
# statistics), no dropout: does the classifier separate the rows it is trained on?
pnet.training(); cnet.training()
pnet.drop_masks = [np.ones(64, np.float32), np.ones(128, np.float32), np.ones(256, np.float32), np.ones(384, np.float32)]
hit = tot = negok = negtot = 0
conf = np.zeros((3, 4), dtype=np.int64)
arng = F.MT19937(5)
for fn in data["validation_set"]:
    img, rois = it.processImage(it.load_image(fn), [F.Roi(r.rect.clone(), r.class_index) for r in gt[fn]["rois"]])
    outs = pnet.forward(img)
    fm = outs[-1]; fmC, fmH, fmW = fm.shape
    pos, neg = F.assemble_examples(it.anchors, cfg, rois, 800, 450, arng)
    rects = np.array([(e[1].rect.minX, e[1].rect.minY, e[1].rect.maxX, e[1].rect.maxY) for e in pos] +
                     [(e[0].minX, e[0].minY, e[0].maxX, e[0].maxY) for e in neg], dtype=np.float64)
    wins = roi_windows(rects, loc, fmH, fmW)
    dw = F.DeviceTensor.from_numpy(wins.astype(np.int32))
    R = len(rects); D = fmC * 36
    cin = F.DeviceTensor.empty((R, D)); pidx = F.DeviceTensor.empty((R, D), np.int32)
    F._lib.call("frcnn_roi_pool_forward", F.ptr(fm), fmC, fmH, fmW, F.ptr(dw), R, 6, 6, F.ptr(cin), F.ptr(pidx), F.stream_ptr())
    cnet.drop_masks = [np.ones((R, 1024), np.float32), np.ones((R, 512), np.float32)]
    bbox, cls = cnet.forward(cin)
    pred = cls.numpy().argmax(1) + 1
    want = np.array([e[1].class_index for e in pos] + [4] * len(neg))
    hit += int((pred[:len(pos)] == want[:len(pos)]).sum()); tot += len(pos)
    for a_, b_ in zip(want[:len(pos)], pred[:len(pos)]): conf[a_ - 1, b_ - 1] += 1
    negok += int((pred[len(pos):] == 4).sum()); negtot += len(neg)
print("confusion (rows: true class 1..3, columns: predicted 1..3, background):", conf.tolist())
print("training-shaped batches (batch-statistics BN, no dropout): positives %d / %d correct, negatives %d / %d background" % (hit, tot, negok, negtot))
pnet.drop_masks = None; cnet.drop_masks = None

det = F.Detector(model)
tp = fp = ngt = 0
nscan = ncand = 0; predhist = np.zeros(6, dtype=np.int64)
ious = []
for fn in data["validation_set"]:
    img, rois = it.processImage(it.load_image(fn), [F.Roi(r.rect.clone(), r.class_index) for r in gt[fn]["rois"]])
    winners = det.detect(img)
    nscan += det.last_scan["n"] if det.last_scan else 0; ncand += len(det.last_pick) if det.last_pick is not None else 0
    if det.last_cnet is not None: pc = det.last_cnet["cls"].argmax(1) + 1; predhist += np.bincount(pc, minlength=6)[:6]
    ngt += len(rois)
    used = set()
    for x in winners:
        best, bj = 0.0, -1
        for j, r in enumerate(rois):
            v = F.Rect.IoU(x["r2"], r.rect)
            if j not in used and r.class_index == x["class"] and v > best:
                best, bj = v, j
        if best >= 0.5:
            tp += 1; used.add(bj); ious.append(best)
        else:
            fp += 1
print("held-out frames %d: ground-truth boxes %d, detections %d, correct (class + IoU >= 0.5) %d -> recall %.2f precision %.2f, mean IoU of the hits %.2f" % (
    len(data["validation_set"]), ngt, tp + fp, tp, tp / max(ngt, 1), tp / max(tp + fp, 1), float(np.mean(ious)) if ious else 0.0))
print("anchors passing p > 0.95: %d, candidates after NMS: %d, cnet class histogram of the candidates (1..3 objects, 4 background): %s" % (nscan, ncand, predhist[1:5].tolist()))

# the evaluation loop proper (frcnn_amd/evaluation.py): validation losses and VOC-style mAP over nextValidation
nval = len(data["validation_set"])
vl = F.validation_losses(model, it, nval)
print("validation losses over %d held-out frames (evaluate mode): pcls %.4f preg %.4f dcls %.4f dreg %.4f (%d examples, %d positive)" % (
    vl["images"], vl["pcls"], vl["preg"], vl["dcls"], vl["dreg"], vl["examples"], vl["positives"]))
ev = F.evaluate_detections(det, it, nval)
print("mAP@0.5 over %d frames: %.3f (per class %s; %d detections, %d ground-truth boxes, tp %d fp %d)" % (
    ev["images"], ev["mAP"], dict((k, round(v, 3)) for k, v in ev["ap"].items()), ev["detections"], ev["ground_truth"], ev["tp"], ev["fp"]))
