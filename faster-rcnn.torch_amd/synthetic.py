"""Synthetic stand-in for BatchIterator (BatchIterator.lua is a file-IO data loader and out of
scope): produces batches with the structure objective.lua:64-69 consumes --
{img, positive, negative} -- from seeded synthetic frames and ground-truth boxes, running the same
example assembly as BatchIterator.lua:198-225 (findPositive, 16 sampled negatives, nearby_aversion)
through the host-side Anchors mirror.  Inputs follow SURVEY 8d: image iid N(0,1) with seed
1000+index; 4 boxes per image drawn from the config's scales x {1:1, 2:1, 1:2} with +-20% jitter,
fully inside the image, class uniform in 1..class_count, seed 7."""
import math

import numpy as np

from .Anchors import Anchors, MT19937
from .Rect import Rect


class Roi(object):
    def __init__(self, rect, class_index):
        self.rect = rect
        self.class_index = class_index


def synthetic_rois(cfg, W, H, count=4, seed=7, index=0):
    rng = np.random.RandomState(seed + 7919 * index)
    rois = []
    scales = [s for s in cfg["scales"] if s * 1.2 * math.sqrt(2) < min(W, H)] or [min(cfg["scales"])]
    for _ in range(count):
        s = scales[rng.randint(len(scales))]
        a = s / math.sqrt(2)
        bw, bh = [(s, s), (2 * a, a), (a, 2 * a)][rng.randint(3)]
        bw *= rng.uniform(0.8, 1.2); bh *= rng.uniform(0.8, 1.2)
        bw = min(bw, W - 2); bh = min(bh, H - 2)
        x = rng.uniform(0, W - bw); y = rng.uniform(0, H - bh)
        r = Rect(math.floor(x), math.floor(y), math.floor(x + bw), math.floor(y + bh))
        rois.append(Roi(r, int(rng.randint(1, cfg["class_count"] + 1))))
    return rois


def synthetic_image(H, W, index=0):
    return np.random.RandomState(1000 + index).randn(3, H, W).astype(np.float32)


def assemble_examples_native(anchors, cfg, rois, W, H, rng, negatives=16):
    """BatchIterator.lua:198-225 for one image through frcnn_anchors_assemble (host-side native code, the same lists as
    the Python path below draw for draw)."""
    import ctypes as C
    from . import _lib
    nroi = len(rois)
    ra = np.array([(r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY) for r in rois], dtype=np.float64).reshape(-1, 4)
    cap = 8192
    ex = np.empty((cap, 5), dtype=np.int32); er = np.empty((cap, 4), dtype=np.float64)
    npos, nneg = C.c_int(0), C.c_int(0)
    _lib.call("frcnn_anchors_assemble", anchors.native(), ra.ctypes.data_as(C.c_void_p), nroi, float(W), float(H),
              float(cfg["positive_threshold"]), float(cfg["negative_threshold"]), int(bool(cfg["best_match"])),
              int(bool(cfg.get("nearby_aversion"))), int(negatives), rng.state.ctypes.data_as(C.c_void_p), C.byref(rng.cidx),
              ex.ctypes.data_as(C.c_void_p), er.ctypes.data_as(C.c_void_p), cap, C.byref(npos), C.byref(nneg))
    tag = Anchors._tag
    exl, erl = ex[:npos.value + nneg.value].tolist(), er[:npos.value + nneg.value].tolist()
    positive = [(tag(Rect(*erl[k]), exl[k][0], exl[k][1], exl[k][2], exl[k][3]), rois[exl[k][4] - 1]) for k in range(npos.value)]
    negative = [(tag(Rect(*erl[k]), exl[k][0], exl[k][1], exl[k][2], exl[k][3]),) for k in range(npos.value, npos.value + nneg.value)]
    return positive, negative


def assemble_examples(anchors, cfg, rois, W, H, rng, negatives=16, native=None):
    """BatchIterator.lua:198-225 for one image.  native=None: the native twin unless FRCNN_NATIVE_ASSEMBLE=0."""
    import os
    if native is None:
        native = os.environ.get("FRCNN_NATIVE_ASSEMBLE", "1") != "0"
    if native:
        return assemble_examples_native(anchors, cfg, rois, W, H, rng, negatives)
    img_rect = Rect(0, 0, W, H)
    positive = anchors.findPositive(rois, img_rect, cfg["positive_threshold"], cfg["negative_threshold"], cfg["best_match"])
    negative = anchors.sampleNegative(img_rect, rois, cfg["negative_threshold"], negatives, rng)
    count = len(positive) + len(negative)
    if cfg.get("nearby_aversion"):
        # every anchor that shares a bin pair with a positive's centre and overlaps it by less than the negative threshold
        # (BatchIterator.lua:204-216).  Vectorised: the candidates of all positives in one IoU evaluation (Rect.IoU's
        # arithmetic in float64, candidates in the order of the nested Lua loops); Rect objects are only built for the
        # few candidates that survive the shuffle.
        nearby = []
        if positive:
            parts = [anchors.findNearbyArrays(*p[0].center()) for p in positive]
            cnt = np.array([len(m) for m, _ in parts])
            if cnt.sum():
                M = np.concatenate([m for m, _ in parts]); R = np.concatenate([r for _, r in parts])
                P = np.repeat(np.array([(p[0].minX, p[0].minY, p[0].maxX, p[0].maxY) for p in positive], dtype=np.float64), cnt, axis=0)
                minx = np.maximum(P[:, 0], R[:, 0]); miny = np.maximum(P[:, 1], R[:, 1])
                maxx = np.minimum(P[:, 2], R[:, 2]); maxy = np.minimum(P[:, 3], R[:, 3])
                ok = (maxx >= minx) & (maxy >= miny)
                inter = np.where(ok, (maxx - minx) * (maxy - miny), 0.0)
                area = lambda A: (A[:, 2] - A[:, 0]) * (A[:, 3] - A[:, 1])
                with np.errstate(divide="ignore", invalid="ignore"):
                    iou = inter / (area(P) + area(R) - inter)
                Mk = M[iou < cfg["negative_threshold"]]
                nearby = list(range(len(Mk)))   # (the shuffle permutes positions; the rows are looked up afterwards)
        c = min(len(positive), count)
        c = min(c, len(nearby))
        # shuffle_n (utilities.lua:31-42) with the MT19937 stream instead of LuaJIT's math.random
        r = len(nearby)
        for i in range(c):
            j = rng.random() % r + i
            nearby[i], nearby[j] = nearby[j], nearby[i]
            r -= 1
        negative.extend((anchors.get(*(int(v) for v in Mk[t])),) for t in nearby[:c])
    return positive, negative


def output_map_sizes(model, H, W):
    """(h, w) of the pnet outputs 1..nheads for an HxW input: ceil-mode 2x2 pooling per block
    (model_utilities.lua:23) then the valid kxk head convolution (:31)."""
    sizes = []
    h, w = H, W
    per_block = []
    for l in model["layers"]:
        for _ in range(l["conv_steps"]):
            h = h + 2 * l["padH"] - l["kH"] + 1; w = w + 2 * l["padW"] - l["kW"] + 1
        h = int(math.ceil((h - 2) / 2.0)) + 1; w = int(math.ceil((w - 2) / 2.0)) + 1
        per_block.append((h, w))
    for a in model["anchor_nets"]:
        bh, bw = per_block[a["input"] - 1]
        sizes.append((bh - a["kW"] + 1, bw - a["kW"] + 1))
    return sizes


def clean_examples(examples, sizes):
    """cleanAnchors (objective.lua:32-43) as a pure function: drops examples whose anchor index lies
    outside its head map."""
    return [e for e in examples if not (e[0].index[1] > sizes[e[0].layer - 1][0] or e[0].index[2] > sizes[e[0].layer - 1][1])]


class SyntheticBatchIterator(object):
    """nextTraining() -> list of {img, positive, negative}.  `images_per_batch` images per call; with
    data parallelism rank r of world_size W takes images r, r+W, ... of the step's list."""

    def __init__(self, model, H=450, W=800, images_per_batch=1, rank=0, world_size=1, pool=4, device_images=True):
        self.cfg = model["cfg"]
        self.anchors = Anchors(model["pnet"], self.cfg["scales"])
        self.H, self.W = H, W
        self.images_per_batch = images_per_batch
        self.rank, self.world_size = rank, world_size
        self.i = 0
        self.pool = []
        rng = MT19937(7)
        for k in range(pool):
            idx = rank * pool + k
            rois = synthetic_rois(self.cfg, W, H, 4, 7, idx)
            pos, neg = assemble_examples(self.anchors, self.cfg, rois, W, H, rng)
            img = synthetic_image(H, W, idx)
            if device_images:
                from .tensor import DeviceTensor
                img = DeviceTensor.from_numpy(img)
            self.pool.append(dict(img=img, positive=pos, negative=neg, rois=rois))

    def nextTraining(self, count=None):
        """images_per_batch images (benchmarks: 1); with images_per_batch=None the reference's
        rule (BatchIterator.lua:166-268): keep adding images until they carry `count` (default cfg.batch_size)
        examples in total."""
        batch = []
        if self.images_per_batch is None:
            count = count or self.cfg["batch_size"]
            while count > 0:
                x = self.pool[self.i % len(self.pool)]
                self.i += 1
                batch.append(x)
                count -= max(1, len(x["positive"]) + len(x["negative"]))   # (an empty image still advances the loop)
            return batch
        for _ in range(self.images_per_batch):
            batch.append(self.pool[self.i % len(self.pool)])
            self.i += 1
        return batch
