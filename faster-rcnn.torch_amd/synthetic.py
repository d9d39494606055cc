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
from .BatchIterator import assemble_examples, assemble_examples_native  # (BatchIterator.lua:198-225 lives there)
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
