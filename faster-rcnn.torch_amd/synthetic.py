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

    def __init__(self, model, H=450, W=800, images_per_batch=1, rank=0, world_size=1, pool=4, device_images=True,
                 upload=False):
        """device_images: the frames live in HBM (the bench contract: inputs resident when the timed region starts).
        upload: the frames live in page-locked host memory and every nextTraining() brings its frame over PCIe -- the
        `x.img:cuda()` of objective.lua:66 -- asynchronously: a copy stream fills a ring of three device buffers one step
        ahead (a buffer is rewritten only after the step that read it has run), the consumer's stream waits for the
        copy's event."""
        self.cfg = model["cfg"]
        self.upload = bool(upload)
        self._ring = None
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
            if self.upload:
                import torch
                img = torch.from_numpy(img).pin_memory()
            elif device_images:
                from .tensor import DeviceTensor
                img = DeviceTensor.from_numpy(img)
            self.pool.append(dict(img=img, positive=pos, negative=neg, rois=rois))

    def _begin_uploads(self, frames_per_call):
        """Once per nextTraining() call in upload mode.  Call c is made when step c-1 is fully queued; the buffers it fills
        were read by step c-3, i.e. by work queued before the mark recorded at call c-2."""
        import torch
        if self._ring is None:
            proto = self.pool[0]["img"]
            self._ring = dict(bufs=[torch.empty_like(proto, device="cuda") for _ in range(3 * frames_per_call)], k=0,
                              stream=torch.cuda.Stream(), marks=[])
        r = self._ring
        mark = torch.cuda.Event(); mark.record(torch.cuda.current_stream())
        if len(r["marks"]) >= 2:
            r["stream"].wait_event(r["marks"][-2])
        r["marks"] = (r["marks"] + [mark])[-2:]

    def _uploaded(self, x):
        """x with img replaced by a device buffer that the copy queued here will have filled when the consumer's stream
        gets to it (objective.lua:66 `x.img:cuda()`, asynchronous)."""
        import torch
        r = self._ring
        k = r["k"]; r["k"] = (k + 1) % len(r["bufs"])
        with torch.cuda.stream(r["stream"]):
            r["bufs"][k].copy_(x["img"], non_blocking=True)
            done = torch.cuda.Event(); done.record(r["stream"])
        torch.cuda.current_stream().wait_event(done)
        y = dict(x); y["img"] = r["bufs"][k]
        return y

    def nextTraining(self, count=None):
        """images_per_batch images (benchmarks: 1); with images_per_batch=None the reference's
        rule (BatchIterator.lua:166-268): keep adding images until they carry `count` (default cfg.batch_size)
        examples in total."""
        batch = []
        if self.images_per_batch is None:
            assert not self.upload, "upload mode draws a fixed number of frames per call"
            count = count or self.cfg["batch_size"]
            while count > 0:
                x = self.pool[self.i % len(self.pool)]
                self.i += 1
                batch.append(x)
                count -= max(1, len(x["positive"]) + len(x["negative"]))   # (an empty image still advances the loop)
            return batch
        if self.upload:
            self._begin_uploads(self.images_per_batch)
        for _ in range(self.images_per_batch):
            x = self.pool[self.i % len(self.pool)]
            batch.append(self._uploaded(x) if self.upload else x)
            self.i += 1
        return batch
