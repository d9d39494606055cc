"""Data-parallel training step on real kernels: two processes share the one GPU of the box (gloo moves the CUDA
buffers through the host, the code path -- bucketed asynchronous all-reduce beside the backward pass, folded
gradient:div, RMSprop -- is the one RCCL runs under).  Rank r takes image r; the result must equal the single-process
step on the two-image batch (objective.lua:49,65,189,200)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 128, 176


def _setup():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frcnn_amd as F
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    anchors = F.Anchors(model["pnet"], cfg["scales"])
    sizes = F.output_map_sizes(model, H, W)
    images = []
    mt = F.MT19937(7)
    for k in range(2):
        rois = F.synthetic_rois(cfg, W, H, 3, 7, k)
        pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, mt, negatives=8)
        pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
        images.append(dict(img=F.synthetic_image(H, W, k), positive=pos, negative=neg))
    rng = np.random.RandomState(3)
    pm = [None if l["dropout"] <= 0 else (rng.rand(l["filters"]) > l["dropout"]).astype(np.float32) for l in model["layers"]]
    cms = []
    for x in images:
        R = len(x["positive"]) + len(x["negative"])
        cms.append([(rng.rand(R, 1024) > 0.5).astype(np.float32), (rng.rand(R, 512) > 0.5).astype(np.float32)])
    return F, model, weights, gradient, anchors, images, pm, cms


class _Batch(object):
    def __init__(self, batch):
        self.batch = batch

    def nextTraining(self, count=None):
        return self.batch


def _step(F, model, weights, gradient, batch, pm, cms):
    """one F.rmsprop step with explicit dropout masks (one cnet mask set per image, in order)"""
    model["pnet"].drop_masks = pm
    cnet = model["cnet"]
    orig = cnet.forward
    it = iter(cms)

    def fwd(x):
        cnet.drop_masks = next(it)
        return orig(x)
    cnet.forward = fwd
    try:
        stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
        f = F.create_objective(model, weights, gradient, _Batch(batch), stats)
        F.rmsprop(f, weights, dict(learningRate=1e-4, alpha=0.9))
    finally:
        cnet.forward = orig
        cnet.drop_masks = None
        model["pnet"].drop_masks = None
    return [stats[k][-1] for k in ("pcls", "preg", "dcls", "dreg")]


def _worker(rank, world, port, out_dir, empty_rank1=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F, model, weights, gradient, anchors, images, pm, cms = _setup()
    if empty_rank1:   # rank 1's image carries no example: its collectives must still match rank 0's
        images[1] = dict(img=images[1]["img"], positive=[], negative=[])
    st = _step(F, model, weights, gradient, [images[rank]], pm, [cms[rank]])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "g%d.npy" % rank), gradient.cpu().numpy())
    np.save(os.path.join(out_dir, "w%d.npy" % rank), weights.cpu().numpy())
    np.save(os.path.join(out_dir, "s%d.npy" % rank), np.array(st))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert np.array_equal(g0, g1) and np.array_equal(w0, w1) and np.array_equal(s0, s1)   # replicas stay identical
    F, model, weights, gradient, anchors, images, pm, cms = _setup()
    w_init = weights.cpu().numpy().copy()
    st = _step(F, model, weights, gradient, images, pm, cms)
    g = gradient.cpu().numpy(); w = weights.cpu().numpy()
    assert np.allclose(s0, st, rtol=1e-6, atol=0)
    assert np.linalg.norm(g0 - g) <= 1e-5 * np.linalg.norm(g)
    assert np.abs(w - w_init).max() > 0
    # (first RMSprop step ~ lr * sign(g) / sqrt(0.1): compare against the size of the update, see test_gpu_model)
    assert np.linalg.norm(w0 - w) <= 1e-3 * np.linalg.norm(w - w_init)


@pytest.mark.timeout(900)
def test_rank_without_examples_issues_the_same_collectives(tmp_path):
    """One rank's image has neither positives nor negatives (objective.lua loops over zero examples): the bucketed
    exchange must not depend on that (a missing collective on one rank would hang RCCL), and the result equals the
    single-process step on {image 0, empty image}."""
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)
    F, model, weights, gradient, anchors, images, pm, cms = _setup()
    images[1] = dict(img=images[1]["img"], positive=[], negative=[])
    _step(F, model, weights, gradient, images, pm, [cms[0]])
    g = gradient.cpu().numpy()
    assert np.isfinite(g0).all() and np.abs(g0).max() > 0
    assert np.linalg.norm(g0 - g) <= 1e-5 * np.linalg.norm(g)
