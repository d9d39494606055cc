"""Data-parallel exchange step on CPU (gloo, world_size 2): per-rank gradients and the 8 fp64
accumulators are summed by allreduce_gradient_and_stats(); the normalised result must equal the
single-process result on the concatenated batch (objective.lua:49,65,189,200).  The per-rank
gradients come from the CPU oracle (no GPU in this test); what is tested is the product's exchange
code path and its sharding rule."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_inputs(k):
    """Deterministic examples for image k on the tiny model."""
    import pyoracle as O
    from util import TINY_CLS, TINY_HEADS, TINY_LAYERS
    cfg = dict(class_count=5, scales=[32, 64, 128, 256], roi_pooling=dict(kw=2, kh=2))
    m = O.make_model(TINY_LAYERS, TINY_HEADS, TINY_CLS, cfg)
    n, pn = O.param_count(m)
    w = (np.random.RandomState(1).randn(n) * 0.1).astype(np.float32)
    H, W = 128, 160
    img = np.random.RandomState(1000 + k).randn(3, H, W).astype(np.float32)
    A = O.Anchors(m)
    rois = np.array([[20 + 10 * k, 30, 90 + 10 * k, 100], [60, 40 + 5 * k, 150, 110]], dtype=np.float64)
    pidx, prect = A.find_positive(rois, (0, 0, W, H), 0.5, 0.25, True)
    nidx, nrect = A.sample_negative((0, 0, W, H), rois, 0.25, 6, O.MT(7 + k))
    R = len(pidx) + len(nidx)
    rs = np.random.RandomState(50 + k)
    pm = [None, (rs.rand(12) > 0.4).astype(np.float32), (rs.rand(16) > 0.4).astype(np.float32), (rs.rand(20) > 0.4).astype(np.float32)]
    cm = [(rs.rand(R, 48) > 0.5).astype(np.float32), (rs.rand(R, 32) > 0.5).astype(np.float32)]
    return m, w, img, pidx, prect, rois, np.array([1, 3], np.int32), nidx, nrect, pm, cm


def _local(k, grad, acc, bn):
    import pyoracle as O
    m, w, img, pidx, prect, rois, rcls, nidx, nrect, pm, cm = _rank_inputs(k)
    O.train_image(m, w, grad, img, pidx, prect, rois, rcls, nidx, nrect, pm, cm, bn, acc)
    return w.size


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import frcnn_amd as F
    import pyoracle as O
    m = _rank_inputs(0)[0]
    n, _ = O.param_count(m)
    grad = np.zeros(n, np.float32); acc = np.zeros(8)
    bn = np.concatenate([np.zeros(48, np.float32), np.ones(48, np.float32)])
    for k in range(rank, 4, world):          # rank r takes images r, r+W, ...
        _local(k, grad, acc, bn)
    g = torch.from_numpy(grad)
    # the bucket that is final early (the cnet slice in lossAndGradient) starts first, the rest + the 8
    # accumulators follow in allreduce_gradient_and_stats
    pending = [F.allreduce_begin(g, n // 3, n - 5)]
    assert pending[0] is not None
    tot = F.allreduce_gradient_and_stats(g, acc, pending)
    g /= tot[2]
    np.save(os.path.join(out_dir, "g%d.npy" % rank), g.numpy())
    np.save(os.path.join(out_dir, "t%d.npy" % rank), tot)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp_two_ranks_equals_single_process(tmp_path, O):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    t0, t1 = np.load(tmp_path / "t0.npy"), np.load(tmp_path / "t1.npy")
    assert np.array_equal(g0, g1) and np.array_equal(t0, t1)       # every rank ends with the same result
    m = _rank_inputs(0)[0]
    n, _ = O.param_count(m)
    grad = np.zeros(n, np.float32); acc = np.zeros(8)
    for k in range(4):                                             # single process, concatenated batch
        _local(k, grad, acc, np.concatenate([np.zeros(48, np.float32), np.ones(48, np.float32)]))
    grad /= acc[2]
    assert np.allclose(t0, acc, rtol=1e-12, atol=0)
    err = np.linalg.norm(g0 - grad) / np.linalg.norm(grad)
    assert err < 1e-5, err                                          # summation order differs
    assert t0[2] > 0 and t0[7] == 4


def test_no_process_group_is_a_noop(F):
    import torch
    g = torch.ones(10)
    tot = F.allreduce_gradient_and_stats(g, np.arange(8.0))
    assert tot.tolist() == list(np.arange(8.0)) and g.sum().item() == 10
