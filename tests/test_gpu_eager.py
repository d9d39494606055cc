"""The update beside the backward pass (include/frcnn_hip.h, "the update beside the backward pass"; main.lua:133's
optim.rmsprop applied slice by slice while objective.lua:189's pnet:backward is still running) leaves the SAME BITS as
the whole-vector update after the pass: per element the arithmetic is one and the same inline function, the packs the next
forward reads are made from the same weights, and the gradient vector ends up holding the same scaled gradient."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_slice_updates_tile_the_whole_vector_bit_for_bit(F):
    """frcnn_scale_rmsprop_slice over a partition with ragged bounds (inside 16-byte groups, empty, shorter than a group)
    == frcnn_scale_rmsprop over the whole vector: x, m and the scaled g, bit for bit; gscale = 1 leaves g alone."""
    import torch
    n = 1_000_003
    g0 = torch.randn(n, device="cuda") * 3.0
    x0 = torch.randn(n, device="cuda")
    m0 = torch.rand(n, device="cuda")
    s = F.stream_ptr()
    for gscale in (1.0 / 41.0, 1.0):
        a = [t.clone() for t in (x0, g0, m0)]
        b = [t.clone() for t in (x0, g0, m0)]
        F._lib.call("frcnn_scale_rmsprop", F.ptr(a[0]), F.ptr(a[1]), gscale, F.ptr(a[2]), n, 1e-4, 0.9, 1e-8, s)
        cuts = [0, 1, 2, 2, 7, 9, 10, 12, 4099, 4100, 300001, 300006, 999999, n]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            F._lib.call("frcnn_scale_rmsprop_slice", F.ptr(b[0]), F.ptr(b[1]), gscale, F.ptr(b[2]), lo, hi, 1e-4, 0.9, 1e-8, s)
        torch.cuda.synchronize()
        for u, v, name in zip(a, b, ("x", "g", "m")):
            assert torch.equal(u, v), "%s differs (gscale %r)" % (name, gscale)
        assert not torch.equal(a[0], x0)
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_scale_rmsprop_slice", F.ptr(x0), F.ptr(g0), 1.0, F.ptr(m0), 5, 4, 1e-4, 0.9, 1e-8, s)


def _train(F, eager, steps, H, W, det):
    import torch
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=11)
    it = F.SyntheticBatchIterator(model, H=H, W=W, pool=2)
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    f = F.create_objective(model, w, g, it, stats)
    st = dict(learningRate=1e-4, alpha=0.9, eager=eager)
    rng = np.random.RandomState(5)
    nat = model["native"]
    F._lib.call("frcnn_set_option", b"deterministic", det)
    out = []
    try:
        for k in range(steps):
            # explicit dropout masks: the two runs must draw the same ones
            model["pnet"].drop_masks = [None] + [(rng.rand(c) > 0.4).astype(np.float32) for c in (128, 256, 384)]
            E = len(F.clean_examples(it.pool[k % 2]["positive"], F.output_map_sizes(model, H, W))) + \
                len(F.clean_examples(it.pool[k % 2]["negative"], F.output_map_sizes(model, H, W)))
            model["cnet"].drop_masks = [(rng.rand(E, 1024) > 0.5).astype(np.float32), (rng.rand(E, 512) > 0.5).astype(np.float32)]
            _, fx = F.rmsprop(f, w, st)
            torch.cuda.synchronize()
            out.append((fx[0], w.cpu().numpy().copy(), g.cpu().numpy().copy(), st["m"].cpu().numpy().copy()))
    finally:
        F._lib.call("frcnn_set_option", b"deterministic", 0)
        model["pnet"].drop_masks = None
        model["cnet"].drop_masks = None
    return out, nat


@pytest.mark.parametrize("size", [(225, 400)])
def test_update_beside_the_backward_pass_equals_the_update_after_it(F, size):
    """Three optimiser steps (the second and third run on packs renewed beside the previous pass), deterministic mode (no
    fp32 atomics: the gradient itself is bit-reproducible): losses, weights, RMSprop state and the scaled gradient of every step
    are identical whether the step is applied slice by slice beside the backward pass or as one pass after it."""
    H, W = size
    a, _ = _train(F, True, 3, H, W, 1)
    b, _ = _train(F, False, 3, H, W, 1)
    for k, (ra, rb) in enumerate(zip(a, b)):
        assert ra[0] == rb[0], "loss of step %d: %r vs %r" % (k, ra[0], rb[0])
        for name, u, v in zip(("weights", "gradient", "rmsprop state"), ra[1:], rb[1:]):
            assert np.array_equal(u, v), "%s differ after step %d (%d elements)" % (name, k, int((u != v).sum()))
    assert not np.array_equal(a[0][1], a[2][1])


def test_default_mode_agrees_to_rounding_and_a_foreign_write_is_noticed(F):
    """Default mode (atomics: last-bit differences between any two runs): the two ways agree like two runs of one way do.  And
    a write to the weights that does not come from the optimiser (torch counts it) makes the next pass re-pack: the step after
    weights.copy_(...) equals the same step taken by a fresh objective."""
    import torch
    H, W = 225, 400
    a, _ = _train(F, True, 2, H, W, 0)
    b, _ = _train(F, False, 2, H, W, 0)
    # (step 0's gradient precedes every update; step 1 starts from weights that differ in their last bits, and a 2x2 pooling
    # winner or a PReLU branch that flips carries that into the 1e-5s -- a pack made from stale weights would show at 1e-2)
    for k, (ra, rb) in enumerate(zip(a, b)):
        assert abs(ra[0] - rb[0]) <= (1e-6, 1e-5)[k] * abs(rb[0])
        assert np.linalg.norm(ra[2].astype(np.float64) - rb[2]) <= (1e-5, 3e-4)[k] * np.linalg.norm(rb[2].astype(np.float64))
    # foreign write
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=11)
    it = F.SyntheticBatchIterator(model, H=H, W=W, pool=1)
    f = F.create_objective(model, w, g, it, dict(pcls=[], preg=[], dcls=[], dreg=[]))
    st = dict(learningRate=1e-4, alpha=0.9)
    F._lib.call("frcnn_set_option", b"deterministic", 1)
    try:
        rng = np.random.RandomState(9)
        E = len(F.clean_examples(it.pool[0]["positive"], F.output_map_sizes(model, H, W))) + \
            len(F.clean_examples(it.pool[0]["negative"], F.output_map_sizes(model, H, W)))
        model["pnet"].drop_masks = [None] + [(rng.rand(c) > 0.4).astype(np.float32) for c in (128, 256, 384)]
        model["cnet"].drop_masks = [(rng.rand(E, 1024) > 0.5).astype(np.float32), (rng.rand(E, 512) > 0.5).astype(np.float32)]
        w_start = w.clone()
        bn0 = model["native"].bn_running.clone()
        F.rmsprop(f, w, st)                       # packs now belong to the updated weights ...
        w.copy_(w_start * 1.5)                    # ... which somebody overwrites
        model["native"].bn_running.copy_(bn0)
        loss_a, _ = f(w)
        ga = g.cpu().numpy().copy()
        # reference: the same weights through an objective that has no promise to rely on
        f2 = F.create_objective(model, w, g, it, dict(pcls=[], preg=[], dcls=[], dreg=[]))
        F._lib.call("frcnn_pnet_invalidate_packs", model["native"].h)
        model["native"].bn_running.copy_(bn0)
        loss_b, _ = f2(w)
        gb = g.cpu().numpy().copy()
        assert loss_a == loss_b and np.array_equal(ga, gb)
    finally:
        F._lib.call("frcnn_set_option", b"deterministic", 0)
        model["pnet"].drop_masks = None
        model["cnet"].drop_masks = None
