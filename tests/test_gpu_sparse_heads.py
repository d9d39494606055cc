"""Option "sparse_heads" (include/frcnn_hip.h, csrc/heads.hip): a training pass computes the anchor nets
(models/model_utilities.lua:31-34) at the sampled anchors only -- objective.lua:91-140 reads their outputs nowhere else and
delta_outputs[1..4] are zero elsewhere -- as one chain of grouped launches.  Same loss, same gradient (to the rounding of an
fp32 product against a split-operand one) as the dense convolutions; more positions than the sparse path takes fall back to
them inside the same pass.  The oracle-backed parity tests run with the option on, its default."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _One(object):
    def __init__(self, batch):
        self.batch = batch

    def nextTraining(self, count=None):
        return self.batch


def _step(F, sparse, H, W, negatives):
    import torch
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=5)
    anchors = F.Anchors(model["pnet"], cfg["scales"])
    rois = F.synthetic_rois(cfg, W, H, 4, 7, 2)
    pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, F.MT19937(23), negatives=negatives)
    sizes = F.output_map_sizes(model, H, W)
    pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
    batch = [dict(img=F.synthetic_image(H, W, 3), positive=pos, negative=neg)]
    f = F.create_objective(model, w, g, _One(batch), dict(pcls=[], preg=[], dcls=[], dreg=[]))
    E = len(pos) + len(neg)
    rng = np.random.RandomState(1)
    F._lib.call("frcnn_set_option", b"sparse_heads", 1 if sparse else 0)
    try:
        model["pnet"].drop_masks = [None if l["dropout"] <= 0 else (rng.rand(l["filters"]) > l["dropout"]).astype(np.float32)
                                    for l in model["layers"]]
        model["cnet"].drop_masks = [(rng.rand(E, l["n"]) > 0.5).astype(np.float32) for l in model["class_layers"]]
        loss, grad = f(w)
        torch.cuda.synchronize()
        per_head = [sum(1 for e in pos + neg if e[0].layer == l + 1) for l in range(4)]
        return loss, grad.cpu().numpy().copy(), model, per_head
    finally:
        F._lib.call("frcnn_set_option", b"sparse_heads", 1)
        model["pnet"].drop_masks = None
        model["cnet"].drop_masks = None


@pytest.mark.parametrize("size,negatives", [((225, 400), 64), ((450, 800), 256), ((450, 800), 3000)])
def test_sparse_anchor_nets_equal_the_dense_convolutions(F, size, negatives):
    H, W = size
    la, ga, model, per_head = _step(F, True, H, W, negatives)
    lb, gb, _, _ = _step(F, False, H, W, negatives)
    if negatives >= 3000:
        assert max(per_head) > 512, per_head      # (this case is the fall-back to the dense convolutions inside a deferred pass)
    else:
        assert 0 < max(per_head) <= 512, per_head
    assert abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    nat = model["native"]
    for off, cnt, kind, aux in nat.param_table:
        a, b = ga[off:off + cnt].astype(np.float64), gb[off:off + cnt].astype(np.float64)
        if np.linalg.norm(b) < 1e-4:
            continue
        assert np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(b), (off, cnt, kind, np.linalg.norm(a - b) / np.linalg.norm(b))
    lo, hi = model["pnet"].heads_param_range()
    assert np.abs(ga[lo:hi]).max() > 0


def _batch_step(F, sparse, compact):
    """Two images in one step (the gradient accumulates over them, objective.lua:64-198), the second without any example: its
    anchor nets are never launched in the sparse mode, its backward pass finds nothing to do for them."""
    import torch
    H, W = 225, 400
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=6)
    anchors = F.Anchors(model["pnet"], cfg["scales"])
    rois = F.synthetic_rois(cfg, W, H, 3, 7, 5)
    pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, F.MT19937(29), negatives=48)
    sizes = F.output_map_sizes(model, H, W)
    pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
    batch = [dict(img=F.synthetic_image(H, W, 7), positive=pos, negative=neg),
             dict(img=F.synthetic_image(H, W, 8), positive=[], negative=[])]
    f = F.create_objective(model, w, g, _One(batch), dict(pcls=[], preg=[], dcls=[], dreg=[]))
    E = len(pos) + len(neg)
    rng = np.random.RandomState(2)
    F._lib.call("frcnn_set_option", b"sparse_heads", 1 if sparse else 0)
    F._lib.call("frcnn_set_option", b"drop_compact", 1 if compact else 0)
    try:
        model["pnet"].drop_masks = [None if l["dropout"] <= 0 else (rng.rand(l["filters"]) > l["dropout"]).astype(np.float32)
                                    for l in model["layers"]]
        model["cnet"].drop_masks = [(rng.rand(E, l["n"]) > 0.5).astype(np.float32) for l in model["class_layers"]]
        loss, grad = f(w)
        torch.cuda.synchronize()
        return loss, grad.cpu().numpy().copy(), model
    finally:
        F._lib.call("frcnn_set_option", b"sparse_heads", 1)
        F._lib.call("frcnn_set_option", b"drop_compact", 1)
        model["pnet"].drop_masks = None
        model["cnet"].drop_masks = None


def test_two_images_one_of_them_without_examples(F):
    la, ga, model = _batch_step(F, True, False)
    lb, gb, _ = _batch_step(F, False, False)
    lc, gc, _ = _batch_step(F, True, True)
    assert np.isfinite(la) and abs(la - lb) <= 1e-6 * abs(lb) and abs(lc - lb) <= 1e-6 * abs(lb)
    nb = np.linalg.norm(gb.astype(np.float64))
    assert np.linalg.norm(ga.astype(np.float64) - gb) <= 2e-5 * nb          # sparse against dense anchor nets: rounding
    assert np.linalg.norm(gc.astype(np.float64) - gb) <= 1e-3 * nb          # ... and the dropped channels left out (decision flips)
    lo, hi = model["pnet"].heads_param_range()
    assert np.abs(ga[lo:hi]).max() > 0
