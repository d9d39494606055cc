"""The end-to-end golden fixtures of tests/golden/make_golden_e2e.py (an independent PyTorch-autograd restatement of
objective.lua:45-218 and Detector.lua:17-141, generated in the authoring container) replayed through the HIP path: no
oracle in this comparison at all -- the expected numbers are data."""
import os

import numpy as np
import pytest

from util import TINY_CLS, TINY_HEADS, TINY_LAYERS, assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(class_count=5, scales=[32, 64, 128, 256], roi_pooling=dict(kw=2, kh=2))


class _OneBatch(object):
    def __init__(self, batch, anchors):
        self.batch, self.anchors = batch, anchors

    def nextTraining(self, count=None):
        return self.batch


def _model(F, weights_host):
    import torch
    model = F.create_model(dict(CFG), TINY_LAYERS, TINY_HEADS, TINY_CLS)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=1)
    assert model["native"].total_params == weights_host.size == 64079
    weights.copy_(torch.from_numpy(weights_host))
    return model, weights, gradient


def test_train_fixture_through_the_hip_path(F):
    """lossAndGradient on the fixture's two-image batch: the four statistics 1e-5, the flat gradient 1e-3 relative L2 per
    tensor and 1e-4 elementwise, the five pnet outputs of image 0 at 1e-4."""
    g = np.load(os.path.join(GOLD, "train_tiny.npz"))
    model, weights, gradient = _model(F, g["weights"])
    nat = model["native"]
    anchors = F.Anchors(model["pnet"], CFG["scales"])
    batch, cms = [], []
    for k in range(int(g["n_images"])):
        rois = [F.Roi(F.Rect(*[float(v) for v in r]), int(c)) for r, c in zip(g["rois%d" % k], g["roi_class%d" % k])]
        pos = [(anchors.get(int(l), int(a), int(y), int(x)), rois[int(ri) - 1]) for l, a, y, x, ri in g["pos%d" % k]]
        neg = [(anchors.get(int(l), int(a), int(y), int(x)),) for l, a, y, x in g["neg%d" % k]]
        # the anchors of the host mirror ARE the fixture's (naive restatement) anchor rects
        for (an, _), want in zip(pos, g["pos_rect%d" % k]):
            assert [an.minX, an.minY, an.maxX, an.maxY] == want.tolist()
        batch.append(dict(img=g["img%d" % k], positive=pos, negative=neg))
        cms.append([g["cmask%d_0" % k], g["cmask%d_1" % k]])
    # the fixture draws one SpatialDropout mask set per image; the device takes one per call: one objective call per image,
    # gradients and statistics combined exactly as objective.lua:194-205 combines them over a batch
    cnet = model["cnet"]
    orig_forward = cnet.forward
    total = np.zeros(nat.total_params, np.float64)
    acc = dict(cls=0, reg=0, imgs=0, pcls=0.0, preg=0.0, dcls=0.0, dreg=0.0)
    outs0 = None
    try:
        for k, b in enumerate(batch):
            model["pnet"].drop_masks = [None] + [g["pmask%d_%d" % (k, j)] for j in (1, 2, 3)]

            def fwd(x, k=k):
                cnet.drop_masks = cms[k]
                return orig_forward(x)
            cnet.forward = fwd
            stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
            f = F.create_objective(model, weights, gradient, _OneBatch([b], anchors), stats)
            loss, grad = f(weights)
            n_p, n_n = len(b["positive"]), len(b["negative"])
            total += grad.cpu().numpy().astype(np.float64) * (n_p + n_n)       # undo this call's gradient:div(cls_count)
            acc["pcls"] += stats["pcls"][-1] * (n_p + n_n); acc["preg"] += stats["preg"][-1] * n_p
            acc["dcls"] += stats["dcls"][-1]; acc["dreg"] += stats["dreg"][-1] * n_p
            acc["cls"] += n_p + n_n; acc["reg"] += n_p; acc["imgs"] += 1
            if k == 0:
                model["pnet"].training()
                outs0 = [o.numpy() for o in model["pnet"].forward(b["img"])]
    finally:
        cnet.forward = orig_forward
        cnet.drop_masks = None
        model["pnet"].drop_masks = None
    got_stats = np.array([acc["pcls"] / acc["cls"], acc["preg"] / acc["reg"], acc["dcls"] / acc["imgs"], acc["dreg"] / acc["reg"]])
    assert np.all(np.abs(got_stats - g["stats"]) <= 1e-5 * np.maximum(1.0, np.abs(g["stats"]))), (got_stats, g["stats"])
    got = total / acc["cls"]
    want = g["gradient"].astype(np.float64)
    assert np.linalg.norm(got - want) <= 1e-3 * np.linalg.norm(want)
    for off, cnt, kind, aux in nat.param_table:
        a, b = got[off:off + cnt], want[off:off + cnt]
        if cnt > 1:
            # (floor: the bias in front of the BatchNormalization has an exactly-zero gradient, fp32 leaves 1e-8 there)
            assert np.linalg.norm(a - b) <= 1e-3 * np.linalg.norm(b) + 1e-6 * np.sqrt(cnt), ("tensor at", off, kind)
        assert_close(a, b, 1e-4, "gradient tensor at %d (kind %d)" % (off, kind))
    for i, o in enumerate(outs0):
        assert_close(o, g["pnet_out%d" % (i + 1)], 1e-4, "pnet output %d" % (i + 1))


def test_detect_fixture_through_the_hip_path(F):
    """Detector:detect on the fixture frame: match indices, NMS survivor ids, winner classes identical; values at fp32 accuracy."""
    import torch
    g = np.load(os.path.join(GOLD, "detect_tiny.npz"))
    model, weights, gradient = _model(F, g["weights"])
    model["native"].bn_running.copy_(torch.from_numpy(g["bn_running"]))
    d = F.Detector(model)
    winners = d.detect(g["img"])
    m = d.last_scan
    assert np.array_equal(m["idx"].numpy(), g["match_idx"])
    assert_close(m["p"].numpy(), g["match_p"], 1e-4, "match log-probabilities")
    assert_close(m["rect"].numpy(), g["match_rect"], 1e-3, "decoded rects")
    assert d.last_pick.tolist() == g["cand_ids"].tolist()
    assert_close(d.last_cnet["bbox"], g["cand_bbox"], 1e-3, "cnet bbox (evaluate mode)")
    assert_close(d.last_cnet["cls"], g["cand_cls"], 1e-3, "cnet log-probabilities (evaluate mode)")
    fw = g["winners"]     # candidate (1-based), class, log-confidence, r2
    assert [x["class"] for x in winners] == fw[:, 1].astype(int).tolist()
    assert_close([x["confidence"] for x in winners], fw[:, 2], 1e-3, "winner confidence")
    assert_close([[x["r2"].minX, x["r2"].minY, x["r2"].maxX, x["r2"].maxY] for x in winners], fw[:, 3:7], 1e-3, "winner boxes")
