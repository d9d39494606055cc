"""Evaluation loop (README.md:11,13; BatchIterator.lua:279-317; main.lua:183-216) on a toy validation set,
against an evaluation done on the oracle side:
  * validation losses: the oracle's pnet / cnet forward passes in evaluate mode (orc_pnet_forward / orc_cnet_forward, ROI
    windows and adaptive max pooling) and the per-example loss arithmetic of objective.lua:91-177 restated here in numpy;
  * mean average precision: orc_detect on every frame + a deliberately naive AP computation written for this test."""
import numpy as np
import pytest

from util import oracle_model
from test_gpu_model import _amplified_weights

pytestmark = pytest.mark.gpu
H, W = 128, 176


class _Val(object):
    """nextValidation(count) -> [{img, rois}] over a fixed list (what BatchIterator.lua:279-317 hands out)."""

    def __init__(self, anchors, items):
        self.anchors, self.items, self.i = anchors, items, 0

    def nextValidation(self, count=1):
        out = []
        for _ in range(count):
            out.append(self.items[self.i % len(self.items)])
            self.i += 1
        return out


@pytest.fixture(scope="module")
def setup(F, O):
    import torch
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    nat = model["native"]
    w = _amplified_weights(nat, weights.cpu().numpy(), 17, cls_gain=200.0)
    weights.copy_(torch.from_numpy(w))
    # running statistics that differ from their initial values (evaluate mode reads them)
    rng = np.random.RandomState(5)
    bn = np.concatenate([rng.randn(1024) * 0.1, rng.uniform(0.5, 1.5, 1024)]).astype(np.float32)
    nat.bn_running.copy_(torch.from_numpy(bn))
    return dict(cfg=cfg, model=model, weights=weights, w=w, bn=bn, om=oracle_model(O, cfg),
                anchors=F.Anchors(model["pnet"], cfg["scales"]))


def _smooth_l1(z):
    z = np.abs(z)
    return np.where(z < 1, 0.5 * z * z, z - 0.5).sum()


def oracle_validation_losses(F, O, s, items, seed=1234, negatives=16):
    cfg, om, w, bn = s["cfg"], s["om"], s["w"], s["bn"]
    rng = F.MT19937(seed)
    bgclass = cfg["class_count"] + 1
    kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
    layers5 = O.model_localizer_layers(om, 5)
    acc = dict(cls=0.0, reg=0.0, creg=0.0, ccls=0.0)
    cls_count = reg_count = ccls_count = 0
    for x in items:
        outs, _ = O.pnet_forward(om, w, x["img"], False, None)           # evaluate mode
        sizes = [(o.shape[1], o.shape[2]) for o in outs[:4]]
        pos, neg = F.assemble_examples(s["anchors"], cfg, x["rois"], W, H, rng, negatives=negatives, native=False)
        pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
        fm = outs[4]
        rows, crt, cct = [], [], []
        for e in list(pos) + list(neg):
            a = e[0]
            v = outs[a.layer - 1][(a.aspect - 1) * 6:(a.aspect - 1) * 6 + 6, a.index[1] - 1, a.index[2] - 1].astype(np.float64)
            target = 0 if len(e) > 1 else 1                               # foreground = class 1, background = class 2
            lse = np.log(np.exp(v[0] - v[:2].max()) + np.exp(v[1] - v[:2].max())) + v[:2].max()
            acc["cls"] += lse - v[target]                                 # nn.CrossEntropyCriterion (:104, :132)
            arect = [a.minX, a.minY, a.maxX, a.maxY]
            if len(e) > 1:
                roi = e[1]
                rr = [roi.rect.minX, roi.rect.minY, roi.rect.maxX, roi.rect.maxY]
                t = O.input_to_anchor(arect, rr).astype(np.float64)       # FloatTensor target (:110)
                acc["reg"] += _smooth_l1(outs[a.layer - 1][(a.aspect - 1) * 6 + 2:(a.aspect - 1) * 6 + 6, a.index[1] - 1, a.index[2] - 1].astype(np.float64) - t) * 10
                prop = O.anchor_to_input(arect, v[2:6].astype(np.float32))   # reg_proposal (:111)
                crt.append(O.input_to_anchor(prop, rr)); cct.append(roi.class_index)
                pooled_rect = rr                                          # positives pool the ground-truth rect (:117)
            else:
                crt.append(np.zeros(4, np.float32)); cct.append(bgclass)
                pooled_rect = arect                                       # negatives pool the anchor rect (:137)
            win = O.extract_roi_window(layers5, pooled_rect, fm.shape[1], fm.shape[2])
            rows.append(O.adaptive_max_pool_fwd(fm, win, kh, kw)[0].reshape(-1))
        ccls_count += 1
        if not rows:
            continue
        bbox, lsm, _ = O.cnet_forward(om, w, np.stack(rows), False, None, bn.copy())
        bbox = bbox.astype(np.float64).copy()
        bbox[len(pos):] = 0                                               # :170
        acc["creg"] += _smooth_l1(bbox - np.stack(crt).astype(np.float64)) * 10
        acc["ccls"] += float(np.mean([-lsm[i, c - 1] for i, c in enumerate(cct)]))   # ClassNLL: mean over the batch
        reg_count += len(pos); cls_count += len(rows)
    return dict(pcls=acc["cls"] / cls_count, preg=acc["reg"] / reg_count, dcls=acc["ccls"] / ccls_count,
                dreg=acc["creg"] / reg_count, examples=cls_count, positives=reg_count)


def test_validation_losses(F, O, setup):
    s = setup
    items = [dict(img=F.synthetic_image(H, W, 40 + k), rois=F.synthetic_rois(s["cfg"], W, H, 3, 7, 40 + k)) for k in range(3)]
    from frcnn_amd.evaluation import validation_losses
    got = validation_losses(s["model"], _Val(s["anchors"], items), 3)
    want = oracle_validation_losses(F, O, s, items)
    assert got["examples"] == want["examples"] and got["positives"] == want["positives"] and got["positives"] > 0
    for k in ("pcls", "preg", "dcls", "dreg"):
        assert abs(got[k] - want[k]) <= 1e-4 * max(1.0, abs(want[k])), (k, got[k], want[k])
    # the networks are handed back in the mode they were in
    assert s["model"]["pnet"].train and s["model"]["cnet"].train


def _naive_map(dets, gts, thr=0.5):
    """Independent AP: per class, walk the detections by decreasing confidence, area under the interpolated PR curve."""
    from frcnn_amd.Rect import Rect
    aps = []
    for c in sorted(set(g[1] for g in gts)):
        G = [g for g in gts if g[1] == c]
        taken = [False] * len(G)
        D = sorted([d for d in dets if d[1] == c], key=lambda d: -d[2])
        pts, tp, fp = [], 0, 0
        for d in D:
            best, bj = -1.0, -1
            for j, g in enumerate(G):
                if g[0] != d[0]:
                    continue
                v = Rect.IoU(d[3], g[2])
                if v > best:
                    best, bj = v, j
            if best >= thr and not taken[bj]:
                taken[bj] = True; tp += 1
            else:
                fp += 1
            pts.append((tp / len(G), tp / (tp + fp)))
        ap, prev_r = 0.0, 0.0
        for i, (r, p) in enumerate(pts):
            if r > prev_r:
                ap += (r - prev_r) * max(q for _, q in pts[i:])
                prev_r = r
        aps.append(ap)
    return float(np.mean(aps)) if aps else float("nan")


def test_mean_average_precision_of_detect(F, O, setup):
    from frcnn_amd.Rect import Rect
    from frcnn_amd.evaluation import evaluate_detections
    s = setup
    frames = [F.synthetic_image(H, W, 60 + k) for k in range(4)]
    # oracle side: Detector:detect on every frame
    odet = []
    for k, img in enumerate(frames):
        ref = O.detect(s["om"], s["w"], s["bn"], img)
        for row in ref["winners"]:
            odet.append((k, int(row[0]), float(row[1]), Rect(*row[2:6])))
    assert len(odet) >= 8, "toy set produced too few detections"
    # ground truth of the toy set: every third detection's box (its own class: a true positive), shifted copies that
    # overlap by less than one half (missed boxes), and a box of a class nobody detects
    items, gts = [], []
    for k, img in enumerate(frames):
        rois = []
        mine = [d for d in odet if d[0] == k]
        for j, d in enumerate(mine[::3]):
            r = d[3]
            rois.append(F.Roi(Rect(r.minX, r.minY, r.maxX, r.maxY) if j % 2 == 0 else r.offset(r.width() * 0.8, 0), d[1]))
        rois.append(F.Roi(Rect(5, 5, 40, 40), 16))
        items.append(dict(img=img, rois=rois))
        gts += [(k, r.class_index, r.rect) for r in rois]
    d = F.Detector(s["model"])
    got = evaluate_detections(d, _Val(s["anchors"], items), len(items))
    assert got["detections"] == len(odet) and got["ground_truth"] == len(gts)
    want = _naive_map(odet, gts)
    assert got["tp"] > 0 and got["fp"] > 0 and 0.0 < want < 1.0
    assert abs(got["mAP"] - want) <= 1e-6, (got["mAP"], want)
