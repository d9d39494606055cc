"""Shared helpers for the parity tests."""
import numpy as np

VGG_SMALL_LAYERS = [
    dict(filters=64, kW=3, kH=3, padW=1, padH=1, dropout=0.0, conv_steps=1),
    dict(filters=128, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
    dict(filters=256, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
    dict(filters=384, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
]
VGG_SMALL_HEADS = [dict(kW=3, n=256, input=3), dict(kW=3, n=256, input=4), dict(kW=5, n=256, input=4),
                   dict(kW=7, n=256, input=4)]
VGG_SMALL_CLS = [dict(n=1024, dropout=0.5, batch_norm=True), dict(n=512, dropout=0.5)]

# a narrow model with the same topology (parity tests at sizes the oracle finishes in seconds)
TINY_LAYERS = [
    dict(filters=8, kW=3, kH=3, padW=1, padH=1, dropout=0.0, conv_steps=1),
    dict(filters=12, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
    dict(filters=16, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
    dict(filters=20, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
]
TINY_HEADS = [dict(kW=3, n=24, input=3), dict(kW=3, n=24, input=4), dict(kW=5, n=24, input=4), dict(kW=7, n=24, input=4)]
TINY_CLS = [dict(n=48, dropout=0.5, batch_norm=True), dict(n=32, dropout=0.5)]


def oracle_model(O, cfg, layers=VGG_SMALL_LAYERS, heads=VGG_SMALL_HEADS, cls=VGG_SMALL_CLS):
    return O.make_model(layers, heads, cls, cfg)


def rel_close(a, b, tol=1e-4):
    """SURVEY 8d tolerance: |a-b| <= tol * max(1, |b|) elementwise; returns (ok, worst)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    worst = float(err.max()) if err.size else 0.0
    return worst <= tol, worst


def assert_close(a, b, tol=1e-4, what=""):
    ok, worst = rel_close(a, b, tol)
    assert ok, "%s: worst |a-b|/max(1,|b|) = %.3e > %.1e" % (what, worst, tol)


def random_boxes(rng, n, w=800, h=450, unique_y2=True):
    x1 = rng.uniform(-20, w - 10, n); y1 = rng.uniform(-20, h - 10, n)
    bw = rng.uniform(4, 200, n); bh = rng.uniform(4, 200, n)
    b = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
    if unique_y2:
        # unique fp32 sort keys (SURVEY hard part 4: TH's tie order is unpinned by the reference)
        for _ in range(100):
            _, first = np.unique(b[:, 3], return_index=True)
            dup = np.ones(n, bool); dup[first] = False
            if not dup.any():
                break
            b[dup, 3] += rng.uniform(0.01, 1.0, int(dup.sum())).astype(np.float32)
        assert len(np.unique(b[:, 3])) == n
    return b


def oracle_tables(pos, neg, rois):
    """The example tables O.train_image takes, from the host mirror's lists: positives (anchor, roi) and
    negatives (anchor,) as built by assemble_examples (BatchIterator.lua:198-225), already cleaned."""
    pos_idx = np.array([[a.layer, a.aspect, a.index[1], a.index[2], rois.index(r) + 1] for a, r in pos], dtype=np.int32).reshape(-1, 5)
    pos_rect = np.array([[a.minX, a.minY, a.maxX, a.maxY] for a, r in pos], dtype=np.float64).reshape(-1, 4)
    neg_idx = np.array([[e[0].layer, e[0].aspect, e[0].index[1], e[0].index[2]] for e in neg], dtype=np.int32).reshape(-1, 4)
    neg_rect = np.array([[e[0].minX, e[0].minY, e[0].maxX, e[0].maxY] for e in neg], dtype=np.float64).reshape(-1, 4)
    roi_rect = np.array([[r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY] for r in rois], dtype=np.float64)
    roi_cls = np.array([r.class_index for r in rois], dtype=np.int32)
    return pos_idx, pos_rect, roi_rect, roi_cls, neg_idx, neg_rect
