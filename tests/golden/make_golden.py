"""Generates the committed golden fixtures (run in the authoring container; outputs are data only).

The reference ships no golden vectors and cannot run here (Lua/Torch7 absent), so these fixtures pin
the CPU oracle itself: every vector is produced by the oracle AND cross-checked by an independent
implementation before it is written -- the naive numpy restatement (oracle/naive_np.py) for the
integer/bit-exact pieces, PyTorch-CPU for the floating-point layer math."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import naive_np  # noqa: E402
import pyoracle as O  # noqa: E402
from util import VGG_SMALL_CLS, VGG_SMALL_HEADS, VGG_SMALL_LAYERS, random_boxes  # noqa: E402


def nms_cases():
    rng = np.random.RandomState(2015)
    cases = [dict(name="appendix_b_quirk", boxes=[0, 0, 10, 10, 1, 1, 11, 11, 50, 50, 60, 70], ncols=4, overlap=0.25, key="y2")]
    for n, thr, key, ncols in [(17, 0.25, "y2", 4), (64, 0.1, "y2", 4), (65, 0.25, "area", 4), (200, 0.1, "5", 5),
                               (333, 0.25, "y2", 5), (1, 0.25, "y2", 4), (2, 0.0, "y2", 4)]:
        b = random_boxes(rng, n)
        if ncols == 5:
            b = np.concatenate([b, rng.permutation(n).astype(np.float32)[:, None] / n], 1)
        cases.append(dict(name="rand_n%d_%s_%g" % (n, key, thr), boxes=b.ravel().tolist(), ncols=ncols, overlap=thr, key=key))
    for c in cases:
        b = np.array(c["boxes"], dtype=np.float32).reshape(-1, c["ncols"])
        km = dict(y2=(0, 0), area=(1, 0)).get(c["key"], (2, int(c["key"]) if c["key"].isdigit() else 0))
        a = O.nms(b, c["overlap"], km[0], km[1]).tolist()
        n2 = naive_np.nms(b, c["overlap"], c["key"]).tolist()
        assert a == n2, c["name"]
        c["pick"] = a
    return cases


def anchors_fixture():
    cfg = dict(class_count=16, scales=[32, 64, 128, 256], roi_pooling=dict(kw=6, kh=6))
    m = O.make_model(VGG_SMALL_LAYERS, VGG_SMALL_HEADS, VGG_SMALL_CLS, cfg)
    A = O.Anchors(m)
    lay = [O.model_localizer_layers(m, i + 1) for i in range(5)]
    wn, hn = naive_np.anchor_tables([l.tolist() for l in lay[:4]], cfg["scales"])
    assert np.array_equal(wn, A.w_table) and np.array_equal(hn, A.h_table)
    img = (0, 0, 800, 450)
    rng = np.random.RandomState(7)
    rois = []
    for _ in range(6):
        x = rng.uniform(0, 600); y = rng.uniform(0, 300); w = rng.uniform(30, 200); h = rng.uniform(30, 140)
        rois.append([float(np.floor(x)), float(np.floor(y)), float(np.floor(x + w)), float(np.floor(y + h))])
    idx, rc = A.find_positive(rois, img, 0.5, 0.25, True)
    idx2 = naive_np.find_positive(wn, hn, rois, img, 0.5, 0.25, True)
    assert idx.tolist() == idx2, "find_positive: oracle vs naive numpy"
    mt = O.MT(7)
    nidx, nrc = A.sample_negative(img, rois, 0.25, 16, mt)
    windows = [O.extract_roi_window(lay[4], r, 29, 50).tolist() for r in rois + [list(img), [790, 440, 800, 450], [0, 0, 16, 16]]]
    windows2 = [naive_np.roi_window(lay[4].tolist(), r, 29, 50) for r in rois + [list(img), [790, 440, 800, 450], [0, 0, 16, 16]]]
    assert windows == windows2
    return dict(localizer_layers=[l.tolist() for l in lay],
                w_sum=float(A.w_table.astype(np.float64).sum()), h_sum=float(A.h_table.astype(np.float64).sum()),
                w_l1a2=A.w_table[0, 1, :4].tolist(), h_l4a1=A.h_table[3, 0, :6].tolist(),
                rois=rois, positive_idx=idx.tolist(), negative_idx_mt7=nidx.tolist(), roi_windows=windows,
                mt19937_seed7_first5=[O.MT(7).random() for _ in range(1)] + [])


def layer_fixture():
    """Small conv / pool / adaptive-pool / linear case: oracle output, cross-checked with PyTorch-CPU."""
    import torch
    import torch.nn.functional as Fn
    rng = np.random.RandomState(11)
    x = rng.randn(5, 9, 11).astype(np.float32); w = (rng.randn(7, 5, 3, 3) * 0.3).astype(np.float32); b = rng.randn(7).astype(np.float32)
    y = O.conv2d_fwd(x, w, b, 1)
    yt = Fn.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)[0].numpy()
    assert np.abs(y - yt).max() < 1e-5
    p, pi = O.maxpool_fwd(y)
    pt = Fn.max_pool2d(torch.from_numpy(y)[None], 2, 2, ceil_mode=True)[0].numpy()
    assert np.array_equal(p, pt)
    return dict(x=x.ravel().tolist(), w=w.ravel().tolist(), b=b.tolist(), y=y.ravel().tolist(), pool=p.ravel().tolist(),
                x_shape=list(x.shape), w_shape=list(w.shape), y_shape=list(y.shape), pool_shape=list(p.shape))


def image_fixture():
    """processImage pieces on a tiny frame: numpy restatement (oracle/orc_image.py); the up-scaling branch of image.scale is
    cross-checked with PyTorch's bilinear interpolation with align_corners=True (the same (src-1)/(dst-1) rule), the
    contrastive normalisation with a direct float64 evaluation of the two modules."""
    import torch
    import torch.nn.functional as Fn
    import orc_image as OI
    rng = np.random.RandomState(21)
    rgb = rng.rand(3, 10, 14).astype(np.float32)
    yuv = OI.rgb2yuv(rgb)
    up = OI.scale_bilinear(yuv, 20, 15)
    ut = Fn.interpolate(torch.from_numpy(yuv).double()[None], size=(15, 20), mode="bilinear", align_corners=True)[0].numpy()
    assert np.abs(up - ut).max() < 1e-5
    down = OI.scale_bilinear(yuv, 9, 7)
    norm = OI.center_and_scale(down)
    k = OI.gaussian1d(7)
    cn = OI.contrastive_norm(norm[0], k)
    kk = k.astype(np.float64) / k.astype(np.float64).sum()
    H, W = norm[0].shape

    def est(x):
        o = np.zeros((H, W))
        for y in range(H):
            for xx in range(W):
                for jy in range(7):
                    for jx in range(7):
                        sy, sx = y + jy - 3, xx + jx - 3
                        if 0 <= sy < H and 0 <= sx < W:
                            o[y, xx] += kk[jy] * kk[jx] * x[sy, sx]
        return o
    coef = est(np.ones((H, W))); sub = norm[0] - est(norm[0].astype(np.float64)) / coef
    sd = np.sqrt(est(sub * sub)) / coef
    assert np.abs(cn - sub / np.where(sd > 1e-4, sd, 1e-4)).max() < 1e-4
    return dict(rgb=rgb.ravel().tolist(), yuv=yuv.ravel().tolist(), up_15x20=up.ravel().tolist(), down_7x9=down.ravel().tolist(),
                normalized=norm.ravel().tolist(), contrastive_y=cn.ravel().tolist(), gaussian1d_7=k.tolist())


if __name__ == "__main__":
    json.dump(image_fixture(), open(os.path.join(HERE, "image_small.json"), "w"))
    json.dump(nms_cases(), open(os.path.join(HERE, "nms_cases.json"), "w"))
    json.dump(anchors_fixture(), open(os.path.join(HERE, "anchors_vgg_small.json"), "w"))
    json.dump(layer_fixture(), open(os.path.join(HERE, "layers_small.json"), "w"))
    print("golden fixtures written")
