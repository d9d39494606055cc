"""lossAndGradient (objective.lua:45-218) on the BENCHMARKED workload -- vgg_small, one synthetic 3x450x800 frame,
config/duplo.lua values, the inputs bench.py's cpu_baseline leg builds (weights seed 42, image seed 1000,
4 boxes seed 7, examples from findPositive + 16 sampled negatives with MT19937 seed 7, explicit dropout masks) --
against the CPU oracle's train_image on the same inputs.  The oracle needs ~8 s for this frame on the GPU box's
host cores.  Bars: the four losses 1e-5 relative; flat gradient per tensor on the L2 norm."""
import numpy as np
import pytest

import decisions
from util import oracle_model, oracle_tables, assert_close
from test_gpu_model import _compare_gradient, _masks, _OneBatch

pytestmark = pytest.mark.gpu
H, W = 450, 800


def fullsize_inputs(F, cfg, model, H=H, W=W, sizes_want=[(55, 98), (27, 48), (25, 46), (23, 44)]):
    """Exactly what SyntheticBatchIterator / bench.cpu_baseline use for image 0."""
    anchors = F.Anchors(model["pnet"], cfg["scales"])
    rois = F.synthetic_rois(cfg, W, H, 4, 7, 0)
    pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, F.MT19937(7))
    sizes = F.output_map_sizes(model, H, W)
    assert sizes == sizes_want
    pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
    return anchors, rois, pos, neg, F.synthetic_image(H, W, 0)


@pytest.mark.parametrize("split", [1, 0])
def test_fullsize_loss_and_gradient(F, O, split):
    """split = 1 (the default): the 3x3 layers whose shapes fit in the split-bf16 operand form (option "split_bf16");
    split = 0: fp32 matrix-core kernels only -- same bars for both."""
    F._lib.call("frcnn_set_option", b"split_bf16", split)
    try:
        _fullsize(F, O)
    finally:
        F._lib.call("frcnn_set_option", b"split_bf16", 1)


def test_vgg_large_fullsize_step(F, O):
    """BASELINE config 5's one-GPU workload at FULL size: vgg_large (64/128/256/512, 2-2-3-3; models/vgg_large.lua:5-22), one
    3x600x1000 frame, config/imagenet.lua values (200 classes, scales 48..384, 6x6 ROI pooling), 45 015 anchors.  The
    oracle needs ~30 s for this frame on the GPU box's host cores.  Same bars as the vgg_small frame."""
    _fullsize(F, O, large=True)


def test_vgg_large_config5_as_stated(F, O):
    """BASELINE config 5 exactly as BASELINE.json states it: vgg_large 3x600x1000, 7x7 ROI pooling
    (config/imagenet.lua:9-12, README.md:19) and an example list of R = 300 (negatives sampled up to that count)."""
    _fullsize(F, O, large=True, pool=7, examples=300)


def test_fullsize_detect(F, O):
    """BASELINE config 2 at its stated size: Detector:detect (Detector.lua:17-141) on synthetic 3x450x800 frames against
    orc_detect, every stage (test_gpu_model.check_detect: lists may differ only where a border case is shown)."""
    import torch
    from test_gpu_model import _amplified_weights, check_detect
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    om = oracle_model(O, cfg)
    w = _amplified_weights(model["native"], weights.cpu().numpy().copy(), cfg["class_count"] + 1, cls_gain=200.0)
    # (x60 on the anchor nets' class logits passes most of the 26 544 anchors; x30 -- bench.py's gain -- about 8 000)
    for off, cnt, kind, aux in model["native"].param_table:
        if kind == 0 and aux == 18:
            v = w[off:off + cnt].reshape(18, -1)
            for a in range(3):
                v[a * 6:a * 6 + 2] *= 0.5
    weights.copy_(torch.from_numpy(w))
    r = check_detect(F, O, model, om, w, range(2), 450, 800)
    print("full-size detect: frame %(seed)d, %(matches)d matches, %(candidates)d candidates, %(winners)d winners, %(frames_compared)d frames" % r)
    assert r["matches"] > 1000 and r["winners"] > 0


def _fullsize(F, O, large=False, pool=None, examples=None):
    import torch
    H, W = (600, 1000) if large else (450, 800)
    cfg = dict(F.imgnet_cfg if large else F.duplo_cfg)
    if pool:
        cfg["roi_pooling"] = dict(kw=pool, kh=pool)
    model = (F.vgg_large if large else F.vgg_small)(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    nat = model["native"]
    om = oracle_model(O, cfg, model["layers"], model["anchor_nets"], model["class_layers"])
    w = weights.cpu().numpy().copy()
    anchors, rois, pos, neg, img = fullsize_inputs(
        F, cfg, model, H, W, [(73, 123), (36, 61), (34, 59), (32, 57)] if large else [(55, 98), (27, 48), (25, 46), (23, 44)])
    if examples:   # pad the list with further sampled negatives (Anchors.lua:197-235, the same generator stream going on)
        mt = F.MT19937(11)
        more = F.clean_examples(anchors.sampleNegative(F.Rect(0, 0, W, H), rois, cfg["negative_threshold"], 4 * examples, mt),
                                F.output_map_sizes(model, H, W))
        seen = set((n[0].layer, n[0].aspect, n[0].index[1], n[0].index[2]) for n in neg)
        for n in more:
            k = (n[0].layer, n[0].aspect, n[0].index[1], n[0].index[2])
            if len(pos) + len(neg) < examples and k not in seen:
                neg.append(n); seen.add(k)
        assert len(pos) + len(neg) == examples
    if large:
        assert 3 * sum(h * w_ for h, w_ in [(73, 123), (36, 61), (34, 59), (32, 57)]) == 45015
    R = len(pos) + len(neg)
    assert len(pos) > 0 and len(neg) >= 16      # 16 sampled negatives + the nearby-aversion ones
    rng = np.random.RandomState(0)
    pm = _masks(rng, model)
    cm = [(rng.rand(R, l["n"]) > 0.5).astype(np.float32) for l in model["class_layers"]]
    bn0 = nat.bn_running.cpu().numpy().copy()
    # ---- the product path --------------------------------------------------------------------------------
    model["pnet"].drop_masks = pm
    model["cnet"].drop_masks = cm
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    try:
        f = F.create_objective(model, weights, gradient, _OneBatch([dict(img=img, positive=pos, negative=neg)], anchors), stats)
        with decisions.CaptureBeforeBackward(F, model, f) as cap:    # the device's pool winners / PReLU branches
            loss, grad = f(weights)
        model["pnet"].training()
        outs_dev = [o.numpy() for o in model["pnet"].forward(img)]   # the five outputs of this very frame (same masks)
    finally:
        model["pnet"].drop_masks = None
        model["cnet"].drop_masks = None
    # ---- oracle, with the device's discrete decisions taken as given (tests/decisions.py) -----------------
    g_want = np.zeros_like(w); acc = np.zeros(8); bn_o = bn0.copy()
    own = decisions.blank_like(cap.captured[0])
    with O.decisions(inject=cap.captured[0], record=own):
        O.train_image(om, w, g_want, img, *oracle_tables(pos, neg, rois), pm, cm, bn_o, acc)
        outs_want, _ = O.pnet_forward(om, w, img, True, pm)
    # the forward pass itself, directly: the five pnet outputs of the frame at 1e-4 (not only through the four losses)
    for i, (a, b) in enumerate(zip(outs_dev, outs_want)):
        assert a.shape == b.shape
        assert_close(a, b, 1e-4, "pnet output %d of the full-size frame" % (i + 1))
    differing = decisions.count_differences(cap.captured[0], own)
    g_want /= acc[2]
    want = dict(pcls=acc[0] / acc[2], preg=acc[1] / acc[3], dcls=acc[6] / acc[7], dreg=acc[4] / acc[5])
    assert acc[2] == R and acc[3] == len(pos)
    g = grad.cpu().numpy()
    for k in ("pcls", "preg", "dcls", "dreg"):
        assert abs(stats[k][-1] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, stats[k][-1], want[k])
    assert abs(loss - (want["pcls"] + want["preg"])) <= 1e-5 * max(1.0, abs(loss))
    assert_close(nat.bn_running.cpu().numpy(), bn_o, 1e-5, "bn running statistics")
    # whole vector, then per tensor
    rel = np.linalg.norm(g.astype(np.float64) - g_want) / np.linalg.norm(g_want.astype(np.float64))
    worst = 0.0
    for off, cnt, kind, aux in nat.param_table:
        a, b = g[off:off + cnt].astype(np.float64), g_want[off:off + cnt].astype(np.float64)
        nb = np.linalg.norm(b)
        if nb > 1e-6 * np.sqrt(cnt):    # (a Linear bias feeding BatchNorm has a true gradient of ~0: rounding noise only)
            worst = max(worst, np.linalg.norm(a - b) / nb)
    print("full-size step: %d examples, loss %.6f (oracle %.6f), gradient rel-L2 %.2e whole vector, %.2e worst tensor"
          % (R, loss, want["pcls"] + want["preg"], rel, worst))
    assert rel <= 1e-3, rel
    # SURVEY 8d on every tensor, nothing set aside: 1e-3 relative L2 per tensor + 1e-4 elementwise
    _compare_gradient(nat, g, g_want, 0, nat.total_params, tol_l2=1e-3, elementwise=True,
                      slope_terms=decisions.slope_terms(nat, model, own["slope_abs"], acc[2]))
    print("decisions the oracle would have taken differently: %s" % {k: "%d of %d" % v for k, v in differing.items()})
    for kind, (nd, nt) in differing.items():
        assert nd <= max(4, 2e-5 * nt), (kind, nd, nt)
    nat.bn_running.copy_(torch.from_numpy(bn0))


def test_frames_with_more_blocks_than_a_magnitude_record_holds(F, small_cfg):
    """Two-plane fp16 form: a launch records one maximum per block in a record of 16 384 entries; a 3000 x 3000 frame gives the
    first layers more blocks than that, and they take the magnitude in a pass of their own.  The anchor nets' outputs must agree
    with the three-bf16-plane form (which needs no magnitudes) to the usual bar."""
    import ctypes
    model = F.vgg_small(dict(small_cfg))
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"])
    pnet = model["pnet"]
    pnet.evaluate()
    img = F.to_device(F.synthetic_image(3000, 3000, 3))
    before = ctypes.c_int(0)
    F._lib.call("frcnn_get_option", b"x3_f16", ctypes.byref(before))
    res = {}
    try:
        for form in (1, 0):
            F._lib.call("frcnn_set_option", b"x3_f16", form)
            res[form] = [o.numpy().copy() for o in pnet.forward(img)]
    finally:
        F._lib.call("frcnn_set_option", b"x3_f16", before.value)
    for a, b in zip(res[1], res[0]):
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
