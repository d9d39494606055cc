"""Edge cases of the hot path on the GPU: images whose sides are odd / not multiples of the pooling stride,
frames with no examples at all, negatives only, empty / single-box NMS, an ROI that collapses to one cell."""
import numpy as np
import pytest

from util import VGG_SMALL_CLS, VGG_SMALL_HEADS, VGG_SMALL_LAYERS, assert_close, oracle_model
from test_gpu_model import _OneBatch, _compare_gradient, _masks, check_pnet_forward_backward

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(F, O):
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    om = oracle_model(O, cfg)
    return dict(cfg=cfg, model=model, weights=weights, gradient=gradient, om=om, w=weights.cpu().numpy().copy())


@pytest.mark.parametrize("H,W", [(131, 173), (97, 211)])
def test_odd_image_sizes(F, O, setup, H, W):
    """ceil-mode pooling on odd maps (131 -> 66 -> 33 -> 17 -> 9), partial tiles on every edge."""
    s = setup
    rng = np.random.RandomState(H)
    img = F.synthetic_image(H, W, 4)
    check_pnet_forward_backward(F, O, s, img, _masks(rng, s["model"]), rng, what="odd size")


def test_no_examples_and_negatives_only(F, setup):
    """objective.lua:45-218 with an image that contributes nothing (no positives, no negatives): zero gradient,
    NaN statistics exactly like 0/0 in Lua; negatives only: finite loss, zero regression gradient paths."""
    s = setup
    model, cfg = s["model"], s["cfg"]
    H, W = 128, 176
    anchors = F.Anchors(model["pnet"], cfg["scales"])
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    empty = [dict(img=F.synthetic_image(H, W, 0), positive=[], negative=[])]
    f = F.create_objective(model, s["weights"], s["gradient"], _OneBatch(empty, anchors), stats)
    loss, grad = f(s["weights"])
    g = grad.cpu().numpy()
    assert not g.any()
    assert np.isnan(loss) and np.isnan(stats["pcls"][-1])
    # negatives only
    rois = F.synthetic_rois(cfg, W, H, 2, 7, 0)
    pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, F.MT19937(3), negatives=8)
    sizes = F.output_map_sizes(model, H, W)
    neg = F.clean_examples(neg, sizes)
    assert len(neg) > 0
    only_neg = [dict(img=F.synthetic_image(H, W, 0), positive=[], negative=neg)]
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    f = F.create_objective(model, s["weights"], s["gradient"], _OneBatch(only_neg, anchors), stats)
    loss, grad = f(s["weights"])
    g = grad.cpu().numpy()
    assert np.isfinite(stats["pcls"][-1]) and stats["pcls"][-1] > 0
    assert np.isnan(stats["preg"][-1])           # reg_loss / reg_count = 0 / 0 (objective.lua:203)
    assert np.isfinite(g).all() and g.any()


def test_nms_empty_and_single(F, O):
    assert len(F.nms(np.zeros((0, 4), np.float32), 0.5, None)) == 0          # nms.lua:26-28
    one = np.array([[1, 2, 30, 40]], np.float32)
    assert list(F.nms(one, 0.5, None)) == [1]
    two = np.array([[1, 2, 30, 40], [1, 2, 30, 40.5]], np.float32)           # IoU ~0.99 -> the larger y2 survives
    assert list(F.nms(two, 0.5, None)) == [2]
    assert list(F.nms(two, 0.5, None)) == O.nms(two, 0.5).tolist()


def test_roi_smaller_than_pool_grid(F, O):
    """extract_roi_pooling_input on a rect that maps to a 1x1 / 2x3 region of the map (objective.lua:5-13):
    SpatialAdaptiveMaxPooling repeats the cells (windows overlap)."""
    rng = np.random.RandomState(0)
    C_, fh, fw, kh, kw = 5, 29, 50, 6, 6
    fm = rng.randn(C_, fh, fw).astype(np.float32)
    wins = np.array([[7, 7, 9, 9], [3, 4, 10, 12], [29, 29, 50, 50], [1, 29, 1, 50]], dtype=np.int32)
    dfm = F.DeviceTensor.from_numpy(fm); dw = F.DeviceTensor.from_numpy(wins)
    R = len(wins); D = C_ * kh * kw
    out = F.DeviceTensor.empty((R, D)); idx = F.DeviceTensor.empty((R, D), np.int32)
    F._lib.call("frcnn_roi_pool_forward", F.ptr(dfm), C_, fh, fw, F.ptr(dw), R, kh, kw, F.ptr(out), F.ptr(idx), F.stream_ptr())
    want = np.stack([O.adaptive_max_pool_fwd(fm, w, kh, kw)[0].reshape(-1) for w in wins])
    assert np.array_equal(out.numpy(), want)
    # backward: overlapping windows accumulate
    g = rng.randn(R, D).astype(np.float32)
    gm = F.DeviceTensor.zeros((C_, fh, fw))
    F._lib.call("frcnn_roi_pool_backward", F.ptr(gm), C_, fh, fw, F.ptr(F.DeviceTensor.from_numpy(g)) if False else F.ptr(_keep(F, g)), F.ptr(idx), R, kh, kw, F.stream_ptr())
    want_g = np.zeros((C_, fh, fw), np.float32)
    for r, w in enumerate(wins):
        _, ix = O.adaptive_max_pool_fwd(fm, w, kh, kw)
        O.adaptive_max_pool_bwd(want_g, g[r].reshape(C_, kh, kw), ix)
    assert_close(gm.numpy(), want_g, 1e-5, "roi backward with repeated cells")


_KEEP = []


def _keep(F, a):
    t = F.DeviceTensor.from_numpy(a)
    _KEEP.append(t)
    return t


def test_async_heads_forward_equals_plain_forward(F, setup):
    """frcnn_pnet_forward_async_heads leaves the anchor nets on the side stream; once joined (frcnn_pnet_backward, a
    new forward, or a device synchronisation) every output equals the plain training-mode forward.  An abandoned
    asynchronous forward followed by a plain one must be safe (the next forward joins first)."""
    import torch
    s = setup
    rng = np.random.RandomState(31)
    img = F.synthetic_image(140, 190, 9)
    pnet = s["model"]["pnet"]
    pnet.training(); pnet.drop_masks = _masks(rng, s["model"])
    try:
        plain = [o.numpy().copy() for o in pnet.forward(img)]
        outs = pnet.forward(img, async_heads=True)
        torch.cuda.synchronize()                      # (device-wide: covers the library's side stream)
        for a, b in zip(plain, outs):
            assert np.array_equal(a, b.numpy())
        pnet.forward(img, async_heads=True)           # abandoned: never joined by the caller
        again = [o.numpy().copy() for o in pnet.forward(img)]
        for a, b in zip(plain, again):
            assert np.array_equal(a, b)
    finally:
        pnet.drop_masks = None


def test_wait_block_gradients_contract(F, setup):
    """frcnn_pnet_wait_block_gradients: rejected before any backward pass and for blocks that do not exist; after a
    backward pass the waiting stream sees the block's final gradient slice."""
    import torch
    s = setup
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)                          # a fresh model: no backward pass has run on it yet
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=5)
    pnet = model["pnet"]
    with pytest.raises(F.FrcnnError):
        pnet.wait_block_gradients(3)
    img = F.synthetic_image(140, 190, 3)
    pnet.training()
    outs = pnet.forward(img)
    deltas = pnet.delta_outputs(zero=True)
    rng = np.random.RandomState(2)
    for d in deltas:
        d.copy_from_numpy((rng.randn(*d.shape) / np.sqrt(d.numel())).astype(np.float32))
    g.zero_()
    pnet.backward(img, deltas)
    aux = torch.cuda.Stream()
    lo, hi = pnet.block_param_range(3)
    with torch.cuda.stream(aux):
        pnet.wait_block_gradients(3)
        early = g[lo:hi].clone()                      # copied on the auxiliary stream, behind the block's event only
    torch.cuda.synchronize()
    assert torch.equal(early, g[lo:hi]) and float(early.abs().max()) > 0
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_pnet_wait_block_gradients", model["native"].h, 9, F.stream_ptr())
    assert pnet.block_param_range(0)[0] == 0 and pnet.block_param_range(3)[1] == pnet.heads_param_range()[0]


def test_uploading_iterator_equals_resident_frames(F, setup):
    """SyntheticBatchIterator(upload=True) -- frames in page-locked host memory, uploaded every step on a copy stream into a
    ring of device buffers, the consumer's stream waiting for the copy's event (objective.lua:66 `x.img:cuda()`) -- feeds the
    training step the same bytes as the resident frames: identical gradients and weights over several steps.  Bit-for-bit
    equality of two runs is a property of the "deterministic" option only (the default folds some sums with fp32 atomics,
    whose order varies from run to run), so the comparison runs under it."""
    import torch
    s = setup
    model = s["model"]
    res = {}
    F._lib.call("frcnn_set_option", b"deterministic", 1)
    try:
        _upload_vs_resident(F, s, model, res)
    finally:
        F._lib.call("frcnn_set_option", b"deterministic", 0)
    s["weights"].copy_(torch.from_numpy(s["w"]))
    assert res["resident"][2] == res["upload"][2]
    assert np.array_equal(res["resident"][1], res["upload"][1]) and np.array_equal(res["resident"][0], res["upload"][0])


def _upload_vs_resident(F, s, model, res):
    import torch
    for mode in ("resident", "upload"):
        s["weights"].copy_(torch.from_numpy(s["w"]))
        it = F.SyntheticBatchIterator(model, H=128, W=176, images_per_batch=1, pool=3, upload=(mode == "upload"))
        rng = np.random.RandomState(1)
        model["pnet"].drop_masks = _masks(rng, model)
        stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
        cnet = model["cnet"]
        orig = cnet.forward

        def fwd(x, orig=orig):   # deterministic cnet dropout masks sized for whatever batch arrives
            R = x.shape[0]
            r2 = np.random.RandomState(R)
            cnet.drop_masks = [(r2.rand(R, 1024) > 0.5).astype(np.float32), (r2.rand(R, 512) > 0.5).astype(np.float32)]
            return orig(x)
        cnet.forward = fwd
        try:
            f = F.create_objective(model, s["weights"], s["gradient"], it, stats)
            state = dict(learningRate=1e-4, alpha=0.9)
            for _ in range(5):
                F.rmsprop(f, s["weights"], state)
            torch.cuda.synchronize()
        finally:
            cnet.forward = orig
            cnet.drop_masks = None
            model["pnet"].drop_masks = None
        res[mode] = (s["weights"].cpu().numpy().copy(), s["gradient"].cpu().numpy().copy(), list(stats["pcls"]))
