"""vgg_large topology (models/vgg_large.lua:5-22: 2-2-3-3 conv steps) on the GPU against the
oracle, in a narrow variant the oracle finishes in seconds, with the 7x7 ROI pooling that README.md:19 lists as an
experiment (kh, kw are parameters: imagenet.lua:9 says 6x6)."""
import numpy as np
import pytest

from util import assert_close, oracle_model
from test_gpu_model import _compare_gradient, _masks

pytestmark = pytest.mark.gpu
H, W = 120, 168

LARGE_NARROW = [
    dict(filters=8, kW=3, kH=3, padW=1, padH=1, dropout=0.0, conv_steps=2),
    dict(filters=16, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
    dict(filters=24, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=3),
    dict(filters=40, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=3),
]
HEADS = [dict(kW=3, n=24, input=3), dict(kW=3, n=24, input=4), dict(kW=5, n=24, input=4), dict(kW=7, n=24, input=4)]
CLS = [dict(n=48, dropout=0.5, batch_norm=True), dict(n=32, dropout=0.5)]


@pytest.fixture(scope="module")
def setup(F, O):
    cfg = dict(F.imgnet_cfg)
    cfg["roi_pooling"] = dict(kw=7, kh=7)
    model = F.create_model(cfg, LARGE_NARROW, HEADS, CLS)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=3)
    om = oracle_model(O, cfg, LARGE_NARROW, HEADS, CLS)
    assert O.param_count(om) == (model["native"].total_params, model["native"].pnet_params)
    return dict(cfg=cfg, model=model, weights=weights, gradient=gradient, om=om, w=weights.cpu().numpy().copy())


def test_large_topology_pnet(F, O, setup):
    s = setup
    rng = np.random.RandomState(1)
    img = F.synthetic_image(H, W, 2)
    masks = _masks(rng, s["model"])
    pnet = s["model"]["pnet"]
    pnet.training()
    pnet.drop_masks = masks
    try:
        outs = pnet.forward(img)
        want, st = O.pnet_forward(s["om"], s["w"], img, True, masks)
        assert [o.shape for o in outs] == [w.shape for w in want]
        for i, (o, w) in enumerate(zip(outs, want)):
            assert_close(o.numpy(), w, 1e-4, "pnet output %d" % (i + 1))
        deltas = [(rng.randn(*w.shape) / np.sqrt(w.size)).astype(np.float32) for w in want]
        g_want = np.zeros_like(s["w"])
        O.pnet_backward(s["om"], s["w"], st, deltas, g_want)
        s["gradient"].zero_()
        dev = pnet.delta_outputs(zero=True)
        for d, h in zip(dev, deltas):
            d.copy_from_numpy(h)
        pnet.backward(img, dev)
        _compare_gradient(s["model"]["native"], s["gradient"].cpu().numpy(), g_want, lo=0, hi=s["model"]["native"].pnet_params)
    finally:
        pnet.drop_masks = None


def test_roi_pool_7x7_and_cnet(F, O, setup):
    """ROI windows -> 7x7 adaptive max pooling (one launch) -> cnet forward/backward, R = 300 (imagenet.lua:12)."""
    s = setup
    model, cfg = s["model"], s["cfg"]
    rng = np.random.RandomState(4)
    img = F.synthetic_image(H, W, 3)
    pnet, cnet = model["pnet"], model["cnet"]
    pnet.evaluate()
    fm = pnet.forward(img)[-1]
    C_, fh, fw = fm.shape
    R = 300
    x0 = rng.uniform(0, W - 20, R); y0 = rng.uniform(0, H - 20, R)
    rects = np.stack([x0, y0, np.minimum(x0 + rng.uniform(8, 100, R), W), np.minimum(y0 + rng.uniform(8, 100, R), H)], 1)
    loc = F.Localizer(pnet.outnode.children[4])
    wins = F.roi_windows(rects, loc, fh, fw)
    layers = O.model_localizer_layers(s["om"], 5)
    want_wins = np.array([O.extract_roi_window(layers, r, fh, fw) for r in rects], dtype=np.int32)
    assert np.array_equal(wins, want_wins)
    kh = kw = 7
    D = C_ * kh * kw
    out = F.DeviceTensor.empty((R, D)); idx = F.DeviceTensor.empty((R, D), np.int32)
    dw = F.DeviceTensor.from_numpy(wins)
    F._lib.call("frcnn_roi_pool_forward", F.ptr(fm), C_, fh, fw, F.ptr(dw), R, kh, kw, F.ptr(out), F.ptr(idx), F.stream_ptr())
    fm_h = fm.numpy()
    want = np.stack([O.adaptive_max_pool_fwd(fm_h, w, kh, kw)[0].reshape(-1) for w in wins])
    assert np.array_equal(out.numpy(), want)   # a gather of existing values: exact
    # cnet on the pooled rows
    cnet.training()
    cm = [(rng.rand(R, 48) > 0.5).astype(np.float32), (rng.rand(R, 32) > 0.5).astype(np.float32)]
    cnet.drop_masks = cm
    native = model["native"]
    bn0 = native.bn_running.cpu().numpy().copy()
    try:
        bbox, cls = cnet.forward(out)
        bn_o = bn0.copy()
        wb, wc, st = O.cnet_forward(s["om"], s["w"], want, True, cm, bn_o)
        assert_close(bbox.numpy(), wb, 1e-4, "cnet bbox")
        assert_close(cls.numpy(), wc, 1e-4, "cnet cls")
        gb = (rng.randn(R, 4) / R).astype(np.float32); gc = (rng.randn(*wc.shape) / R).astype(np.float32)
        g_want = np.zeros_like(s["w"])
        gx_want = O.cnet_backward(s["om"], s["w"], st, gb, gc, g_want, D)
        s["gradient"].zero_()
        gx = cnet.backward(out, [F.DeviceTensor.from_numpy(gb), F.DeviceTensor.from_numpy(gc)])
        assert_close(gx.numpy(), gx_want, 1e-4, "cnet gradInput")
        _compare_gradient(native, s["gradient"].cpu().numpy(), g_want, lo=native.pnet_params, hi=native.total_params)
    finally:
        cnet.drop_masks = None
        import torch
        native.bn_running.copy_(torch.from_numpy(bn0))
