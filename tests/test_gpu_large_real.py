"""BASELINE config 5 at the REAL widths of models/vgg_large.lua:5-22 (64/128/256/512 filters, 2-2-3-3 conv steps,
config/imagenet.lua:2-12: 200 classes, scales 48..384, 6x6 ROI pooling) against the CPU oracle:
  * the whole proposal net (forward, and backward with dense deltas) on a frame the oracle finishes in seconds;
  * one lossAndGradient step and one Detector:detect with the 200-class per-class NMS on the same frame size;
  * at the full 3x600x1000 layer shapes the size-independent adjoint property of the three conv kernels.
(The 512-filter layer shapes alone are in test_gpu_conv.py's case list.)"""
import numpy as np
import pytest

from util import assert_close, oracle_model
from test_gpu_conv import test_conv_full_size_adjoint as _adjoint
from test_gpu_model import _amplified_weights, _compare_gradient, _masks, check_detect, check_loss_and_gradient, check_pnet_forward_backward

pytestmark = pytest.mark.gpu
H, W = 128, 176      # the smallest frame class whose 8x11 last map still feeds the 7x7 anchor net


@pytest.fixture(scope="module")
def setup(F, O):
    cfg = dict(F.imgnet_cfg)
    model = F.vgg_large(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=11)
    om = oracle_model(O, cfg, model["layers"], model["anchor_nets"], model["class_layers"])
    nat = model["native"]
    assert O.param_count(om) == (nat.total_params, nat.pnet_params)
    assert [l["filters"] for l in model["layers"]] == [64, 128, 256, 512]
    assert [l["conv_steps"] for l in model["layers"]] == [2, 2, 3, 3]
    return dict(cfg=cfg, model=model, weights=weights, gradient=gradient, om=om, w=weights.cpu().numpy().copy())


def test_vgg_large_pnet_forward_backward(F, O, setup):
    s = setup
    rng = np.random.RandomState(2)
    img = F.synthetic_image(H, W, 4)
    pnet = s["model"]["pnet"]
    outs = check_pnet_forward_backward(F, O, s, img, _masks(rng, s["model"]), rng, what="vgg_large pnet")
    assert outs[-1].shape[0] == 512
    pnet.evaluate()
    outs = pnet.forward(img)
    want, _ = O.pnet_forward(s["om"], s["w"], img, False, None)
    for i, (o, w) in enumerate(zip(outs, want)):
        assert_close(o.numpy(), w, 1e-4, "vgg_large pnet eval output %d" % (i + 1))


def test_vgg_large_loss_and_gradient(F, O, setup):
    r = check_loss_and_gradient(F, O, setup, H, W, nimages=1, nrois=3, negatives=8)
    assert r["examples"] > 8


def test_vgg_large_detect_200_classes(F, O, setup):
    """Detector.lua:101-136 with class_count = 200: up to 200 per-class NMS problems per frame."""
    import torch
    s = setup
    nat = s["model"]["native"]
    w = _amplified_weights(nat, s["w"], 201, cls_gain=400.0)   # (the arg-max of 201 log-probs must pass p > 0.2)
    s["weights"].copy_(torch.from_numpy(w))
    try:
        r = check_detect(F, O, s["model"], s["om"], w, range(20, 26), H, W)
        assert r["winners"] > 0
        classes = sorted(set(x["class"] for x in r["got"]))
        print("vgg_large detect: frame %d, %d matches, %d candidates, %d winners in %d classes"
              % (r["seed"], r["matches"], r["candidates"], r["winners"], len(classes)))
        assert len(classes) > 1, "per-class NMS ran on a single class only"
    finally:
        s["weights"].copy_(torch.from_numpy(s["w"]))


FULL_1000x600 = [
    # vgg_large 3x600x1000 layer shapes: C, H, W, O, k, pad
    (64, 600, 1000, 64, 3, 1),     # b1c2
    (256, 75, 125, 512, 3, 1),     # b4c1
    (512, 75, 125, 512, 3, 1),     # b4c2 / b4c3
    (512, 38, 63, 256, 5, 0),      # 5x5 anchor net on the 512-plane map
]


@pytest.mark.parametrize("C_,H_,W_,O_,k,pad", FULL_1000x600)
def test_vgg_large_full_size_adjoint(F, C_, H_, W_, O_, k, pad):
    _adjoint(F, C_, H_, W_, O_, k, pad)
