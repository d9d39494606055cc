"""nn.SpatialConvolution forward / updateGradInput / accGradParameters: HIP implicit-GEMM kernels
(fp32 MFMA) through the C ABI vs the CPU oracle (fp64 accumulation).  Tolerance:
|a-b| <= 1e-4 * max(1, |b|)."""
import ctypes as C

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu

CASES = [
    # C, H, W, O, k, pad
    (3, 37, 53, 64, 3, 1),      # first layer shape class (Cin = 3, M = 64)
    (16, 29, 50, 40, 3, 1),     # 29x50 map (block-4 geometry), M not a multiple of 32
    (64, 57, 100, 128, 3, 1),
    (24, 19, 23, 18, 1, 0),     # 1x1 head (M = 18)
    (20, 29, 50, 32, 3, 0),     # valid 3x3 head
    (10, 29, 50, 24, 5, 0),     # valid 5x5 head
    (6, 29, 50, 24, 7, 0),      # valid 7x7 head
    (130, 15, 17, 130, 3, 1),   # channel counts that are not multiples of the chunk / tile
    (256, 38, 63, 512, 3, 1),   # vgg_large.lua:9 widths: 256 -> 512 ...
    (512, 38, 63, 512, 3, 1),   # ... and 512 -> 512 (M = 512, K = 4608)
    (512, 38, 63, 256, 7, 0),   # 7x7 anchor net on the 512-plane map (K = 25 088)
]


def _dev(F, a):
    return F.DeviceTensor.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


@pytest.mark.parametrize("C_,H,W,O_,k,pad", CASES)
def test_conv_forward(F, O, C_, H, W, O_, k, pad):
    rng = np.random.RandomState(C_ * 1000 + k)
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, k, k) * np.sqrt(2.0 / (k * k * O_))).astype(np.float32)
    b = rng.randn(O_).astype(np.float32)
    want = O.conv2d_fwd(x, w, b, pad)
    out = F.DeviceTensor.empty(want.shape)
    dx, dw, db = _dev(F, x), _dev(F, w), _dev(F, b)
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, None, None, F.ptr(dw), F.ptr(db),
                O_, k, pad, F.ptr(out), F.stream_ptr())
    assert_close(out.numpy(), want, 1e-4, "conv fwd")


def test_conv_forward_fused_activation(F, O):
    """The producing layer's PReLU + SpatialDropout scale are applied by the consumer's loader."""
    rng = np.random.RandomState(1)
    C_, H, W, O_, k, pad = 12, 21, 34, 20, 3, 1
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, k, k) * 0.2).astype(np.float32)
    b = rng.randn(O_).astype(np.float32)
    a = np.float32(0.25)
    scale = (rng.rand(C_) > 0.4).astype(np.float32)
    act = np.where(x > 0, x, a * x) * scale[:, None, None]
    want = O.conv2d_fwd(act, w, b, pad)
    out = F.DeviceTensor.empty(want.shape)
    dx, da, ds, dw, db = _dev(F, x), _dev(F, [a]), _dev(F, scale), _dev(F, w), _dev(F, b)
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, F.ptr(da), F.ptr(ds),
                F.ptr(dw), F.ptr(db), O_, k, pad, F.ptr(out), F.stream_ptr())
    assert_close(out.numpy(), want, 1e-4, "conv fwd + act")


@pytest.mark.parametrize("C_,H,W,O_,k,pad", CASES[1:])
def test_conv_backward_input(F, O, C_, H, W, O_, k, pad):
    rng = np.random.RandomState(C_ * 77 + k)
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    g = rng.randn(O_, Ho, Wo).astype(np.float32)
    w = (rng.randn(O_, C_, k, k) * np.sqrt(2.0 / (k * k * O_))).astype(np.float32)
    want = O.conv2d_bwd_input(g, w, pad, H, W)
    gin = F.DeviceTensor.empty((C_, H, W))
    dg, dw = _dev(F, g), _dev(F, w)
    F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, Ho, Wo, F.ptr(dw), C_, k, pad, F.ptr(gin), 0,
                F.stream_ptr())
    assert_close(gin.numpy(), want, 1e-4, "conv dgrad")
    # accumulate flag: gin += ...
    F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, Ho, Wo, F.ptr(dw), C_, k, pad, F.ptr(gin), 1,
                F.stream_ptr())
    assert_close(gin.numpy(), 2 * want, 2e-4, "conv dgrad accumulate")


@pytest.mark.parametrize("C_,H,W,O_,k,pad", CASES)
def test_conv_backward_weight(F, O, C_, H, W, O_, k, pad):
    rng = np.random.RandomState(C_ * 13 + k)
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    x = rng.randn(C_, H, W).astype(np.float32)
    g = (rng.randn(O_, Ho, Wo) / np.sqrt(Ho * Wo)).astype(np.float32)
    gw_want, gb_want = O.conv2d_bwd_weight(x, g, k, k, pad)
    gw = F.DeviceTensor.zeros((O_, C_, k, k)); gb = F.DeviceTensor.zeros((O_,))
    dx, dg = _dev(F, x), _dev(F, g)
    F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, H, W, None, None, F.ptr(dg), O_, k, pad,
                F.ptr(gw), F.ptr(gb), F.stream_ptr())
    assert_close(gw.numpy(), gw_want, 1e-4, "conv wgrad")
    assert_close(gb.numpy(), gb_want, 1e-4, "conv bias grad")


def test_conv_full_size_linearity(F):
    """b2c1 geometry at full size (64->128 @ 225x400): conv(2x) == 2*conv(x) exactly and a delta
    image reproduces the filter taps -- properties that need no oracle run at this size."""
    rng = np.random.RandomState(0)
    C_, H, W, O_, k, pad = 64, 225, 400, 128, 3, 1
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, k, k) * 0.05).astype(np.float32)
    dw = _dev(F, w)
    import ctypes
    opt = ctypes.c_int(0)
    F._lib.call("frcnn_get_option", b"x3_f16", ctypes.byref(opt))
    f16 = bool(opt.value)   # the two-plane fp16 operand form (convx.hip) is selected
    o1 = F.DeviceTensor.empty((O_, H, W)); o2 = F.DeviceTensor.empty((O_, H, W))
    dx1, dx2 = _dev(F, x), _dev(F, 2 * x)
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx1), C_, H, W, None, None, F.ptr(dw), None, O_, k, pad, F.ptr(o1), F.stream_ptr())
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx2), C_, H, W, None, None, F.ptr(dw), None, O_, k, pad, F.ptr(o2), F.stream_ptr())
    a1, a2 = o1.numpy(), o2.numpy()
    assert np.array_equal(a2, 2 * a1)
    d = np.zeros((C_, H, W), np.float32); d[5, 100, 200] = 1.0
    dd = _dev(F, d)
    F._lib.call("frcnn_conv2d_forward", F.ptr(dd), C_, H, W, None, None, F.ptr(dw), None, O_, k, pad, F.ptr(o1), F.stream_ptr())
    r = o1.numpy()
    for ky in range(3):
        for kx in range(3):
            got, tap = r[:, 100 + 1 - ky, 200 + 1 - kx], w[:, 5, ky, kx]
            if f16:   # two fp16 planes hold 22 significand bits of a weight scaled to the tensor's largest magnitude
                assert np.all(np.abs(got.astype(np.float64) - tap) <= 2.0 ** -22 * np.abs(tap) + 2.0 ** -38 * np.abs(w).max())
            else:     # the three bf16 planes sum to the fp32 tap exactly
                assert np.array_equal(got, tap)
    r[:, 99:102, 199:202] = 0
    assert np.abs(r).max() == 0


FULL_LAYERS = [
    # vgg_small 800x450 shapes: C, H, W, O, k, pad
    (128, 225, 400, 128, 3, 1),   # b2c2: one wave of 1440 blocks, no split
    (256, 57, 100, 384, 3, 1),    # b4c1: split-K slabs + reduce
    (384, 29, 50, 256, 7, 0),     # a4: 7x7 anchor net, 24 splits
    (256, 55, 98, 18, 1, 0),      # 1x1 head, M = 18
    (64, 300, 500, 64, 3, 1),     # vgg_large.lua block 1 widths on a 500x300 map: 64-filter blocks of the split form, both ways
    (64, 225, 400, 128, 3, 1),    # b2c1: 128-filter blocks forward, 64-filter blocks for the input gradient
]


@pytest.mark.parametrize("C_,H,W,O_,k,pad", FULL_LAYERS)
def test_conv_full_size_adjoint(F, C_, H, W, O_, k, pad):
    """At BASELINE.json's full layer sizes the three kernels must be each other's adjoints:
    <conv(x; w), g> == <x, updateGradInput(g; w)> == <w, accGradParameters(x, g)>  (a size-independent property
    that exercises the split-K slabs, the XCD block order and the tile edges without an oracle run)."""
    rng = np.random.RandomState(k * 1000 + C_)
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    x = rng.randn(C_, H, W).astype(np.float32)
    g = rng.randn(O_, Ho, Wo).astype(np.float32)
    w = (rng.randn(O_, C_, k, k) / np.sqrt(C_ * k * k)).astype(np.float32)
    dx, dg, dw = _dev(F, x), _dev(F, g), _dev(F, w)
    out = F.DeviceTensor.empty((O_, Ho, Wo)); gin = F.DeviceTensor.empty((C_, H, W)); gw = F.DeviceTensor.zeros((O_, C_, k, k))
    s = F.stream_ptr()
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, None, None, F.ptr(dw), None, O_, k, pad, F.ptr(out), s)
    F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, Ho, Wo, F.ptr(dw), C_, k, pad, F.ptr(gin), 0, s)
    F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, H, W, None, None, F.ptr(dg), O_, k, pad, F.ptr(gw), None, s)
    a = float(np.dot(out.numpy().astype(np.float64).ravel(), g.astype(np.float64).ravel()))
    b = float(np.dot(gin.numpy().astype(np.float64).ravel(), x.astype(np.float64).ravel()))
    c = float(np.dot(gw.numpy().astype(np.float64).ravel(), w.astype(np.float64).ravel()))
    # each inner product sums ~1e7 terms of size ~1: its own fp32 rounding noise is ~1e-4 of sqrt(#terms)
    scale = np.sqrt(float(out.numel())) * 4.0
    assert abs(a - b) <= 1e-4 * scale and abs(a - c) <= 1e-4 * scale, (a, b, c, scale)
