"""Split-bf16 operand form of the 3x3 convolution (csrc/convx.hip, option "split_bf16"): fp32 tensors in and out, every
product formed from six exact bf16 x bf16 partial products on the bf16 matrix cores.  The claim under test: it is an fp32
convolution -- against the CPU oracle (fp64 accumulation) its error is within the tolerance of the fp32 matrix-core kernel
(|a-b| <= 1e-4 max(1,|b|)) AND comparable with that kernel's own error on the same inputs: per case at most
RMS_GATE (1.75) times its root-mean-square error, and over all cases of the file no larger on the geometric mean (MEAN_GATE;
round 6 measured 0.97 -- see the comment at RMS_GATE).  Both split forms (two fp16 planes / three bf16 planes) run every case."""
import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


def _dev(F, a):
    return F.DeviceTensor.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _option(F, name, value=None):
    import ctypes as C
    if value is None:
        v = C.c_int(0)
        F._lib.call("frcnn_get_option", name.encode(), C.byref(v))
        return v.value
    F._lib.call("frcnn_set_option", name.encode(), int(value))


@pytest.fixture
def both_forms(F):
    """-> run(fn): (result with the split form, result with the fp32 matrix-core kernel)."""
    before = _option(F, "split_bf16")

    def run(fn):
        _option(F, "split_bf16", 1)
        a = fn()
        _option(F, "split_bf16", 0)
        b = fn()
        _option(F, "split_bf16", 1)
        return a, b
    yield run
    _option(F, "split_bf16", before)


@pytest.fixture(params=[0, 1], ids=["bf16x6", "f16x3"])
def x3_form(F, request):
    """Both operand forms of the split launches: three bf16 planes / six partial products (exact split), two fp16 planes / three."""
    before = _option(F, "x3_f16")
    _option(F, "x3_f16", request.param)
    yield request.param
    _option(F, "x3_f16", before)


def _rms(a, want):
    return float(np.sqrt(np.mean((a.astype(np.float64) - want) ** 2)))


# The split forms must be no worse than the fp32 matrix-core kernel of conv.hip against the fp64 oracle.  Both accumulate in fp32 in
# different orders, so on one small shape the ratio of the two RMS errors is a random draw around 1 (round 6, MI355X, the 83 cases of
# this file: 0.68 ... 1.63, geometric mean 0.97; gpurun_out/exp2/convx_ratios.txt).  What is asserted (round 5
# allowed 2.0 per case and nothing overall): every case <= RMS_GATE, and over the whole file the geometric mean <= MEAN_GATE
# (test_zz_error_ratio_over_the_file) -- "no larger on average, never much larger".  MAX_GATE: the same for a maximum over ~10^5 outputs.
RMS_GATE = 1.75
MEAN_GATE = 1.05
RATIOS = []


def _gate(es, ed, what=""):
    r = es / max(ed, 1e-30)
    RATIOS.append(r)
    print('rms ratio split/fp32-mfma: %.3f %s' % (r, what))
    assert es <= RMS_GATE * ed + 1e-9, (es, ed, what)
STRESS_ROUNDS = 25    # repetitions of every full-size launch shape (the round-5 staging fault was timing dependent)
MAX_GATE = 1.5

SHAPES = [
    # C, H, W, O, pad      (C % 16 == 0 and O % 128 == 0: the shapes the split form takes)
    (64, 57, 100, 128, 1),      # b2 widths on a 57x100 map: ragged 5 x 25 tiles
    (128, 29, 50, 256, 1),      # split-K slabs
    (256, 38, 63, 512, 1),      # vgg_large widths
    (16, 9, 11, 128, 1),        # one chunk, a map smaller than a tile
    (48, 31, 45, 128, 0),       # valid convolution (anchor-net geometry)
    (384, 29, 50, 256, 0),      # the 3x3 anchor net on the last map
    (64, 57, 100, 64, 1),       # 64 filters: 64-filter blocks (1 x 4 waves of 64 x 32)
    (32, 23, 37, 192, 1),       # 192 = 3 x 64 filters
]


@pytest.mark.parametrize("C_,H,W,O_,pad", SHAPES)
def test_forward_is_an_fp32_convolution(F, O, both_forms, x3_form, C_, H, W, O_, pad):
    rng = np.random.RandomState(C_ + H)
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, 3, 3) * np.sqrt(2.0 / (9 * O_))).astype(np.float32)
    b = rng.randn(O_).astype(np.float32)
    want = O.conv2d_fwd(x, w, b, pad)
    dx, dw, db = _dev(F, x), _dev(F, w), _dev(F, b)

    def fn():
        out = F.DeviceTensor.empty(want.shape)
        F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, None, None, F.ptr(dw), F.ptr(db), O_, 3, pad, F.ptr(out),
                    F.stream_ptr())
        return out.numpy()
    split, direct = both_forms(fn)
    assert not np.array_equal(split, direct), "the option did not switch the kernel"
    assert_close(split, want, 1e-4, "split-bf16 conv fwd")
    es, ed = _rms(split, want), _rms(direct, want)
    _gate(es, ed)


def test_wide_dynamic_range(F, O, both_forms, x3_form):
    """Magnitudes spread over ~12 decades (log-normal): the three-way split keeps 24 significand bits of every operand
    whatever its exponent, so the error stays that of an fp32 product."""
    rng = np.random.RandomState(3)
    C_, H, W, O_, pad = 64, 23, 37, 128, 1
    x = (rng.randn(C_, H, W) * np.exp(3 * rng.randn(C_, H, W))).astype(np.float32)
    w = (rng.randn(O_, C_, 3, 3) * np.exp(2 * rng.randn(O_, C_, 3, 3)) * 0.01).astype(np.float32)
    want = O.conv2d_fwd(x, w, np.zeros(O_, np.float32), pad)
    dx, dw = _dev(F, x), _dev(F, w)

    def fn():
        out = F.DeviceTensor.empty(want.shape)
        F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, None, None, F.ptr(dw), None, O_, 3, pad, F.ptr(out), F.stream_ptr())
        return out.numpy()
    split, direct = both_forms(fn)
    # error relative to the sum of magnitudes (the natural scale of a dot product's rounding error)
    mag = O.conv2d_fwd(np.abs(x), np.abs(w), np.zeros(O_, np.float32), pad)
    rs = np.abs(split - want) / mag
    rd = np.abs(direct - want) / mag
    print("max error / sum of magnitudes: split %.3e  fp32 MFMA %.3e" % (rs.max(), rd.max()))
    assert rs.max() <= 4e-6, rs.max()
    assert rs.max() <= MAX_GATE * rd.max() + 1e-9, (rs.max(), rd.max())


def test_fused_activation_of_the_producing_layer(F, O, both_forms, x3_form):
    rng = np.random.RandomState(1)
    C_, H, W, O_, pad = 32, 21, 34, 128, 1
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, 3, 3) * 0.1).astype(np.float32)
    b = rng.randn(O_).astype(np.float32)
    a = np.float32(0.25)
    scale = (rng.rand(C_) > 0.4).astype(np.float32)
    for slope, sc in ((a, scale), (a, None), (None, scale)):
        act = x if slope is None else np.where(x > 0, x, slope * x)
        if sc is not None:
            act = act * sc[:, None, None]
        want = O.conv2d_fwd(act.astype(np.float32), w, b, pad)
        out = F.DeviceTensor.empty(want.shape)
        dx, dw, db = _dev(F, x), _dev(F, w), _dev(F, b)
        da = _dev(F, [slope]) if slope is not None else None
        ds = _dev(F, sc) if sc is not None else None
        F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, F.ptr(da) if da else None, F.ptr(ds) if ds else None,
                    F.ptr(dw), F.ptr(db), O_, 3, pad, F.ptr(out), F.stream_ptr())
        assert_close(out.numpy(), want, 1e-4, "split-bf16 conv fwd + act")


@pytest.mark.parametrize("C_,H,W,O_,pad", [(128, 57, 100, 64, 1), (256, 38, 63, 512, 1), (128, 31, 45, 48, 0),
                                            (64, 57, 100, 128, 1)])   # the last: M = 64 input channels (64-filter blocks)
def test_input_gradient(F, O, both_forms, x3_form, C_, H, W, O_, pad):
    """updateGradInput: M = C input channels (a multiple of 64), K = O filters (a multiple of 16)."""
    rng = np.random.RandomState(C_ * 7 + O_)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    g = rng.randn(O_, Ho, Wo).astype(np.float32)
    w = (rng.randn(O_, C_, 3, 3) * np.sqrt(2.0 / (9 * O_))).astype(np.float32)
    want = O.conv2d_bwd_input(g, w, pad, H, W)
    dg, dw = _dev(F, g), _dev(F, w)

    def fn():
        gin = F.DeviceTensor.empty((C_, H, W))
        F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, Ho, Wo, F.ptr(dw), C_, 3, pad, F.ptr(gin), 0, F.stream_ptr())
        first = gin.numpy()
        F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, Ho, Wo, F.ptr(dw), C_, 3, pad, F.ptr(gin), 1, F.stream_ptr())
        return first, gin.numpy()
    (split, split2), (direct, _) = both_forms(fn)
    assert not np.array_equal(split, direct)
    assert_close(split, want, 1e-4, "split-bf16 conv dgrad")
    assert_close(split2, 2 * want, 2e-4, "split-bf16 conv dgrad accumulate")
    _gate(_rms(split, want), _rms(direct, want))


def _tap_equal(got, tap, wmax, f16):
    if not f16:   # h + m + l == x: bit for bit
        return np.array_equal(got, tap)
    # two fp16 planes of w * 2^e (the tensor's largest magnitude just below 2^15): 22 significand bits, absolute floor at the
    # smallest fp16 subnormal
    return bool(np.all(np.abs(got.astype(np.float64) - tap) <= 2.0 ** -22 * np.abs(tap) + 2.0 ** -38 * wmax))


def test_exactness_properties_at_full_size(F, x3_form):
    """b2c2 at the benchmarked size (128 -> 128 @ 225x400), properties that need no oracle: conv(2x) == 2 conv(x) exactly in
    both forms (a power of two moves through either split untouched); with three bf16 planes the split of a value is exact
    (h + m + l == x), so a delta image returns the filter taps BIT FOR BIT; with two fp16 planes to 22 bits."""
    f16 = bool(x3_form)
    rng = np.random.RandomState(0)
    C_, H, W, O_ = 128, 225, 400, 128
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, 3, 3) * 0.05).astype(np.float32)
    dw = _dev(F, w)
    o1 = F.DeviceTensor.empty((O_, H, W)); o2 = F.DeviceTensor.empty((O_, H, W))
    dx1, dx2 = _dev(F, x), _dev(F, 2 * x)
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx1), C_, H, W, None, None, F.ptr(dw), None, O_, 3, 1, F.ptr(o1), F.stream_ptr())
    F._lib.call("frcnn_conv2d_forward", F.ptr(dx2), C_, H, W, None, None, F.ptr(dw), None, O_, 3, 1, F.ptr(o2), F.stream_ptr())
    assert np.array_equal(o2.numpy(), 2 * o1.numpy())
    d = np.zeros((C_, H, W), np.float32); d[5, 100, 200] = 1.0; d[77, 224, 399] = 1.0; d[100, 0, 0] = -2.0
    dd = _dev(F, d)
    F._lib.call("frcnn_conv2d_forward", F.ptr(dd), C_, H, W, None, None, F.ptr(dw), None, O_, 3, 1, F.ptr(o1), F.stream_ptr())
    r = o1.numpy()
    for ky in range(3):
        for kx in range(3):
            wmax = float(np.abs(w).max())
            assert _tap_equal(r[:, 100 + 1 - ky, 200 + 1 - kx], w[:, 5, ky, kx], wmax, f16)
            if ky >= 1 and kx >= 1:
                assert _tap_equal(r[:, 224 + 1 - ky, 399 + 1 - kx], w[:, 77, ky, kx], wmax, f16)
            if ky <= 1 and kx <= 1:
                assert _tap_equal(r[:, 0 + 1 - ky, 0 + 1 - kx], -2 * w[:, 100, ky, kx], 2 * wmax, f16)
    r[:, 99:102, 199:202] = 0; r[:, 223:225, 398:400] = 0; r[:, 0:2, 0:2] = 0
    assert not r.any()


@pytest.mark.parametrize("xs,ws,gs", [(0.0, 1.0, 0.0), (1e-30, 1.0, 1e10), (1e30, 1e-8, 1e-20), (1e-20, 1e-15, 1e20), (3e4, 7e3, 1.0)])
def test_two_plane_form_at_the_ends_of_the_range(F, O, xs, ws, gs):
    """The fp16 form scales each tensor by a power of two taken from its largest magnitude: an all-zero tensor, magnitudes near the
    ends of the fp32 range and products that would overflow fp16 many times over must all come out as the fp32 convolution
    (forward, input gradient and weight gradient; relative to the largest output; the scales are chosen so that the RESULTS are
    fp32 numbers -- the sum of the two tensors' exponents is not, which is why the kernels undo the scaling in two halves)."""
    before = _option(F, "x3_f16")
    _option(F, "x3_f16", 1)
    try:
        rng = np.random.RandomState(11)
        C_, H, W, O_, pad = 64, 19, 28, 128, 1
        x = (rng.randn(C_, H, W) * xs).astype(np.float32)
        w = (rng.randn(O_, C_, 3, 3) * ws).astype(np.float32)
        g = (rng.randn(O_, H, W) * gs).astype(np.float32)
        dx, dw, dg = _dev(F, x), _dev(F, w), _dev(F, g)
        out = F.DeviceTensor.empty((O_, H, W)); gin = F.DeviceTensor.empty((C_, H, W)); gw = F.DeviceTensor.zeros((O_, C_, 3, 3))
        F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, None, None, F.ptr(dw), None, O_, 3, pad, F.ptr(out), F.stream_ptr())
        F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, H, W, F.ptr(dw), C_, 3, pad, F.ptr(gin), 0, F.stream_ptr())
        F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, H, W, None, None, F.ptr(dg), O_, 3, pad, F.ptr(gw), None, F.stream_ptr())
        x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
        for got, name in ((out.numpy(), "forward"), (gin.numpy(), "input gradient"), (gw.numpy(), "weight gradient")):
            assert np.isfinite(got).all(), "%s: not finite" % name
        if xs == 0.0:
            assert not out.numpy().any() and not gin.numpy().any() and not gw.numpy().any()
            return
        # references in fp64 by the definition (small shape)
        xp = np.pad(x64, ((0, 0), (1, 1), (1, 1)))
        ref_y = np.zeros((O_, H, W)); ref_gw = np.zeros((O_, C_, 3, 3))
        for ky in range(3):
            for kx in range(3):
                patch = xp[:, ky:ky + H, kx:kx + W]
                ref_y += np.einsum("oc,chw->ohw", w64[:, :, ky, kx], patch)
                ref_gw[:, :, ky, kx] = np.einsum("ohw,chw->oc", g64, patch)
        gp = np.pad(g64, ((0, 0), (1, 1), (1, 1)))
        ref_gin = np.zeros((C_, H, W))
        for ky in range(3):
            for kx in range(3):
                ref_gin += np.einsum("oc,ohw->chw", w64[:, :, 2 - ky, 2 - kx], gp[:, ky:ky + H, kx:kx + W])
        for got, ref, name in ((out.numpy(), ref_y, "forward"), (gin.numpy(), ref_gin, "input gradient"), (gw.numpy(), ref_gw, "weight gradient")):
            scale = np.abs(ref).max()
            assert scale > 0
            assert np.abs(got.astype(np.float64) - ref).max() <= 1e-5 * scale, (name, np.abs(got - ref).max() / scale)
    finally:
        _option(F, "x3_f16", before)


@pytest.mark.parametrize("C_,H,W,O_,pad", [(64, 57, 100, 128, 1), (128, 29, 50, 64, 1), (256, 38, 63, 512, 1), (64, 9, 11, 64, 1),
                                            (64, 31, 45, 64, 0), (64, 75, 125, 64, 1), (128, 37, 250, 64, 1)])
def test_weight_gradient(F, O, both_forms, x3_form, C_, H, W, O_, pad):
    """accGradParameters (csrc/wgradx.hip): both operands split while they are staged into pixel-contiguous planes."""
    rng = np.random.RandomState(C_ * 13 + O_)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    x = rng.randn(C_, H, W).astype(np.float32)
    g = (rng.randn(O_, Ho, Wo) / np.sqrt(Ho * Wo)).astype(np.float32)
    gw_want, gb_want = O.conv2d_bwd_weight(x, g, 3, 3, pad)
    dx, dg = _dev(F, x), _dev(F, g)

    def fn():
        gw = F.DeviceTensor.zeros((O_, C_, 3, 3)); gb = F.DeviceTensor.zeros((O_,))
        F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, H, W, None, None, F.ptr(dg), O_, 3, pad, F.ptr(gw), F.ptr(gb),
                    F.stream_ptr())
        first = gw.numpy()
        F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, H, W, None, None, F.ptr(dg), O_, 3, pad, F.ptr(gw), F.ptr(gb),
                    F.stream_ptr())
        return first, gw.numpy()
    (split, split2), (direct, _) = both_forms(fn)
    assert not np.array_equal(split, direct), "the option did not switch the kernel"
    assert_close(split, gw_want, 1e-4, "split-bf16 conv wgrad")
    assert_close(split2, 2 * gw_want, 2e-4, "split-bf16 conv wgrad accumulates")
    _gate(_rms(split, gw_want), _rms(direct, gw_want))


@pytest.mark.parametrize("W", [34, 36, 48])   # 36 / 48: the 16-byte-segment loader (W % 4 = 0), 34: the per-element one
def test_weight_gradient_with_fused_activation(F, O, x3_form, W):
    rng = np.random.RandomState(2)
    C_, H, O_, pad = 64, 21, 64, 1
    x = rng.randn(C_, H, W).astype(np.float32)
    g = (rng.randn(O_, H, W) * 0.05).astype(np.float32)
    a = np.float32(0.25)
    scale = (rng.rand(C_) > 0.4).astype(np.float32)
    act = (np.where(x > 0, x, a * x) * scale[:, None, None]).astype(np.float32)
    gw_want, _ = O.conv2d_bwd_weight(act, g, 3, 3, pad)
    gw = F.DeviceTensor.zeros((O_, C_, 3, 3)); gb = F.DeviceTensor.zeros((O_,))
    dx, dg, da, ds = _dev(F, x), _dev(F, g), _dev(F, [a]), _dev(F, scale)
    F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, H, W, F.ptr(da), F.ptr(ds), F.ptr(dg), O_, 3, pad, F.ptr(gw),
                F.ptr(gb), F.stream_ptr())
    assert_close(gw.numpy(), gw_want, 1e-4, "split-bf16 conv wgrad + act")


@pytest.mark.parametrize("C_,H,W,O_,k", [(384, 29, 50, 256, 5), (384, 29, 50, 256, 7), (32, 17, 23, 128, 5), (16, 9, 30, 128, 7)])
def test_anchor_net_kernel_sizes(F, O, both_forms, x3_form, C_, H, W, O_, k):
    """The 5x5 and 7x7 valid convolutions of the anchor nets (models/model_utilities.lua:31, vgg_small.lua:13-14) in the
    same form: forward and input gradient."""
    rng = np.random.RandomState(C_ + k)
    x = rng.randn(C_, H, W).astype(np.float32)
    w = (rng.randn(O_, C_, k, k) * np.sqrt(2.0 / (k * k * O_))).astype(np.float32)
    b = rng.randn(O_).astype(np.float32)
    want = O.conv2d_fwd(x, w, b, 0)
    dx, dw, db = _dev(F, x), _dev(F, w), _dev(F, b)

    def fwd():
        out = F.DeviceTensor.empty(want.shape)
        F._lib.call("frcnn_conv2d_forward", F.ptr(dx), C_, H, W, None, None, F.ptr(dw), F.ptr(db), O_, k, 0, F.ptr(out), F.stream_ptr())
        return out.numpy()
    split, direct = both_forms(fwd)
    assert not np.array_equal(split, direct), "the option did not switch the kernel"
    assert_close(split, want, 1e-4, "split-bf16 %dx%d conv fwd" % (k, k))
    _gate(_rms(split, want), _rms(direct, want))
    if C_ % 128 == 0 and O_ % 16 == 0:   # input gradient: M = C
        Ho, Wo = H - k + 1, W - k + 1
        g = rng.randn(O_, Ho, Wo).astype(np.float32)
        gwant = O.conv2d_bwd_input(g, w, 0, H, W)
        dg = _dev(F, g)

        def bwd():
            gin = F.DeviceTensor.empty((C_, H, W))
            F._lib.call("frcnn_conv2d_backward_input", F.ptr(dg), O_, Ho, Wo, F.ptr(dw), C_, k, 0, F.ptr(gin), 0, F.stream_ptr())
            return gin.numpy()
        s2, d2 = both_forms(bwd)
        assert not np.array_equal(s2, d2)
        assert_close(s2, gwant, 1e-4, "split-bf16 %dx%d conv dgrad" % (k, k))


MODEL_LAYERS = [  # name, Cin, H, W, Cout, pad -- the 3x3 launches of a vgg_small 800x450 step (both block shapes, split-K, valid)
    ("b2c1", 64, 225, 400, 128, 1), ("b2c2", 128, 225, 400, 128, 1), ("b3c1", 128, 113, 200, 256, 1),
    ("b3c2", 256, 113, 200, 256, 1), ("b4c1", 256, 57, 100, 384, 1), ("b4c2", 384, 57, 100, 384, 1),
    ("a1", 256, 57, 100, 256, 0), ("a2", 384, 29, 50, 256, 0),
]


@pytest.mark.parametrize("name,C_,H,W,O_,pad", MODEL_LAYERS)
def test_model_layer_shapes_full_size(F, x3_form, name, C_, H, W, O_, pad):
    """Every 3x3 launch shape of the benchmarked step at FULL size -- forward without and with the fused input activation
    (PReLU slope + dropout scale: the <SLOPE, SCALE> instantiations), and the input gradient -- against the fp32 matrix-core
    kernel on the same inputs, STRESS_ROUNDS (25) times each (round 5 ran three: VERDICT r5 next 4).  The small shapes above did not catch a timing-dependent fault of the
    activation path that only showed with hundreds of blocks in flight (round 5: packed multiplies, convx.hip store_patch);
    the oracle is too slow for these sizes, the fp32 kernel (itself oracle-checked above and in test_gpu_conv.py) is not."""
    rng = np.random.RandomState(1)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    x, g = _dev(F, rng.randn(C_, H, W).astype(np.float32)), _dev(F, rng.randn(O_, Ho, Wo).astype(np.float32))
    w, b = _dev(F, (rng.randn(O_, C_, 3, 3) * 0.05).astype(np.float32)), _dev(F, rng.randn(O_).astype(np.float32))
    slope, scale = _dev(F, [np.float32(0.25)]), _dev(F, (rng.rand(C_) > 0.4).astype(np.float32))
    s = F.stream_ptr()

    def fwd(act):
        out = F.DeviceTensor.empty((O_, Ho, Wo))
        F._lib.call("frcnn_conv2d_forward", F.ptr(x), C_, H, W, F.ptr(slope) if act else None, F.ptr(scale) if act else None,
                    F.ptr(w), F.ptr(b), O_, 3, pad, F.ptr(out), s)
        return out.numpy()

    def dgrad():
        gin = F.DeviceTensor.empty((C_, H, W))
        F._lib.call("frcnn_conv2d_backward_input", F.ptr(g), O_, Ho, Wo, F.ptr(w), C_, 3, pad, F.ptr(gin), 0, s)
        return gin.numpy()
    before = _option(F, "split_bf16", 0)
    try:
        want = [fwd(0), fwd(1), dgrad()]
        _option(F, "split_bf16", 1)
        for _ in range(STRESS_ROUNDS):
            for got, ref, what in zip([fwd(0), fwd(1), dgrad()], want, ("forward", "forward + fused activation", "input gradient")):
                assert_close(got, ref, 2e-4, "%s %s, split form vs fp32 matrix-core kernel" % (name, what))
    finally:
        _option(F, "split_bf16", before)


@pytest.mark.parametrize("k", [5, 7])
def test_anchor_net_shapes_full_size(F, x3_form, k):
    """The 5x5 / 7x7 anchor nets of vgg_small at the benchmarked size (384 -> 256 on the 29x50 map, valid): since round 5 their
    forward pass takes the split form by default (8x16-pixel tiles, 240 / 308-position patch, up to 24 K splits + fold).
    Against the fp32 matrix-core kernel, three runs."""
    rng = np.random.RandomState(k)
    C_, H, W, O_ = 384, 29, 50, 256
    Ho, Wo = H - k + 1, W - k + 1
    x = _dev(F, rng.randn(C_, H, W).astype(np.float32))
    w, b = _dev(F, (rng.randn(O_, C_, k, k) * 0.03).astype(np.float32)), _dev(F, rng.randn(O_).astype(np.float32))

    def fwd():
        out = F.DeviceTensor.empty((O_, Ho, Wo))
        F._lib.call("frcnn_conv2d_forward", F.ptr(x), C_, H, W, None, None, F.ptr(w), F.ptr(b), O_, k, 0, F.ptr(out), F.stream_ptr())
        return out.numpy()
    before = _option(F, "split_bf16", 0)
    try:
        want = fwd()
        _option(F, "split_bf16", 1)
        for _ in range(3):
            got = fwd()
            assert not np.array_equal(got, want), "the option did not switch the kernel"
            assert_close(got, want, 2e-4, "%dx%d anchor net, split form vs fp32 matrix-core kernel" % (k, k))
    finally:
        _option(F, "split_bf16", before)


def test_zz_error_ratio_over_the_file(F):
    """Runs last (file order): over every comparison of this file the split forms' RMS error against the fp64 oracle is, on the
    geometric mean, no larger than the fp32 matrix-core kernel's (MEAN_GATE leaves 5 % for the draw of one run)."""
    if len(RATIOS) < 20:
        pytest.skip("only %d comparisons ran (a filtered session)" % len(RATIOS))
    gm = float(np.exp(np.mean(np.log(np.maximum(RATIOS, 1e-12)))))
    print("split / fp32-MFMA RMS error over %d comparisons: geometric mean %.3f, min %.3f, max %.3f" % (len(RATIOS), gm, min(RATIOS), max(RATIOS)))
    assert gm <= MEAN_GATE, (gm, len(RATIOS))
