"""Pool / activation / ROI pooling / Linear / optimiser kernels through the C ABI vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


def _dev(F, a, dt=np.float32):
    return F.DeviceTensor.from_numpy(np.ascontiguousarray(a, dtype=dt))


@pytest.mark.parametrize("C_,H,W", [(5, 9, 13), (8, 57, 100), (3, 450, 800), (4, 2, 2), (2, 3, 2)])
def test_maxpool_act_forward_backward(F, O, C_, H, W):
    rng = np.random.RandomState(H * W)
    x = rng.randn(C_, H, W).astype(np.float32)
    a = np.float32(0.25)
    scale = (rng.rand(C_) > 0.3).astype(np.float32) if C_ > 3 else None
    act = np.where(x > 0, x, a * x) * (scale[:, None, None] if scale is not None else 1)
    want, widx = O.maxpool_fwd(act)
    Ho, Wo = want.shape[1:]
    out = F.DeviceTensor.empty(want.shape); idx = F.DeviceTensor.empty(want.shape, np.uint8)
    ds = _dev(F, scale) if scale is not None else None
    da = _dev(F, [a])
    dx = _dev(F, x)
    F._lib.call("frcnn_maxpool_act_forward", F.ptr(dx), C_, H, W, F.ptr(da), F.ptr(ds), F.ptr(out), F.ptr(idx), F.stream_ptr())
    got = out.numpy()
    assert np.array_equal(got, want)
    code = idx.numpy().astype(np.int64)
    oy, ox = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
    flat = (oy[None] * 2 + code // 2) * W + (ox[None] * 2 + code % 2)
    dropped = np.zeros(C_, bool) if scale is None else scale == 0
    assert np.array_equal(flat[~dropped], widx[~dropped])   # argmax (no ties in random data)
    # backward: gx = unpool(g) * scale * prelu'(x), bias/slope gradients
    g = rng.randn(*want.shape).astype(np.float32)
    gact = O.maxpool_bwd(g, flat.astype(np.int32), H, W) * (scale[:, None, None] if scale is not None else 1)
    gx_want = np.where(x > 0, gact, a * gact).astype(np.float32)
    gb_want = gx_want.reshape(C_, -1).astype(np.float64).sum(1)
    ga_want = float((np.where(x > 0, 0, x.astype(np.float64) * gact)).sum())
    gx = F.DeviceTensor.empty(x.shape); gb = F.DeviceTensor.zeros((C_,)); ga = F.DeviceTensor.zeros((1,))
    dg = _dev(F, g)
    F._lib.call("frcnn_maxpool_act_backward", F.ptr(dg), F.ptr(idx), F.ptr(dx), C_, H, W, F.ptr(da), F.ptr(ds),
                F.ptr(gx), F.ptr(gb), F.ptr(ga), F.stream_ptr())
    assert np.array_equal(gx.numpy(), gx_want)
    assert_close(gb.numpy(), gb_want, 1e-4, "bias grad")
    assert_close(ga.numpy()[0], ga_want, 1e-4, "slope grad")


def test_act_backward(F):
    rng = np.random.RandomState(2)
    C_, hw = 7, 1234
    x = rng.randn(C_, hw).astype(np.float32); g = rng.randn(C_, hw).astype(np.float32)
    a = np.float32(-0.3)  # negative slope: the sign of x cannot be recovered from y, x itself must be used
    gx_want = np.where(x > 0, g, a * g).astype(np.float32)
    dg = _dev(F, g)
    gb = F.DeviceTensor.zeros((C_,)); ga = F.DeviceTensor.zeros((1,))
    dx, da = _dev(F, x), _dev(F, [a])
    F._lib.call("frcnn_act_backward", F.ptr(dg), F.ptr(dx), C_, hw, F.ptr(da), None, F.ptr(dg), F.ptr(gb),
                F.ptr(ga), F.stream_ptr())
    assert np.array_equal(dg.numpy(), gx_want)
    assert_close(gb.numpy(), gx_want.astype(np.float64).sum(1), 1e-4)
    assert_close(ga.numpy()[0], float(np.where(x > 0, 0, x.astype(np.float64) * g).sum()), 1e-4)


def test_roi_pool_forward_backward(F, O):
    rng = np.random.RandomState(4)
    C_, H, W, kh, kw = 24, 29, 50, 6, 6
    fmap = rng.randn(C_, H, W).astype(np.float32)
    wins = np.array([[1, 29, 1, 50], [5, 15, 5, 15], [1, 12, 1, 7], [26, 29, 48, 50], [1, 3, 1, 3], [7, 7, 9, 9],
                     [2, 4, 10, 30], [29, 29, 1, 50]], dtype=np.int32)  # incl. windows smaller than 6x6
    R = len(wins)
    out = F.DeviceTensor.empty((R, C_ * kh * kw)); idx = F.DeviceTensor.empty((R, C_ * kh * kw), np.int32)
    dfm = _dev(F, fmap)
    dwins = _dev(F, wins, np.int32)
    F._lib.call("frcnn_roi_pool_forward", F.ptr(dfm), C_, H, W, F.ptr(dwins), R, kh, kw, F.ptr(out),
                F.ptr(idx), F.stream_ptr())
    got, gidx = out.numpy(), idx.numpy()
    gmap_want = np.zeros((C_, H, W), dtype=np.float32)
    gout = rng.randn(R, C_ * kh * kw).astype(np.float32)
    for r in range(R):
        wo, wi = O.adaptive_max_pool_fwd(fmap, wins[r], kh, kw)
        assert np.array_equal(got[r], wo.ravel())
        assert np.array_equal(gidx[r], wi.ravel())
        O.adaptive_max_pool_bwd(gmap_want, gout[r].reshape(C_, kh, kw), wi)
    gmap = F.DeviceTensor.zeros((C_, H, W))
    dgout = _dev(F, gout)
    F._lib.call("frcnn_roi_pool_backward", F.ptr(gmap), C_, H, W, F.ptr(dgout), F.ptr(idx), R, kh, kw, F.stream_ptr())
    assert_close(gmap.numpy(), gmap_want, 1e-5, "roi pool bwd")


@pytest.mark.parametrize("R,I,Oo", [(7, 100, 33), (96, 864, 48), (130, 13824, 64), (1, 512, 17)])
def test_linear_forward_backward(F, O, R, I, Oo):
    rng = np.random.RandomState(R)
    x = rng.randn(R, I).astype(np.float32)
    w = (rng.randn(Oo, I) / np.sqrt(I)).astype(np.float32); b = rng.randn(Oo).astype(np.float32)
    want = O.linear_fwd(x, w, b)
    y = F.DeviceTensor.empty((R, Oo))
    dx, dw = _dev(F, x), _dev(F, w)
    db = _dev(F, b)
    F._lib.call("frcnn_linear_forward", F.ptr(dx), R, I, F.ptr(dw), F.ptr(db), Oo, F.ptr(y), F.stream_ptr())
    assert_close(y.numpy(), want, 1e-4, "linear fwd")
    gy = rng.randn(R, Oo).astype(np.float32)
    gx = F.DeviceTensor.empty((R, I)); gw = F.DeviceTensor.zeros((Oo, I)); gb = F.DeviceTensor.zeros((Oo,))
    dgy = _dev(F, gy)
    F._lib.call("frcnn_linear_backward", F.ptr(dx), F.ptr(dgy), R, I, F.ptr(dw), Oo, F.ptr(gx), F.ptr(gw), F.ptr(gb),
                F.stream_ptr())
    g64, x64, w64 = gy.astype(np.float64), x.astype(np.float64), w.astype(np.float64)
    assert_close(gx.numpy(), g64 @ w64, 1e-4, "linear dgrad")
    assert_close(gw.numpy(), g64.T @ x64, 1e-4, "linear wgrad")
    assert_close(gb.numpy(), g64.sum(0), 1e-4, "linear bias grad")


@pytest.mark.parametrize("R", [138, 280, 560])
def test_linear_fc1_split_bf16_form(F, O, R):
    """nn.Linear(13824, 1024) (models/model_utilities.lua:82; cnet layer 1 of vgg_small / duplo) in its three roles --
    forward, input gradient, weight gradient -- at the row counts of a training step.  These products take the split-bf16
    operand form of gemm.hip (six exact bf16 x bf16 partial products per fp32 product): against fp64 they must meet the
    1e-4 bar AND be no worse than the fp32 matrix-core kernel on the same inputs (root-mean-square error <= 2x)."""
    I, Oo = 13824, 1024
    rng = np.random.RandomState(R)
    x = rng.randn(R, I).astype(np.float32)
    x[:, ::7] *= 1e-3; x[3] *= 1e-2         # a few decades of dynamic range (downwards: the bar is relative to max(1, |y|))
    w = (rng.randn(Oo, I) / np.sqrt(I)).astype(np.float32); b = rng.randn(Oo).astype(np.float32)
    gy = (rng.randn(R, Oo) / R).astype(np.float32)
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64)
    want = dict(fwd=x64 @ w64.T + b, dgrad=g64 @ w64, wgrad=g64.T @ x64)
    dx, dw, db, dgy = _dev(F, x), _dev(F, w), _dev(F, b), _dev(F, gy)
    res = {}
    for split in (1, 0):
        F._lib.call("frcnn_set_option", b"gemm_x_roles", 7 if split else 0)    # all three roles in the split form / none
        try:
            y = F.DeviceTensor.empty((R, Oo)); gx = F.DeviceTensor.empty((R, I)); gw = F.DeviceTensor.zeros((Oo, I)); gb = F.DeviceTensor.zeros((Oo,))
            F._lib.call("frcnn_linear_forward", F.ptr(dx), R, I, F.ptr(dw), F.ptr(db), Oo, F.ptr(y), F.stream_ptr())
            F._lib.call("frcnn_linear_backward", F.ptr(dx), F.ptr(dgy), R, I, F.ptr(dw), Oo, F.ptr(gx), F.ptr(gw), F.ptr(gb), F.stream_ptr())
            res[split] = dict(fwd=y.numpy(), dgrad=gx.numpy(), wgrad=gw.numpy())
        finally:
            F._lib.call("frcnn_set_option", b"gemm_x_roles", -1)
    for k in ("fwd", "dgrad", "wgrad"):
        assert_close(res[1][k], want[k], 1e-4, "FC1 %s (split form)" % k)
        assert_close(res[0][k], want[k], 1e-4, "FC1 %s (fp32 kernel)" % k)
        e1 = np.sqrt(np.mean((res[1][k] - want[k]) ** 2)); e0 = np.sqrt(np.mean((res[0][k] - want[k]) ** 2))
        print("FC1 %s R=%d: rms error split %.3e, fp32 kernel %.3e" % (k, R, e1, e0))
        assert e1 <= 2.0 * e0 + 1e-12, (k, e1, e0)
        assert not np.array_equal(res[1][k], res[0][k]), "the split form was not taken"


@pytest.mark.parametrize("R,tm", [(33, None), (193, None), (561, 64), (561, 128), (561, 192), (561, 256), (1398, None)])
def test_linear_fc1_split_form_tile_heights_and_ragged_rows(F, O, R, tm, monkeypatch):
    """The split-form Linear kernels (gemmx.hip) with row counts that are no multiple of any tile (33: less than one tile;
    193 / 561: one row into a new tile; 1398: the candidate count of an inference frame) and with every tile height forced
    in turn (FRCNN_GX_TM; 256 rows exist for the planes x planes weight gradient only -- the other roles then keep their
    own choice): forward, input gradient and weight gradient against fp64 at the 1e-4 bar."""
    I, Oo = 13824, 1024
    if tm is not None:
        monkeypatch.setenv("FRCNN_GX_TM", str(tm))
    rng = np.random.RandomState(R + (tm or 0))
    x = rng.randn(R, I).astype(np.float32)
    w = (rng.randn(Oo, I) / np.sqrt(I)).astype(np.float32); b = rng.randn(Oo).astype(np.float32)
    gy = (rng.randn(R, Oo) / R).astype(np.float32)
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64)
    gw0 = rng.randn(Oo, I).astype(np.float32)            # accGradParameters ADDS to what is there
    dx, dw, db, dgy = _dev(F, x), _dev(F, w), _dev(F, b), _dev(F, gy)
    F._lib.call("frcnn_set_option", b"gemm_x_roles", 7)
    try:
        y = F.DeviceTensor.empty((R, Oo)); gx = F.DeviceTensor.empty((R, I)); gw = _dev(F, gw0); gb = F.DeviceTensor.zeros((Oo,))
        F._lib.call("frcnn_linear_forward", F.ptr(dx), R, I, F.ptr(dw), F.ptr(db), Oo, F.ptr(y), F.stream_ptr())
        F._lib.call("frcnn_linear_backward", F.ptr(dx), F.ptr(dgy), R, I, F.ptr(dw), Oo, F.ptr(gx), F.ptr(gw), F.ptr(gb), F.stream_ptr())
        got = dict(fwd=y.numpy(), dgrad=gx.numpy(), wgrad=gw.numpy())
    finally:
        F._lib.call("frcnn_set_option", b"gemm_x_roles", -1)
    assert_close(got["fwd"], x64 @ w64.T + b, 1e-4, "FC1 forward R=%d tm=%s" % (R, tm))
    assert_close(got["dgrad"], g64 @ w64, 1e-4, "FC1 input gradient R=%d tm=%s" % (R, tm))
    assert_close(got["wgrad"], gw0.astype(np.float64) + g64.T @ x64, 1e-4, "FC1 weight gradient R=%d tm=%s" % (R, tm))


def test_linear_split_form_with_weights_at_a_4_byte_offset(F, O):
    """inside the flat parameter vector the weights of a Linear start at ANY multiple of 4 bytes (single PReLU
    slopes sit in front of them), while gemm_planes_kernel fetches them with 16-byte LDS-DMA.  gfx950 takes dword-aligned
    b128 loads; this pins it: W (and the gradient it accumulates into) one float off a 16-byte boundary."""
    import ctypes as C
    R, I, Oo = 320, 13824, 1024
    rng = np.random.RandomState(4)
    x = rng.randn(R, I).astype(np.float32)
    w = (rng.randn(Oo, I) / np.sqrt(I)).astype(np.float32); b = rng.randn(Oo).astype(np.float32)
    gy = (rng.randn(R, Oo) / R).astype(np.float32)
    wbuf = _dev(F, np.concatenate([[0.25], w.ravel()]).astype(np.float32))       # a slope, then the weights
    gwbuf = F.DeviceTensor.zeros((1 + Oo * I,))
    off = lambda t: C.c_void_p(F.ptr(t).value + 4)
    assert off(wbuf).value % 16 == 4
    dx, db, dgy = _dev(F, x), _dev(F, b), _dev(F, gy)
    F._lib.call("frcnn_set_option", b"gemm_x_roles", 7)
    try:
        y = F.DeviceTensor.empty((R, Oo)); gx = F.DeviceTensor.empty((R, I)); gb = F.DeviceTensor.zeros((Oo,))
        F._lib.call("frcnn_linear_forward", F.ptr(dx), R, I, off(wbuf), F.ptr(db), Oo, F.ptr(y), F.stream_ptr())
        F._lib.call("frcnn_linear_backward", F.ptr(dx), F.ptr(dgy), R, I, off(wbuf), Oo, F.ptr(gx), off(gwbuf), F.ptr(gb), F.stream_ptr())
    finally:
        F._lib.call("frcnn_set_option", b"gemm_x_roles", -1)
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64)
    assert_close(y.numpy(), x64 @ w64.T + b, 1e-4, "forward, W at +4 bytes")
    assert_close(gx.numpy(), g64 @ w64, 1e-4, "input gradient, W at +4 bytes")
    gw = gwbuf.numpy()
    assert gw[0] == 0.0
    assert_close(gw[1:].reshape(Oo, I), g64.T @ x64, 1e-4, "weight gradient at +4 bytes")


def test_rmsprop_and_scale(F, O):
    rng = np.random.RandomState(9)
    n = 100003
    x = rng.randn(n).astype(np.float32); g = rng.randn(n).astype(np.float32); m = rng.rand(n).astype(np.float32)
    import torch
    tx, tg, tm = [torch.from_numpy(v.copy()).cuda() for v in (x, g, m)]
    F._lib.call("frcnn_rmsprop", F.ptr(tx), F.ptr(tg), F.ptr(tm), n, 1e-4, 0.9, 1e-8, F.stream_ptr())
    O.rmsprop(x, g, m, 1e-4, 0.9, 1e-8)
    assert_close(tx.cpu().numpy(), x, 1e-6, "rmsprop x")
    assert_close(tm.cpu().numpy(), m, 1e-6, "rmsprop m")
    F._lib.call("frcnn_scale", F.ptr(tg), n, 1.0 / 37.0, F.stream_ptr())
    assert_close(tg.cpu().numpy(), g * np.float32(1.0 / 37.0), 1e-6, "scale")


def test_scale_rmsprop_with_the_divisor_on_the_device(F):
    """frcnn_scale_rmsprop_dev (the data-parallel tail: gradient:div(n) with n = the all-reduced count in device memory) ==
    frcnn_scale_rmsprop with 1 / n computed on the host, bit for bit; a count of 0 leaves the gradient unscaled."""
    import torch
    rng = np.random.RandomState(11)
    n = 100003
    x = rng.randn(n).astype(np.float32); g = rng.randn(n).astype(np.float32); m = rng.rand(n).astype(np.float32)
    for count in (560.0, 3.0, 0.0):
        a = [torch.from_numpy(v.copy()).cuda() for v in (x, g, m)]
        b = [torch.from_numpy(v.copy()).cuda() for v in (x, g, m)]
        cnt = torch.tensor([1.5, count, 7.0], dtype=torch.float64, device="cuda")
        F._lib.call("frcnn_scale_rmsprop_dev", F.ptr(a[0]), F.ptr(a[1]), C.c_void_p(cnt.data_ptr() + 8), F.ptr(a[2]), n, 1e-4, 0.9, 1e-8,
                    F.stream_ptr())
        F._lib.call("frcnn_scale_rmsprop", F.ptr(b[0]), F.ptr(b[1]), (1.0 / count) if count > 0 else 1.0, F.ptr(b[2]), n, 1e-4, 0.9, 1e-8,
                    F.stream_ptr())
        torch.cuda.synchronize()
        for u, v in zip(a, b):
            assert torch.equal(u, v), count
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_scale_rmsprop_dev", F.ptr(a[0]), F.ptr(a[1]), None, F.ptr(a[2]), n, 1e-4, 0.9, 1e-8, F.stream_ptr())


def test_scale_rmsprop_equals_the_two_calls(F):
    """frcnn_scale_rmsprop == frcnn_scale followed by frcnn_rmsprop, bit for bit (x, m and the scaled g)."""
    import torch
    rng = np.random.RandomState(10)
    n = 100003
    x = rng.randn(n).astype(np.float32); g = rng.randn(n).astype(np.float32); m = rng.rand(n).astype(np.float32)
    a = [torch.from_numpy(v.copy()).cuda() for v in (x, g, m)]
    b = [torch.from_numpy(v.copy()).cuda() for v in (x, g, m)]
    s = F.stream_ptr()
    F._lib.call("frcnn_scale", F.ptr(a[1]), n, 1.0 / 41.0, s)
    F._lib.call("frcnn_rmsprop", F.ptr(a[0]), F.ptr(a[1]), F.ptr(a[2]), n, 1e-4, 0.9, 1e-8, s)
    F._lib.call("frcnn_scale_rmsprop", F.ptr(b[0]), F.ptr(b[1]), 1.0 / 41.0, F.ptr(b[2]), n, 1e-4, 0.9, 1e-8, s)
    for u, v, what in zip(a, b, ("x", "g", "m")):
        assert torch.equal(u, v), what


@pytest.mark.parametrize("H,W", [(29, 50), (100, 120)])    # the LDS path (plane <= 64 KB of 64-bit words) and the global one
def test_deterministic_roi_pool_backward_and_partials(F, O, H, W):
    """Deterministic mode op by op: ROI-pooling backward in 64-bit fixed point, bias / slope partials folded in order --
    equal to the oracle like the default mode, and bit-identical from run to run."""
    rng = np.random.RandomState(H)
    C_, kh, kw, R = 12, 6, 6, 40
    fmap = rng.randn(C_, H, W).astype(np.float32)
    y0 = rng.randint(1, H - 8, R); x0 = rng.randint(1, W - 8, R)
    wins = np.stack([y0, y0 + rng.randint(1, 8, R), x0, x0 + rng.randint(1, 8, R)], 1).astype(np.int32)
    out = F.DeviceTensor.empty((R, C_ * kh * kw)); idx = F.DeviceTensor.empty((R, C_ * kh * kw), np.int32)
    dfm, dwins = _dev(F, fmap), _dev(F, wins, np.int32)
    F._lib.call("frcnn_roi_pool_forward", F.ptr(dfm), C_, H, W, F.ptr(dwins), R, kh, kw, F.ptr(out), F.ptr(idx), F.stream_ptr())
    gout = (rng.randn(R, C_ * kh * kw) * 1e-3).astype(np.float32)
    gidx = idx.numpy()
    want = np.zeros((C_, H, W), dtype=np.float32)
    for r in range(R):
        O.adaptive_max_pool_bwd(want, gout[r].reshape(C_, kh, kw), gidx[r].reshape(C_, kh, kw))
    dgout = _dev(F, gout)
    x = rng.randn(C_, H * W).astype(np.float32); g = rng.randn(C_, H * W).astype(np.float32)
    res = []
    F._lib.call("frcnn_set_option", b"deterministic", 1)
    try:
        for _ in range(2):
            gmap = F.DeviceTensor.zeros((C_, H, W))
            F._lib.call("frcnn_roi_pool_backward", F.ptr(gmap), C_, H, W, F.ptr(dgout), F.ptr(idx), R, kh, kw, F.stream_ptr())
            dg, dx, da = _dev(F, g), _dev(F, x), _dev(F, [0.25])
            gb = F.DeviceTensor.zeros((C_,)); ga = F.DeviceTensor.zeros((1,))
            F._lib.call("frcnn_act_backward", F.ptr(dg), F.ptr(dx), C_, H * W, F.ptr(da), None, F.ptr(dg), F.ptr(gb), F.ptr(ga), F.stream_ptr())
            res.append((gmap.numpy(), gb.numpy(), ga.numpy()))
    finally:
        F._lib.call("frcnn_set_option", b"deterministic", 0)
    assert_close(res[0][0], want, 1e-5, "deterministic roi pool bwd")
    gx = np.where(x > 0, g, 0.25 * g)
    assert_close(res[0][1], gx.astype(np.float64).sum(1), 1e-4)
    assert_close(res[0][2][0], float(np.where(x > 0, 0, x.astype(np.float64) * g).sum()), 1e-4)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
