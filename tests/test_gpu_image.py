"""GPU parity of the image-preparation kernels (image.hip) against oracle/orc_image.py, through
the C ABI (frcnn_image_*) and through the host mirror BatchIterator.processImage / nextTraining.
Tolerances: gathers (crop / flips) bit-exact; float pipelines 1e-5 relative (the kernels keep the operation
order of the C originals and are compiled without FMA contraction, so most results are bit-identical; the fp64
reductions are summed in a different order than numpy's)."""
import ctypes as C

import numpy as np
import pytest

import orc_image as OI
from util import assert_close

pytestmark = pytest.mark.gpu


def _dev(F, a):
    return F.DeviceTensor.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def test_rgb2yuv(F):
    rng = np.random.RandomState(0)
    img = rng.rand(3, 37, 53).astype(np.float32)
    out = F.DeviceTensor.empty((3, 37, 53))
    d = _dev(F, img)   # (keep the input alive until the result has been read back)
    F._lib.call("frcnn_image_rgb2yuv", F.ptr(d), F.ptr(out), 37, 53, F.stream_ptr())
    assert np.array_equal(out.numpy(), OI.rgb2yuv(img))


@pytest.mark.parametrize("space", ["hsv", "lab"])
def test_rgb2hsv_and_rgb2lab(F, space):
    """The other two colour spaces of load_image (utilities.lua:212-215).  hsv is float arithmetic in a fixed order: bit
    exact.  lab goes through pow(): 2e-5 absolute on values of magnitude <= 128 (the cube roots differ in the last bits)."""
    rng = np.random.RandomState(3)
    img = rng.rand(3, 61, 83).astype(np.float32)
    img[:, 0, :8] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [0, 0, 0], [.5, .5, .5], [.02, .03, .01], [1, 0, 1]], np.float32).T
    img[:, 1, :] = img[0, 1, :]   # a grey row
    out = F.DeviceTensor.empty((3, 61, 83))
    d = _dev(F, img)
    F._lib.call("frcnn_image_rgb2" + space, F.ptr(d), F.ptr(out), 61, 83, F.stream_ptr())
    got, want = out.numpy(), getattr(OI, "rgb2" + space)(img)
    if space == "hsv":
        assert np.array_equal(got, want)
    else:
        assert np.abs(got - want).max() < 2e-5
    with pytest.raises(F._lib.FrcnnError):
        F._lib.call("frcnn_image_rgb2" + space, F.ptr(d), F.ptr(d), 61, 83, F.stream_ptr())


@pytest.mark.parametrize("src,dst", [((3, 40, 64), (90, 150)), ((3, 90, 150), (40, 64)), ((3, 108, 192), (45, 80)),
                                     ((3, 50, 70), (50, 33)), ((3, 31, 47), (77, 47)), ((1, 1, 1), (5, 4)),
                                     ((3, 33, 100), (33, 100)), ((2, 64, 3), (7, 200))])
def test_scale_matches_oracle(F, src, dst):
    rng = np.random.RandomState(1)
    Cn, H, W = src
    dH, dW = dst
    img = rng.randn(*src).astype(np.float32)
    out = F.DeviceTensor.empty((Cn, dH, dW)); tmp = F.DeviceTensor.empty((Cn * H * dW,))
    d = _dev(F, img)
    F._lib.call("frcnn_image_scale", F.ptr(d), Cn, H, W, F.ptr(out), dH, dW, F.ptr(tmp), 0, F.stream_ptr())
    want = OI.scale_bilinear(img, dW, dH)
    got = out.numpy()
    assert got.shape == want.shape
    assert_close(got, want, 1e-6, "image.scale %s -> %s" % (src, dst))


def test_scale_full_size_properties(F):
    """1080p -> 800x450 (the bench frame's source size): a constant stays constant, the result is linear in the
    input, and equals the oracle on a strip."""
    rng = np.random.RandomState(2)
    H, W, dH, dW = 1080, 1920, 450, 800
    a = rng.rand(3, H, W).astype(np.float32); b = rng.rand(3, H, W).astype(np.float32)
    tmp = F.DeviceTensor.empty((3 * H * dW,))

    def run(x):
        out = F.DeviceTensor.empty((3, dH, dW))
        d = _dev(F, x)
        F._lib.call("frcnn_image_scale", F.ptr(d), 3, H, W, F.ptr(out), dH, dW, F.ptr(tmp), 0, F.stream_ptr())
        return out.numpy()
    ra, rb, rab = run(a), run(b), run(a + 2 * b)
    assert_close(rab, ra + 2 * rb, 1e-5, "linearity")
    assert_close(run(np.full((3, H, W), 0.375, np.float32)), np.full((3, dH, dW), 0.375, np.float32), 1e-6, "constant")
    want = OI.scale_bilinear(a[:1], dW, dH)
    assert_close(ra[:1], want, 1e-6, "1080p -> 800x450 channel 0")


def test_scale_with_fused_rgb2yuv_is_bit_identical(F):
    rng = np.random.RandomState(9)
    for (H, W, dH, dW) in [(108, 192, 45, 80), (40, 64, 90, 150), (50, 70, 50, 70), (1080, 1920, 450, 800)]:
        rgb = rng.rand(3, H, W).astype(np.float32)
        d = _dev(F, rgb); yuv = F.DeviceTensor.empty((3, H, W)); tmp = F.DeviceTensor.empty((3 * H * dW,))
        a = F.DeviceTensor.empty((3, dH, dW)); b = F.DeviceTensor.empty((3, dH, dW))
        F._lib.call("frcnn_image_rgb2yuv", F.ptr(d), F.ptr(yuv), H, W, F.stream_ptr())
        F._lib.call("frcnn_image_scale", F.ptr(yuv), 3, H, W, F.ptr(a), dH, dW, F.ptr(tmp), 0, F.stream_ptr())
        F._lib.call("frcnn_image_scale", F.ptr(d), 3, H, W, F.ptr(b), dH, dW, F.ptr(tmp), 1, F.stream_ptr())
        assert np.array_equal(a.numpy(), b.numpy())


def test_crop_flip_exact(F):
    rng = np.random.RandomState(3)
    img = rng.randn(3, 60, 90).astype(np.float32)
    d = _dev(F, img)
    for (x0, y0, w, h, hf, vf) in [(0, 0, 90, 60, 1, 0), (0, 0, 90, 60, 0, 1), (5, 7, 40, 33, 0, 0), (11, 3, 64, 50, 1, 1),
                                   (89, 59, 1, 1, 1, 1)]:
        out = F.DeviceTensor.empty((3, h, w))
        F._lib.call("frcnn_image_crop_flip", F.ptr(d), 3, 60, 90, x0, y0, w, h, hf, vf, F.ptr(out), F.stream_ptr())
        want = OI.crop(img, x0, y0, x0 + w, y0 + h)
        if hf: want = OI.hflip(want)
        if vf: want = OI.vflip(want)
        assert np.array_equal(out.numpy(), want)
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_image_crop_flip", F.ptr(d), 3, 60, 90, 80, 0, 20, 10, 0, 0, F.ptr(d), F.stream_ptr())


@pytest.mark.parametrize("flags", [(1, 1), (1, 0), (0, 1)])
def test_normalize_matches_oracle(F, flags):
    rng = np.random.RandomState(4)
    img = (rng.rand(3, 450, 800) * np.array([1, 5, 0.1])[:, None, None] + np.array([3, -2, 0.5])[:, None, None]).astype(np.float32)
    d = _dev(F, img)
    wsb = F._lib.load().frcnn_image_normalize_workspace_bytes(3)
    ws = F.DeviceTensor.empty(((wsb + 3) // 4,))
    F._lib.call("frcnn_image_normalize", F.ptr(d), 3, 450, 800, flags[0], flags[1], F.ptr(ws), wsb, F.stream_ptr())
    got = d.numpy()
    assert_close(got, OI.center_and_scale(img, bool(flags[0]), bool(flags[1])), 1e-6, "centre/scale %s" % (flags,))
    if flags == (1, 1):
        for c in range(3):
            assert abs(got[c].astype(np.float64).mean()) < 1e-6 and abs(got[c].astype(np.float64).std(ddof=1) - 1) < 1e-6
    # a flat channel is centred but not divided by its (zero) standard deviation
    flat = _dev(F, np.full((3, 16, 16), 2.0, np.float32))
    F._lib.call("frcnn_image_normalize", F.ptr(flat), 3, 16, 16, 1, 1, F.ptr(ws), wsb, F.stream_ptr())
    assert np.array_equal(flat.numpy(), np.zeros((3, 16, 16), np.float32))


@pytest.mark.parametrize("shape,K", [((45, 80), 7), ((33, 31), 7), ((5, 200), 7), ((64, 64), 3), ((40, 37), 15), ((3, 3), 7)])
def test_contrastive_norm_matches_oracle(F, shape, K):
    rng = np.random.RandomState(5)
    plane = rng.randn(*shape).astype(np.float32)
    k = OI.gaussian1d(K)
    out = F.DeviceTensor.empty(shape); tmp = F.DeviceTensor.empty(shape)
    d = _dev(F, plane)
    F._lib.call("frcnn_image_contrastive_norm", F.ptr(d), shape[0], shape[1], k.ctypes.data_as(C.c_void_p), K,
                1e-4, F.ptr(out), F.ptr(tmp), F.stream_ptr())
    assert_close(out.numpy(), OI.contrastive_norm(plane, k), 1e-5, "contrastive %s K=%d" % (shape, K))


def test_contrastive_norm_full_size_in_place(F):
    """450x800 luminance plane, in == out (how processImage calls it): equals the oracle; scaling the input by a
    positive factor does not change the result (every local deviation is far above the threshold)."""
    rng = np.random.RandomState(6)
    plane = rng.randn(450, 800).astype(np.float32)
    k = OI.gaussian1d(7)
    tmp = F.DeviceTensor.empty((450, 800))

    def run(x):
        d = _dev(F, x)
        F._lib.call("frcnn_image_contrastive_norm", F.ptr(d), 450, 800, k.ctypes.data_as(C.c_void_p), 7, 1e-4, F.ptr(d),
                    F.ptr(tmp), F.stream_ptr())
        return d.numpy()
    got = run(plane)
    assert_close(got, OI.contrastive_norm(plane, k), 1e-5, "contrastive 450x800")
    assert_close(run(plane * 4.0), got, 1e-5, "scale invariance")
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_image_contrastive_norm", F.ptr(tmp), 450, 800, k.ctypes.data_as(C.c_void_p), 8, 1e-4, F.ptr(tmp),
                    F.ptr(tmp), F.stream_ptr())


def _tiny_model(F, cfg):
    return F.vgg_small(cfg)


def test_process_image_matches_oracle(F, small_cfg):
    """BatchIterator.processImage on a 1080p YUV frame against the oracle's process_image with the same flip
    decisions (drawn from an identical MT19937 stream), incl. the ROI transforms."""
    model = _tiny_model(F, small_cfg)
    rng = np.random.RandomState(7)
    frame = OI.rgb2yuv(rng.rand(3, 540, 960).astype(np.float32))
    data = dict(ground_truth={}, training_set=["a"], validation_set=[], background_files=[])
    for seed in (1, 2, 3, 4):
        it = F.BatchIterator(model, data, seed=seed)
        twin = F.MT19937(seed); twin.randperm(1)    # (the constructor shuffles the one-element training set: no draw)
        rois = [F.Roi(F.Rect(96, 54, 480, 270), 2)]
        img, out_rois = it.processImage(frame, rois)
        hf = twin.uniform() < small_cfg["augmentation"]["hflip"]
        vf = twin.uniform() < small_cfg["augmentation"]["vflip"]
        want = OI.process_image(frame, small_cfg, hf, vf)
        assert img.shape == want.shape == (3, 450, 800)
        assert_close(img.numpy(), want, 2e-5, "processImage seed %d (hflip %s, vflip %s)" % (seed, hf, vf))
        r = out_rois[0].rect
        sx, sy = 800 / 960, 450 / 540
        x0, x1, y0, y1 = 96 * sx, 480 * sx, 54 * sy, 270 * sy
        if hf: x0, x1 = 800 - x1, 800 - x0
        if vf: y0, y1 = 450 - y1, 450 - y0
        assert np.allclose([r.minX, r.minY, r.maxX, r.maxY], [x0, y0, x1, y1], atol=1e-9)


def test_next_training_batches(F, small_cfg):
    """nextTraining on in-memory frames: batch rule (images are added until they carry cfg.batch_size examples),
    one background image with 5 % of the examples, shapes, determinism under a seed, and the result feeds
    lossAndGradient."""
    model = _tiny_model(F, small_cfg)
    rng = np.random.RandomState(8)
    frames = {"img%d" % i: rng.rand(3, 540, 960).astype(np.float32) for i in range(3)}
    frames["bg0"] = rng.rand(3, 600, 600).astype(np.float32)
    frames["small"] = rng.rand(3, 100, 900).astype(np.float32)       # 450 px smaller side -> capped at 1000 wide, 111 high: skipped
    gt = {k: dict(rois=[F.Roi(F.Rect(100 + 50 * j, 80 + 40 * j, 300 + 60 * j, 260 + 50 * j), 1 + j) for j in range(3)])
          for k in frames}
    data = dict(ground_truth=gt, training_set=["img0", "img1", "img2", "small"], validation_set=["img1"], background_files=["bg0"])

    def make(seed):
        return F.BatchIterator(model, data, load_image=lambda fn: frames[fn], seed=seed)
    it = make(21)
    batch = it.nextTraining()
    assert len(batch) >= 2
    bg, rest = batch[0], batch[1:]
    assert bg["positive"] == [] and len(bg["negative"]) == int(small_cfg["batch_size"] * 0.05) and bg["img"].shape == (3, 450, 450)
    total = sum(len(x["positive"]) + len(x["negative"]) for x in rest)
    assert total >= small_cfg["batch_size"] - len(bg["negative"])
    assert total - (len(rest[-1]["positive"]) + len(rest[-1]["negative"])) < small_cfg["batch_size"] - len(bg["negative"])
    for x in rest:
        assert x["img"].shape == (3, 450, 800) and len(x["negative"]) >= 16
        a = x["img"].numpy()
        assert np.isfinite(a).all() and abs(a[1].mean()) < 1e-4 and abs(a[2].std(ddof=1) - 1) < 1e-4
    again = make(21).nextTraining()
    assert [len(x["positive"]) for x in again] == [len(x["positive"]) for x in batch]
    assert np.array_equal(again[1]["img"].numpy(), batch[1]["img"].numpy())
    val = it.nextValidation(2)
    assert len(val) == 2 and val[0]["img"].shape == (3, 450, 800) and len(val[0]["rois"]) == 3
    # the batches are what create_objective consumes
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=1)
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    f = F.create_objective(model, w, g, make(5), stats)
    loss, grad = f(w)
    assert np.isfinite(loss) and bool(grad.isfinite().all())


def test_training_data_file_feeds_batch_iterator(F, small_cfg, tmp_path):
    """The whole loader chain of SURVEY 8f: boxes.csv -> create_training_data -> torch object file -> load_training_data
    -> BatchIterator.nextTraining (frames decoded elsewhere) -> lossAndGradient."""
    rng = np.random.RandomState(11)
    rows = []
    frames = {}
    for i in range(4):
        name = "img%d.png" % i
        frames[name] = rng.rand(3, 540, 960).astype(np.float32)
        for j in range(2):
            x0, y0 = 100 + 200 * j + 10 * i, 80 + 60 * j
            rows.append('"%s", %d, %d, %d, %d, "Brick%d", %d, "Red", 4' % (name, x0, y0, x0 + 220, y0 + 200, j, j + 1))
    (tmp_path / "boxes.csv").write_text("\n".join(rows) + "\n")
    fn = str(tmp_path / "duplo.t7")
    F.traindata.create_training_data("unit", str(tmp_path / "boxes.csv"), None, fn, validation_size=1, seed=2)
    data = F.traindata.load_training_data(fn)
    assert len(data["training_set"]) == 3 and len(data["validation_set"]) == 1
    model = F.vgg_small(small_cfg)
    it = F.BatchIterator(model, data, load_image=lambda f: frames[f], seed=4)
    batch = it.nextTraining(64)
    assert batch and all(x["img"].shape == (3, 450, 800) for x in batch) and sum(len(x["positive"]) for x in batch) > 0
    assert all(p[1].class_index in (1, 2) for x in batch for p in x["positive"])
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=1)
    f = F.create_objective(model, w, g, F.BatchIterator(model, data, load_image=lambda f: frames[f], seed=4),
                           dict(pcls=[], preg=[], dcls=[], dreg=[]))
    small_cfg2 = dict(small_cfg)
    loss, grad = f(w)
    assert np.isfinite(loss) and bool(grad.isfinite().all())


def test_batch_iterator_decodes_image_files(F, small_cfg, tmp_path):
    """Default loader: PNG files under cfg.examples_base_path are decoded on the host (Pillow), everything after that
    runs on the device; the prepared frame equals the oracle's processImage of the decoded pixels."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(13)
    px = rng.randint(0, 256, size=(270, 480, 3)).astype(np.uint8)
    Image.fromarray(px).save(str(tmp_path / "frame.png"))
    cfg = dict(small_cfg); cfg["examples_base_path"] = str(tmp_path)
    cfg["augmentation"] = dict(vflip=0, hflip=0, random_scaling=0.0, aspect_jitter=0.0)
    model = F.vgg_small(cfg)
    data = dict(ground_truth={"frame.png": dict(rois=[F.Roi(F.Rect(40, 30, 200, 150), 1)])}, training_set=["frame.png"],
                validation_set=["frame.png"], background_files=[])
    it = F.BatchIterator(model, data)
    val = it.nextValidation(1)
    img = val[0]["img"]
    want = OI.process_image(OI.rgb2yuv((px.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1)), cfg)
    assert img.shape == want.shape == (3, 450, 800)
    assert_close(img.numpy(), want, 2e-5, "decoded + prepared frame")
    r = val[0]["rois"][0].rect
    assert np.allclose([r.minX, r.minY, r.maxX, r.maxY], [40 * 800 / 480, 30 * 450 / 270, 200 * 800 / 480, 150 * 450 / 270])
    # decode-ahead pool: worker threads decode, the frame crosses PCIe as 8-bit RGB and is converted inside the row pass
    # (frcnn_image_scale_u8) -- the same arithmetic, so the prepared frame is bit-identical
    it2 = F.BatchIterator(model, data, workers=3, prefetch=4)
    for _ in range(3):
        v2 = it2.nextValidation(1)
        assert np.array_equal(v2[0]["img"].numpy(), img.numpy())
    cfg_rgb = dict(cfg); cfg_rgb["color_space"] = "rgb"
    m2 = dict(model); m2["cfg"] = cfg_rgb
    a = F.BatchIterator(m2, data).nextValidation(1)[0]["img"].numpy()
    b = F.BatchIterator(m2, data, workers=2).nextValidation(1)[0]["img"].numpy()
    assert np.array_equal(a, b)
    # 'lab' and 'hsv' (utilities.lua:212-215): converted at full resolution before processImage, with or without the pool
    rgb01 = (px.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1)
    for space, tol in (("hsv", 2e-5), ("lab", 2e-4)):
        cfg_s = dict(cfg); cfg_s["color_space"] = space
        ms = dict(model); ms["cfg"] = cfg_s
        a = F.BatchIterator(ms, data).nextValidation(1)[0]["img"].numpy()
        b = F.BatchIterator(ms, data, workers=2).nextValidation(1)[0]["img"].numpy()
        assert np.array_equal(a, b)
        assert_close(a, OI.process_image(getattr(OI, "rgb2" + space)(rgb01), cfg_s), tol, "prepared %s frame" % space)


def test_background_base_path_and_many_frame_sizes(F, small_cfg, tmp_path):
    """Background files resolve against cfg.background_base_path (BatchIterator.lua:255), not the examples' base; and a
    data set in which every frame has its own size does not grow the loader's device cache beyond its bound."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(21)
    ex_dir, bg_dir = tmp_path / "examples", tmp_path / "backgrounds"
    ex_dir.mkdir(); bg_dir.mkdir()
    names = []
    for i in range(12):
        h, w = 200 + 7 * i, 300 + 11 * i
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(str(ex_dir / ("f%d.png" % i)))
        names.append("f%d.png" % i)
    Image.fromarray(rng.randint(0, 256, size=(240, 400, 3)).astype(np.uint8)).save(str(bg_dir / "bg.png"))
    cfg = dict(small_cfg); cfg["examples_base_path"] = str(ex_dir); cfg["background_base_path"] = str(bg_dir)
    model = F.vgg_small(cfg)
    gt = dict((n, dict(rois=[F.Roi(F.Rect(40, 30, 200, 150), 1 + i % 16)])) for i, n in enumerate(names))
    data = dict(ground_truth=gt, training_set=names, validation_set=names, background_files=["bg.png"])
    for workers in (0, 2):
        bound = 24 << 20
        it = F.BatchIterator(model, data, workers=workers, cache_bytes=bound, seed=3)
        logs = []
        it.log = logs.append
        for _ in range(6):
            batch = it.nextTraining(40)
            assert not batch[0]["positive"] and len(batch[0]["negative"]) == 2     # the background image: 5 % of 40
            assert len(batch) >= 2
            shapes = set(tuple(b["img"].shape) for b in batch)
            ptrs = [b["img"].ptr for b in batch]
            assert len(set(ptrs)) == len(ptrs), "two images of one batch share a buffer"
            del batch
            assert it.pool.cached <= bound
        assert not [l for l in logs if "Invalid" in l], logs
        assert it.pool.allocated <= bound + 8 * (3 * 470 * 830 * 4)      # cache + the frames a batch holds


def test_golden_image_fixture_on_device(F):
    """The committed golden vectors (tests/golden/image_small.json) through the kernels."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "image_small.json")))
    rgb = np.array(g["rgb"], np.float32).reshape(3, 10, 14)
    d = _dev(F, rgb); tmp = F.DeviceTensor.empty((3 * 10 * 20,))
    up = F.DeviceTensor.empty((3, 15, 20)); down = F.DeviceTensor.empty((3, 7, 9))
    F._lib.call("frcnn_image_scale", F.ptr(d), 3, 10, 14, F.ptr(up), 15, 20, F.ptr(tmp), 1, F.stream_ptr())
    F._lib.call("frcnn_image_scale", F.ptr(d), 3, 10, 14, F.ptr(down), 7, 9, F.ptr(tmp), 1, F.stream_ptr())
    assert_close(up.numpy().ravel(), np.array(g["up_15x20"], np.float32), 1e-6, "golden up-scaling")
    assert_close(down.numpy().ravel(), np.array(g["down_7x9"], np.float32), 1e-6, "golden down-scaling")
    wsb = F._lib.load().frcnn_image_normalize_workspace_bytes(3)
    ws = F.DeviceTensor.empty(((wsb + 3) // 4,))
    F._lib.call("frcnn_image_normalize", F.ptr(down), 3, 7, 9, 1, 1, F.ptr(ws), wsb, F.stream_ptr())
    assert_close(down.numpy().ravel(), np.array(g["normalized"], np.float32), 1e-5, "golden centring / scaling")
    y = down.offset_view(0, (7, 9)); t2 = F.DeviceTensor.empty((7, 9))
    k = np.array(g["gaussian1d_7"], np.float32)
    F._lib.call("frcnn_image_contrastive_norm", F.ptr(y), 7, 9, k.ctypes.data_as(C.c_void_p), 7, 1e-4, F.ptr(y), F.ptr(t2), F.stream_ptr())
    assert_close(y.numpy().ravel(), np.array(g["contrastive_y"], np.float32), 1e-4, "golden contrastive normalisation")
