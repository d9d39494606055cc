"""CPU tests of the image-preparation restatement (oracle/orc_image.py) and of the host half of
BatchIterator.processImage (ROI transforms, target size, draw order).  The restatement is PARITY UNPINNED (the
torch `image` / `nn` packages are not in /root/reference and no Torch7 runs here): what pins it are the
hand-derived answers below and a second, naive float64 implementation of the normalisation modules."""
import math

import numpy as np

import orc_image as OI


def test_find_target_size_known_answers():  # utilities.lua:188-203, derived by hand
    assert OI.find_target_size(1920, 1080, 450, 1000) == (800, 450)
    assert OI.find_target_size(600, 800, 450, 1000) == (450, 600)
    assert OI.find_target_size(4000, 1000, 450, 1000) == (1000, 250)   # capped by max_pixel_size
    assert OI.find_target_size(500, 500, 450, 1000) == (450, 450)      # square takes the else branch
    from frcnn_amd.BatchIterator import find_target_size
    for w, h in [(1920, 1080), (600, 800), (4000, 1000), (333, 777), (1280, 720), (720, 1280)]:
        assert find_target_size(w, h, 480, 1000) == OI.find_target_size(w, h, 480, 1000)


def test_scaled_size_truncation():
    # BatchIterator.lua:51: w * (tw / w) in double arithmetic, truncated by the tensor constructor
    for w, tw in [(1920, 800), (1080, 450), (777, 333), (4000, 1000), (641, 450)]:
        sw, _ = OI.scaled_size(w, 10, tw / w, 1.0)
        assert sw in (tw, tw - 1) and sw == int(w * (tw / w))


def test_scale_line_known_answers():
    up = OI._scale_line(np.array([[0.0, 1.0]], np.float32), 3)
    assert np.allclose(up, [[0.0, 0.5, 1.0]])
    dn = OI._scale_line(np.array([[1, 2, 3, 4]], np.float32), 2)
    assert np.allclose(dn, [[1.5, 3.5]])
    dn = OI._scale_line(np.array([[1, 2, 3]], np.float32), 2)      # scale 1.5: [1 + 0.5*2]/1.5, [0.5*2 + 3]/1.5
    assert np.allclose(dn, [[4.0 / 3.0, 8.0 / 3.0]], atol=1e-6)
    same = OI._scale_line(np.array([[5, 6, 7]], np.float32), 3)
    assert np.array_equal(same, [[5, 6, 7]])
    one = OI._scale_line(np.array([[2.5]], np.float32), 4)          # single source sample: replicated
    assert np.array_equal(one, [[2.5, 2.5, 2.5, 2.5]])
    # a constant stays constant, a ramp keeps its end points when up-scaling
    c = OI.scale_bilinear(np.full((3, 7, 9), 0.25, np.float32), 20, 4)
    assert c.shape == (3, 4, 20) and np.allclose(c, 0.25, atol=1e-6)
    r = OI._scale_line(np.arange(10, dtype=np.float32)[None], 23)
    assert r[0, 0] == 0 and r[0, -1] == 9 and np.all(np.diff(r[0]) > 0)


def test_gaussian_and_yuv_known_answers():
    g = OI.gaussian1d(7)
    want = [math.exp(-(((i - 4.0) / 1.75) ** 2) / 2) for i in range(1, 8)]
    assert np.allclose(g, want, atol=1e-7) and abs(g[3] - 1.0) < 1e-7 and abs(g[2] - 0.849365) < 1e-5
    from frcnn_amd.BatchIterator import gaussian1D
    assert np.array_equal(gaussian1D(7), g)
    white = OI.rgb2yuv(np.ones((3, 1, 1), np.float32))
    assert abs(white[0, 0, 0] - 1.0) < 1e-6 and abs(white[1, 0, 0]) < 2e-5 and abs(white[2, 0, 0]) < 2e-5   # (the published coefficients sum to 1e-5)


def test_hsv_and_lab_known_answers():
    """image.rgb2hsv / image.rgb2lab (load_image, utilities.lua:212-215) against values anyone can look up: the hue circle's
    corners, and the published CIE L*a*b* coordinates (D65) of the sRGB primaries."""
    def px(fn, r, g, b):
        return fn(np.array([r, g, b], np.float32).reshape(3, 1, 1)).reshape(3).astype(np.float64)
    for rgb, hsv in (((1, 0, 0), (0, 1, 1)), ((1, 1, 0), (1 / 6, 1, 1)), ((0, 1, 0), (1 / 3, 1, 1)), ((0, 1, 1), (1 / 2, 1, 1)),
                     ((0, 0, 1), (2 / 3, 1, 1)), ((1, 0, 1), (5 / 6, 1, 1)), ((0.5, 0.5, 0.5), (0, 0, 0.5)), ((0, 0, 0), (0, 0, 0)),
                     ((0.5, 0.25, 0.25), (0, 0.5, 0.5)), ((0.2, 0.4, 0.8), (11 / 18, 0.75, 0.8))):
        assert np.allclose(px(OI.rgb2hsv, *rgb), hsv, atol=1e-6), (rgb, px(OI.rgb2hsv, *rgb))
    for rgb, lab in (((1, 1, 1), (100, 0, 0)), ((0, 0, 0), (0, 0, 0)), ((1, 0, 0), (53.24, 80.09, 67.20)),
                     ((0, 1, 0), (87.73, -86.18, 83.18)), ((0, 0, 1), (32.30, 79.19, -107.86)), ((0.5, 0.5, 0.5), (53.39, 0, 0))):
        assert np.allclose(px(OI.rgb2lab, *rgb), lab, atol=0.02), (rgb, px(OI.rgb2lab, *rgb))
    # the linear toe of the sRGB curve and of f(t): a very dark grey
    dark = px(OI.rgb2lab, 0.02, 0.02, 0.02)
    assert abs(dark[0] - (24389.0 / 27.0) * (0.02 / 12.92)) < 1e-4 and abs(dark[1]) < 1e-4 and abs(dark[2]) < 1e-4


def test_center_and_scale_properties():
    rng = np.random.RandomState(0)
    img = (rng.rand(3, 40, 50) * np.array([1, 5, 0.1])[:, None, None] + np.array([3, -2, 0.5])[:, None, None]).astype(np.float32)
    out = OI.center_and_scale(img)
    for c in range(3):
        assert abs(out[c].astype(np.float64).mean()) < 1e-6
        assert abs(out[c].astype(np.float64).std(ddof=1) - 1.0) < 1e-6
    flat = OI.center_and_scale(np.full((3, 4, 4), 2.0, np.float32))   # std 0 <= 1e-8: left unscaled
    assert np.array_equal(flat, np.zeros((3, 4, 4), np.float32))


def _naive_contrastive(plane, k1d, thr):
    """second restatement: direct 2-D sums in float64 over the zero-padded plane"""
    k = k1d.astype(np.float64) / k1d.astype(np.float64).sum()
    K = len(k); p = K // 2
    H, W = plane.shape

    def est(x):
        out = np.zeros((H, W))
        for y in range(H):
            for xx in range(W):
                s = 0.0
                for jy in range(K):
                    for jx in range(K):
                        sy, sx = y + jy - p, xx + jx - p
                        if 0 <= sy < H and 0 <= sx < W:
                            s += k[jy] * k[jx] * x[sy, sx]
                out[y, xx] = s
        return out
    coef = est(np.ones((H, W)))
    sub = plane - est(plane.astype(np.float64)) / coef
    sd = np.sqrt(est(sub * sub)) / coef
    sd = np.where(sd > thr, sd, thr)
    return sub / sd


def test_contrastive_norm_against_naive():
    rng = np.random.RandomState(1)
    plane = rng.randn(12, 17).astype(np.float32)
    k = OI.gaussian1d(7)
    got = OI.contrastive_norm(plane, k)
    want = _naive_contrastive(plane, k, 1e-4)
    assert np.abs(got - want).max() < 1e-4
    # scale invariance above the threshold, and the threshold branch on a (numerically) flat plane
    assert np.abs(OI.contrastive_norm(plane * 3.0, k) - got).max() < 1e-4
    flat = OI.contrastive_norm(np.zeros((9, 9), np.float32), k)
    assert np.array_equal(flat, np.zeros((9, 9), np.float32))


def test_roi_transforms_and_draw_order():
    """Host half of processImage: ROI transforms of BatchIterator.lua:49-80 and the order of the random draws."""
    from frcnn_amd.BatchIterator import _transform_rois
    from frcnn_amd import Rect, Roi
    rois = [Roi(Rect(10, 20, 110, 220), 3), Roi(Rect(900, 10, 1000, 50), 1)]
    out = _transform_rois(rois, lambda r, w, h: r.scale(0.5, 0.25), 1000, 400, 500, 100)
    assert [(r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY) for r in out] == [(5, 5, 55, 55), (450, 2.5, 500, 12.5)]
    flipped = _transform_rois(out, lambda r, w, h: Rect(w - r.maxX, r.minY, w - r.minX, r.maxY), 500, 100, 500, 100)
    assert (flipped[0].rect.minX, flipped[0].rect.maxX) == (445, 495)
    vf = _transform_rois(flipped, lambda r, w, h: Rect(r.minX, h - r.maxY, r.maxX, h - r.minY), 500, 100, 500, 100)
    assert (vf[0].rect.minY, vf[0].rect.maxY) == (45, 95)
    # Rect.isEmpty needs BOTH extents to vanish (Rect.lua): the first box collapses to a point and is dropped, the
    # second keeps a height and survives with zero width
    crop = Rect.fromXYWidthHeight(496, 96, 4, 4)
    kept = _transform_rois(vf, lambda r, w, h: r.clip(crop).offset(-crop.minX, -crop.minY), 500, 100, 4, 4)
    assert len(kept) == 1 and kept[0].class_index == 1 and kept[0].rect.width() == 0 and kept[0].rect.height() == 1.5


def test_randperm_is_a_permutation_and_seeded():
    from frcnn_amd import MT19937
    a, b = MT19937(11).randperm(50), MT19937(11).randperm(50)
    assert a == b and sorted(a) == list(range(1, 51)) and a != list(range(1, 51))
    assert MT19937(3).randperm(1) == [1]
    u = [MT19937(5).uniform() for _ in range(3)]
    assert all(0.0 <= v < 1.0 for v in u)


def test_decode_image(tmp_path):
    """image.load(fn, 3, 'float'): 8-bit samples / 255 as float RGB planes; grey images expanded; .npy passed through."""
    import pytest
    Image = pytest.importorskip("PIL.Image")
    from frcnn_amd import decode_image
    rng = np.random.RandomState(3)
    px = rng.randint(0, 256, size=(7, 9, 3)).astype(np.uint8)
    Image.fromarray(px).save(str(tmp_path / "a.png"))
    a = decode_image(str(tmp_path / "a.png"))
    assert a.shape == (3, 7, 9) and a.dtype == np.float32
    assert np.array_equal(a, (px.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1))
    Image.fromarray(px[:, :, 0]).save(str(tmp_path / "g.png"))
    g = decode_image(str(tmp_path / "g.png"))
    assert g.shape == (3, 7, 9) and np.array_equal(g[0], g[1]) and np.array_equal(g[1], g[2])
    np.save(str(tmp_path / "f.npy"), a)
    assert np.array_equal(decode_image(str(tmp_path / "f.npy")), a)


def test_golden_image_fixture():
    """tests/golden/image_small.json (make_golden.py: restatement cross-checked with PyTorch's align_corners bilinear
    up-scaling and a direct float64 evaluation of the normalisation modules) is what the restatement still produces."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "image_small.json")))
    rgb = np.array(g["rgb"], np.float32).reshape(3, 10, 14)
    yuv = OI.rgb2yuv(rgb)
    assert np.array_equal(yuv.ravel(), np.array(g["yuv"], np.float32))
    assert np.array_equal(OI.scale_bilinear(yuv, 20, 15).ravel(), np.array(g["up_15x20"], np.float32))
    down = OI.scale_bilinear(yuv, 9, 7)
    assert np.array_equal(down.ravel(), np.array(g["down_7x9"], np.float32))
    norm = OI.center_and_scale(down)
    assert np.allclose(norm.ravel(), np.array(g["normalized"], np.float32), rtol=0, atol=1e-6)
    assert np.allclose(OI.contrastive_norm(norm[0], OI.gaussian1d(7)).ravel(), np.array(g["contrastive_y"], np.float32), rtol=0, atol=1e-5)
