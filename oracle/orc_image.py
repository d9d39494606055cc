"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of BatchIterator:processImage's image arithmetic
(BatchIterator.lua:101-164).  Imported by tests/ only; the product path never touches it.

PARITY UNPINNED.  processImage is a thin caller of two third-party luarocks that are NOT in /root/reference:
  * torch `image` (image.scale / hflip / vflip / crop / rgb2yuv / rgb2hsv / rgb2lab / gaussian1D), rockspec image-1.1.alpha
  * torch `nn`    (nn.SpatialContrastiveNormalization = Subtractive + Divisive normalisation), nn scm-1
Their algorithms are restated here from the published sources (image/generic/image.c `scaleLinear_rowcol`
+ `Main_scaleBilinear`, image/init.lua `gaussian1D` / `rgb2yuv`, nn/SpatialSubtractiveNormalization.lua,
nn/SpatialDivisiveNormalization.lua, TH `meanall` / `stdall`); no Torch7 runs in this image, so nothing
here was checked against the real packages.  Parity is anchored on the reference's call sites (the
argument values it passes: BatchIterator.lua:51,57,64,71,88-92,146-162) and on hand-derived known answers
(tests/test_oracle_image.py).

All arithmetic that the originals do in C `float` is done in np.float32 in the same order; reductions that TH
does in `accreal` (double for float tensors) are done in float64."""
import math

import numpy as np

f32 = np.float32


# ------------------------------------------------------------------ utilities.lua:188-203
def find_target_size(orig_w, orig_h, target_smaller_side, max_pixel_size):
    if orig_h < orig_w:
        w = min(orig_w * target_smaller_side / orig_h, max_pixel_size)
        h = math.floor(orig_h * w / orig_w + 0.5)
        w = math.floor(w + 0.5)
    else:
        h = min(orig_h * target_smaller_side / orig_w, max_pixel_size)
        w = math.floor(orig_w * h / orig_h + 0.5)
        h = math.floor(h + 0.5)
    assert w >= 1 and h >= 1
    return int(w), int(h)


def scaled_size(w, h, scale_x, scale_y):
    """BatchIterator.lua:51: image.scale(img, math.max(1, w*scaleX), math.max(1, h*scaleY)); the destination
    tensor is allocated with those doubles, which the tensor constructor truncates to integers."""
    return int(max(1, w * scale_x)), int(max(1, h * scale_y))


# ------------------------------------------------------------------ image.rgb2yuv (image/init.lua)
def rgb2yuv(img):
    r, g, b = img[0].astype(f32), img[1].astype(f32), img[2].astype(f32)
    y = f32(0.299) * r + f32(0.587) * g + f32(0.114) * b
    u = f32(-0.14713) * r - f32(0.28886) * g + f32(0.436) * b
    v = f32(0.615) * r - f32(0.51499) * g - f32(0.10001) * b
    return np.stack([y, u, v]).astype(f32)


# ------------------------------------------------------------------ image.rgb2hsv / image.rgb2lab (image/generic/image.c)
def rgb2hsv(img):
    """load_image with color_space 'hsv' (utilities.lua:214-215).  Per pixel, in float: v = max, s = (max-min)/max,
    h = the sextant formula / 6; a grey pixel (max == min) gets h = s = 0."""
    r, g, b = img[0].astype(f32), img[1].astype(f32), img[2].astype(f32)
    mx = np.maximum(np.maximum(r, g), b); mn = np.minimum(np.minimum(r, g), b)
    d = (mx - mn).astype(f32)
    grey = mx == mn
    dd = np.where(grey, f32(1), d)
    h = np.where(mx == r, (g - b) / dd + np.where(g < b, f32(6), f32(0)),
                 np.where(mx == g, (b - r) / dd + f32(2), (r - g) / dd + f32(4))).astype(f32)
    h = (h / f32(6)).astype(f32)
    s = (d / np.where(grey, f32(1), mx)).astype(f32)
    return np.stack([np.where(grey, f32(0), h), np.where(grey, f32(0), s), mx]).astype(f32)


def _srgb_expand(c):
    c64 = c.astype(np.float64)
    return np.where(c <= f32(0.04045), c64 / 12.92, np.power((c64 + 0.055) / 1.055, 2.4)).astype(f32)


def rgb2lab(img):
    """load_image with color_space 'lab' (utilities.lua:212-213).  sRGB gamma expansion (result kept in float), linear
    sRGB -> XYZ and the D65 white point in double, f(t) = t^(1/3) above epsilon = 216/24389 else (kappa t + 16)/116 with
    kappa = 24389/27, L = 116 fy - 16, a = 500 (fx - fy), b = 200 (fy - fz), stored as float."""
    r, g, b = (_srgb_expand(img[k].astype(f32)).astype(np.float64) for k in range(3))
    x = (0.412453 * r + 0.357580 * g + 0.180423 * b) / 0.950456
    y = 0.212671 * r + 0.715160 * g + 0.072169 * b
    z = (0.019334 * r + 0.119193 * g + 0.950227 * b) / 1.088754
    eps, kappa = 216.0 / 24389.0, 24389.0 / 27.0

    def f(t):
        return np.where(t > eps, np.cbrt(np.maximum(t, 0.0)), (kappa * t + 16.0) / 116.0)
    fx, fy, fz = f(x), f(y), f(z)
    return np.stack([116.0 * fy - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)]).astype(f32)


# ------------------------------------------------------------------ image.scale, mode 'bilinear' (the default)
def _scale_line(src, dst_len):
    """generic/image.c scaleLinear_rowcol along the LAST axis of `src` (float32), all other axes batched.
    Up-scaling: linear interpolation with scale (src_len-1)/(dst_len-1), last sample copied.
    Down-scaling: box filter over the source interval [di*scale, (di+1)*scale) with fractional end weights,
    divided by the accumulated weight.  Same length: copy."""
    src = np.ascontiguousarray(src, dtype=f32)
    src_len = src.shape[-1]
    dst = np.empty(src.shape[:-1] + (dst_len,), dtype=f32)
    if dst_len > src_len:
        if src_len == 1:
            dst[...] = src[..., :1]
            return dst
        scale = f32(src_len - 1) / f32(dst_len - 1)
        for di in range(dst_len - 1):
            si_f = f32(di) * scale
            si_i = int(si_f)
            si_f = f32(si_f - f32(si_i))
            dst[..., di] = (f32(1) - si_f) * src[..., si_i] + si_f * src[..., si_i + 1]
        dst[..., dst_len - 1] = src[..., src_len - 1]
    elif dst_len < src_len:
        scale = f32(src_len) / f32(dst_len)
        si0_i, si0_f = 0, f32(0)
        for di in range(dst_len):
            si1_f = f32(di + 1) * scale
            si1_i = int(si1_f)
            si1_f = f32(si1_f - f32(si1_i))
            acc = (f32(1) - si0_f) * src[..., si0_i]
            n = f32(1) - si0_f
            for si in range(si0_i + 1, si1_i):
                acc = acc + src[..., si]
                n = f32(n + f32(1))
            if si1_i < src_len:
                acc = acc + si1_f * src[..., si1_i]
                n = f32(n + si1_f)
            dst[..., di] = acc / n
            si0_i, si0_f = si1_i, si1_f
    else:
        dst[...] = src
    return dst


def scale_bilinear(img, dst_w, dst_h):
    """image.scale(img, dst_w, dst_h): Main_scaleBilinear -- every row to the new width into a temporary
    [C][H][dst_w], then every column of the temporary to the new height."""
    tmp = _scale_line(img, dst_w)                                   # rows
    out = _scale_line(np.swapaxes(tmp, 1, 2), dst_h)                # columns
    return np.ascontiguousarray(np.swapaxes(out, 1, 2))


def crop(img, x0, y0, x1, y1):      # image.crop(img, x0, y0, x1, y1): [x0, x1) x [y0, y1)
    return np.ascontiguousarray(img[:, y0:y1, x0:x1])


def hflip(img):
    return np.ascontiguousarray(img[:, :, ::-1])


def vflip(img):
    return np.ascontiguousarray(img[:, ::-1, :])


# ------------------------------------------------------------------ BatchIterator.lua:146-160
def center_and_scale(img, centering=True, scaling=True):
    """img[i]:add(-img[i]:mean()) then img[i]:div(img[i]:std()) when std > 1e-8.  TH: mean and the unbiased
    standard deviation accumulate in double; the Lua number is narrowed to float for the tensor op."""
    img = img.astype(f32).copy()
    for c in range(img.shape[0]):
        if centering:
            m = img[c].astype(np.float64).sum() / img[c].size
            img[c] = img[c] + f32(-m)
    for c in range(img.shape[0]):
        if scaling:
            x = img[c].astype(np.float64)
            m = x.sum() / x.size
            s = math.sqrt(((x - m) ** 2).sum() / (x.size - 1))
            if s > 1e-8:
                img[c] = img[c] / f32(s)
    return img


# ------------------------------------------------------------------ image.gaussian1D(size) with its defaults
def gaussian1d(size, sigma=0.25, amplitude=1.0, mean=0.5):
    center = mean * size + 0.5
    return np.array([amplitude * math.exp(-(((i - center) / (sigma * size)) ** 2) / 2) for i in range(1, size + 1)],
                    dtype=f32)


# ------------------------------------------------------------------ nn.SpatialContrastiveNormalization(1, k1d)
def _sep_conv_zero_pad(plane, k):
    """meanestimator of the normalisation modules for a 1-D kernel and one plane: zero padding floor(k/2) on
    every side, horizontal pass, vertical pass (float32 accumulation in tap order, as SpatialConvolution*)."""
    K = len(k); p = K // 2
    H, W = plane.shape
    pad = np.zeros((H + 2 * p, W + 2 * p), dtype=f32)
    pad[p:p + H, p:p + W] = plane
    hor = np.zeros((H + 2 * p, W), dtype=f32)
    for j in range(K):
        hor += k[j] * pad[:, j:j + W]
    out = np.zeros((H, W), dtype=f32)
    for j in range(K):
        out += k[j] * hor[j:j + H, :]
    return out


def contrastive_norm(plane, k1d, threshold=1e-4):
    """SpatialSubtractiveNormalization then SpatialDivisiveNormalization on one plane (nInputPlane = 1).
    Both modules normalise the kernel to unit sum and divide their estimate by `coef` = the same estimator
    applied to a plane of ones (border correction); the divisive module divides sqrt(estimate of x^2) by the
    UN-rooted coef and replaces values <= threshold by thresval (= threshold)."""
    plane = plane.astype(f32)
    k = (k1d.astype(f32) / f32(k1d.astype(np.float64).sum())).astype(f32)   # kernel:div(kernel:sum() * 1), sum in accreal
    ones = np.ones_like(plane)
    coef = _sep_conv_zero_pad(ones, k)
    sub = plane - _sep_conv_zero_pad(plane, k) / coef
    std = np.sqrt(_sep_conv_zero_pad(sub * sub, k)) / coef
    std = np.where(std > f32(threshold), std, f32(threshold)).astype(f32)
    return (sub / std).astype(f32)


# ------------------------------------------------------------------ BatchIterator.lua:101-164 without the RNG
def process_image(img, cfg, hflip_on=False, vflip_on=False):
    """processImage for random_scaling = 0 (both shipped configs): resize to find_target_size, optional flips
    (the caller draws math.random() < aug.hflip / aug.vflip), per-channel centring and scaling, contrastive
    normalisation of channel 1."""
    C, H, W = img.shape
    tw, th = find_target_size(W, H, cfg["target_smaller_side"], cfg["max_pixel_size"])
    sw, sh = scaled_size(W, H, tw / W, th / H)
    out = scale_bilinear(img, sw, sh)
    if hflip_on:
        out = hflip(out)
    if vflip_on:
        out = vflip(out)
    nz = cfg["normalization"]
    out = center_and_scale(out, nz.get("centering", False), nz.get("scaling", False))
    if nz.get("method") == "contrastive":
        out[0] = contrastive_norm(out[0], gaussian1d(nz["width"]))
    return out
