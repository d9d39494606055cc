"""BatchIterator.lua:1-317 -- training / validation batch assembly with the image preparation on the device
(SURVEY 8f-1).  The reference loads a JPEG with `image.load`, converts the colour space and runs
`processImage` on the CPU; here the decoded frame goes to HBM once and every step of processImage is a HIP
kernel (frcnn_image_*: rgb2yuv, image.scale, crop + flips, per-channel centring / scaling, contrastive
normalisation of the luminance channel), so the frame is never touched by the host again before pnet:forward.

Differences that are stated rather than hidden:
  * decoding stays on the host: `load_image(fn)` is a callable that returns the DECODED float RGB frame [3][H][W] in
    0..1 (what image.load(fn, 3, 'float') returns); the default decodes with Pillow (`.npy` arrays are read as they
    are).  color_space 'yuv' (both shipped configs) is fused into the scaling pass; 'lab' and 'hsv' are converted at
    full resolution first (frcnn_image_rgb2lab / _rgb2hsv), decode-ahead frames included; anything else stays RGB.
  * `math.random` is LuaJIT's own PRNG (not reproducible outside LuaJIT): every draw comes from the MT19937
    stream also used for torch.random / torch.randperm (same substitution as Anchors.sampleNegative).
The sequence of draws follows the reference line by line (BatchIterator.lua:112-143, :7-25)."""
import copy
import ctypes as C
import math
import os
import sys

import numpy as np

from . import _lib
from .Anchors import Anchors, MT19937
from .Rect import Rect
from .tensor import DeviceTensor, ptr, stream_ptr, to_device


def find_target_size(orig_w, orig_h, target_smaller_side, max_pixel_size):  # utilities.lua:188-203
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


def gaussian1D(size, sigma=0.25, amplitude=1.0, mean=0.5):
    """image.gaussian1D with its defaults (float tensor under main.lua:51)."""
    center = mean * size + 0.5
    return np.array([amplitude * math.exp(-(((i - center) / (sigma * size)) ** 2) / 2) for i in range(1, size + 1)],
                    dtype=np.float32)


def decode_image(fn):
    """image.load(fn, 3, 'float') (utilities.lua load_image): the decoded frame as float RGB [3][H][W] in 0..1.
    `.npy` files hold that array directly; everything else goes through Pillow (8-bit samples / 255, grey and
    palette images expanded to three channels, alpha dropped -- what image.load does for depth 3)."""
    if fn.lower().endswith(".npy"):
        return np.load(fn)
    try:
        from PIL import Image
    except ImportError as e:
        raise _lib.FrcnnError("decoding '%s' needs Pillow (or pass load_image=... / use .npy frames): %s" % (fn, e))
    with Image.open(fn) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.float32) * np.float32(1.0 / 255.0)
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def decode_image_u8(fn):
    """The decoder's own output: 8-bit interleaved RGB [H][W][3] (the device converts it, frcnn_image_scale_u8)."""
    from PIL import Image
    with Image.open(fn) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


class _U8Frame(object):
    """A decoded frame in HBM as 8-bit interleaved RGB; `event` marks the end of its upload on the copy stream."""

    def __init__(self, dev, H, W, event, slot=None, recycle=None):
        self.dev, self.event, self.slot, self.recycle = dev, event, slot, recycle
        self.shape = (3, H, W)

    def __del__(self):
        # a frame that never reached processImage (an exception, a caller that dropped it) still hands its device buffer
        # back to the pool instead of leaving it to the allocator
        r, self.recycle = getattr(self, "recycle", None), None
        if r is not None:   # (processImage clears `recycle` when it has handed the buffer back itself)
            try:
                r()
            except Exception:
                pass


class _DecodeAhead(object):
    """Worker threads decode upcoming files (Pillow releases the GIL while it decodes) and upload the 8-bit frames
    through pinned buffers on a copy stream of their own; the consumer only ever waits for an event.  At 200+ images/s
    per GPU a single-threaded decode (10-20 ms per 1080p JPEG) would otherwise be the bottleneck of the step."""

    def __init__(self, workers, resolve, cache_bytes=2 << 30):
        import concurrent.futures
        import torch
        self.torch = torch
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers)
        # the consumer thread re-acquires the GIL after every C-ABI call; with the default 5 ms switch interval it can wait
        # that long behind a decoding thread that is in a Python-level stretch (convoy effect)
        if sys.getswitchinterval() > 2e-4:
            sys.setswitchinterval(2e-4)
        self.copy_stream = torch.cuda.Stream()
        self.resolve = resolve
        self.jobs = {}
        # Device frames and pinned staging buffers are recycled by capacity bucket with a bound on what is kept (no
        # allocator traffic -- and no hipMalloc synchronisation -- in the steady state, no growth with the number of
        # distinct frame sizes of the data set); a device frame is overwritten only after the kernels that read its
        # previous content (event `consumed`, waited for by the copy stream).
        self.dev_pool = _BufferPool(lambda n: dict(dev=torch.empty((n,), dtype=torch.uint8, device="cuda"), consumed=None),
                                    cache_bytes)
        self.pin_pool = _BufferPool(lambda n: torch.empty((n,), dtype=torch.uint8).pin_memory(), max(cache_bytes // 4, 64 << 20))
        self.inflight = []

    def _work(self, fn, base):
        """worker thread: decode into a pinned host buffer -- no HIP call here (the runtime's locks are shared with the
        consumer thread's kernel launches; uploads issued from 16 threads slowed the training step down)"""
        a = decode_image_u8(self.resolve(fn, base))
        pin, pb = self.pin_pool.take(a.nbytes)
        C.memmove(pin.data_ptr(), a.ctypes.data, a.nbytes)   # (a foreign call: the GIL is released for the 6 MB copy)
        return pin, pb, a.shape

    def request(self, fn, base=""):
        if (fn, base) not in self.jobs:
            self.jobs[(fn, base)] = self.pool.submit(self._work, fn, base)

    def forget(self, fn=None, base=""):
        """Drop prefetched decodes nobody will ask for: one file that was loaded another way (materialize / a custom path),
        or -- fn None -- everything, when the epoch's order is redrawn.  Their pinned buffers go back to the pool."""
        keys = [k for k in self.jobs if fn is None or k == (fn, base)]
        for k in keys:
            fut = self.jobs.pop(k)
            if not fut.cancel():
                def give(f, self=self):
                    try:
                        pin, pb, _ = f.result()
                        self.pin_pool.give(pin, pb)
                    except Exception:
                        pass
                fut.add_done_callback(give)

    def get(self, fn, base=""):
        """consumer thread: wait for the decode, queue the upload on the copy stream, hand out the frame + its event"""
        torch = self.torch
        self.request(fn, base)
        pin, pb, shape = self.jobs.pop((fn, base)).result()
        nbytes = int(shape[0]) * int(shape[1]) * int(shape[2])
        slot, sb = self.dev_pool.take(nbytes)
        with torch.cuda.stream(self.copy_stream):
            if slot["consumed"] is not None:
                self.copy_stream.wait_event(slot["consumed"])
                slot["consumed"] = None
            slot["dev"][:nbytes].copy_(pin[:nbytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self.inflight.append((pin, pb, ev))
        while self.inflight and self.inflight[0][2].query():   # pinned buffers whose upload has landed go back to the pool
            p, b, _ = self.inflight.pop(0)
            self.pin_pool.give(p, b)
        return _U8Frame(slot["dev"], shape[0], shape[1], ev, slot, lambda: self.dev_pool.give(slot, sb))


def _bucket(nbytes):
    """Capacity bucket of a request: the next value of the form (8 + k) / 8 * 2^e (k = 0..7) -- at most 12.5 % slack, and
    frames of nearly the same size (every ImageNet file has its own) share buffers instead of each caching its own."""
    n = max(int(nbytes), 4096)
    e = n.bit_length() - 4          # 8 <= n >> e < 16
    m = -(-n >> e)                  # ceil
    return m << e


class _BufferPool(object):
    """Recycles device (or pinned host) buffers by capacity bucket (hipMalloc / hipFree synchronise the device, so the
    steady state must not allocate) with a BOUND on what it keeps: buffers handed back are cached until the cached bytes
    exceed `max_bytes`, then the least recently used buckets are released.  A buffer is handed back either explicitly
    (intermediates, once the launch that read them is queued on the same stream) or when the last reference to the tensor
    built on it is dropped (`_PooledTensor.__del__`: the batch that held the frame is gone) -- reuse is then ordered
    behind the consumer's kernels by the stream itself."""

    def __init__(self, alloc, max_bytes, release=None):
        import collections
        import threading
        self.alloc, self.release, self.max_bytes = alloc, release, int(max_bytes)
        self.free = collections.OrderedDict()    # bucket -> [buffers], least recently used bucket first
        self.cached = 0
        self.allocated = 0                        # bytes handed out + cached (for tests / diagnostics)
        self.lock = threading.Lock()

    def take(self, nbytes):
        b = _bucket(nbytes)
        with self.lock:
            lst = self.free.get(b)
            if lst:
                buf = lst.pop()
                self.cached -= b
                if lst:
                    self.free.move_to_end(b)
                else:
                    del self.free[b]
                return buf, b
            self.allocated += b
        return self.alloc(b), b

    def give(self, buf, b):
        drop = []
        with self.lock:
            self.free.setdefault(b, []).append(buf)
            self.free.move_to_end(b)
            self.cached += b
            while self.cached > self.max_bytes and self.free:
                ob, lst = next(iter(self.free.items()))
                drop.append(lst.pop())
                self.cached -= ob
                self.allocated -= ob
                if not lst:
                    del self.free[ob]
        for d in drop:
            if self.release is not None:
                self.release(d)


class _PooledTensor(DeviceTensor):
    """A DeviceTensor on a pool buffer; the buffer returns to the pool with the last reference."""

    def __init__(self, pool, shape, dtype=np.float32):
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        self._pool = pool
        self._buf, self._bucket = pool.take(nbytes)
        DeviceTensor.__init__(self, self._buf.ptr, shape, dtype, owner=self._buf)

    def release(self):
        pool, self._pool = self._pool, None
        if pool is not None:
            pool.give(self._buf, self._bucket)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _device_pool(max_bytes):
    return _BufferPool(lambda n: DeviceTensor.empty((n,), np.uint8), max_bytes)   # (dropping the DeviceTensor frees it)


def _copy_rois(rois):  # deep_copy(self.ground_truth[fn].rois) (BatchIterator.lua:172): the rects are rewritten below
    out = []
    for r in rois:
        c = copy.copy(r)
        c.rect = r.rect.clone()
        out.append(c)
    return out


def _transform_rois(rois, froi, old_w, old_h, new_w, new_h):  # BatchIterator.lua:27-47 (the roi half)
    result = []
    img_rect = Rect(0, 0, new_w, new_h)
    for roi in rois or []:
        r = froi(roi.rect, old_w, old_h)
        if r is not None:
            r = r.clip(img_rect)
            if not r.isEmpty():
                roi.rect = r
                result.append(roi)
    return result


# ---- BatchIterator.lua:198-225: the examples of one image -----------------------------------------------
def assemble_examples_native(anchors, cfg, rois, W, H, rng, negatives=16):
    """BatchIterator.lua:198-225 for one image through frcnn_anchors_assemble (host-side native code, the same lists as
    the Python path below draw for draw)."""
    import ctypes as C
    from . import _lib
    nroi = len(rois)
    ra = np.array([(r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY) for r in rois], dtype=np.float64).reshape(-1, 4)
    cap = 8192
    ex = np.empty((cap, 5), dtype=np.int32); er = np.empty((cap, 4), dtype=np.float64)
    npos, nneg = C.c_int(0), C.c_int(0)
    _lib.call("frcnn_anchors_assemble", anchors.native(), ra.ctypes.data_as(C.c_void_p), nroi, float(W), float(H),
              float(cfg["positive_threshold"]), float(cfg["negative_threshold"]), int(bool(cfg["best_match"])),
              int(bool(cfg.get("nearby_aversion"))), int(negatives), rng.state.ctypes.data_as(C.c_void_p), C.byref(rng.cidx),
              ex.ctypes.data_as(C.c_void_p), er.ctypes.data_as(C.c_void_p), cap, C.byref(npos), C.byref(nneg))
    tag = Anchors._tag
    exl, erl = ex[:npos.value + nneg.value].tolist(), er[:npos.value + nneg.value].tolist()
    positive = [(tag(Rect(*erl[k]), exl[k][0], exl[k][1], exl[k][2], exl[k][3]), rois[exl[k][4] - 1]) for k in range(npos.value)]
    negative = [(tag(Rect(*erl[k]), exl[k][0], exl[k][1], exl[k][2], exl[k][3]),) for k in range(npos.value, npos.value + nneg.value)]
    return positive, negative


def assemble_examples(anchors, cfg, rois, W, H, rng, negatives=16, native=None):
    """BatchIterator.lua:198-225 for one image.  native=None: the native twin unless FRCNN_NATIVE_ASSEMBLE=0."""
    import os
    if native is None:
        native = os.environ.get("FRCNN_NATIVE_ASSEMBLE", "1") != "0"
    if native:
        return assemble_examples_native(anchors, cfg, rois, W, H, rng, negatives)
    img_rect = Rect(0, 0, W, H)
    positive = anchors.findPositive(rois, img_rect, cfg["positive_threshold"], cfg["negative_threshold"], cfg["best_match"])
    negative = anchors.sampleNegative(img_rect, rois, cfg["negative_threshold"], negatives, rng)
    count = len(positive) + len(negative)
    if cfg.get("nearby_aversion"):
        # every anchor that shares a bin pair with a positive's centre and overlaps it by less than the negative threshold
        # (BatchIterator.lua:204-216).  Vectorised: the candidates of all positives in one IoU evaluation (Rect.IoU's
        # arithmetic in float64, candidates in the order of the nested Lua loops); Rect objects are only built for the
        # few candidates that survive the shuffle.
        nearby = []
        if positive:
            parts = [anchors.findNearbyArrays(*p[0].center()) for p in positive]
            cnt = np.array([len(m) for m, _ in parts])
            if cnt.sum():
                M = np.concatenate([m for m, _ in parts]); R = np.concatenate([r for _, r in parts])
                P = np.repeat(np.array([(p[0].minX, p[0].minY, p[0].maxX, p[0].maxY) for p in positive], dtype=np.float64), cnt, axis=0)
                minx = np.maximum(P[:, 0], R[:, 0]); miny = np.maximum(P[:, 1], R[:, 1])
                maxx = np.minimum(P[:, 2], R[:, 2]); maxy = np.minimum(P[:, 3], R[:, 3])
                ok = (maxx >= minx) & (maxy >= miny)
                inter = np.where(ok, (maxx - minx) * (maxy - miny), 0.0)
                area = lambda A: (A[:, 2] - A[:, 0]) * (A[:, 3] - A[:, 1])
                with np.errstate(divide="ignore", invalid="ignore"):
                    iou = inter / (area(P) + area(R) - inter)
                Mk = M[iou < cfg["negative_threshold"]]
                nearby = list(range(len(Mk)))   # (the shuffle permutes positions; the rows are looked up afterwards)
        c = min(len(positive), count)
        c = min(c, len(nearby))
        # shuffle_n (utilities.lua:31-42) with the MT19937 stream instead of LuaJIT's math.random
        r = len(nearby)
        for i in range(c):
            j = rng.random() % r + i
            nearby[i], nearby[j] = nearby[j], nearby[i]
            r -= 1
        negative.extend((anchors.get(*(int(v) for v in Mk[t])),) for t in nearby[:c])
    return positive, negative



class _RgbFrame(object):
    """A decoded RGB frame in HBM whose image.rgb2yuv conversion is still pending: processImage folds it into the
    row pass of image.scale (frcnn_image_scale, rgb2yuv = 1) instead of materialising the full-resolution YUV frame."""

    def __init__(self, rgb):
        self.rgb = rgb
        self.shape = tuple(rgb.shape)


class BatchIterator(object):
    def __init__(self, model, training_data, load_image=None, seed=5489, workers=0, prefetch=8, cache_bytes=2 << 30):
        # BatchIterator.lua:82-99.  workers > 0 (default loader only): image files are decoded `prefetch` entries ahead by
        # a pool of threads and uploaded as 8-bit frames (see _DecodeAhead).  cache_bytes bounds the device memory kept
        # for recycling (frames of a batch are held by the batch itself and return to the pool when it is dropped).
        cfg = model["cfg"]
        self.cfg = cfg
        self.ground_truth = training_data["ground_truth"]
        nz = cfg["normalization"]
        self.kernel = gaussian1D(nz["width"]) if nz.get("method") == "contrastive" else None  # :88-92
        self.anchors = Anchors(model["pnet"], cfg["scales"])
        self.rng = MT19937(seed)
        base = cfg.get("examples_base_path") or ""
        bg_base = cfg.get("background_base_path") or ""     # background files have their own base (BatchIterator.lua:255)
        resolve = lambda fn, b: fn if os.path.isabs(fn) or not b else os.path.join(b, fn)
        self.base_of = {"examples": base, "background": bg_base}
        self.load_image_fn = load_image or (lambda fn, b=base: decode_image(resolve(fn, b)))
        self._custom_loader = load_image is not None
        self.training = dict(order=[], list=list(training_data["training_set"]))
        self.validation = dict(order=[], list=list(training_data.get("validation_set", [])))
        self.background = dict(order=[], list=list(training_data.get("background_files") or []))
        self._randomize_order(self.training, self.validation, self.background)
        self.ahead = None
        if workers > 0 and load_image is None:
            self.ahead = _DecodeAhead(workers, resolve, cache_bytes=cache_bytes)
            self.prefetch = prefetch
        self.pool = _device_pool(cache_bytes)
        self.scratch = {}
        self.log = lambda msg: None   # the reference prints one line per image (:249); silent by default

    # ---- BatchIterator.lua:7-25
    def _randomize_order(self, *sets):
        if getattr(self, "ahead", None) is not None:
            self.ahead.forget()    # decodes requested for the old order are not waited for any more
        for x in sets:
            if x["list"]:
                x["order"] = self.rng.randperm(len(x["list"]))
            x["i"] = 1

    def _next_entry(self, s):
        if s["i"] > len(s["list"]):
            self._randomize_order(s)
        fn = s["list"][s["order"][s["i"] - 1] - 1]
        s["i"] += 1
        if self.ahead is not None:   # decode the following entries of this epoch's order ahead of time (no RNG involved)
            base = self.base_of["background" if s is self.background else "examples"]
            for k in range(s["i"], min(s["i"] + self.prefetch, len(s["list"]) + 1)):
                self.ahead.request(s["list"][s["order"][k - 1] - 1], base)
        return fn

    def _tmp(self, name, n):
        t = self.scratch.get(name)
        if t is None or t.numel() < n:
            t = self.scratch[name] = DeviceTensor.empty((max(n, 1),))
        return t

    # ---- utilities.lua load_image: decoded RGB frame -> device, colour space conversion
    def load_image(self, fn, materialize=False, what="examples"):
        """what: 'examples' (cfg.examples_base_path) or 'background' (cfg.background_base_path, BatchIterator.lua:255)."""
        base = self.base_of[what]
        cs = self.cfg.get("color_space", "rgb")
        full_frame = cs in ("lab", "hsv")   # not fused into the scaling pass: converted at full resolution, as load_image does
        if self.ahead is not None and not materialize and not full_frame:
            return self.ahead.get(fn, base)
        if self.ahead is not None:
            self.ahead.forget(fn, base)   # loaded here instead: its prefetched decode (if any) is not kept around
        img = to_device(self.load_image_fn(fn) if self._custom_loader else self.load_image_fn(fn, base))
        if len(img.shape) != 3 or img.shape[0] != 3:
            return img   # the caller reports the unexpected channel count (:185-188)
        if cs == "yuv" and not materialize:
            return _RgbFrame(img)   # converted inside processImage's first pass
        if cs in ("yuv", "lab", "hsv"):   # utilities.lua:210-216
            out = _PooledTensor(self.pool, tuple(img.shape))
            _lib.call("frcnn_image_rgb2" + cs, ptr(img), ptr(out), img.shape[1], img.shape[2], stream_ptr())
            return out
        return img   # 'rgb', and (as in the reference) any other string: the frame stays RGB

    # ---- BatchIterator.lua:101-164
    def processImage(self, img, rois=None):
        cfg, aug, s = self.cfg, self.cfg["augmentation"], stream_ptr()
        u8 = img if isinstance(img, _U8Frame) else None
        to_yuv = isinstance(img, _RgbFrame)
        if u8 is None:
            img = img.rgb if to_yuv else to_device(img)
        Cn, H, W = img.shape
        tw, th = find_target_size(W, H, cfg["target_smaller_side"], cfg["max_pixel_size"])
        scale_X, scale_Y = tw / W, th / H
        if aug.get("random_scaling") and aug["random_scaling"] > 0:   # :112-115 (restated as written)
            scale_X = tw * (self.rng.uniform() - 0.5) * aug["random_scaling"] / W
            scale_Y = scale_X + (self.rng.uniform() - 0.5) * aug["aspect_jitter"]
        # scale (:117, :49-55): the destination size is truncated by the tensor constructor
        sw, sh = int(max(1, W * scale_X)), int(max(1, H * scale_Y))
        cur = _PooledTensor(self.pool, (Cn, sh, sw))
        if u8 is not None:   # 8-bit frame from the decode-ahead pool: float conversion (+ yuv) fused into the row pass
            import torch
            torch.cuda.current_stream().wait_event(u8.event)
            _lib.call("frcnn_image_scale_u8", ptr(u8.dev), H, W, ptr(cur), sh, sw, ptr(self._tmp("scale", Cn * H * sw)),
                      int(cfg.get("color_space", "rgb") == "yuv"), s)
            done = torch.cuda.Event(); done.record()
            u8.slot["consumed"] = done   # the frame's buffer may be overwritten once this launch has read it
            if u8.recycle is not None:
                u8.recycle(); u8.recycle = None
        else:
            _lib.call("frcnn_image_scale", ptr(img), Cn, H, W, ptr(cur), sh, sw, ptr(self._tmp("scale", Cn * H * sw)),
                      int(to_yuv), s)
        rois = _transform_rois(rois, lambda r, w, h: r.scale(scale_X, scale_Y), W, H, sw, sh)
        # crop to the target size if a dimension was up-sampled beyond it (:119-130)
        cw, ch, x0, y0 = sw, sh, 0, 0
        if sw > tw or sh > th:
            cw, ch = min(tw, sw), min(th, sh)
            x0 = int(math.floor(self.rng.uniform() * (sw - cw)))
            y0 = int(math.floor(self.rng.uniform() * (sh - ch)))
            rect = Rect.fromXYWidthHeight(x0, y0, cw, ch)
            rois = _transform_rois(rois, lambda r, w, h: r.clip(rect).offset(-rect.minX, -rect.minY), sw, sh, cw, ch)
        hf = vf = False
        if aug.get("hflip") and aug["hflip"] > 0 and self.rng.uniform() < aug["hflip"]:   # :132-137
            hf = True
            rois = _transform_rois(rois, lambda r, w, h: Rect(w - r.maxX, r.minY, w - r.minX, r.maxY), cw, ch, cw, ch)
        if aug.get("vflip") and aug["vflip"] > 0 and self.rng.uniform() < aug["vflip"]:   # :139-144
            vf = True
            rois = _transform_rois(rois, lambda r, w, h: Rect(r.minX, h - r.maxY, r.maxX, h - r.minY), cw, ch, cw, ch)
        if hf or vf or (cw, ch) != (sw, sh):   # crop and both flips are one gather
            nxt = _PooledTensor(self.pool, (Cn, ch, cw))
            _lib.call("frcnn_image_crop_flip", ptr(cur), Cn, sh, sw, x0, y0, cw, ch, int(hf), int(vf), ptr(nxt), s)
            cur.release()    # an intermediate: reusable behind this launch (same stream)
            cur = nxt
        nz = cfg["normalization"]
        if nz.get("centering") or nz.get("scaling"):   # :146-160
            wsb = _lib.load().frcnn_image_normalize_workspace_bytes(Cn)
            ws = self._tmp("norm", (wsb + 3) // 4)
            _lib.call("frcnn_image_normalize", ptr(cur), Cn, ch, cw, int(bool(nz.get("centering"))),
                      int(bool(nz.get("scaling"))), ptr(ws), wsb, s)
        if self.kernel is not None:   # :162 img[1] = normalization:forward(img[{{1}}])
            y = cur.offset_view(0, (ch, cw))
            _lib.call("frcnn_image_contrastive_norm", ptr(y), ch, cw, self.kernel.ctypes.data_as(C.c_void_p),
                      len(self.kernel), 1e-4, ptr(y), ptr(self._tmp("cn", ch * cw)), s)
        return cur, rois

    # ---- BatchIterator.lua:166-277
    def nextTraining(self, count=None):
        cfg = self.cfg
        batch = []
        count = count or cfg["batch_size"]

        def checked_load(fn, what):
            try:
                img = self.load_image(fn, what=what)
            except Exception as e:   # pcall: ImageNet contains invalid files (:176-181)
                self.log("Invalid image '%s': %s" % (fn, e))
                return None
            if len(img.shape) != 3 or img.shape[0] != 3:
                self.log("Warning: Skipping image '%s'. Unexpected channel count" % fn)
                return None
            return img

        def try_add_next():
            fn = self._next_entry(self.training)
            rois = _copy_rois(self.ground_truth[fn]["rois"])
            img = checked_load(fn, "examples")
            if img is None:
                return 0
            img, rois = self.processImage(img, rois)
            _, h, w = img.shape
            if h < 128 or w < 128:   # :192-196
                self.log("Warning: Skipping image '%s'. Invalid size after process: (%dx%d)" % (fn, w, h))
                return 0
            positive, negative = assemble_examples(self.anchors, cfg, rois, w, h, self.rng)   # :198-225
            batch.append(dict(img=img, positive=positive, negative=negative))
            self.log("'%s' (%dx%d); p: %d; n: %d" % (fn, w, h, len(positive), len(negative)))
            return len(positive) + len(negative)

        if self.background["list"]:   # one background image per batch with 5 % of the examples (:253-270)
            fn = self._next_entry(self.background)
            img = checked_load(fn, "background")
            if img is not None:
                img, _ = self.processImage(img)
                _, h, w = img.shape
                if h >= 128 and w >= 128:
                    negative = self.anchors.sampleNegative(Rect(0, 0, w, h), [], 0, int(math.floor(count * 0.05)), self.rng)
                    batch.append(dict(img=img, positive=[], negative=negative))
                    count -= len(negative)
        guard = 0
        while count > 0:
            n = try_add_next()
            count -= n
            guard = guard + 1 if n == 0 else 0
            if guard > 10 * max(1, len(self.training["list"])):   # (the reference would spin forever on an unusable set)
                raise _lib.FrcnnError("nextTraining: no usable training image")
        return batch

    # ---- BatchIterator.lua:279-317
    def nextValidation(self, count=1):
        batch = []
        while count > 0:
            fn = self._next_entry(self.validation)
            try:
                img = self.load_image(fn)
            except Exception as e:
                self.log("Invalid image '%s': %s" % (fn, e))
                continue
            if len(img.shape) != 3 or img.shape[0] != 3:
                continue
            rois = _copy_rois(self.ground_truth[fn]["rois"])
            img, rois = self.processImage(img, rois)
            _, h, w = img.shape
            if h < 128 or w < 128:
                continue
            batch.append(dict(img=img, rois=rois))
            count -= 1
        return batch
