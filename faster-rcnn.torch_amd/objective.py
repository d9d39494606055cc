"""extract_roi_pooling_input and create_objective -- host-side mirror of objective.lua.

lossAndGradient(w) keeps the structure of objective.lua:45-218 (per image: pnet forward, sparse RPN
loss on the sampled anchors, ROI adaptive max-pooling, cnet forward/backward, ROI-pool backward,
pnet backward; then normalise and log), but every per-example Lua loop that touched the device one
scalar at a time is ONE batched kernel here (frcnn_rpn_loss, frcnn_roi_pool_forward/backward,
frcnn_cnet_losses), and the eight statistics of objective.lua:52-58 stay in a device-side fp64
vector until the single read-back at the end of the call.

Data parallelism (not in the reference): when torch.distributed is initialised, every
rank processes its own images and the flat gradient + the 8 accumulators are all-reduced (sum) just
before `gradient:div(cls_count)` (objective.lua:200), so the result equals the single-process result
on the concatenated batch."""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from .Anchors import Anchors
from .Localizer import Localizer
from .Rect import Rect
from .tensor import DeviceTensor, ptr, stream_ptr, to_device


def roi_window(input_rect, localizer, fm_h, fm_w):
    """The index table of objective.lua:11: rows {min(minY+1,maxY), maxY}, cols {min(minX+1,maxX), maxX}
    (1-based inclusive) of the feature map covered by input_rect."""
    r = localizer.inputToFeatureRect(input_rect)
    r = r.clip(Rect(0, 0, fm_w, fm_h))
    return (int(min(r.minY + 1, r.maxY)), int(r.maxY), int(min(r.minX + 1, r.maxX)), int(r.maxX))


def roi_windows(rects, localizer, fm_h, fm_w):
    """roi_window() for an (n, 4) array of input rects -> int32 (n, 4), vectorised."""
    r = localizer.inputToFeatureRectBatch(rects)
    minX = np.minimum(np.maximum(r[:, 0], 0), fm_w); minY = np.minimum(np.maximum(r[:, 1], 0), fm_h)   # Rect.clip
    maxX = np.maximum(np.minimum(r[:, 2], fm_w), 0); maxY = np.maximum(np.minimum(r[:, 3], fm_h), 0)
    return np.stack([np.minimum(minY + 1, maxY), maxY, np.minimum(minX + 1, maxX), maxX], 1).astype(np.int32)


def extract_roi_pooling_input(input_rect, localizer, feature_layer_output):  # objective.lua:5-13
    """Returns (window, idx): idx = {{}, {row_lo,row_hi}, {col_lo,col_hi}} like the reference; the strided
    view itself is never materialised -- the batched ROI-pool kernel reads the window in place."""
    s = feature_layer_output.shape
    win = roi_window(input_rect, localizer, s[1], s[2])
    idx = ((), (win[0], win[1]), (win[2], win[3]))
    return win, idx


class _Scratch(object):
    """Per-objective device scratch that grows on demand (no allocation in the steady state)."""

    def __init__(self):
        self.bufs = {}

    def get(self, name, shape, dtype=np.float32):
        need = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        b = self.bufs.get(name)
        if b is None or b.nbytes < need:
            b = DeviceTensor.empty((max(need, 256),), np.uint8)
            self.bufs[name] = b
        return DeviceTensor(b.ptr, shape, dtype, owner=b)


class _PinnedRing(object):
    """Page-locked staging buffers for the per-image example tables: an upload from ordinary host memory blocks the host until
    the stream has reached it (the host then issues the launches that follow with the device already waiting: measured as a
    ~0.45 ms hole in every step); from page-locked memory frcnn_memcpy_h2d is asynchronous and the host keeps its lead.  A slot
    is reused only after the copy that read it has run (event)."""

    def __init__(self, slots=4):
        self.slots = [None] * slots
        self.i = 0

    def stage(self, arr):
        """Copy the uint8 array into the next slot; returns (host address, done) -- call done() right after queuing the copy."""
        import torch
        k = self.i
        self.i = (k + 1) % len(self.slots)
        slot = self.slots[k]
        n = int(arr.nbytes)
        if slot is not None and slot["event"] is not None:
            slot["event"].synchronize()
        if slot is None or slot["bytes"] < n:
            if slot is not None:
                _lib.call("frcnn_host_free", C.c_void_p(slot["ptr"]))
            p = C.c_void_p()
            cap = max(n, 1 << 16)
            _lib.call("frcnn_host_alloc", C.byref(p), cap)
            slot = self.slots[k] = dict(ptr=p.value, bytes=cap, event=None)
        C.memmove(slot["ptr"], arr.ctypes.data, n)

        def done():
            ev = torch.cuda.Event()
            ev.record()
            slot["event"] = ev
        return slot["ptr"], done

    def __del__(self):
        for slot in self.slots:
            if slot is not None:
                try:
                    _lib.load().frcnn_host_free(C.c_void_p(slot["ptr"]))
                except Exception:
                    pass


class DeviceDivisor(object):
    """gradient:div(n) (objective.lua:200) with n on the device: the address of an fp64 count (plus what keeps it alive).
    utilities.rmsprop hands it to frcnn_scale_rmsprop_dev."""

    def __init__(self, p, owner):
        self.ptr = int(p)
        self._owner = owner


def _dist():
    """The exchange back end of the step: the library's own communicator when one is active (comm.activate: RCCL through
    the C ABI, what a LuaJIT host uses), else an initialised torch.distributed process group, else None."""
    from . import comm
    c = comm.active()
    if c is not None:
        return c
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
    except ImportError:
        pass
    return None


# bench.py at N = 1: a list that receives (label, lo, hi, waits_on, event) at every point of the pass where the all-reduce of a
# bucket of the flat gradient WOULD start at N > 1 (the same program points, the same streams), plus "step_begin" /
# "backward_end" marks -- the bucket schedule a first multi-GPU run can be compared with.  None: off.
exchange_probe = None


def _probe_mark(label, lo, hi, waits_on):
    import torch
    e = torch.cuda.Event(enable_timing=True)
    e.record()      # (on the current stream: the one the bucket's all-reduce would be ordered behind)
    exchange_probe.append((label, int(lo), int(hi), waits_on, e))


def allreduce_begin(gradient, lo, hi):
    """Start the all-reduce of gradient[lo:hi] (a bucket whose values are final) without waiting for it: RCCL
    runs it on its own stream, ordered after the kernels already queued on the current stream, beside whatever
    the caller queues next.  Returns a handle for allreduce_gradient_and_stats(pending=...) or None (single
    process)."""
    dist = _dist()
    if dist is None or hi <= lo:
        return None
    return (lo, hi, dist.all_reduce(gradient[lo:hi], async_op=True))


def allreduce_begin_rest(gradient, pending):
    """Start the all-reduce of every part of `gradient` that no handle in `pending` covers yet (the whole vector
    is final).  Returns the extended list; lets the exchange run while the host still reads the accumulators."""
    if _dist() is None:
        return list(pending)
    done = sorted((p[0], p[1]) for p in pending if p is not None)
    out = [p for p in pending if p is not None]
    pos = 0
    for lo, hi in done + [(gradient.numel(), gradient.numel())]:
        if lo > pos:
            out.append(allreduce_begin(gradient, pos, lo))
        pos = max(pos, hi)
    return out


def allreduce_gradient_and_stats(gradient, tot, pending=()):
    """Data-parallel exchange step: sum the flat gradient (in place) and the 8 fp64
    accumulators {cls_loss, reg_loss, cls_count, reg_count, creg_loss, creg_count, ccls_loss,
    ccls_count} (objective.lua:52-58) over all ranks.  `gradient` is a torch tensor (CUDA: RCCL over
    xGMI through torch.distributed's "nccl" backend; CPU: gloo, used by the tests).  Buckets already started
    with allreduce_begin() (`pending`) are only waited for; the rest of the vector is reduced here.  No-op
    without an initialised process group."""
    dist = _dist()
    if dist is None:
        return tot
    import torch
    t = torch.from_numpy(np.asarray(tot, dtype=np.float64)).to(gradient.device)
    done = sorted((lo, hi) for lo, hi, _ in [p for p in pending if p is not None])
    works = [w for _, _, w in [p for p in pending if p is not None]]
    pos = 0
    for lo, hi in done + [(gradient.numel(), gradient.numel())]:
        if lo > pos:
            works.append(dist.all_reduce(gradient[pos:lo], async_op=True))
        pos = max(pos, hi)
    works.append(dist.all_reduce(t, async_op=True))
    for w in works:
        w.wait()
    return t.cpu().numpy()


def create_objective(model, weights, gradient, batch_iterator, stats):  # objective.lua:15
    cfg = model["cfg"]
    pnet = model["pnet"]
    cnet = model["cnet"]
    native = model["native"]
    bgclass = cfg["class_count"] + 1  # :20
    localizer = Localizer(pnet.outnode.children[4] if len(pnet.outnode.children) == 5
                          else pnet.outnode.children[-1])  # :22 children[5]
    kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
    cnet_input_planes = model["layers"][-1]["filters"]
    D = kh * kw * cnet_input_planes
    ncls = cfg["class_count"] + 1
    scratch = _Scratch()
    pinned = _PinnedRing()
    import torch
    acc_t = torch.zeros(8, dtype=torch.float64, device="cuda")       # the accumulators of objective.lua:52-58
    acc_dev = DeviceTensor(acc_t.data_ptr(), (8,), np.float64, owner=acc_t)
    acc_pin = torch.zeros(8, dtype=torch.float64).pin_memory()
    acc_event = torch.cuda.Event()
    L = _lib.load()
    # Data parallel, device tail: the four example counts of objective.lua:194-198 are host-side numbers known before
    # any kernel runs.  They are uploaded into the unused slots 2, 3, 6, 7 of the accumulator vector at the start of the
    # pass (instead of zeroing it), travel through the same 8-element fp64 all-reduce as the loss sums, and the optimiser
    # reads the divisor of gradient:div(cls_count) from the device (frcnn_scale_rmsprop_dev): no host read-back and no
    # host-side collective between the backward pass and the update, with RCCL through torch.distributed or through the
    # library's own communicator alike.  FRCNN_DP_SYNC_TAIL=1 keeps the synchronous exchange (read-back, then scale).
    dev_tail_ok = not os.environ.get("FRCNN_DP_SYNC_TAIL")

    # data parallel: the slices of the deep backbone blocks are final long before the backward pass ends (the deepest
    # block's weight gradients come first); their all-reduces start behind a per-block event on an auxiliary stream
    early_blocks = []
    aux_stream = [None]
    if os.environ.get("FRCNN_DP_EARLY_BLOCKS", "1") != "0":
        nblk = int(native.desc.nblocks)
        early_blocks = [b for b in range(nblk - 1, 0, -1)
                        if pnet.block_param_range(b)[1] - pnet.block_param_range(b)[0] >= (1 << 18)]

    def cleanAnchors(examples, outputs):  # objective.lua:32-43
        return [e for e in examples
                if not (e[0].index[1] > outputs[e[0].layer - 1].shape[1] or e[0].index[2] > outputs[e[0].layer - 1].shape[2])]

    prepared = {}   # id(batch entry) -> (the entry itself, tables packed ahead of the pass: finish(), while the host waits for
                    # the device); the entry is held so that its id cannot be re-used by another object while the tables wait

    def map_sizes(H, W):
        """Sizes of the anchor nets' outputs and of the last pooled map for an H x W image (ceil-mode pooling,
        model_utilities.lua:23; valid k x k head convolution, :31) -- what pnet.forward will hand out."""
        from .synthetic import output_map_sizes
        h, w = H, W
        for l in model["layers"]:
            for _ in range(l["conv_steps"]):
                h = h + 2 * l["padH"] - l["kH"] + 1; w = w + 2 * l["padW"] - l["kW"] + 1
            h = -(-(h - 2) // 2) + 1; w = -(-(w - 2) // 2) + 1
        return [tuple(t) for t in output_map_sizes(model, H, W)], (h, w)

    def prepare_examples(x):
        """The host half of objective.lua:74-140 for one image: cleanAnchors, the example tables (anchor / ground-truth
        rectangles, (layer, aspect, y, x), classes), the ROI-pooling windows and the positions the sparse head backward
        will touch, packed into the ONE blob that is uploaded behind the forward pass."""
        from .synthetic import clean_examples
        shp = x["img"].shape
        sizes, (fmH, fmW) = map_sizes(int(shp[1]), int(shp[2]))
        p = clean_examples(x["positive"], sizes)  # :74-75
        n = clean_examples(x["negative"], sizes)
        npos, E = len(p), len(p) + len(n)
        prep = dict(p=p, n=n, sizes=sizes, fm=(fmH, fmW))
        if E == 0:
            return prep
        anch = [e[0] for e in p] + [e[0] for e in n]
        ex_idx = np.array([(a.layer, a.aspect, a.index[1], a.index[2]) for a in anch], dtype=np.int32)
        ex_anchor = np.array([(a.minX, a.minY, a.maxX, a.maxY) for a in anch], dtype=np.float64)
        ex_roi = np.zeros((max(npos, 1), 4), dtype=np.float64)
        ex_class = np.zeros(max(npos, 1), dtype=np.int32)
        if npos:
            ex_roi[:npos] = [(e[1].rect.minX, e[1].rect.minY, e[1].rect.maxX, e[1].rect.maxY) for e in p]
            ex_class[:npos] = [e[1].class_index for e in p]
            if ex_class[:npos].min() < 1 or ex_class[:npos].max() > cfg["class_count"]:
                # nn.ClassNLLCriterion raises on a target outside 1..n (objective.lua:174); the batched loss
                # kernel indexes with it, so a stale index in a training-data file must stop here
                raise _lib.FrcnnError("roi.class_index %d outside 1..%d (class_count)"
                                      % (int(ex_class[:npos].min() if ex_class[:npos].min() < 1 else ex_class[:npos].max()),
                                         cfg["class_count"]))
        # positives pool the GT rect (:117), negatives pool the anchor rect itself (:137)
        wins = roi_windows(np.concatenate([ex_roi[:npos], ex_anchor[npos:]], 0), localizer, fmH, fmW)
        # positions where delta_outputs[l] will be non-zero (hint for the sparse head backward)
        sp = []
        for l in range(4):
            sel = ex_idx[ex_idx[:, 0] == l + 1]
            sp.append(np.unique((sel[:, 2] - 1) * sizes[l][1] + (sel[:, 3] - 1)).astype(np.int32))
        sp_all = np.concatenate(sp)
        blob = np.concatenate([ex_anchor.view(np.uint8).ravel(), ex_roi.view(np.uint8).ravel(),
                               ex_idx.view(np.uint8).ravel(), ex_class.view(np.uint8).ravel(),
                               wins.view(np.uint8).ravel(), sp_all.view(np.uint8).ravel()])
        prep.update(ex_anchor=ex_anchor, ex_roi=ex_roi, ex_idx=ex_idx, ex_class=ex_class, wins=wins, sp=sp, blob=blob)
        return prep

    def run(w, defer, eager=None):
        """Queues the whole pass.  Returns finish() -> (loss, gradient); with defer=True (single process) the
        accumulators travel to pinned host memory asynchronously and finish() only waits for that copy, so the
        caller may queue more work (the optimiser step) before it looks at the loss."""
        if w is not weights:  # :46-48
            weights.copy_(w)
        if packs_promise[0] is not None and packs_promise[0] != getattr(weights, "_version", None):
            # the packs renewed beside the previous pass were made from weights somebody has written since (torch counts
            # in-place writes; the library's own updates do not go through torch): the forward pass re-packs
            _lib.call("frcnn_pnet_invalidate_packs", native.h)
        packs_promise[0] = None
        s = stream_ptr()
        probing = exchange_probe is not None and _dist() is None and getattr(gradient, "is_cuda", False)
        if probing:
            _probe_mark("step_begin", 0, 0, "")
        _lib.call("frcnn_zero", ptr(gradient), gradient.numel() * 4, s)  # :49
        cls_count = reg_count = creg_count = ccls_count = 0
        pnet.training()  # :61-62
        cnet.training()
        batch = next_batch[0] if next_batch[0] is not None else batch_iterator.nextTraining()  # :64
        next_batch[0] = None
        pending = []
        early_copy = False
        # The update beside the backward pass (include/frcnn_hip.h): single process, one image per step, the side streams on.
        # eager = {m, lr, alpha, eps} from utilities.rmsprop; eager["done"] collects the slices updated on the update stream.
        if eager is not None:
            side = C.c_int(0)
            _lib.call("frcnn_get_option", b"side_stream", C.byref(side))
            if not (defer == "fold" and _dist() is None and len(batch) == 1 and side.value and w is weights
                    and getattr(gradient, "is_cuda", False)):
                eager = None
        if eager is not None:
            from .synthetic import clean_examples, output_map_sizes
            x0 = batch[0]
            held0 = prepared.get(id(x0))
            if held0 is not None and held0[0] is x0:
                n_ex = len(held0[1]["p"]) + len(held0[1]["n"])
            else:
                sizes0 = output_map_sizes(model, x0["img"].shape[1], x0["img"].shape[2])
                n_ex = len(clean_examples(x0["positive"], sizes0)) + len(clean_examples(x0["negative"], sizes0))
            eager["gscale"] = (1.0 / n_ex) if n_ex > 0 else 1.0   # gradient:div(cls_count), :200 (nothing to scale without examples)
            eager["done"] = []
            us = C.c_void_p()
            _lib.call("frcnn_model_update_stream", native.h, C.byref(us))
            nblk = int(native.desc.nblocks)

            def eager_slice(lo, hi, on, group=None):
                _lib.call("frcnn_scale_rmsprop_slice", ptr(weights), ptr(gradient), eager["gscale"], ptr(eager["m"]), lo, hi,
                          eager["lr"], eager["alpha"], eager["eps"], on)
                eager["done"].append((int(lo), int(hi)))
                if group is not None:
                    _lib.call("frcnn_pnet_refresh_packs", native.h, ptr(weights), group, on)
                    eager["groups"].add(group)
            eager["groups"] = set()
            eager["slice"] = eager_slice
        dev_tail = (dev_tail_ok and defer == "fold" and _dist() is not None and getattr(gradient, "is_cuda", False))
        c4 = None
        if dev_tail:
            from .synthetic import clean_examples, output_map_sizes
            c4 = [0.0, 0.0, 0.0, 0.0]   # cls, reg, creg, ccls exactly as accumulated below (:194-198)
            for x in batch:
                shp = x["img"].shape
                sizes = output_map_sizes(model, shp[1], shp[2])
                np_, nn_ = len(clean_examples(x["positive"], sizes)), len(clean_examples(x["negative"], sizes))
                c4[0] += np_ + nn_; c4[1] += np_; c4[2] += np_; c4[3] += 1
            init = np.array([0.0, 0.0, c4[0], c4[1], 0.0, 0.0, c4[2], c4[3]], dtype=np.float64)
            hinit, staged = pinned.stage(init.view(np.uint8))
            _lib.call("frcnn_memcpy_h2d", ptr(acc_dev), C.c_void_p(hinit), 64, s)
            staged()
        else:
            acc_dev.zero_()
        for bi, x in enumerate(batch):
            last = bi == len(batch) - 1   # (by position: an iterator may hand out the same pooled image twice)
            # the example tables are host work that needs no device result (the map sizes follow from the image size): they
            # were packed while the host waited for the previous step (finish()), or are packed now -- before the forward
            # pass is queued, not between it and the stage that consumes them, where the device would run dry
            held = prepared.pop(id(x), None)
            prep = held[1] if held is not None and held[0] is x else prepare_examples(x)
            img = to_device(x["img"])  # :66
            outputs = pnet.forward(img, async_heads=True)  # :71 (the anchor nets stay in flight beside the cnet stage)
            if prep["sizes"] != [tuple(o.shape[1:]) for o in outputs[:len(prep["sizes"])]] or prep["fm"] != tuple(outputs[-1].shape[1:]):
                raise _lib.FrcnnError("lossAndGradient: the example tables were packed for other map sizes than this image's")
            p, n = prep["p"], prep["n"]  # :74-75
            delta_outputs = pnet.delta_outputs(zero=True)  # :78-84
            npos, nneg = len(p), len(n)
            E = npos + nneg
            fm = outputs[-1]
            fmC, fmH, fmW = fm.shape
            if E > 0:
                ex_anchor, ex_roi, ex_idx, ex_class, wins, sp, blob = (prep[k] for k in ("ex_anchor", "ex_roi", "ex_idx", "ex_class", "wins", "sp", "blob"))
                dblob = scratch.get("blob", (blob.size,), np.uint8)
                hblob, staged = pinned.stage(blob)   # page-locked: the upload is asynchronous, the host keeps its lead
                _lib.call("frcnn_memcpy_h2d", ptr(dblob), C.c_void_p(hblob), blob.nbytes, s)
                staged()
                o = 0
                d_anchor = dblob.ptr + o; o += ex_anchor.nbytes
                d_roi = dblob.ptr + o; o += ex_roi.nbytes
                d_idx = dblob.ptr + o; o += ex_idx.nbytes
                d_class = dblob.ptr + o; o += ex_class.nbytes
                d_wins = dblob.ptr + o; o += wins.nbytes
                for l in range(4):
                    _lib.call("frcnn_pnet_set_sparse_deltas", native.h, l + 1, C.c_void_p(dblob.ptr + o), len(sp[l]))
                    o += sp[l].nbytes
                # ---- RPN loss on the sampled anchors (objective.lua:91-140) ------------------
                # queued behind the anchor nets on the library's side stream, followed by their backward pass
                # (delta_outputs[1..4] are final after the loop: the fine-tuning stage only touches
                # delta_outputs[5], :182-185); this stream goes on with the cnet stage on the last map
                ex_loss = scratch.get("ex_loss", (E, 2), np.float64)
                crtarget = scratch.get("crtarget", (E, 4))
                cctarget = scratch.get("cctarget", (E,))
                pnet.anchor_loss_begin(d_idx, d_anchor, d_roi, d_class, npos, nneg, bgclass, ex_loss, crtarget,
                                       cctarget, acc_dev)
                # ---- ROI pooling of every example in one launch (:117-119, :137-139) ---------
                cinput = scratch.get("cinput", (E, D))
                pidx = scratch.get("pidx", (E, D), np.int32)
                _lib.call("frcnn_roi_pool_forward", ptr(fm), fmC, fmH, fmW, C.c_void_p(d_wins), E, kh, kw,
                          ptr(cinput), ptr(pidx), s)
                # ---- fine-tuning stage (:146-186) --------------------------------------------
                coutputs = cnet.forward(cinput)  # :164
                crout, ccout = coutputs
                crdelta = scratch.get("crdelta", (E, 4))
                ccdelta = scratch.get("ccdelta", (E, ncls))
                pnet.anchor_loss_wait()   # crtarget is relative to the anchor nets' proposals (:156)
                _lib.call("frcnn_cnet_losses", ptr(crout), ptr(crtarget), ptr(ccout), ptr(cctarget), E, npos, ncls,
                          ptr(crdelta), ptr(ccdelta), C.c_void_p(acc_dev.ptr + 4 * 8), s)  # :170-177
                post_roi_delta = cnet.backward(cinput, [crdelta, ccdelta])  # :179
                _lib.call("frcnn_roi_pool_backward", ptr(delta_outputs[4]), fmC, fmH, fmW, ptr(post_roi_delta),
                          ptr(pidx), E, kh, kw, s)  # :182-185
            if E == 0:
                for l in range(4):
                    _lib.call("frcnn_pnet_set_sparse_deltas", native.h, l + 1, None, 0)
            if last and _dist() is not None:
                # the classification net's slice of the flat gradient (55 % of it) is final, and so is the anchor
                # nets' slice once the side stream's part is joined: their all-reduces run beside the backbone's
                # backward pass; only the backbone's 3.3 M elements remain for the end
                _lib.call("frcnn_cnet_backward_join", native.h, stream_ptr())   # its weight gradients ride on a stream of their own
                pending.append(allreduce_begin(gradient, native.pnet_params, gradient.numel()))
                # The decision must not depend on this rank's data (every rank issues the same collectives): with the
                # side stream on and one image per batch the slice is final here on every rank -- either the
                # anchor nets' backward was started early and is joined now, or the image had no example and
                # contributes nothing to that slice.
                side = C.c_int(0)
                _lib.call("frcnn_get_option", b"side_stream", C.byref(side))
                if len(batch) == 1 and side.value:
                    pnet.backward_heads_join()
                    lo, hi = pnet.heads_param_range()
                    pending.append(allreduce_begin(gradient, lo, hi))
            if last and probing:   # (the same program points as the branch above, nothing exchanged)
                _lib.call("frcnn_cnet_backward_join", native.h, stream_ptr())
                _probe_mark("classification net", native.pnet_params, gradient.numel(), "frcnn_cnet_backward_join (its weight-gradient stream)")
                side = C.c_int(0)
                _lib.call("frcnn_get_option", b"side_stream", C.byref(side))
                if len(batch) == 1 and side.value:
                    pnet.backward_heads_join()
                    lo, hi = pnet.heads_param_range()
                    _probe_mark("anchor nets", lo, hi, "frcnn_pnet_backward_heads_join (the anchor nets' streams)")
            if last and defer and _dist() is None:
                # the eight statistics are final before the backbone's backward pass: their read-back is queued
                # here, so the caller's wait ends mid-step and the host queues the next step while this one
                # is still running (the device never drains between steps)
                acc_pin.copy_(acc_t, non_blocking=True)
                acc_event.record()
                early_copy = True
            debug["E"] = E
            pnet.backward(img, delta_outputs)  # :189
            if eager is not None:
                # On the update stream, once the caller's stream is inside the backbone's backward pass (matrix-core bound: the
                # bandwidth-bound update costs least beside it): the classification net's slice (55 % of the vector; its weight
                # gradients ran on this very stream), the anchor nets' slice and packs, then each backbone block as the pass leaves it
                # (that event follows the whole classification-net stage in stream order: its slice is final, its weights have
                # had their last reader -- also for an image without examples; the anchor nets may still be adding up their
                # parameter gradients)
                _lib.call("frcnn_pnet_wait_backward_begun", native.h, us)
                eager_slice(native.pnet_params, gradient.numel(), us)
                _lib.call("frcnn_pnet_wait_heads_done", native.h, us)
                lo, hi = pnet.heads_param_range()
                eager_slice(lo, hi, us, nblk)
                for b in range(nblk, 1, -1):   # deepest block first: the order in which the pass leaves them; block 1 ends the pass
                    _lib.call("frcnn_pnet_wait_block_done", native.h, b, us)
                    lo, hi = pnet.block_param_range(b - 1)
                    eager_slice(lo, hi, us, b - 1)
                _lib.call("frcnn_model_update_join", native.h, stream_ptr())
            if last and _dist() is not None and early_blocks and getattr(gradient, "is_cuda", False):
                if aux_stream[0] is None:
                    aux_stream[0] = torch.cuda.Stream()
                for b in early_blocks:   # deepest first: the order in which their gradients become final
                    lo, hi = pnet.block_param_range(b)
                    with torch.cuda.stream(aux_stream[0]):
                        pnet.wait_block_gradients(b)
                        pending.append(allreduce_begin(gradient, lo, hi))
            if last and probing:
                covered = []
                if early_blocks:
                    if aux_stream[0] is None:
                        aux_stream[0] = torch.cuda.Stream()
                    for b in early_blocks:
                        lo, hi = pnet.block_param_range(b)
                        with torch.cuda.stream(aux_stream[0]):
                            pnet.wait_block_gradients(b)
                            _probe_mark("backbone block %d" % (b + 1), lo, hi, "frcnn_pnet_wait_block_gradients(%d) on an auxiliary stream" % (b + 1))
                        covered.append((lo, hi))
                _probe_mark("backward_end", 0, 0, "")
                lo_rest = 0
                hi_rest = min([c[0] for c in covered] + [pnet.heads_param_range()[0]])
                _probe_mark("rest of the backbone", lo_rest, hi_rest, "end of frcnn_pnet_backward on the caller's stream")
            reg_count += npos  # :194-198
            cls_count += npos + nneg
            creg_count += npos
            ccls_count += 1

        # ---- statistics: one read-back per call ------------------------------------------
        single = _dist() is None
        fold = single and defer == "fold" and cls_count > 0   # the optimiser folds gradient:div into its own pass
        if single and cls_count > 0 and not fold:  # the divisor is host-known: queue the scaling before the read-back
            _lib.call("frcnn_scale", ptr(gradient), gradient.numel(), 1.0 / cls_count, stream_ptr())  # :200
        counts = (cls_count, reg_count, creg_count, ccls_count)
        if not single:   # the rest of the gradient is final too: its exchange starts before the host read-back below
            pending[:] = allreduce_begin_rest(gradient, pending)
        if dev_tail:
            # asynchronous data-parallel tail: loss sums AND counts reduced on the device in one 8-element all-reduce
            assert [float(v) for v in (cls_count, reg_count, creg_count, ccls_count)] == c4, "count bookkeeping diverged"
            acc_work = _dist().all_reduce(acc_t, async_op=True)
            for pnd in pending:
                pnd[2].wait()           # RCCL: the current stream waits, the host does not
            acc_work.wait()
            acc_pin.copy_(acc_t, non_blocking=True)
            acc_event.record()
            fin = lambda: finish(None, None, (), True, reduced=True)
            return (fin, DeviceDivisor(acc_dev.ptr + 2 * 8, acc_t))   # slot 2 = the all-reduced cls_count (:200)
        if single and defer:
            if not early_copy:
                acc_pin.copy_(acc_t, non_blocking=True)
                acc_event.record()
            fin = lambda: finish(None, counts, pending, single)
            if eager is not None:
                assert cls_count == 0 or eager["gscale"] == 1.0 / cls_count, "example count bookkeeping diverged"
                return (fin, eager["gscale"] if cls_count > 0 else None)
            return (fin, 1.0 / cls_count) if fold else fin
        if defer == "fold" and not single:   # the all-reduced count is known after finish(): scaling left to the caller
            res = finish(acc_dev.numpy(), counts, pending, single, fold=True)
            gs = dp_gscale[0]
            return ((lambda: res), gs) if gs is not None else (lambda: res)
        return (lambda r: (lambda: r))(finish(acc_dev.numpy(), counts, pending, single))

    dp_gscale = [None]
    packs_promise = [None]   # weights._version at the moment every pack group had been renewed beside a pass (else None)
    debug = dict(scratch=scratch, E=0)
    next_batch = [None]
    prefetch_batches = os.environ.get("FRCNN_PREFETCH_BATCH", "1") != "0"

    def finish(a, counts, pending, single, fold=False, reduced=False):
        if a is None:
            # deferred read-back: the whole step is queued, the device is busy -- the next batch (image decoding hand-over,
            # processImage launches, example assembly: host work of the loader) is drawn now instead of at the start of
            # the next call, where the device would wait for it.  Same call sequence on the iterator, one call early.
            if next_batch[0] is None and prefetch_batches:
                next_batch[0] = batch_iterator.nextTraining()
                prepared.clear()
                for xb in next_batch[0]:   # ... and its example tables are packed (host work with no device input)
                    prepared[id(xb)] = (xb, prepare_examples(xb))
            acc_event.synchronize()
            a = acc_pin.numpy().copy()
        if counts is None:   # device tail: the (all-reduced) counts sit in the slots the kernels leave alone
            counts = (a[2], a[3], a[6], a[7])
        cls_count, reg_count, creg_count, ccls_count = counts
        tot = np.array([a[0], a[1], cls_count, reg_count, a[4], creg_count, a[5], ccls_count], dtype=np.float64)
        if not reduced:   # (asynchronous data-parallel tail: gradient, accumulators and counts are already summed)
            tot = allreduce_gradient_and_stats(gradient, tot, pending)  # DP: no-op for a single process
        cls_loss, reg_loss, cls_count, reg_count, creg_loss, creg_count, ccls_loss, ccls_count = tot
        dp_gscale[0] = None
        if not single and cls_count > 0:
            if fold:
                dp_gscale[0] = 1.0 / cls_count
            else:
                _lib.call("frcnn_scale", ptr(gradient), gradient.numel(), 1.0 / cls_count, stream_ptr())  # :200
        with np.errstate(divide="ignore", invalid="ignore"):
            pcls = float(np.float64(cls_loss) / cls_count)  # :202-205
            preg = float(np.float64(reg_loss) / reg_count)
            dcls = float(np.float64(ccls_loss) / ccls_count)
            dreg = float(np.float64(creg_loss) / creg_count)
        if stats.get("verbose"):
            print("prop: cls: %f (%d), reg: %f (%d); det: cls: %f, reg: %f" % (pcls, cls_count, preg, reg_count, dcls, dreg))
        stats["pcls"].append(pcls); stats["preg"].append(preg)  # :211-214
        stats["dcls"].append(dcls); stats["dreg"].append(dreg)
        return pcls + preg, gradient  # :216-217

    def lossAndGradient(w):
        return run(w, False)()

    # private protocol with utilities.rmsprop: begin(w) queues the pass and returns (finish, gradient); the
    # optimiser queues its update and only then calls finish() for the loss (no device idle during the read-back)
    lossAndGradient.begin = lambda w: (run(w, True), gradient)
    # begin_fold(w) -> (finish, gradient, gscale): gradient:div(cls_count) (:200) is left to the caller, who folds it
    # into the first pass of its update (frcnn_scale_rmsprop); gscale is None when there is nothing to scale
    def begin_fold(w, eager=None):
        """eager = dict(m=, lr=, alpha=, eps=) (utilities.rmsprop): the pass may apply the optimiser's step to slices of the
        vectors as they become final (the update beside the backward pass, include/frcnn_hip.h); eager["done"] then lists the
        slices it has updated, eager["slice"](lo, hi, stream, group) updates another one, eager["groups"] the pack groups
        already renewed, and the caller finishes with eager["complete"]()."""
        r = run(w, "fold", eager)
        if eager is not None and "done" in eager:
            ngroups = int(native.desc.nblocks) + 1

            def complete():
                """The caller has updated the rest of the vector: renew the packs nobody has renewed yet (on the current
                stream) and remember which weights every pack now belongs to."""
                for g in range(ngroups):
                    if g not in eager["groups"]:
                        _lib.call("frcnn_pnet_refresh_packs", native.h, ptr(weights), g, stream_ptr())
                        eager["groups"].add(g)
                packs_promise[0] = getattr(weights, "_version", None)
            eager["complete"] = complete
        return (r[0], gradient, r[1]) if isinstance(r, tuple) else (r, gradient, None)
    lossAndGradient.begin_fold = begin_fold
    lossAndGradient.debug = debug   # (tests: the scratch buffers and the example count of the image being processed)
    return lossAndGradient


def _accumulate_losses(ex_loss, E, acc_dev, s):
    """acc[0] += sum cls, acc[1] += sum reg (objective.lua:104,112,132), on device, fixed order."""
    _lib.call("frcnn_loss_accumulate", ptr(ex_loss), E, ptr(acc_dev), s)
