"""The path's DISCRETE decisions -- max-pool window winners, adaptive max-pool cell winners, PReLU branches -- read from
the device after a forward pass (frcnn_model_debug_buffer) in the layout the oracle's decision injection takes
(oracle/frcnn_oracle.h orc_set_decisions).  With them injected, the oracle's gradient follows exactly the routes the
device took, and the comparison holds SURVEY 8d's strict bars (1e-3 per tensor, 1e-4 elementwise) on every tensor with
no exclusions; the number of decisions the oracle would have taken differently is counted and bounded separately."""
import ctypes as C
import math

import numpy as np


def _dev_array(F, nat, kind, index, dtype, shape):
    p = C.c_void_p(); n = C.c_longlong()
    F._lib.call("frcnn_model_debug_buffer", nat.h, kind, index, C.byref(p), C.byref(n))
    t = F.DeviceTensor(p.value, shape, dtype)
    assert t.nbytes == n.value, (kind, index, shape, n.value)
    return t.numpy()


def _pool_out(n):
    return int(math.ceil((n - 2) / 2.0)) + 1


def capture(F, model, H, W, R=0, roi_idx=None):
    """-> dict for O.decisions(inject=...) of the LAST forward pass of `model` on an H x W image (and of the last cnet
    forward with R rows, if R > 0)."""
    nat = model["native"]
    layers = model["layers"]
    d = dict(pool_idx=[], conv_pos=[], head_pos=[], cnet_pos=[], roi_idx=None, conv_ignore=[])
    h, w = H, W
    ci = 0
    block_hw = []
    for b, l in enumerate(layers):
        for _ in range(l["conv_steps"]):
            h = h + 2 * l["padH"] - l["kH"] + 1; w = w + 2 * l["padW"] - l["kW"] + 1
            x = _dev_array(F, nat, 0, ci, np.float32, (l["filters"], h, w))
            d["conv_pos"].append(np.ascontiguousarray((x > 0).astype(np.uint8)))
            # Channels a SpatialDropout behind this convolution dropped in the pass (a training pass does not even compute them,
            # option "drop_compact": they read 0): their PReLU branch is multiplied by the zero scale forward and backward, so it is
            # no decision of the path -- count_differences leaves them out
            ign = None
            if _ == 0 and l.get("dropout", 0) > 0 and l["conv_steps"] >= 2 and model["pnet"].train:
                keep = _dev_array(F, nat, 4, b, np.float32, (l["filters"],))
                ign = keep == 0
            d["conv_ignore"].append(ign)
            ci += 1
        hp, wp = _pool_out(h), _pool_out(w)
        code = _dev_array(F, nat, 1, b, np.uint8, (l["filters"], hp, wp)).astype(np.int32)
        oy = np.arange(hp, dtype=np.int32)[None, :, None]; ox = np.arange(wp, dtype=np.int32)[None, None, :]
        d["pool_idx"].append(np.ascontiguousarray((2 * oy + (code >> 1)) * w + (2 * ox + (code & 1))).astype(np.int32))
        h, w = hp, wp
        block_hw.append((h, w))
    for i, a in enumerate(model["anchor_nets"]):
        bh, bw = block_hw[a["input"] - 1]
        x = _dev_array(F, nat, 2, i, np.float32, (a["n"], bh - a["kW"] + 1, bw - a["kW"] + 1))
        d["head_pos"].append(np.ascontiguousarray((x > 0).astype(np.uint8)))
    if R > 0:
        for i, l in enumerate(model["class_layers"]):
            x = _dev_array(F, nat, 3, i, np.float32, (R, l["n"]))
            d["cnet_pos"].append(np.ascontiguousarray((x > 0).astype(np.uint8)))
        if roi_idx is not None:
            d["roi_idx"] = np.ascontiguousarray(roi_idx, dtype=np.int32)
    return d


def blank_like(d):
    """Arrays of the same shapes for O.decisions(record=...)."""
    out = dict(slope_abs=np.zeros(48))   # (record only: the size of the terms each PReLU slope gradient sums)
    for k, v in d.items():
        if k == "conv_ignore":
            continue
        if isinstance(v, list):
            out[k] = [np.zeros_like(a) for a in v]
        else:
            out[k] = None if v is None else np.zeros_like(v)
    return out


def count_differences(a, b):
    """-> dict(kind -> (differing, total)) between two decision sets."""
    res = {}
    ignore = a.get("conv_ignore") or b.get("conv_ignore")
    for k in a:
        if k in ("slope_abs", "conv_ignore") or k not in b:
            continue
        va, vb = a[k], b[k]
        if va is None or vb is None:
            continue
        if not isinstance(va, list):
            va, vb = [va], [vb]
        if k == "conv_pos" and ignore:   # (channels dropped by a SpatialDropout: see capture)
            diff = tot = 0
            for x, y, ign in zip(va, vb, ignore):
                ne = x != y
                if ign is not None:
                    ne = ne[~ign]
                    tot += int((~ign).sum()) * int(x[0].size)
                else:
                    tot += x.size
                diff += int(ne.sum())
            res[k] = (diff, tot)
            continue
        res[k] = (int(sum((x != y).sum() for x, y in zip(va, vb))), int(sum(x.size for x in va)))
    return res


class CaptureBeforeBackward(object):
    """Wraps model['pnet'].backward: right before the backward pass of every image (all forward state of that image is in
    HBM, the cnet has run) the decisions are copied to the host.  self.captured[i] = decisions of image i."""

    def __init__(self, F, model, objective=None):
        self.F, self.model, self.objective = F, model, objective
        self.captured = []

    def __enter__(self):
        pnet = self.model["pnet"]
        self._orig = pnet.backward
        F, model = self.F, self.model

        def backward(img, deltas):
            import torch
            torch.cuda.synchronize()
            _, H, W = img.shape
            R = int(self.objective.debug["E"]) if self.objective is not None else 0
            roi = None
            if R > 0:
                D = model["cfg"]["roi_pooling"]["kh"] * model["cfg"]["roi_pooling"]["kw"] * model["layers"][-1]["filters"]
                sc = self.objective.debug["scratch"]
                roi = sc.get("pidx", (R, D), np.int32).numpy()
            self.captured.append(capture(F, model, H, W, R=R, roi_idx=roi))
            return self._orig(img, deltas)
        pnet.backward = backward
        return self

    def __exit__(self, *a):
        self.model["pnet"].backward = self._orig
        return False


def slope_terms(native, model, slope_abs, divisor=1.0):
    """{flat offset of a PReLU slope: sum |x * gy| / divisor} from a recorded slope_abs (48 doubles: backbone convolutions,
    anchor nets, classification layers -- the order in which the slopes appear in the flat parameter vector)."""
    nconv = sum(l["conv_steps"] for l in model["layers"])
    order = list(range(nconv)) + [32 + h for h in range(len(model["anchor_nets"]))] + [40 + l for l in range(len(model["class_layers"]))]
    offs = [off for off, cnt, kind, aux in native.param_table if kind == 2]
    assert len(offs) == len(order), (len(offs), len(order))
    return {off: float(slope_abs[i]) / divisor for off, i in zip(offs, order)}
