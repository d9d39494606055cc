"""Generates tests/golden/train_tiny.npz and detect_tiny.npz: END-TO-END golden vectors of the hot path from a SECOND,
independent implementation -- a PyTorch-CPU float64 *autograd* restatement of

    models/model_utilities.lua:3-124   (pnet with its nngraph fan-out, cnet)
    objective.lua:5-13, 45-218         (ROI window, one lossAndGradient over a batch of images)
    Detector.lua:17-141                (detect)

written from the Lua files, NOT from oracle/orc_*.c: no backward pass is written here at all (autograd derives every
gradient from the forward composition), the sparse anchor losses are plain indexing into the output maps, the ROI
pooling is torch's adaptive_max_pool2d on a slice.  Geometry that the repo's naive restatement already pins against
SURVEY Appendix B (anchor tables, positives, NMS: oracle/naive_np.py) is imported from there; Localizer / ROI window /
anchor transforms are restated again below.

Run in the authoring container (`python tests/golden/make_golden_e2e.py`); the outputs are DATA (inputs + expected
numbers).  tests/test_oracle_pinning.py replays them through the C oracle, tests/test_gpu_golden_e2e.py through the HIP
path.  This is still not the reference's own vector (Torch7 cannot run here): the assumed Torch7 layer semantics are
those of oracle/ASSUMPTIONS.md rows 1-12 -- what these fixtures remove is "one author wrote both sides of every
end-to-end comparison IN THE SAME WAY": composition, fan-out sums, loss bookkeeping and normalisation come out of
autograd here.

Flat parameter order (the repo's convention, include/frcnn_hip.h frcnn_model_param_table / ASSUMPTIONS.md row 18): backbone
convolutions in order (W, b, PReLU slope), anchor nets in order (W kxk, b, slope, W 1x1, b), classification layers
(W, b, [BN weight, BN bias], slope), bbox head (W, b), class head (W, b)."""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import naive_np  # noqa: E402  (second restatement of Anchors / nms, pinned against SURVEY Appendix B)
from util import TINY_CLS, TINY_HEADS, TINY_LAYERS  # noqa: E402  (topology tables: data)

CFG = dict(class_count=5, scales=[32, 64, 128, 256], roi_pooling=dict(kw=2, kh=2))
T = torch.float64


# ------------------------------------------------------------------------------------------ parameters
def param_shapes(layers, heads, cls, cfg):
    """[(name, shape)] in flat order."""
    out = []
    cin = 3
    for b, l in enumerate(layers):
        for s in range(l["conv_steps"]):
            out += [("c%d_%d.w" % (b, s), (l["filters"], cin, l["kH"], l["kW"])), ("c%d_%d.b" % (b, s), (l["filters"],)),
                    ("c%d_%d.a" % (b, s), (1,))]
            cin = l["filters"]
    for h, a in enumerate(heads):
        c = layers[a["input"] - 1]["filters"]
        out += [("h%d.w" % h, (a["n"], c, a["kW"], a["kW"])), ("h%d.b" % h, (a["n"],)), ("h%d.a" % h, (1,)),
                ("h%d.w1" % h, (18, a["n"], 1, 1)), ("h%d.b1" % h, (18,))]
    prev = cfg["roi_pooling"]["kh"] * cfg["roi_pooling"]["kw"] * layers[-1]["filters"]   # model_utilities.lua:127
    for i, l in enumerate(cls):
        out += [("l%d.w" % i, (l["n"], prev)), ("l%d.b" % i, (l["n"],))]
        if l.get("batch_norm"):
            out += [("l%d.bnw" % i, (l["n"],)), ("l%d.bnb" % i, (l["n"],))]
        out += [("l%d.a" % i, (1,))]
        prev = l["n"]
    out += [("bbox.w", (4, prev)), ("bbox.b", (4,)), ("cls.w", (cfg["class_count"] + 1, prev)), ("cls.b", (cfg["class_count"] + 1,))]
    return out


def unflatten(flat, shapes):
    P, o = {}, 0
    for name, shp in shapes:
        n = int(np.prod(shp))
        P[name] = flat[o:o + n].reshape(shp)
        o += n
    assert o == flat.numel()
    return P


# ------------------------------------------------------------------------------------------ networks
def pnet_forward(P, img, layers, heads, training, masks):
    """create_proposal_net, model_utilities.lua:3-58.  masks[b]: per-channel keep mask of block b's SpatialDropout."""
    x = img[None]
    conv_outputs = []
    for b, l in enumerate(layers):
        for s in range(l["conv_steps"]):
            x = Fn.conv2d(x, P["c%d_%d.w" % (b, s)], P["c%d_%d.b" % (b, s)], padding=(l["padH"], l["padW"]))   # :8
            x = Fn.prelu(x, P["c%d_%d.a" % (b, s)])                                                              # :9
            if s == 0 and l.get("dropout", 0) > 0:                                                                  # :10-12, :20
                x = x * masks[b][None, :, None, None] if training else x * (1.0 - l["dropout"])                     # [ext] 2015 SpatialDropout
        x = Fn.max_pool2d(x, 2, 2, ceil_mode=True)                                                               # :23
        conv_outputs.append(x)
    outs = []
    for h, a in enumerate(heads):                                                                                   # :29-35, :52-55
        y = Fn.conv2d(conv_outputs[a["input"] - 1], P["h%d.w" % h], P["h%d.b" % h])
        y = Fn.conv2d(Fn.prelu(y, P["h%d.a" % h]), P["h%d.w1" % h], P["h%d.b1" % h])
        outs.append(y[0])
    outs.append(conv_outputs[-1][0])                                                                                # :56
    return outs


def cnet_forward(P, x, cls, training, masks, bn_running):
    """create_classification_net, model_utilities.lua:76-124.  masks[i]: R x n keep masks of layer i's nn.Dropout."""
    k = 0
    for i, l in enumerate(cls):
        x = Fn.linear(x, P["l%d.w" % i], P["l%d.b" % i])                                   # :82
        if l.get("batch_norm"):                                                          # :83-85
            n = l["n"]
            if training:
                x = Fn.batch_norm(x, None, None, P["l%d.bnw" % i], P["l%d.bnb" % i], training=True, eps=1e-5)
            else:
                x = Fn.batch_norm(x, bn_running[k:k + n].clone(), bn_running[k + n:k + 2 * n].clone(), P["l%d.bnw" % i], P["l%d.bnb" % i],
                                  training=False, eps=1e-5)
            k += 2 * n
        x = Fn.prelu(x, P["l%d.a" % i])                                                   # :86
        if l.get("dropout", 0) > 0 and training:                                         # :87-89, [ext] nn.Dropout v2: mask / (1 - p)
            x = x * masks[i] / (1.0 - l["dropout"])
    return Fn.linear(x, P["bbox.w"], P["bbox.b"]), Fn.log_softmax(Fn.linear(x, P["cls.w"], P["cls.b"]), dim=1)   # :99-105


# ------------------------------------------------------------------------------------------ geometry (restated from the Lua)
def localizer_layers(layers, upto_block):
    """Localizer.lua:6-39 for the chain input -> conv_outputs[upto_block]: rows kW,kH,dW,dH,padW,padH."""
    out = []
    for b in range(upto_block):
        l = layers[b]
        out += [[l["kW"], l["kH"], 1, 1, l["padW"], l["padH"]]] * l["conv_steps"]
        out += [[2, 2, 2, 2, 0, 0]]
    return out


def input_to_feature_rect(lay, r):
    """Localizer.lua:41-67, kept with its dW/dH mix-ups (all strides are square here)."""
    minX, minY, maxX, maxY = [float(v) for v in r]
    lmod = lambda a, b: a - math.floor(a / b) * b
    for kW, kH, dW, dH, padW, padH in lay:
        if dW < kW:
            minX -= kW - dW; minY -= kH - dH; maxX += kW - dW; maxY += kH - dH       # :45 inflate
        minX += padW; maxX += padW; minY += padH; maxY += padH                       # :48 offset
        minX = minX / dH; minY = minY / dH                                           # :51-52
        maxX = max((maxX - kW) / dW + 1, minX + 1) if lmod(maxX - kW, dW) == 0 else max(math.ceil((maxX - kW) / dW) + 1, minX + 1)
        maxY = max((maxY - kH) / dW + 1, minY + 1) if lmod(maxY - kH, dH) == 0 else max(math.ceil((maxY - kH) / dH) + 1, minY + 1)
    return [math.floor(minX), math.floor(minY), math.ceil(maxX), math.ceil(maxY)]     # :66 snapToInt


def roi_slice(lay, rect, fm):
    """extract_roi_pooling_input, objective.lua:5-13: the sub-window of fm (C x H x W) as a view."""
    r = input_to_feature_rect(lay, rect)
    H, W = fm.shape[1], fm.shape[2]
    minX = min(max(r[0], 0), W); minY = min(max(r[1], 0), H); maxX = max(min(r[2], W), 0); maxY = max(min(r[3], H), 0)   # Rect:clip
    y0, y1 = min(minY + 1, maxY), maxY       # 1-based inclusive
    x0, x1 = min(minX + 1, maxX), maxX
    return fm[:, y0 - 1:y1, x0 - 1:x1], (y0, y1, x0, x1)


def input_to_anchor(a, r):     # Anchors.lua:237-243 (a, r: minX, minY, maxX, maxY)
    aw, ah = a[2] - a[0], a[3] - a[1]
    return [(r[0] - a[0]) / aw, (r[1] - a[1]) / ah, math.log((r[2] - r[0]) / aw), math.log((r[3] - r[1]) / ah)]


def anchor_to_input(a, t):     # Anchors.lua:245-252
    aw, ah = a[2] - a[0], a[3] - a[1]
    x, y = t[0] * aw + a[0], t[1] * ah + a[1]
    return [x, y, x + math.exp(t[2]) * aw, y + math.exp(t[3]) * ah]


def f32(v):
    """Values that pass through a torch.FloatTensor / CudaTensor in the reference are fp32 numbers."""
    return float(np.float32(v))


# ------------------------------------------------------------------------------------------ objective.lua:45-218
def loss_and_gradient(flat_w, shapes, layers, heads, cls, cfg, batch):
    """batch: list of dict(img, pos=[(layer, aspect, y, x, roi_index)], pos_rect, neg=[(layer, aspect, y, x)], neg_rect, rois, roi_class,
    pmasks, cmasks).  Returns the four statistics, the flat gradient (already divided by cls_count) and intermediates of image 0."""
    w = torch.tensor(flat_w, dtype=T, requires_grad=True)
    P = unflatten(w, shapes)
    kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
    lay5 = localizer_layers(layers, len(layers))
    bg = cfg["class_count"] + 1
    cls_loss = reg_loss = creg_loss = ccls_loss = 0.0
    cls_count = reg_count = creg_count = ccls_count = 0
    keep = {}
    for bi, x in enumerate(batch):
        outs = pnet_forward(P, torch.tensor(x["img"], dtype=T), layers, heads, True, [None if m is None else torch.tensor(m, dtype=T) for m in x["pmasks"]])
        pooled, targets_c, targets_r, is_pos = [], [], [], []
        for (l, a, yy, xx, ri), arect in zip(x["pos"], x["pos_rect"]):                    # :91-120
            v = outs[l - 1][(a - 1) * 6:a * 6, yy - 1, xx - 1]
            cls_loss = cls_loss + Fn.cross_entropy(v[None, 0:2], torch.tensor([0]))       # :104 target 1
            roi = x["rois"][ri - 1]
            tgt = [f32(t) for t in input_to_anchor(arect, roi)]                           # :110 (a FloatTensor)
            reg_loss = reg_loss + 10.0 * Fn.smooth_l1_loss(v[2:6], torch.tensor(tgt, dtype=T), reduction="sum")   # :112
            prop = anchor_to_input(arect, [f32(t) for t in v[2:6].detach().tolist()])    # :111 reg_proposal (values, no gradient)
            win, _ = roi_slice(lay5, roi, outs[4])                                        # :117
            pooled.append(Fn.adaptive_max_pool2d(win[None], (kh, kw))[0].reshape(-1))     # :118
            targets_c.append(x["roi_class"][ri - 1])                                      # :154
            targets_r.append([f32(t) for t in input_to_anchor(prop, roi)])                # :156
            is_pos.append(True)
        for (l, a, yy, xx), arect in zip(x["neg"], x["neg_rect"]):                        # :123-140
            v = outs[l - 1][(a - 1) * 6:a * 6, yy - 1, xx - 1]
            cls_loss = cls_loss + Fn.cross_entropy(v[None, 0:2], torch.tensor([1]))       # :132 target 2
            win, _ = roi_slice(lay5, arect, outs[4])                                      # :137 (the anchor is the rect)
            pooled.append(Fn.adaptive_max_pool2d(win[None], (kh, kw))[0].reshape(-1))
            targets_c.append(bg); targets_r.append([0.0, 0.0, 0.0, 0.0]); is_pos.append(False)
        R = len(pooled)
        if R > 0:                                                                         # :146-186
            cinput = torch.stack(pooled)
            crout, ccout = cnet_forward(P, cinput, cls, True, [torch.tensor(m, dtype=T) for m in x["cmasks"]], None)
            posm = torch.tensor(is_pos)[:, None].to(T)
            crz = crout * posm                                                            # :169 rows of negatives zeroed
            creg_loss = creg_loss + 10.0 * Fn.smooth_l1_loss(crz, torch.tensor(targets_r, dtype=T), reduction="sum")   # :170
            ccls_loss = ccls_loss + Fn.nll_loss(ccout, torch.tensor(targets_c) - 1)       # :174-175 (sizeAverage: mean over R)
            if bi == 0:
                keep.update(cinput=cinput.detach().numpy().copy(), crout=crout.detach().numpy().copy(), ccout=ccout.detach().numpy().copy())
        if bi == 0:
            keep.update(outs=[o.detach().numpy().copy() for o in outs])
        n_p, n_n = len(x["pos"]), len(x["neg"])
        reg_count += n_p; cls_count += n_p + n_n; creg_count += n_p; ccls_count += 1      # :191-195
    total = cls_loss + reg_loss + creg_loss + ccls_loss      # what pnet:backward / cnet:backward accumulate the gradient of
    total.backward()
    grad = (w.grad / cls_count).numpy().copy()                                            # :200
    cls_loss, reg_loss, creg_loss, ccls_loss = [float(v.detach()) if torch.is_tensor(v) else float(v) for v in (cls_loss, reg_loss, creg_loss, ccls_loss)]
    stats = dict(pcls=cls_loss / cls_count, preg=reg_loss / max(reg_count, 1e-300), dcls=ccls_loss / ccls_count, dreg=creg_loss / max(creg_count, 1e-300))
    acc = np.array([cls_loss, reg_loss, cls_count, reg_count, creg_loss, creg_count, ccls_loss, ccls_count])
    return stats, acc, grad, keep


# ------------------------------------------------------------------------------------------ Detector.lua:17-141
def detect(flat_w, shapes, layers, heads, cls, cfg, img, bn_running, wtab, htab):
    with torch.no_grad():
        P = unflatten(torch.tensor(flat_w, dtype=T), shapes)
        H, W = img.shape[1], img.shape[2]
        outs = pnet_forward(P, torch.tensor(img, dtype=T), layers, heads, False, None)    # :31-33
        matches = []
        for i in range(4):                                                                # :39-66
            lsm = torch.stack([Fn.log_softmax(outs[i][a * 6:a * 6 + 2], dim=0) for a in range(3)])   # [aspect][2][y][x]
            o = outs[i]
            for y in range(o.shape[1]):
                for x in range(o.shape[2]):
                    for a in range(3):
                        c1 = f32(lsm[a, 0, y, x])
                        if math.exp(c1) > 0.95:                                           # :54
                            anchor = [float(wtab[i, a, x, 0]), float(htab[i, a, y, 0]), float(wtab[i, a, x, 1]), float(htab[i, a, y, 1])]   # Anchors.lua:60-67
                            r = anchor_to_input(anchor, [f32(t) for t in o[a * 6 + 2:a * 6 + 6, y, x].tolist()])
                            if r[0] < W and r[2] > 0 and r[1] < H and r[3] > 0:           # :58 overlaps(input_rect)
                                matches.append(dict(p=c1, idx=[i + 1, a + 1, y + 1, x + 1], r=r))
        res = dict(match_idx=np.array([m["idx"] for m in matches], np.int32).reshape(-1, 4), match_p=np.array([m["p"] for m in matches], np.float32),
                   match_rect=np.array([m["r"] for m in matches], np.float64).reshape(-1, 4))
        if not matches:
            return res
        bb = np.array([m["r"] for m in matches], dtype=np.float32)                        # :74-79 (a FloatTensor, main.lua:51)
        pick = naive_np.nms(bb, 0.25, "y2")                                               # :82 a tensor `scores` falls through to y2 (nms.lua:37-43)
        cands = [matches[int(p) - 1] for p in pick]
        lay5 = localizer_layers(layers, len(layers))
        kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
        cin = torch.stack([Fn.adaptive_max_pool2d(roi_slice(lay5, c["r"], outs[4])[0][None], (kh, kw))[0].reshape(-1) for c in cands])   # :93-98
        bbox, clsout = cnet_forward(P, cin, cls, False, None, torch.tensor(bn_running, dtype=T))   # :101-103
        res.update(cand_ids=np.array(pick, np.int64), cand_bbox=bbox.numpy().copy(), cand_cls=clsout.numpy().copy())
        bg = cfg["class_count"] + 1
        yclass = {}
        for i, c in enumerate(cands):                                                     # :105-123
            c["r2"] = anchor_to_input(c["r"], [f32(t) for t in bbox[i].tolist()])
            cp = np.array([f32(t) for t in clsout[i].tolist()], np.float32)
            k = int(np.argmax(cp)) + 1                                                    # :110 sort descending, first
            c["cls"], c["conf"] = k, float(cp[k - 1])
            if k != bg and math.exp(c["conf"]) > 0.2:
                yclass.setdefault(k, []).append(i)
        winners = []
        for k in sorted(yclass):                                                          # :125-136 (pairs(): ascending class here)
            ids = yclass[k]
            b5 = np.array([cands[i]["r2"] + [cands[i]["conf"]] for i in ids], dtype=np.float32)
            for p in naive_np.nms(b5, 0.1, "y2"):                                         # :133 tensor scores -> y2 again
                i = ids[int(p) - 1]
                winners.append([i + 1, k, cands[i]["conf"]] + cands[i]["r2"])
        res["winners"] = np.array(winners, np.float64).reshape(-1, 7)   # candidate (1-based), class, log-confidence, r2
        return res


# ------------------------------------------------------------------------------------------ inputs
def make_inputs(k, layers, heads, H, W, wtab, htab):
    """Image k of the batch: a seeded frame, two ground-truth boxes, positives by the brute-force restatement of findPositive,
    negatives = a fixed pick among the anchors inside the image that overlap no box (any such list is a valid batch)."""
    img = np.random.RandomState(1000 + k).randn(3, H, W).astype(np.float32)
    rois = [[20.0 + 10 * k, 30.0, 90.0 + 10 * k, 100.0], [60.0, 40.0 + 5 * k, 120.0, 104.0]]
    roi_class = [1 + k, 3]
    pos = naive_np.find_positive(wtab, htab, rois, [0, 0, W, H], 0.5, 0.25, True)
    sizes = feature_sizes(layers, heads, H, W)
    pos = [p for p in pos if p[2] <= sizes[p[0] - 1][0] and p[3] <= sizes[p[0] - 1][1]]   # cleanAnchors, objective.lua:32-43
    arect = lambda l, a, y, x: [float(wtab[l - 1, a - 1, x - 1, 0]), float(htab[l - 1, a - 1, y - 1, 0]), float(wtab[l - 1, a - 1, x - 1, 1]), float(htab[l - 1, a - 1, y - 1, 1])]
    rng = np.random.RandomState(77 + k)
    neg = []
    while len(neg) < 6:
        l = int(rng.randint(1, 5)); a = int(rng.randint(1, 4)); y = int(rng.randint(1, sizes[l - 1][0] + 1)); x = int(rng.randint(1, sizes[l - 1][1] + 1))
        r = arect(l, a, y, x)
        inside = r[0] >= 0 and r[1] >= 0 and r[2] <= W and r[3] <= H
        if inside and all(naive_np._iou(roi, r) < 0.25 for roi in rois) and [l, a, y, x] not in neg:
            neg.append([l, a, y, x])
    R = len(pos) + len(neg)
    rs = np.random.RandomState(50 + k)
    pm = [None] + [(rs.rand(l["filters"]) > 0.4).astype(np.float32) for l in layers[1:]]
    cm = [(rs.rand(R, c["n"]) > 0.5).astype(np.float32) for c in TINY_CLS]
    return dict(img=img, rois=rois, roi_class=roi_class, pos=pos, pos_rect=[arect(*p[:4]) for p in pos], neg=neg, neg_rect=[arect(*n) for n in neg],
                pmasks=pm, cmasks=cm)


def feature_sizes(layers, heads, H, W):
    hw = []
    h, w = H, W
    for l in layers:
        h, w = (h + 1) // 2, (w + 1) // 2     # 3x3 pad 1 keeps the size, the ceil-mode pool halves it
        hw.append((h, w))
    return [(hw[a["input"] - 1][0] - a["kW"] + 1, hw[a["input"] - 1][1] - a["kW"] + 1) for a in heads]


def main():
    layers, heads, cls, cfg = TINY_LAYERS, TINY_HEADS, TINY_CLS, CFG
    shapes = param_shapes(layers, heads, cls, cfg)
    n = sum(int(np.prod(s)) for _, s in shapes)
    lay_scale = [localizer_layers(layers, a["input"]) + [[a["kW"], a["kW"], 1, 1, 0, 0], [1, 1, 1, 1, 0, 0]] for a in heads]
    wtab, htab = naive_np.anchor_tables(lay_scale, cfg["scales"])
    H, W = 112, 128
    # ---- training: a batch of two images (the cross-image accumulation and the single division are part of the composition)
    wts = (np.random.RandomState(1).randn(n) * 0.1).astype(np.float32)
    batch = [make_inputs(k, layers, heads, H, W, wtab, htab) for k in range(2)]
    stats, acc, grad, keep = loss_and_gradient(wts, shapes, layers, heads, cls, cfg, batch)
    out = dict(weights=wts, H=H, W=W, n_images=len(batch), stats=np.array([stats["pcls"], stats["preg"], stats["dcls"], stats["dreg"]]), acc=acc,
               gradient=grad.astype(np.float32), gradient_l2=np.linalg.norm(grad))
    for k, x in enumerate(batch):
        out.update({"img%d" % k: x["img"], "rois%d" % k: np.array(x["rois"]), "roi_class%d" % k: np.array(x["roi_class"], np.int32),
                    "pos%d" % k: np.array(x["pos"], np.int32).reshape(-1, 5), "pos_rect%d" % k: np.array(x["pos_rect"]).reshape(-1, 4),
                    "neg%d" % k: np.array(x["neg"], np.int32).reshape(-1, 4), "neg_rect%d" % k: np.array(x["neg_rect"]).reshape(-1, 4)})
        for b, m in enumerate(x["pmasks"]):
            if m is not None:
                out["pmask%d_%d" % (k, b)] = m
        for i, m in enumerate(x["cmasks"]):
            out["cmask%d_%d" % (k, i)] = m
    for i, o in enumerate(keep["outs"]):
        out["pnet_out%d" % (i + 1)] = o.astype(np.float32)
    out.update(cinput=keep["cinput"].astype(np.float32), crout=keep["crout"].astype(np.float32), ccout=keep["ccout"].astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "train_tiny.npz"), **out)
    print("train_tiny: %d parameters, %d + %d positives, stats %s, |g| %.6g" % (n, len(batch[0]["pos"]), len(batch[1]["pos"]), stats, out["gradient_l2"]))
    # ---- detect: weights whose anchor nets are confident somewhere (plain N(0, 0.1) weights pass p > 0.95 nowhere)
    rng = np.random.RandomState(3)
    wd = (rng.randn(n) * 0.1).astype(np.float32)
    o = 0
    for name, shp in shapes:
        cnt = int(np.prod(shp))
        if name.endswith(".w1") or name.startswith("cls."):
            wd[o:o + cnt] *= 12.0
        if name.endswith(".b1"):           # foreground logits of the three aspects lifted: about a third of the anchors pass p > 0.95
            wd[o + 0:o + cnt:6] += 3.0
        o += cnt
    img = np.random.RandomState(2000).randn(3, H, W).astype(np.float32)
    bn = np.concatenate([np.zeros(cls[0]["n"], np.float32), np.ones(cls[0]["n"], np.float32)])
    res = detect(wd, shapes, layers, heads, cls, cfg, img, bn, wtab, htab)
    print("detect_tiny: %d matches, %d candidates, %d winners" % (len(res["match_idx"]), len(res.get("cand_ids", [])), len(res.get("winners", []))))
    np.savez_compressed(os.path.join(HERE, "detect_tiny.npz"), weights=wd, img=img, bn_running=bn, H=H, W=W,
                        **{k: (v.astype(np.float32) if v.dtype == np.float64 and k in ("cand_bbox", "cand_cls") else v) for k, v in res.items()})


if __name__ == "__main__":
    main()
