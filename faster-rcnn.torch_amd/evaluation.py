"""Evaluation loop -- what the reference lists as still to do (README.md:11,13) and only sketches as
`evaluation_demo` (main.lua:183-216: `nextValidation(1)` -> `Detector:detect` -> draw): validation losses and a
PASCAL-VOC style mean average precision over `BatchIterator:nextValidation` (BatchIterator.lua:279-317).

  validation_losses(model, batch_iterator, count)  -> {pcls, preg, dcls, dreg, ...}: the four statistics of
      objective.lua:202-205 on `count` validation images, forward only, networks in evaluate() mode (SpatialDropout x(1-p),
      Dropout identity, BatchNormalization running statistics); the examples of an image are assembled exactly like the
      training ones (BatchIterator.lua:198-225) from a generator of their own, so the training stream is not disturbed.
  evaluate_detections(detector, batch_iterator, count) -> {mAP, ap: {class: AP}, ...}: Detector:detect on `count`
      validation images against their ground-truth boxes.  A detection (class c, box r2, confidence) is a true positive when
      its Rect.IoU (Rect.lua:138-141) with a not yet matched ground-truth box of class c in the same image is >= iou_threshold
      (0.5); detections are taken in order of decreasing confidence; AP is the area under the monotone precision envelope
      (VOC2010+) or the 11-point average (use_07_metric=True); mAP averages the classes that have ground truth.

All arithmetic of the networks runs through the C ABI (frcnn_pnet_forward, frcnn_rpn_loss, frcnn_loss_accumulate,
frcnn_roi_pool_forward, frcnn_cnet_forward, frcnn_cnet_losses, Detector.detect); the bookkeeping here is host code."""
import ctypes as C

import numpy as np

from . import _lib
from .Anchors import MT19937
from .BatchIterator import assemble_examples
from .Localizer import Localizer
from .Rect import Rect
from .objective import roi_windows
from .synthetic import clean_examples, output_map_sizes
from .tensor import DeviceTensor, ptr, stream_ptr, to_device


def validation_losses(model, batch_iterator, count, seed=1234, negatives=16):
    """-> dict(pcls, preg, dcls, dreg, images, examples, positives).  `batch_iterator` needs nextValidation(n) -> [{img, rois}]
    and .anchors (BatchIterator / any object with the same two members)."""
    import torch
    cfg = model["cfg"]
    pnet, cnet, native = model["pnet"], model["cnet"], model["native"]
    bgclass = cfg["class_count"] + 1
    ncls = cfg["class_count"] + 1
    kh, kw = cfg["roi_pooling"]["kh"], cfg["roi_pooling"]["kw"]
    planes = model["layers"][-1]["filters"]
    D = kh * kw * planes
    localizer = Localizer(pnet.outnode.children[-1])
    anchors = batch_iterator.anchors
    rng = MT19937(seed)
    acc = torch.zeros(8, dtype=torch.float64, device="cuda")     # {cls, reg, -, -, creg, ccls, -, -} like the objective
    cls_count = reg_count = creg_count = ccls_count = 0
    was_training = (pnet.train, cnet.train)
    pnet.evaluate(); cnet.evaluate()
    s = stream_ptr()
    keep = []
    try:
        images = 0
        while images < count:
            for x in batch_iterator.nextValidation(1):
                img = to_device(x["img"])
                _, H, W = img.shape
                outputs = pnet.forward(img)
                sizes = [(o.shape[1], o.shape[2]) for o in outputs[:-1]]
                pos, neg = assemble_examples(anchors, cfg, x["rois"], W, H, rng, negatives=negatives)
                pos, neg = clean_examples(pos, sizes), clean_examples(neg, sizes)   # cleanAnchors, objective.lua:74-75
                npos, nneg = len(pos), len(neg)
                E = npos + nneg
                images += 1
                ccls_count += 1        # objective.lua:198: one per image
                if E == 0:
                    continue
                anch = [e[0] for e in pos] + [e[0] for e in neg]
                ex_idx = np.array([(a.layer, a.aspect, a.index[1], a.index[2]) for a in anch], dtype=np.int32)
                ex_anchor = np.array([(a.minX, a.minY, a.maxX, a.maxY) for a in anch], dtype=np.float64)
                ex_roi = np.zeros((max(npos, 1), 4), dtype=np.float64)
                ex_class = np.zeros(max(npos, 1), dtype=np.int32)
                if npos:
                    ex_roi[:npos] = [(e[1].rect.minX, e[1].rect.minY, e[1].rect.maxX, e[1].rect.maxY) for e in pos]
                    ex_class[:npos] = [e[1].class_index for e in pos]
                fm = outputs[-1]
                fmC, fmH, fmW = fm.shape
                wins = roi_windows(np.concatenate([ex_roi[:npos], ex_anchor[npos:]], 0), localizer, fmH, fmW)
                d_idx, d_anchor = DeviceTensor.from_numpy(ex_idx), DeviceTensor.from_numpy(ex_anchor)
                d_roi, d_class, d_wins = DeviceTensor.from_numpy(ex_roi), DeviceTensor.from_numpy(ex_class), DeviceTensor.from_numpy(wins)
                # anchor losses on the sampled anchors (objective.lua:91-140); the gradients it also writes go to scratch maps
                maps = (C.c_void_p * 4)(*[outputs[l].ptr for l in range(4)])
                scratch = [DeviceTensor.zeros(outputs[l].shape) for l in range(4)]
                deltas = (C.c_void_p * 4)(*[t.ptr for t in scratch])
                Hs = (C.c_int * 4)(*[outputs[l].shape[1] for l in range(4)])
                Ws = (C.c_int * 4)(*[outputs[l].shape[2] for l in range(4)])
                ex_loss = DeviceTensor.empty((E, 2), np.float64)
                crtarget = DeviceTensor.empty((E, 4)); cctarget = DeviceTensor.empty((E,))
                _lib.call("frcnn_rpn_loss", maps, deltas, Hs, Ws, ptr(d_idx), ptr(d_anchor), ptr(d_roi), ptr(d_class), npos, nneg,
                          bgclass, ptr(ex_loss), ptr(crtarget), ptr(cctarget), s)
                _lib.call("frcnn_loss_accumulate", ptr(ex_loss), E, C.c_void_p(acc.data_ptr()), s)
                # region classification on the pooled examples (:117-119, :137-139, :146-177), forward only
                cinput = DeviceTensor.empty((E, D)); pidx = DeviceTensor.empty((E, D), np.int32)
                _lib.call("frcnn_roi_pool_forward", ptr(fm), fmC, fmH, fmW, ptr(d_wins), E, kh, kw, ptr(cinput), ptr(pidx), s)
                crout, ccout = cnet.forward(cinput)
                crdelta = DeviceTensor.empty((E, 4)); ccdelta = DeviceTensor.empty((E, ncls))
                _lib.call("frcnn_cnet_losses", ptr(crout), ptr(crtarget), ptr(ccout), ptr(cctarget), E, npos, ncls,
                          ptr(crdelta), ptr(ccdelta), C.c_void_p(acc.data_ptr() + 4 * 8), s)
                keep.append((d_idx, d_anchor, d_roi, d_class, d_wins, scratch, ex_loss, crtarget, cctarget, cinput, pidx, crdelta, ccdelta))
                reg_count += npos; cls_count += E; creg_count += npos      # :194-197
        a = acc.cpu().numpy()
    finally:
        pnet.train, cnet.train = was_training
    with np.errstate(divide="ignore", invalid="ignore"):
        return dict(pcls=float(np.float64(a[0]) / cls_count), preg=float(np.float64(a[1]) / reg_count),
                    dcls=float(np.float64(a[5]) / ccls_count), dreg=float(np.float64(a[4]) / creg_count),
                    images=images, examples=cls_count, positives=reg_count)


def voc_ap(recall, precision, use_07_metric=False):
    """Average precision of one class from its recall / precision points (in order of decreasing confidence)."""
    recall = np.asarray(recall, dtype=np.float64); precision = np.asarray(precision, dtype=np.float64)
    if use_07_metric:   # VOC2007: mean over t = 0, 0.1, ..., 1 of the best precision at recall >= t
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = precision[recall >= t].max() if np.any(recall >= t) else 0.0
            ap += p / 11.0
        return float(ap)
    mrec = np.concatenate([[0.0], recall, [1.0]])
    mpre = np.concatenate([[0.0], precision, [0.0]])
    for i in range(len(mpre) - 2, -1, -1):      # monotone precision envelope
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def mean_average_precision(detections, ground_truth, iou_threshold=0.5, use_07_metric=False):
    """detections: list of (image_id, class, confidence, Rect); ground_truth: list of (image_id, class, Rect).
    -> dict(mAP, ap={class: AP}, npos={class: #ground truth}, tp, fp)."""
    gt = {}
    for image_id, cls, rect in ground_truth:
        gt.setdefault(cls, {}).setdefault(image_id, []).append(rect)
    ap, npos_of = {}, {}
    tp_total = fp_total = 0
    for cls in sorted(gt):
        npos = sum(len(v) for v in gt[cls].values())
        npos_of[cls] = npos
        dets = [d for d in detections if d[1] == cls]
        dets.sort(key=lambda d: -d[2])            # stable: equal confidences keep their detection order
        used = dict((image_id, [False] * len(v)) for image_id, v in gt[cls].items())
        tp = np.zeros(len(dets)); fp = np.zeros(len(dets))
        for k, (image_id, _, conf, rect) in enumerate(dets):
            best, best_j = -1.0, -1
            for j, g in enumerate(gt[cls].get(image_id, [])):
                iou = Rect.IoU(rect, g)
                if iou > best:
                    best, best_j = iou, j
            if best >= iou_threshold and not used[image_id][best_j]:
                tp[k] = 1; used[image_id][best_j] = True
            else:
                fp[k] = 1                           # low overlap, or a second detection of an already matched box
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        recall = ctp / float(npos)
        precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        ap[cls] = voc_ap(recall, precision, use_07_metric) if len(dets) else 0.0
        tp_total += int(tp.sum()); fp_total += int(fp.sum())
    return dict(mAP=float(np.mean(list(ap.values()))) if ap else float("nan"), ap=ap, npos=npos_of, tp=tp_total, fp=fp_total)


def evaluate_detections(detector, batch_iterator, count, iou_threshold=0.5, use_07_metric=False):
    """Detector:detect on `count` validation images (main.lua:198-206) scored against their ground truth."""
    detections, ground_truth = [], []
    images = 0
    while images < count:
        for x in batch_iterator.nextValidation(1):
            image_id = images
            images += 1
            for roi in x["rois"]:
                ground_truth.append((image_id, int(roi.class_index), roi.rect))
            for w in detector.detect(x["img"]):
                detections.append((image_id, int(w["class"]), float(w["confidence"]), w["r2"]))
    res = mean_average_precision(detections, ground_truth, iou_threshold, use_07_metric)
    res.update(images=images, detections=len(detections), ground_truth=len(ground_truth))
    return res
