"""Second, deliberately naive restatement (numpy / pure-Python loops) of the integer-exact pieces of
the reference: nms.lua, Localizer.lua, Anchors.lua:7-58,86-195 and objective.lua:5-13.

TEST INFRASTRUCTURE ONLY.  It exists to pin the C oracle (oracle/*.c): both were written
independently from the Lua sources and must agree bit for bit (tests/test_oracle_pinning.py,
tests/golden/make_golden.py).  Every tensor op of nms.lua is one explicit np.float32 operation."""
import math

import numpy as np

f32 = np.float32


def nms(boxes, overlap, key="y2"):
    """nms.lua:23-102 with torch tensor ops spelled out as numpy float32 array ops."""
    boxes = np.asarray(boxes, dtype=f32)
    if boxes.size == 0:
        return np.zeros(0, dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = ((x2 - x1) + f32(1)) * ((y2 - y1) + f32(1))  # :35
    if key == "y2":
        scores = y2  # :42
    elif key == "area":
        scores = area
    else:
        scores = boxes[:, int(key) - 1]
    order = sorted(range(len(scores)), key=lambda i: (scores[i], i))  # ascending, ties by index (documented rule)
    I = list(order)
    pick = []
    ov = f32(overlap)
    while len(I) > 0:
        i = I[-1]
        pick.append(i + 1)
        if len(I) == 1:
            break
        I = I[:-1]
        idx = np.array(I, dtype=np.int64)
        xx1 = np.maximum(x1[idx], x1[i]); yy1 = np.maximum(y1[idx], y1[i])  # :78-79
        xx2 = np.minimum(x2[idx], x2[i]); yy2 = np.minimum(y2[idx], y2[i])  # :80-81
        w = np.maximum((xx2 + f32(-1) * xx1) + f32(1), f32(0))              # :85
        h = np.maximum((yy2 + f32(-1) * yy1) + f32(1), f32(0))              # :86
        inter = w * h                                                       # :89
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / ((area[idx] + area[i]) - inter)                   # :94
        I = [j for j, v in zip(I, iou) if v <= ov]                          # :96
    return np.array(pick, dtype=np.int64)


def _lua_mod(a, b):
    return a - math.floor(a / b) * b


def input_to_feature(layers, rect):
    """Localizer.lua:41-67; layers rows = [kW,kH,dW,dH,padW,padH]."""
    minX, minY, maxX, maxY = [float(v) for v in rect]
    for kW, kH, dW, dH, padW, padH in layers:
        if dW < kW:
            minX -= kW - dW; minY -= kH - dH; maxX += kW - dW; maxY += kH - dH
        minX += padW; minY += padH; maxX += padW; maxY += padH
        minX /= dH; minY /= dH
        maxX = max((maxX - kW) / dW + 1, minX + 1) if _lua_mod(maxX - kW, dW) == 0 else max(math.ceil((maxX - kW) / dW) + 1, minX + 1)
        maxY = max((maxY - kH) / dW + 1, minY + 1) if _lua_mod(maxY - kH, dH) == 0 else max(math.ceil((maxY - kH) / dH) + 1, minY + 1)
    return [math.floor(minX), math.floor(minY), math.ceil(maxX), math.ceil(maxY)]


def feature_to_input(layers, minX, minY, maxX, maxY):
    """Localizer.lua:69-79"""
    for kW, kH, dW, dH, padW, padH in reversed(layers):
        minX = minX * dW - padW
        minY = minY * dH - padW
        maxX = maxX * dW - padH + kW - dW
        maxY = maxY * dH - padH + kH - dH
    return [minX, minY, maxX, maxY]


def roi_window(layers, rect, fmH, fmW):
    """objective.lua:5-13 -> [row_lo,row_hi,col_lo,col_hi] (1-based inclusive)"""
    r = input_to_feature(layers, rect)
    c = [min(max(r[0], 0), fmW), min(max(r[1], 0), fmH), max(min(r[2], fmW), 0), max(min(r[3], fmH), 0)]
    return [int(min(c[1] + 1, c[3])), int(c[3]), int(min(c[0] + 1, c[2])), int(c[2])]


def anchor_tables(layers_per_scale, scales):
    """Anchors.lua:14-57 -> w,h float32 [n][3][200][2]"""
    n = len(scales)
    w = np.zeros((n, 3, 200, 2), dtype=f32); h = np.zeros((n, 3, 200, 2), dtype=f32)
    for i, s in enumerate(scales):
        a = s / math.sqrt(2)
        for j, (bw, bh) in enumerate(((s, s), (2 * a, a), (a, 2 * a))):
            for y in range(1, 201):
                r = feature_to_input(layers_per_scale[i], 0, y - 1, 0, y)
                cy = (r[1] + r[3]) / 2
                h[i, j, y - 1, 0] = cy - bh * 0.5
                h[i, j, y - 1, 1] = (cy - bh * 0.5) + bh
            for x in range(1, 201):
                r = feature_to_input(layers_per_scale[i], x - 1, 0, x, 0)
                cx = (r[0] + r[2]) / 2
                w[i, j, x - 1, 0] = cx - bw * 0.5
                w[i, j, x - 1, 1] = (cx - bw * 0.5) + bw
    return w, h


def _iou(a, b):
    minx = max(a[0], b[0]); miny = max(a[1], b[1]); maxx = min(a[2], b[2]); maxy = min(a[3], b[3])
    i = (maxx - minx) * (maxy - miny) if (maxx >= minx and maxy >= miny) else 0.0
    return i / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i)


def find_positive(w, h, rois, clip, pos_thr, neg_thr, include_best):
    """Anchors.lua:147-195 by brute force: every anchor of every (scale, aspect) is tested against the
    range predicates of findRangesXY (:112-135) one by one instead of by binary search."""
    out = []
    for ri, roi in enumerate(rois):
        best, best_iou, have = [], -1.0, bool(include_best)
        for i in range(4):
            for j in range(3):
                xs = [x for x in range(200) if float(w[i, j, x, 1]) > roi[0] and float(w[i, j, x, 0]) < roi[2]
                      and float(w[i, j, x, 0]) >= clip[0] and float(w[i, j, x, 1]) <= clip[2]]
                ys = [y for y in range(200) if float(h[i, j, y, 1]) > roi[1] and float(h[i, j, y, 0]) < roi[3]
                      and float(h[i, j, y, 0]) >= clip[1] and float(h[i, j, y, 1]) <= clip[3]]
                for y in ys:
                    for x in xs:
                        a = [float(w[i, j, x, 0]), float(h[i, j, y, 0]), float(w[i, j, x, 1]), float(h[i, j, y, 1])]
                        v = _iou(roi, a)
                        if v > pos_thr:
                            out.append([i + 1, j + 1, y + 1, x + 1, ri + 1]); have = False
                        elif v > neg_thr and have and v >= best_iou:
                            if v - 0.025 > best_iou:
                                best = []
                            best.append([i + 1, j + 1, y + 1, x + 1, ri + 1]); best_iou = v
        if have and best_iou > 0:
            out.extend(best)
    return out
