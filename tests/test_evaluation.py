"""mean_average_precision / voc_ap (frcnn_amd/evaluation.py): hand-computed known answers (CPU)."""
import numpy as np

from frcnn_amd.Rect import Rect
from frcnn_amd.evaluation import mean_average_precision, voc_ap


def _toy():
    A, B, Cc, D = Rect(0, 0, 10, 10), Rect(100, 100, 110, 110), Rect(0, 0, 10, 10), Rect(50, 50, 60, 60)
    gt = [(0, 1, A), (0, 1, B), (1, 1, Cc), (1, 2, D)]
    det = [(1, 1, 0.5, Rect(0, 0, 10, 9)),        # d5: IoU 0.9 with C -> TP
           (0, 1, 0.9, Rect(0, 0, 10, 8)),        # d1: IoU 0.8 with A -> TP
           (1, 1, 0.7, Rect(0, 0, 10, 3)),        # d3: IoU 0.3 with C -> FP (low overlap)
           (0, 1, 0.8, Rect(0, 0, 10, 6)),        # d2: IoU 0.6 with A, already matched -> FP (duplicate)
           (0, 1, 0.6, Rect(100, 100, 110, 105.5)),   # d4: IoU 0.55 with B -> TP
           (0, 3, 0.99, Rect(0, 0, 10, 10))]      # a class without ground truth: ignored
    return det, gt


def test_known_answer_voc2010_and_voc2007():
    det, gt = _toy()
    # class 1 in confidence order: TP FP FP TP TP -> recall 1/3 1/3 1/3 2/3 1, precision 1 1/2 1/3 1/2 3/5
    r = mean_average_precision(det, gt)
    assert abs(r["ap"][1] - (1 / 3 * 1.0 + 1 / 3 * 0.6 + 1 / 3 * 0.6)) < 1e-12
    assert r["ap"][2] == 0.0 and set(r["ap"]) == {1, 2}
    assert abs(r["mAP"] - r["ap"][1] / 2) < 1e-12
    assert (r["tp"], r["fp"]) == (3, 2) and r["npos"] == {1: 3, 2: 1}
    r07 = mean_average_precision(det, gt, use_07_metric=True)
    assert abs(r07["ap"][1] - (4 * 1.0 + 3 * 0.6 + 4 * 0.6) / 11) < 1e-12


def test_threshold_and_edge_cases():
    det, gt = _toy()
    r = mean_average_precision(det, gt, iou_threshold=0.58)     # d4 (0.55) drops out
    assert (r["tp"], r["fp"]) == (2, 3)
    assert mean_average_precision([], gt)["mAP"] == 0.0
    assert np.isnan(mean_average_precision(det, [])["mAP"])
    assert voc_ap([1.0], [1.0]) == 1.0 and abs(voc_ap([0.5, 1.0], [1.0, 1.0], True) - 1.0) < 1e-12
