"""NMS micro-benchmark: n in {300, 2000, 6000, 26544} boxes with unique y2 keys, thresholds 0.25 / 0.1;
device NMS (frcnn_nms_device through nms()) vs the CPU oracle, ids compared."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import frcnn_amd as F
import pyoracle as O
from util import random_boxes

for n in (300, 2000, 6000, 26544):
    rng = np.random.RandomState(n)
    b = random_boxes(rng, n)
    db = F.DeviceTensor.from_numpy(b)
    for thr in (0.25, 0.1):
        pick = F.nms(db, thr, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 10
        for _ in range(reps):
            pick = F.nms(db, thr, None)
        torch.cuda.synchronize()
        gpu = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        want = O.nms(b, thr)
        cpu = time.perf_counter() - t0
        same = list(pick) == want.tolist()
        print("n=%6d thr=%.2f  GPU %8.3f ms   CPU oracle %9.2f ms   kept %5d  ids identical: %s" % (n, thr, gpu * 1e3, cpu * 1e3, len(want), same))
