"""NMS survivor ids: HIP kernels (through the C ABI) vs the CPU oracle -- bit-exact (nms.lua:23-102)."""
import json
import os

import numpy as np
import pytest

from util import random_boxes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_nms_known_answer_quirk(F):
    # scores passed as a tensor are ignored, key = y2 -> picks {3, 2}
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 70]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.1], dtype=np.float32)
    assert F.nms(boxes, 0.25, scores).tolist() == [3, 2]
    assert F.nms(boxes, 0.25).tolist() == [3, 2]
    assert F.nms(np.zeros((0, 4), np.float32), 0.25).tolist() == []
    assert F.nms(boxes[:1], 0.25).tolist() == [1]


@pytest.mark.parametrize("n", [2, 63, 64, 65, 129, 300, 2000, 6000])
@pytest.mark.parametrize("thr", [0.25, 0.1])
def test_nms_matches_oracle(F, O, n, thr):
    rng = np.random.RandomState(n)
    b = random_boxes(rng, n)
    got = F.nms(b, thr)
    want = O.nms(b, thr)
    assert got.tolist() == want.tolist()


def test_nms_key_modes_and_five_columns(F, O):
    rng = np.random.RandomState(7)
    b = random_boxes(rng, 777)
    conf = rng.permutation(777).astype(np.float32)[:, None] / 777
    b5 = np.concatenate([b, conf], 1)
    assert F.nms(b5, 0.1, b5[:, 4]).tolist() == O.nms(b5, 0.1, 0).tolist()          # tensor -> y2
    assert F.nms(b5, 0.1, 5).tolist() == O.nms(b5, 0.1, 2, 5).tolist()              # column 5
    assert F.nms(b5, 0.3, "area").tolist() == O.nms(b5, 0.3, 1).tolist()            # 'area'


def test_nms_ties_follow_documented_rule(F, O):
    rng = np.random.RandomState(3)
    b = random_boxes(rng, 500, unique_y2=False)
    b[:, 3] = np.round(b[:, 3] / 8) * 8  # many equal keys
    assert F.nms(b, 0.25).tolist() == O.nms(b, 0.25).tolist()


def test_nms_device_pointer_variant(F, O):
    rng = np.random.RandomState(11)
    b = random_boxes(rng, 1500)
    d = F.DeviceTensor.from_numpy(b)
    assert F.nms(d, 0.25).tolist() == O.nms(b, 0.25).tolist()


def test_nms_full_size_properties(F, O):
    """n = 26 544 (every anchor of an 800x450 frame): properties that need no oracle run of that size."""
    rng = np.random.RandomState(5)
    n = 26544
    b = random_boxes(rng, n)
    pick = F.nms(b, 0.25)
    ids = pick - 1
    assert len(set(ids.tolist())) == len(ids)
    keys = b[ids, 3]
    assert np.all(keys[:-1] >= keys[1:])                    # pick order = descending key
    assert ids[0] == int(np.argmax(b[:, 3]))                # the max-key box always survives
    again = F.nms(b[ids], 0.25)                             # idempotence: survivors do not suppress each other
    assert again.tolist() == list(range(1, len(ids) + 1))
    sub = rng.choice(n, 3000, replace=False)                # exact parity on a 3000-box subset
    assert F.nms(b[sub], 0.25).tolist() == O.nms(b[sub], 0.25).tolist()


def test_nms_golden_fixture(F):
    with open(os.path.join(GOLD, "nms_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        b = np.array(c["boxes"], dtype=np.float32).reshape(-1, c["ncols"])
        key = c["key"]
        scores = None if key == "y2" else ("area" if key == "area" else int(key))
        assert F.nms(b, c["overlap"], scores).tolist() == c["pick"], c["name"]


@pytest.mark.parametrize("n,nclass", [(7, 3), (500, 16), (3000, 200), (4000, 1)])
def test_class_aware_nms_equals_one_nms_per_class(F, O, n, nclass):
    """frcnn_nms_device_classes: the per-class loop of Detector.lua:125-136 in one pass.  A stable partition of its picks
    by class must be, per class, exactly nms() on that class's rows (5 columns: box + confidence, key = max-y)."""
    rng = np.random.RandomState(n + nclass)
    b = np.concatenate([random_boxes(rng, n), rng.rand(n, 1).astype(np.float32)], 1)
    cls = rng.randint(1, nclass + 1, n).astype(np.int32)
    db, dc = F.DeviceTensor.from_numpy(b), F.DeviceTensor.from_numpy(cls)
    wsb = F._lib.load().frcnn_nms_workspace_bytes(n)
    ws = F.DeviceTensor.empty((wsb,), np.uint8); pick = F.DeviceTensor.empty((n,), np.int64); cnt = F.DeviceTensor.empty((1,), np.int32)
    import ctypes as C
    F._lib.call("frcnn_nms_device_classes", F.ptr(db), n, 5, C.c_float(0.1), 0, 0, F.ptr(dc), F.ptr(pick), F.ptr(cnt),
                F.ptr(ws), wsb, F.stream_ptr())
    got = pick.numpy()[:int(cnt.numpy()[0])]
    total = 0
    for c in range(1, nclass + 1):
        rows = np.nonzero(cls == c)[0]
        want = rows[O.nms(b[rows], 0.1) - 1] + 1 if len(rows) else np.zeros(0, np.int64)
        mine = np.array([g for g in got if cls[g - 1] == c], dtype=np.int64)
        assert mine.tolist() == want.tolist(), "class %d" % c
        total += len(want)
    assert total == len(got)
    if nclass == 1:   # one class: the plain NMS
        assert got.tolist() == O.nms(b, 0.1).tolist()
