"""nms(boxes, overlap, scores) -- host-side mirror of nms.lua:23-102 over frcnn_nms_host /
frcnn_nms_device.  Key dispatch is the reference's (nms.lua:37-43): a number selects that column
(1-based), the string 'area' selects the area, ANYTHING ELSE -- including a score tensor or None,
which is what both call sites of the reference pass (Detector.lua:82,133) -- sorts by max-y.
Returns the 1-based row ids of the survivors in pick order (a LongTensor in the reference)."""
import ctypes as C
import numbers

import numpy as np

from . import _lib
from .tensor import DeviceTensor, ptr, stream_ptr


def _key(scores):
    if isinstance(scores, numbers.Number) and not isinstance(scores, bool):
        return 2, int(scores)
    if isinstance(scores, str) and scores == "area":
        return 1, 0
    return 0, 0  # nms.lua:42 "use max_y"


def nms(boxes, overlap, scores=None):
    key_mode, key_col = _key(scores)
    on_device = isinstance(boxes, DeviceTensor) or (hasattr(boxes, "is_cuda") and boxes.is_cuda)
    if on_device:
        n = boxes.shape[0] if len(boxes.shape) == 2 else 0
        if n == 0 or int(np.prod(boxes.shape)) == 0:
            return np.zeros(0, dtype=np.int64)
        ncols = boxes.shape[1]
        wsb = _lib.load().frcnn_nms_workspace_bytes(n)
        ws = DeviceTensor.empty((wsb,), np.uint8)
        # picks and their count in ONE device buffer (the count behind the n ids): one read-back, one wait for the device
        pick = DeviceTensor.empty((n + 1,), np.int64)
        _lib.call("frcnn_nms_device", ptr(boxes), n, ncols, C.c_float(overlap), key_mode, key_col, ptr(pick),
                  C.c_void_p(pick.ptr + 8 * n), ptr(ws), wsb, stream_ptr())
        host = pick.numpy()
        k = int(host[n:].view(np.int32)[0])
        return host[:k].copy()
    b = np.ascontiguousarray(boxes.detach().cpu().numpy() if hasattr(boxes, "detach") else boxes, dtype=np.float32)
    if b.size == 0:  # nms.lua:26-28
        return np.zeros(0, dtype=np.int64)
    n, ncols = b.shape
    pick = np.zeros(n, dtype=np.int64)
    count = C.c_int(0)
    _lib.call("frcnn_nms_host", b.ctypes.data_as(C.c_void_p), n, ncols, C.c_float(overlap), key_mode, key_col,
              pick.ctypes.data_as(C.c_void_p), C.byref(count))
    return pick[:count.value].copy()
