"""Minimal device-tensor plumbing for the host mirror: a raw HIP device buffer with a shape
(what `torch.CudaTensor` is to the reference's Lua code).  PyTorch tensors are accepted wherever a
device pointer is needed (`.data_ptr()`); nothing here computes anything."""
import ctypes as C

import numpy as np

from . import _lib


def stream_ptr(stream=None):
    """hipStream_t of torch's current stream (as void*), or the given raw handle."""
    if stream is not None:
        return C.c_void_p(stream)
    try:
        import torch
        if torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    except ImportError:
        pass
    return C.c_void_p(0)


def ptr(x):
    """Device pointer (c_void_p) of a DeviceTensor / torch CUDA tensor / int / None."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, DeviceTensor):
        return C.c_void_p(x.ptr)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        if hasattr(x, "is_cuda") and not x.is_cuda:
            raise _lib.FrcnnError("expected a device tensor, got a host tensor")
        return C.c_void_p(x.data_ptr())
    raise TypeError("cannot take a device pointer of %r" % type(x))


class DeviceTensor(object):
    """A view of (or an owned allocation in) HBM: pointer + shape + dtype."""

    def __init__(self, p, shape, dtype=np.float32, owner=None, owned=False):
        self.ptr = int(p)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner
        self._owned = owned

    @staticmethod
    def empty(shape, dtype=np.float32):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _lib.call("frcnn_malloc", C.byref(p), max(nbytes, 16))
        return DeviceTensor(p.value, shape, dtype, owned=True)

    @staticmethod
    def zeros(shape, dtype=np.float32):
        t = DeviceTensor.empty(shape, dtype)
        t.zero_()
        return t

    @staticmethod
    def from_numpy(a, stream=None):
        a = np.ascontiguousarray(a)
        t = DeviceTensor.empty(a.shape, a.dtype)
        t.copy_from_numpy(a, stream)
        return t

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            try:
                _lib.load().frcnn_free(C.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = 0

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def numel(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def data_ptr(self):
        return self.ptr

    def view(self, *shape):
        return DeviceTensor(self.ptr, shape, self.dtype, owner=self)

    def offset_view(self, elem_offset, shape):
        return DeviceTensor(self.ptr + elem_offset * self.dtype.itemsize, shape, self.dtype, owner=self)

    def zero_(self, stream=None):
        _lib.call("frcnn_zero", C.c_void_p(self.ptr), self.nbytes, stream_ptr(stream))
        return self

    def copy_from_numpy(self, a, stream=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        s = stream_ptr(stream)
        _lib.call("frcnn_memcpy_h2d", C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, s)
        _lib.call("frcnn_stream_sync", s)  # `a` may be a temporary
        return self

    def copy_(self, src, stream=None):
        _lib.call("frcnn_memcpy_d2d", C.c_void_p(self.ptr), ptr(src), self.nbytes, stream_ptr(stream))
        return self

    def numpy(self, stream=None):
        out = np.empty(self.shape, dtype=self.dtype)
        s = stream_ptr(stream)
        if self.nbytes:
            _lib.call("frcnn_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.nbytes, s)
        _lib.call("frcnn_stream_sync", s)
        return out

    cpu = numpy

    def clone(self, stream=None):
        t = DeviceTensor.empty(self.shape, self.dtype)
        t.copy_(self, stream)
        return t


def to_device(x, dtype=np.float32):
    """x:cuda() -- accept numpy / torch CPU / torch CUDA / DeviceTensor, return something ptr() accepts."""
    if isinstance(x, DeviceTensor):
        return x
    if hasattr(x, "is_cuda"):
        if x.is_cuda:
            return x.contiguous()
        return DeviceTensor.from_numpy(x.detach().cpu().numpy().astype(dtype, copy=False))
    return DeviceTensor.from_numpy(np.asarray(x, dtype=dtype))


def shape_of(x):
    return tuple(x.shape)
