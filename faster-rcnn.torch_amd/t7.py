"""Torch7 object serialisation (`torch.DiskFile:writeObject / readObject`) for the snapshot files of the
reference: `save_model` writes `{version = 0, weights, options, stats}` with `save_obj`
(utilities.lua:113-134) and `graph_training` restores `weights` from it (main.lua:94-98).  `save_obj` opens the
DiskFile WITHOUT `:binary()`, so the reference's snapshots are in torch's ASCII mode; both modes are implemented.

The format lives in the torch7 package (File.lua `writeObject`/`readObject`, Tensor.c / Storage.c `write`,
DiskFile.c), which is not part of /root/reference: it is restated here from the published sources, pinned by
hand-derived byte strings (tests/test_t7.py).  PARITY UNPINNED in the sense of the task: no file written by a real
Torch7 was available to read back.

  object   := TYPE(int) payload
  TYPE     0 nil | 1 number (double) | 2 string (int length, raw chars) | 3 table | 4 torch object | 5 boolean (int)
  table    := index(int) [ count(int) { key-object value-object } ]      -- body only the first time an index appears
  torch    := index(int) [ "V 1" className  class-specific body ]        -- both strings as (int length, raw chars)
  Tensor   := nDim(int) size(long x nDim) stride(long x nDim) storageOffset(long, 1-based) storage-object
  Storage  := n(long) data(n elements)
ASCII mode: every scalar or array is printed with C formats (%d, %ld, %.9g float, %.17g double), array elements
separated by one blank, each write followed by a newline; raw chars are written as they are.
Lua tables map to dicts; tables whose keys are exactly 1..n are returned as lists (and lists are written that way)."""
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

_STORAGE = {"Float": np.float32, "Double": np.float64, "Long": np.int64, "Int": np.int32, "Short": np.int16,
            "Byte": np.uint8, "Char": np.int8}


def _cuda_kind(k):
    """'Cuda' -> 'Float', 'CudaLong' -> 'Long', ... (cutorch's device classes serialise like their host twins)."""
    if k == "Cuda":
        return "Float"
    return k[4:] if k.startswith("Cuda") else k


_BY_DTYPE = {np.dtype(v): k for k, v in _STORAGE.items()}
_BIN = {np.dtype(np.float32): "f", np.dtype(np.float64): "d", np.dtype(np.int64): "q", np.dtype(np.int32): "i",
        np.dtype(np.int16): "h", np.dtype(np.uint8): "B", np.dtype(np.int8): "b"}


def _fmt(v, dtype):
    if dtype == np.float32:
        return "%.9g" % float(v)
    if dtype == np.float64:
        return "%.17g" % float(v)
    return "%d" % int(v)


class Writer(object):
    def __init__(self, f, ascii=True):
        self.f, self.ascii = f, ascii
        self.seen = {}     # id(object) -> index
        self.keep = []     # (objects stay alive while their id() is a key)

    # ---- scalars / arrays
    def _array(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype).ravel()
        if self.ascii:
            if a.size > 1024 and a.dtype.kind == "f":   # (vectorised formatting for the weight vector)
                txt = np.char.mod("%.9g" if a.dtype == np.float32 else "%.17g", a.astype(np.float64))
                self.f.write((" ".join(txt.tolist()) + "\n").encode("latin-1"))
            elif a.size:
                self.f.write((" ".join(_fmt(v, np.dtype(dtype)) for v in a) + "\n").encode("latin-1"))
        else:
            self.f.write(a.astype(np.dtype(dtype).newbyteorder("<")).tobytes())

    def int(self, v): self._array([v], np.int32)
    def long(self, v): self._array([v], np.int64)
    def double(self, v): self._array([v], np.float64)

    def chars(self, s):
        b = s if isinstance(s, bytes) else s.encode("latin-1")
        self.f.write(b + (b"\n" if self.ascii and b else b""))   # (DiskFile adds the newline only after n > 0 elements)

    def string(self, s):
        b = s if isinstance(s, bytes) else s.encode("latin-1")
        self.int(len(b)); self.chars(b)

    # ---- objects
    def _index(self, obj):
        """-> (index, first_time)"""
        k = id(obj)
        if k in self.seen:
            return self.seen[k], False
        self.seen[k] = len(self.seen) + 1
        self.keep.append(obj)
        return self.seen[k], True

    def storage(self, a):
        a = np.ascontiguousarray(a).ravel()
        name = _BY_DTYPE[a.dtype]
        self.int(TYPE_TORCH)
        idx, first = self._index(a)
        self.int(idx)
        if first:
            self.string("V 1"); self.string("torch.%sStorage" % name)
            self.long(a.size); self._array(a, a.dtype)

    def tensor(self, a):
        name = _BY_DTYPE[a.dtype]
        self.int(TYPE_TORCH)
        idx, first = self._index(a)
        self.int(idx)
        if not first:
            return
        self.string("V 1"); self.string("torch.%sTensor" % name)
        c = np.ascontiguousarray(a)
        self.int(c.ndim)
        self._array(c.shape, np.int64)
        self._array([s // c.itemsize for s in c.strides], np.int64)
        self.long(1)
        if c.ndim == 0 or c.size == 0:
            self.int(TYPE_NIL)
        else:
            self.storage(c.ravel())

    def object(self, o):
        if o is None:
            self.int(TYPE_NIL)
        elif isinstance(o, (bool, np.bool_)):
            self.int(TYPE_BOOLEAN); self.int(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.int(TYPE_NUMBER); self.double(float(o))
        elif isinstance(o, (str, bytes)):
            self.int(TYPE_STRING); self.string(o)
        elif isinstance(o, np.ndarray):
            self.tensor(o)
        elif type(o).__name__ == "Rect" and hasattr(o, "minX") or isinstance(o, T7Object):
            cls = "Rect" if not isinstance(o, T7Object) else o.torch_class
            fields = dict(o) if isinstance(o, T7Object) else dict(minX=o.minX, minY=o.minY, maxX=o.maxX, maxY=o.maxY)
            self.int(TYPE_TORCH)
            idx, first = self._index(o)
            self.int(idx)
            if first:
                self.string("V 1"); self.string(cls)
                self.keep.append(fields)
                self.object(fields)
        elif isinstance(o, (dict, list, tuple)):
            self.int(TYPE_TABLE)
            idx, first = self._index(o)
            self.int(idx)
            if first:
                items = list(o.items()) if isinstance(o, dict) else [(i + 1, v) for i, v in enumerate(o)]
                self.int(len(items))
                for k, v in items:
                    self.object(k); self.object(v)
        elif hasattr(o, "cpu") or hasattr(o, "numpy"):   # torch tensor / DeviceTensor: a FloatTensor on disk
            a = o.detach().cpu().numpy() if hasattr(o, "detach") else o.numpy()
            self.tensor(np.ascontiguousarray(a))
        else:
            raise TypeError("t7: cannot serialise %r" % type(o))


class Reader(object):
    def __init__(self, f, ascii=True):
        self.f, self.ascii = f, ascii
        self.objects = {}

    def _tokens(self, n):
        """n whitespace-delimited ASCII tokens.  Every write group of the format ends with a newline, so whole lines
        are consumed (a 27 M element storage is one line); surplus tokens of a line are kept for the next call."""
        out = self._left if hasattr(self, "_left") else []
        while len(out) < n:
            line = self.f.readline()
            if not line:
                raise EOFError("t7: unexpected end of file")
            out = out + line.split()
        self._left = out[n:]
        return out[:n]

    def _array(self, n, dtype):
        dtype = np.dtype(dtype)
        if n == 0:
            return np.zeros(0, dtype)
        if self.ascii:
            toks = self._tokens(n)
            if dtype.kind == "f":
                return np.array(toks, dtype="S").astype(np.float64).astype(dtype)
            return np.array([int(t) for t in toks], dtype=dtype)
        raw = self.f.read(n * dtype.itemsize)
        if len(raw) != n * dtype.itemsize:
            raise EOFError("t7: unexpected end of file")
        return np.frombuffer(raw, dtype=dtype.newbyteorder("<")).astype(dtype)

    def int(self): return int(self._array(1, np.int32)[0])
    def long(self): return int(self._array(1, np.int64)[0])
    def double(self): return float(self._array(1, np.float64)[0])

    def chars(self, n):
        if getattr(self, "_left", None):
            raise ValueError("t7: raw characters requested in the middle of a number line")
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError("t7: unexpected end of file")
        if self.ascii and n:
            nl = self.f.read(1)
            if nl not in (b"\n", b""):
                self.f.seek(-1, 1)
        return b

    def string(self):
        return self.chars(self.int()).decode("latin-1")

    def object(self):
        t = self.int()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = self.double()
            return int(v) if v == int(v) and abs(v) < 2 ** 53 else v
        if t == TYPE_BOOLEAN:
            return self.int() != 0
        if t == TYPE_STRING:
            return self.string()
        if t == TYPE_TABLE:
            idx = self.int()
            if idx in self.objects:
                return self.objects[idx]
            d = {}
            self.objects[idx] = d
            for _ in range(self.int()):
                k = self.object(); d[k] = self.object()
            n = len(d)
            if n and all(isinstance(k, int) for k in d) and sorted(d) == list(range(1, n + 1)):
                lst = [d[i] for i in range(1, n + 1)]
                self.objects[idx] = lst
                return lst
            return d
        if t == TYPE_TORCH:
            idx = self.int()
            if idx in self.objects:
                return self.objects[idx]
            version = self.string()
            cls = self.string() if version.startswith("V ") else version
            kind = cls.split(".")[-1]
            if kind.endswith("Storage") and _cuda_kind(kind[:-7]) in _STORAGE:
                # (cutorch writes a CudaStorage as its size followed by the float data -- the reference's snapshots
                # hold `weights` as a torch.CudaTensor over a torch.CudaStorage, main.lua:86-92 / utilities.lua:126-134)
                n = self.long()
                a = self._array(n, _STORAGE[_cuda_kind(kind[:-7])])
            elif kind.endswith("Tensor") and _cuda_kind(kind[:-6]) in _STORAGE:
                nd = self.int()
                size = self._array(nd, np.int64); stride = self._array(nd, np.int64)
                off = self.long() - 1
                st = self.object()
                if st is None or nd == 0:
                    a = np.zeros(tuple(size) if nd else (0,), _STORAGE[_cuda_kind(kind[:-6])])
                else:
                    a = np.lib.stride_tricks.as_strided(st[off:], shape=tuple(int(s) for s in size),
                                                        strides=tuple(int(s) * st.itemsize for s in stride)).copy()
            elif cls.startswith("torch.") or cls.startswith("nn.") or cls.startswith("cudnn."):
                raise ValueError("t7: unsupported class '%s'" % cls)
            else:
                # a torch.class object without a write() method (e.g. the reference's `Rect`, Rect.lua:5): File.lua
                # serialises the object's fields as one table
                fields = self.object()
                a = _make_object(cls, fields)
            self.objects[idx] = a
            return a
        raise ValueError("t7: unknown type tag %d" % t)


class T7Object(dict):
    """Fields of a torch.class object of an unknown class (`.torch_class` holds its name)."""
    torch_class = None


def _make_object(cls, fields):
    if cls == "Rect" and isinstance(fields, dict) and all(k in fields for k in ("minX", "minY", "maxX", "maxY")):
        from .Rect import Rect
        return Rect(fields["minX"], fields["minY"], fields["maxX"], fields["maxY"])
    o = T7Object(fields if isinstance(fields, dict) else {"value": fields})
    o.torch_class = cls
    return o


def save_obj(file_name, obj, ascii=True):  # utilities.lua:113-117 (ASCII is DiskFile's default mode)
    with open(file_name, "wb") as f:
        Writer(f, ascii).object(obj)


def load_obj(file_name, ascii=True):  # utilities.lua:119-124
    with open(file_name, "rb") as f:
        return Reader(f, ascii).object()


def save_model(file_name, weights, options, stats, ascii=True):  # utilities.lua:126-134
    save_obj(file_name, dict(version=0, weights=weights, options=options, stats=stats), ascii)


def restore_weights(file_name, weights, ascii=True):
    """main.lua:94-98: `local stored = load_obj(opt.restore); weights:copy(stored.weights)` -> the stored table."""
    stored = load_obj(file_name, ascii)
    w = np.asarray(stored["weights"], dtype=np.float32).ravel()
    n = weights.numel() if hasattr(weights, "numel") else weights.size
    if w.size != n:
        raise ValueError("restore: the snapshot holds %d weights, the model has %d" % (w.size, n))
    if hasattr(weights, "copy_from_numpy"):
        weights.copy_from_numpy(w.reshape(weights.shape))
    elif hasattr(weights, "copy_"):
        import torch
        weights.copy_(torch.from_numpy(w).view(weights.shape))
    else:
        weights[...] = w.reshape(weights.shape)
    return stored
