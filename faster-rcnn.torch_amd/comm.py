"""Data-parallel exchange through the library's own communicator (include/frcnn_hip.h: frcnn_comm_*,
frcnn_allreduce_f32 / _f64, frcnn_broadcast_f32 -- RCCL over xGMI, one process per GPU), i.e. the calls a LuaJIT
host makes (bindings/objective_hip.lua).  `Comm` is the host-side handle; `activate(comm)` makes create_objective
use it for the step's all-reduces instead of torch.distributed (same bucket schedule: an all-reduce is queued on a
stream of the communicator's own behind an event of the caller's stream and joined by an event again, so it runs
beside the rest of the backward pass).

Not in the reference (single process, single device: main.lua:52); SURVEY 8e."""
import ctypes as C
import os

from . import _lib
from .tensor import ptr, stream_ptr

_active = None


class _Work(object):
    def __init__(self, event):
        self.event = event

    def wait(self):
        """The CURRENT stream waits for the collective (the host does not block), like a torch NCCL work."""
        import torch
        torch.cuda.current_stream().wait_event(self.event)
        return True


class Comm(object):
    def __init__(self, rank, world_size, path=None, id_bytes=None, timeout_ms=120000):
        """Collective: every rank constructs its Comm (after frcnn_set_device / torch.cuda.set_device).  Rendezvous
        through `path` (a file every rank can see; rank 0 creates it) or an id obtained from Comm.unique_id()."""
        import torch
        self.rank, self.world_size = int(rank), int(world_size)
        h = C.c_void_p()
        if id_bytes is not None:
            buf = C.create_string_buffer(bytes(id_bytes), 128)
            _lib.call("frcnn_comm_init_rank_timeout", C.byref(h), self.world_size, self.rank, buf, int(timeout_ms))
        else:
            if not path:
                raise _lib.FrcnnError("Comm needs a rendezvous path or an id")
            _lib.call("frcnn_comm_init_rank_file", C.byref(h), self.world_size, self.rank, path.encode(), int(timeout_ms))
        self.h = h
        self.path = path
        self.stream = torch.cuda.Stream()   # collectives run here, beside the caller's kernels

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _lib.call("frcnn_comm_get_unique_id", buf)
        return buf.raw

    @staticmethod
    def from_env(timeout_ms=120000):
        """RANK / WORLD_SIZE / MASTER_PORT as set by torch.distributed.run (or any launcher).  The rendezvous file is
        named after the port and the launcher's pid, and the job nonce the library checks in it (FRCNN_COMM_NONCE, see
        frcnn_comm_exchange_id_file) defaults to the same pair plus the launcher's run id: an id file that an earlier
        job left behind is neither named like this job's nor accepted if it is."""
        rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
        os.environ.setdefault("FRCNN_COMM_NONCE", "%s:%s:%d:%s" % (
            os.environ.get("MASTER_ADDR", ""), os.environ.get("MASTER_PORT", "0"), os.getppid(),
            os.environ.get("TORCHELASTIC_RUN_ID", "")))
        path = os.environ.get("FRCNN_COMM_FILE") or os.path.join(
            os.environ.get("TMPDIR", "/tmp"), "frcnn_comm_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
        return Comm(rank, world, path=path, timeout_ms=timeout_ms)

    def get_world_size(self):
        return self.world_size

    def query(self):
        """(count, user rank, device) as RCCL reports them for this communicator (ncclCommCount / UserRank / CuDevice)."""
        n, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.call("frcnn_comm_query", self.h, C.byref(n), C.byref(r), C.byref(d))
        return n.value, r.value, d.value

    def gather_ints(self, value):
        """[value of rank 0, ..., value of rank W-1] on every rank (one-hot slots summed by the all-reduce)."""
        import torch
        t = torch.zeros(self.world_size, dtype=torch.float64, device="cuda")
        t[self.rank] = float(value)
        self.all_reduce(t)
        return [int(v) for v in t.cpu().tolist()]

    def _call(self, t, fn, *extra):
        import torch
        if not getattr(t, "is_cuda", False):
            raise _lib.FrcnnError("the native communicator reduces device tensors only")
        if not t.is_contiguous():
            raise _lib.FrcnnError("all_reduce needs a contiguous tensor (a slice of the flat vector is)")
        timed = getattr(self, "timing", None) is not None     # bench.py: per-bucket durations on the communicator's stream
        done = torch.cuda.Event(enable_timing=timed)
        ready = torch.cuda.Event()
        ready.record()                        # the operand is final where the caller's stream stands now
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            if timed:
                start = torch.cuda.Event(enable_timing=True)
                start.record()
            fn(t, stream_ptr(), *extra)
            done.record()
        if timed:
            self.timing.append((t.numel() * t.element_size(), start, done))
        t.record_stream(self.stream)
        return _Work(done)

    def bucket_times(self):
        """[(bytes, launches, mean ms)] of the collectives recorded since `self.timing = []` (after a device synchronize):
        what each bucket of the exchange step costs on the communicator's stream, beside the kernels of the step."""
        import torch
        torch.cuda.synchronize()
        acc = {}
        for nbytes, a, b in (getattr(self, "timing", None) or []):
            n, ms = acc.get(nbytes, (0, 0.0))
            acc[nbytes] = (n + 1, ms + a.elapsed_time(b))
        return [dict(bytes=k, launches=n, mean_ms=round(ms / n, 4)) for k, (n, ms) in sorted(acc.items(), reverse=True)]

    def all_reduce(self, t, async_op=False, group=None):
        """In-place sum over the ranks of a float32 / float64 device tensor."""
        import torch
        if t.dtype == torch.float32:
            name = "frcnn_allreduce_f32"
        elif t.dtype == torch.float64:
            name = "frcnn_allreduce_f64"
        else:
            raise _lib.FrcnnError("all_reduce: float32 / float64 only (got %s)" % t.dtype)
        w = self._call(t, lambda x, s: _lib.call(name, self.h, ptr(x), x.numel(), s))
        if not async_op:
            w.wait()
        return w

    def broadcast(self, t, root=0):
        """main.lua:92-98 under data parallelism: every replica starts from rank `root`'s flat weights."""
        import torch
        if t.dtype != torch.float32:   # frcnn_broadcast_f32 counts 4-byte elements
            raise _lib.FrcnnError("broadcast: float32 only (got %s)" % t.dtype)
        w = self._call(t, lambda x, s: _lib.call("frcnn_broadcast_f32", self.h, ptr(x), x.numel(), int(root), s))
        w.wait()
        return t

    def barrier(self):
        import torch
        t = torch.zeros(1, dtype=torch.float32, device="cuda")
        self.all_reduce(t)
        torch.cuda.synchronize()

    def gather_max(self, value):
        """max over the ranks of a host number (sum of one-hot slots, then max): bench.py's max-over-ranks time."""
        import torch
        t = torch.zeros(self.world_size, dtype=torch.float64, device="cuda")
        t[self.rank] = float(value)
        self.all_reduce(t)
        return float(t.max().item())

    def destroy(self):
        global _active
        if _active is self:
            _active = None
        if getattr(self, "h", None):
            import torch
            torch.cuda.synchronize()
            _lib.call("frcnn_comm_destroy", self.h)
            self.h = None
            if self.rank == 0 and self.path:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass


def activate(comm):
    """create_objective() uses `comm` for the exchange step from now on (None: back to torch.distributed / none)."""
    global _active
    _active = comm


def active():
    """The communicator the objective should use, if it spans more than one rank -- or any, when FRCNN_COMM_FORCE=1
    (exercises the exchange path on a single GPU: every all-reduce is then an identity)."""
    c = _active
    if c is not None and (c.world_size > 1 or os.environ.get("FRCNN_COMM_FORCE") == "1"):
        return c
    return None
