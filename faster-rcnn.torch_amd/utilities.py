"""combine_and_flatten_parameters (utilities.lua:136-147) and the optimiser step of
main.lua:122,133 (optim.rmsprop).  The flat weight / gradient vectors are torch CUDA tensors so that
`torch.distributed` (RCCL) can all-reduce the gradient in place."""
import numpy as np

from . import _lib
from .tensor import ptr, stream_ptr


def combine_and_flatten_parameters(pnet, cnet, seed=42, weights_host=None):
    """-> weights, gradient: ONE contiguous device vector each (pnet parameters first, then cnet).
    Both nets are re-pointed at the flat storage (nn.Module.flatten semantics)."""
    import torch
    native = pnet.native
    assert cnet.native is native
    if not torch.cuda.is_available():
        raise _lib.FrcnnError("no HIP device: the product path has no CPU fallback")
    w0 = native.init_parameters(seed) if weights_host is None else np.asarray(weights_host, dtype=np.float32)
    assert w0.size == native.total_params
    weights = torch.from_numpy(w0).cuda()
    gradient = torch.zeros_like(weights)
    native.weights, native.gradient = weights, gradient
    nbn = native.bn_running_size()
    if nbn:
        bn = np.zeros(nbn, dtype=np.float32)
        o = 0
        for i in range(native.desc.ncls):
            if native.desc.cls_bn[i]:
                n = native.desc.cls_n[i]
                bn[o + n:o + 2 * n] = 1.0  # running_mean 0, running_var 1
                o += 2 * n
        native.bn_running = torch.from_numpy(bn).cuda()
    return weights, gradient


import os
EAGER_DEFAULT = os.environ.get("FRCNN_EAGER_UPDATE", "0") == "1"


def rmsprop(opfunc, x, state):
    """optim.rmsprop(opfunc, x, state) [ext]: state.learningRate (1e-2), state.alpha (0.99),
    state.epsilon (1e-8); m = alpha*m + (1-alpha)*g^2 ; x -= lr * g / (sqrt(m) + eps).
    Returns x, [f(x)] like the Lua function (main.lua:133)."""
    import torch
    lr = state.get("learningRate", 1e-2); alpha = state.get("alpha", 0.99); eps = state.get("epsilon", 1e-8)
    if "m" not in state:
        state["m"] = torch.zeros_like(x)
    begin = getattr(opfunc, "begin_fold", None)
    timing = state.get("_timing")     # bench.py: seconds the host spends queueing a step / waiting for its statistics
    if timing is not None:
        import time
        t_in = time.perf_counter()
    if begin is not None:   # create_objective's closure: the loss is read back AFTER the update has been queued,
        # and gradient:div(n) rides on the update's own pass over the vectors.  state["eager"] (default off: measured neutral on
        # one GPU, EXPERIMENTS.md round 6 -- the backward pass has no idle registers for the update to run in): the pass may
        # apply this very step slice by slice, beside its own backward half, as slices of the gradient become final
        eager = dict(m=state["m"], lr=lr, alpha=alpha, eps=eps) if state.get("eager", EAGER_DEFAULT) else None
        finish, dfdx, gscale = begin(x, eager) if eager is not None else begin(x)
        if eager is not None and "done" in eager:
            # what the pass has not updated (the shallowest block: its gradients end the pass; everything, for an image without
            # examples), on the caller's stream, followed by the packs made from it
            done = sorted(eager["done"])
            rest, at = [], 0
            for lo, hi in done:
                if lo > at:
                    rest.append((at, lo))
                at = max(at, hi)
            if at < x.numel():
                rest.append((at, x.numel()))
            for lo, hi in rest:
                eager["slice"](lo, hi, stream_ptr())
            eager["complete"]()
        elif hasattr(gscale, "ptr"):   # data parallel: the divisor is the all-reduced count, still on the device
            _lib.call("frcnn_scale_rmsprop_dev", ptr(x), ptr(dfdx), ptr(gscale.ptr), ptr(state["m"]), x.numel(), lr, alpha,
                      eps, stream_ptr())
        elif gscale is None:
            _lib.call("frcnn_rmsprop", ptr(x), ptr(dfdx), ptr(state["m"]), x.numel(), lr, alpha, eps, stream_ptr())
        else:
            _lib.call("frcnn_scale_rmsprop", ptr(x), ptr(dfdx), gscale, ptr(state["m"]), x.numel(), lr, alpha, eps,
                      stream_ptr())
        if timing is not None:
            t_q = time.perf_counter()
        fx, _ = finish()
        if timing is not None:
            timing["enqueue"] += t_q - t_in; timing["wait"] += time.perf_counter() - t_q; timing["steps"] += 1
        return x, [fx]
    fx, dfdx = opfunc(x)
    _lib.call("frcnn_rmsprop", ptr(x), ptr(dfdx), ptr(state["m"]), x.numel(), lr, alpha, eps, stream_ptr())
    return x, [fx]


def reverse(array):  # utilities.lua:79-87
    array.reverse()
    return array
