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
        finish, dfdx, gscale = begin(x)   # and gradient:div(n) rides on the update's own pass over the vectors
        if hasattr(gscale, "ptr"):   # data parallel: the divisor is the all-reduced count, still on the device
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
