"""create_proposal_net / create_classification_net / create_model -- host-side mirror of
models/model_utilities.lua:3-136.  The two returned objects satisfy the nn.Module call sites of
the reference (forward / backward / training / evaluate / parameters / cuda, and the
`outnode.children[i]` introspection Localizer needs) and run on the native model runtime of
libfrcnn_hip.so (frcnn_pnet_* / frcnn_cnet_*)."""
import ctypes as C

import numpy as np

from . import _lib
from .tensor import DeviceTensor, ptr, stream_ptr, to_device


class _Node(object):
    """Stand-in for an nngraph node: carries the conv/pool geometry list of the path from the input
    to this output (what Localizer.lua:8-36 extracts by walking node.children[1])."""

    def __init__(self, layers):
        self.layers = layers


class _OutNode(object):
    def __init__(self, children):
        self.children = children  # 0-based python list: children[i-1] is output i of the Lua graph


class NativeModel(object):
    """Owns the frcnn_model handle shared by pnet and cnet."""

    def __init__(self, cfg, layers, anchor_nets, class_layers):
        d = _lib.ModelDesc()
        d.nblocks = len(layers)
        for i, l in enumerate(layers):
            assert l["kW"] == l["kH"] and l["padW"] == l["padH"], "square kernels only"
            d.filters[i] = l["filters"]; d.ksize[i] = l["kW"]; d.pad[i] = l["padW"]
            d.conv_steps[i] = l["conv_steps"]; d.dropout[i] = float(l.get("dropout") or 0.0)
        d.nheads = len(anchor_nets)
        for i, a in enumerate(anchor_nets):
            d.head_k[i] = a["kW"]; d.head_n[i] = a["n"]; d.head_input[i] = a["input"]
        d.ncls = len(class_layers)
        for i, l in enumerate(class_layers):
            d.cls_n[i] = l["n"]; d.cls_bn[i] = 1 if l.get("batch_norm") else 0
            d.cls_dropout[i] = float(l.get("dropout") or 0.0)
        d.class_count = cfg["class_count"]
        d.kh = cfg["roi_pooling"]["kh"]; d.kw = cfg["roi_pooling"]["kw"]
        self.desc = d
        h = C.c_void_p()
        _lib.call("frcnn_model_create", C.byref(d), C.byref(h))
        self.h = h
        tot = C.c_longlong(); pn = C.c_longlong()
        _lib.call("frcnn_model_param_count", self.h, C.byref(tot), C.byref(pn))
        self.total_params, self.pnet_params = tot.value, pn.value
        tab = np.zeros((256, 4), dtype=np.int64); n = C.c_int()
        _lib.call("frcnn_model_param_table", self.h, tab.ctypes.data_as(C.c_void_p), 256, C.byref(n))
        self.param_table = tab[:n.value].copy()
        self.weights = None   # flat device vectors, bound by combine_and_flatten_parameters
        self.gradient = None
        self.bn_running = None
        self.seed = 1

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _lib.load().frcnn_model_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def localizer_layers(self, output_index):
        buf = np.zeros((64, 6), dtype=np.int32); n = C.c_int()
        _lib.call("frcnn_model_localizer_layers", self.h, output_index, buf.ctypes.data_as(C.c_void_p), 64, C.byref(n))
        return buf[:n.value].copy()

    def init_parameters(self, seed=42):
        """Host-side initial values in flat order: conv weights N(0, sqrt(2/(kW*kH*nOut))), conv bias 0
        (model_utilities.lua:60-71); PReLU 0.25; Linear U(+-1/sqrt(fan_in)); BatchNorm weight U(0,1),
        bias 0 (Torch defaults of the period, [ext])."""
        rng = np.random.RandomState(seed)
        w = np.zeros(self.total_params, dtype=np.float32)
        for off, cnt, kind, aux in self.param_table:
            if kind == 0:
                w[off:off + cnt] = rng.normal(0.0, np.sqrt(2.0 / aux), cnt).astype(np.float32)
            elif kind == 2:
                w[off:off + cnt] = 0.25
            elif kind in (3, 4):
                s = 1.0 / np.sqrt(aux)
                w[off:off + cnt] = rng.uniform(-s, s, cnt).astype(np.float32)
            elif kind == 5:
                w[off:off + cnt] = rng.uniform(0.0, 1.0, cnt).astype(np.float32)
        return w

    def bn_running_size(self):
        return int(sum(2 * self.desc.cls_n[i] for i in range(self.desc.ncls) if self.desc.cls_bn[i]))


def _mask_ptrs(masks, n):
    """host array of n device pointers (or NULL) for the dropout-mask arguments."""
    if masks is None:
        return None, None
    keep = [to_device(m) if m is not None else None for m in masks]
    arr = (C.c_void_p * n)()
    for i in range(n):
        arr[i] = ptr(keep[i]).value if i < len(keep) and keep[i] is not None else None
    return arr, keep


class ProposalNet(object):
    """pnet: VGG-style conv blocks + anchor heads; forward(img) -> 5 outputs (4 head maps of 18 planes +
    the last feature map), model_utilities.lua:3-74."""

    def __init__(self, native):
        self.native = native
        self.train = True
        n_out = native.desc.nheads + 1
        self.outnode = _OutNode([_Node(native.localizer_layers(i + 1)) for i in range(n_out)])
        self.drop_masks = None   # optional explicit SpatialDropout keep masks (parity runs)
        self.output = None

    def cuda(self):
        return self

    def training(self):
        self.train = True

    def evaluate(self):
        self.train = False

    def forward(self, img, async_heads=False):
        """pnet:forward(img).  async_heads (training only, used by the objective): the anchor nets stay in flight
        on the library's side stream -- only the last output is final in stream order until anchor_loss_begin /
        backward have been called (frcnn_pnet_forward_async_heads)."""
        nat = self.native
        if nat.weights is None:
            raise _lib.FrcnnError("pnet:forward before combine_and_flatten_parameters()")
        img = to_device(img)
        assert len(img.shape) == 3 and img.shape[0] == 3, "expected a 3xHxW image"
        _, H, W = img.shape
        arr, keep = _mask_ptrs(self.drop_masks, nat.desc.nblocks)
        nat.seed += 1
        if async_heads and self.train:
            _lib.call("frcnn_pnet_forward_async_heads", nat.h, ptr(nat.weights), ptr(img), H, W,
                      C.cast(arr, C.c_void_p) if arr is not None else None, nat.seed, stream_ptr())
        else:
            _lib.call("frcnn_pnet_forward", nat.h, ptr(nat.weights), ptr(img), H, W, 1 if self.train else 0,
                      C.cast(arr, C.c_void_p) if arr is not None else None, nat.seed, stream_ptr())
        outs = []
        for i in range(1, nat.desc.nheads + 2):
            p = C.c_void_p(); c = C.c_int(); h = C.c_int(); w = C.c_int()
            _lib.call("frcnn_pnet_output", nat.h, i, C.byref(p), C.byref(c), C.byref(h), C.byref(w))
            outs.append(DeviceTensor(p.value, (c.value, h.value, w.value), np.float32, owner=nat))
        self.output = outs
        return outs

    def delta_outputs(self, zero=True):
        """The gradient buffers matching forward()'s outputs (objective.lua:78-84)."""
        nat = self.native
        if zero:
            _lib.call("frcnn_pnet_zero_deltas", nat.h, stream_ptr())
        res = []
        for i, o in enumerate(self.output):
            p = C.c_void_p()
            _lib.call("frcnn_pnet_delta", nat.h, i + 1, C.byref(p))
            res.append(DeviceTensor(p.value, o.shape, np.float32, owner=nat))
        return res

    def backward(self, img, delta_outputs):
        """pnet:backward(img, delta_outputs): accumulates into the flat gradient (objective.lua:189)."""
        nat = self.native
        own = self.delta_outputs(zero=False)
        for mine, given in zip(own, delta_outputs):
            if ptr(given).value != mine.ptr:
                mine.copy_(given)
        _lib.call("frcnn_pnet_backward", nat.h, ptr(nat.weights), ptr(nat.gradient), stream_ptr())
        return None  # gradInput of the first convolution is unused by the reference and not computed

    def backward_heads_begin(self):
        """Optional early start of :backward: the anchor nets' part runs on the library's side stream as soon
        as delta_outputs[1..nheads] are final (they are, after objective.lua:140), beside the cnet stage."""
        nat = self.native
        _lib.call("frcnn_pnet_backward_heads_begin", nat.h, ptr(nat.weights), ptr(nat.gradient), stream_ptr())

    def anchor_loss_begin(self, d_idx, d_anchor, d_roi, d_class, npos, nneg, bgclass, ex_loss, crtarget, cctarget, acc):
        """objective.lua:91-140 on this net's outputs[1..nheads] / delta_outputs[1..nheads], then the anchor nets'
        part of :backward, on the library's side stream (frcnn_pnet_anchor_loss_begin).  d_* are device addresses
        of the example tables (layout of frcnn_rpn_loss)."""
        nat = self.native
        _lib.call("frcnn_pnet_anchor_loss_begin", nat.h, ptr(nat.weights), ptr(nat.gradient), C.c_void_p(d_idx),
                  C.c_void_p(d_anchor), C.c_void_p(d_roi), C.c_void_p(d_class), npos, nneg, bgclass, ptr(ex_loss),
                  ptr(crtarget), ptr(cctarget), ptr(acc), stream_ptr())

    def anchor_loss_wait(self):
        """The caller's stream waits for the losses / cnet targets of anchor_loss_begin()."""
        _lib.call("frcnn_pnet_anchor_loss_wait", self.native.h, stream_ptr())

    def backward_heads_join(self):
        """The caller's stream waits for backward_heads_begin()'s work.  True: the anchor nets' gradient slice is
        final in stream order; False: nothing had been started (backward() will compute it)."""
        joined = C.c_int(0)
        _lib.call("frcnn_pnet_backward_heads_join", self.native.h, stream_ptr(), C.byref(joined))
        return bool(joined.value)

    def block_param_range(self, b):
        """[lo, hi) of backbone block b's parameters (0-based b) in the flat vector: three tensors per convolution."""
        d = self.native.desc
        first = sum(int(d.conv_steps[i]) for i in range(b))
        n = int(d.conv_steps[b])
        t = self.native.param_table
        last = t[3 * (first + n) - 1]
        return int(t[3 * first][0]), int(last[0]) + int(last[1])

    def wait_block_gradients(self, b):
        """The CURRENT stream waits until block b's (0-based) gradients of the queued backward pass are final."""
        _lib.call("frcnn_pnet_wait_block_gradients", self.native.h, b + 1, stream_ptr())

    def heads_param_range(self):
        """[lo, hi) of the anchor nets' parameters in the flat vector (they follow the backbone convolutions)."""
        d = self.native.desc
        nconv = sum(int(d.conv_steps[b]) for b in range(d.nblocks))
        return int(self.native.param_table[3 * nconv][0]), int(self.native.pnet_params)

    def parameters(self):
        return _param_views(self.native, 0, self.native.pnet_params)


class ClassificationNet(object):
    """cnet: Linear(+BN)+PReLU+Dropout stack with a bbox head and a LogSoftMax class head,
    model_utilities.lua:76-124.  forward(R x D) -> [R x 4, R x (classes+1)]."""

    def __init__(self, native):
        self.native = native
        self.train = True
        self.drop_masks = None
        self.output = None
        self._bufs = {}
        self._pending = None

    def _buf(self, name, shape):
        """Module-owned output / gradInput buffers, reused by the next call like nn.Module's (hipMalloc /
        hipFree inside a step would synchronise the device)."""
        need = int(np.prod(shape)) * 4
        b = self._bufs.get(name)
        if b is None or b.nbytes < need:
            b = DeviceTensor.empty((max(need, 256),), np.uint8)
            self._bufs[name] = b
        return DeviceTensor(b.ptr, shape, np.float32, owner=b)

    def cuda(self):
        return self

    def training(self):
        self.train = True

    def evaluate(self):
        self.train = False

    def forward(self, cinput):
        nat = self.native
        cinput = to_device(cinput)
        R, D = cinput.shape
        nc = nat.desc.class_count + 1
        bbox = self._buf("bbox", (R, 4)); cls = self._buf("cls", (R, nc))
        arr, keep = _mask_ptrs(self.drop_masks, nat.desc.ncls)
        nat.seed += 1
        self._input = cinput
        _lib.call("frcnn_cnet_forward", nat.h, ptr(nat.weights), ptr(cinput), R, 1 if self.train else 0,
                  C.cast(arr, C.c_void_p) if arr is not None else None, nat.seed, ptr(nat.bn_running), ptr(bbox),
                  ptr(cls), stream_ptr())
        self.output = [bbox, cls]
        return self.output

    def backward(self, cinput, grad_outputs):
        nat = self.native
        R, D = self._input.shape
        gx = self._buf("gx", (R, D))
        _lib.call("frcnn_cnet_backward", nat.h, ptr(nat.weights), ptr(grad_outputs[0]), ptr(grad_outputs[1]), ptr(gx),
                  ptr(nat.gradient), stream_ptr())
        # the library's weight-gradient stream reads these until the join (include/frcnn_hip.h, LIFETIME): keep them
        # alive -- and out of the caching allocator's hands -- until the next forward / join replaces the references
        self._pending = (self._input, grad_outputs[0], grad_outputs[1])
        return gx

    def join_backward(self):
        """frcnn_cnet_backward_join on the current stream; the buffers held for the asynchronous part are released."""
        _lib.call("frcnn_cnet_backward_join", self.native.h, stream_ptr())
        self._pending = None

    def parameters(self):
        return _param_views(self.native, self.native.pnet_params, self.native.total_params)


def _param_views(native, lo, hi):
    if native.weights is None:
        raise _lib.FrcnnError("parameters() before combine_and_flatten_parameters()")
    ws, gs = [], []
    for off, cnt, kind, aux in native.param_table:
        if lo <= off < hi:
            ws.append(native.weights[off:off + cnt])
            gs.append(native.gradient[off:off + cnt])
    return ws, gs


def create_model(cfg, layers, anchor_nets, class_layers):  # model_utilities.lua:126-136
    native = NativeModel(cfg, layers, anchor_nets, class_layers)
    return dict(cfg=cfg, layers=layers, anchor_nets=anchor_nets, class_layers=class_layers,
                pnet=ProposalNet(native), cnet=ClassificationNet(native), native=native)
