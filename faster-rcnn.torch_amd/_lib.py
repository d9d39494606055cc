"""ctypes binding of libfrcnn_hip.so (include/frcnn_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, an
exception is raised (the Lua surface would `error()`)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("FRCNN_LIB_PATH") or os.path.join(_HERE, "libfrcnn_hip.so")   # (FRCNN_LIB_PATH: A/B builds of the library)

KC_NAMES = ["conv_igemm_k3", "conv_igemm_other", "conv_wgrad_k3", "conv_wgrad_other", "gemm", "elemwise",
            "roi", "rpn", "nms", "optim", "image", "conv_x3", "conv_wgradx"]


class FrcnnError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [
        ("nblocks", C.c_int),
        ("filters", C.c_int * 8), ("ksize", C.c_int * 8), ("pad", C.c_int * 8), ("conv_steps", C.c_int * 8),
        ("dropout", C.c_float * 8),
        ("nheads", C.c_int),
        ("head_k", C.c_int * 8), ("head_n", C.c_int * 8), ("head_input", C.c_int * 8),
        ("ncls", C.c_int),
        ("cls_n", C.c_int * 8), ("cls_bn", C.c_int * 8),
        ("cls_dropout", C.c_float * 8),
        ("class_count", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int),
    ]


vp = C.c_void_p
_SIGS = {
    "frcnn_version": ([], C.c_int),
    "frcnn_set_option": ([C.c_char_p, C.c_int], C.c_int),
    "frcnn_get_option": ([C.c_char_p, C.POINTER(C.c_int)], C.c_int),
    "frcnn_last_error": ([], C.c_char_p),
    "frcnn_device_count": ([C.POINTER(C.c_int)], C.c_int),
    "frcnn_set_device": ([C.c_int], C.c_int),
    "frcnn_device_name": ([C.c_char_p, C.c_int], C.c_int),
    "frcnn_malloc": ([C.POINTER(vp), C.c_size_t], C.c_int),
    "frcnn_free": ([vp], C.c_int),
    "frcnn_host_alloc": ([C.POINTER(vp), C.c_size_t], C.c_int),
    "frcnn_host_free": ([vp], C.c_int),
    "frcnn_memcpy_h2d": ([vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_memcpy_d2h": ([vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_memcpy_d2d": ([vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_stream_sync": ([vp], C.c_int),
    "frcnn_zero": ([vp, C.c_size_t, vp], C.c_int),
    "frcnn_scale": ([vp, C.c_longlong, C.c_float, vp], C.c_int),
    "frcnn_add": ([vp, vp, C.c_longlong, vp], C.c_int),
    "frcnn_prof_enable": ([C.c_int], C.c_int),
    "frcnn_prof_collect": ([vp, vp, vp, vp], C.c_int),
    "frcnn_nms_workspace_bytes": ([C.c_int], C.c_size_t),
    "frcnn_nms_device": ([vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_nms_device_classes": ([vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_nms_device_n": ([vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_roi_windows": ([vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp], C.c_int),
    "frcnn_detect_post": ([vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_detect_gather": ([vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_nms_host": ([vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp], C.c_int),
    "frcnn_conv2d_forward": ([vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp], C.c_int),
    "frcnn_conv2d_backward_input": ([vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp], C.c_int),
    "frcnn_conv2d_backward_weight": ([vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp], C.c_int),
    "frcnn_maxpool_act_forward": ([vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_maxpool_act_backward": ([vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_act_forward": ([vp, C.c_int, C.c_longlong, vp, vp, vp, vp], C.c_int),
    "frcnn_act_backward": ([vp, vp, C.c_int, C.c_longlong, vp, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_roi_pool_forward": ([vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp], C.c_int),
    "frcnn_roi_pool_backward": ([vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp], C.c_int),
    "frcnn_rpn_scan_workspace_bytes": ([vp, vp], C.c_size_t),
    "frcnn_rpn_scan": ([vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_int, vp, vp, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "frcnn_rpn_loss": ([vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp], C.c_int),
    "frcnn_loss_accumulate": ([vp, C.c_int, vp, vp], C.c_int),
    "frcnn_linear_forward": ([vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp], C.c_int),
    "frcnn_linear_backward": ([vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp], C.c_int),
    "frcnn_rmsprop": ([vp, vp, vp, C.c_longlong, C.c_float, C.c_float, C.c_float, vp], C.c_int),
    "frcnn_scale_rmsprop": ([vp, vp, C.c_float, vp, C.c_longlong, C.c_float, C.c_float, C.c_float, vp], C.c_int),
    "frcnn_scale_rmsprop_dev": ([vp, vp, vp, vp, C.c_longlong, C.c_float, C.c_float, C.c_float, vp], C.c_int),
    "frcnn_scale_rmsprop_slice": ([vp, vp, C.c_float, vp, C.c_longlong, C.c_longlong, C.c_float, C.c_float, C.c_float, vp], C.c_int),
    "frcnn_model_update_stream": ([vp, C.POINTER(vp)], C.c_int),
    "frcnn_model_update_fork": ([vp, vp], C.c_int),
    "frcnn_model_update_join": ([vp, vp], C.c_int),
    "frcnn_pnet_wait_backward_begun": ([vp, vp], C.c_int),
    "frcnn_pnet_wait_heads_done": ([vp, vp], C.c_int),
    "frcnn_pnet_wait_block_done": ([vp, C.c_int, vp], C.c_int),
    "frcnn_pnet_refresh_packs": ([vp, vp, C.c_int, vp], C.c_int),
    "frcnn_pnet_invalidate_packs": ([vp], C.c_int),
    "frcnn_model_create": ([C.POINTER(ModelDesc), C.POINTER(vp)], C.c_int),
    "frcnn_model_destroy": ([vp], C.c_int),
    "frcnn_model_param_count": ([vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], C.c_int),
    "frcnn_model_param_table": ([vp, vp, C.c_int, C.POINTER(C.c_int)], C.c_int),
    "frcnn_model_localizer_layers": ([vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)], C.c_int),
    "frcnn_model_debug_buffer": ([vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_longlong)], C.c_int),
    "frcnn_pnet_forward": ([vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_ulonglong, vp], C.c_int),
    "frcnn_pnet_forward_async_heads": ([vp, vp, vp, C.c_int, C.c_int, vp, C.c_ulonglong, vp], C.c_int),
    "frcnn_pnet_output": ([vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "frcnn_pnet_delta": ([vp, C.c_int, C.POINTER(vp)], C.c_int),
    "frcnn_pnet_set_sparse_deltas": ([vp, C.c_int, vp, C.c_int], C.c_int),
    "frcnn_pnet_zero_deltas": ([vp, vp], C.c_int),
    "frcnn_pnet_backward_heads_begin": ([vp, vp, vp, vp], C.c_int),
    "frcnn_pnet_anchor_loss_begin": ([vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_pnet_anchor_loss_wait": ([vp, vp], C.c_int),
    "frcnn_pnet_backward_heads_join": ([vp, vp, C.POINTER(C.c_int)], C.c_int),
    "frcnn_pnet_backward": ([vp, vp, vp, vp], C.c_int),
    "frcnn_pnet_wait_block_gradients": ([vp, C.c_int, vp], C.c_int),
    "frcnn_cnet_forward": ([vp, vp, vp, C.c_int, C.c_int, vp, C.c_ulonglong, vp, vp, vp, vp], C.c_int),
    "frcnn_cnet_backward": ([vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "frcnn_cnet_backward_join": ([vp, vp], C.c_int),
    "frcnn_cnet_losses": ([vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp], C.c_int),
    "frcnn_cnet_decode": ([vp, C.c_int, C.c_int, vp, vp, vp], C.c_int),
    "frcnn_anchors_create": ([vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(vp)], C.c_int),
    "frcnn_anchors_destroy": ([vp], C.c_int),
    "frcnn_anchors_assemble": ([vp, vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, vp,
                               C.POINTER(C.c_int), vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "frcnn_comm_get_unique_id": ([vp], C.c_int),
    "frcnn_comm_exchange_id_file": ([C.c_char_p, C.c_int, vp, C.c_int], C.c_int),
    "frcnn_comm_init_rank": ([C.POINTER(vp), C.c_int, C.c_int, vp], C.c_int),
    "frcnn_comm_init_rank_timeout": ([C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int], C.c_int),
    "frcnn_comm_init_rank_file": ([C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, C.c_int], C.c_int),
    "frcnn_comm_destroy": ([vp], C.c_int),
    "frcnn_comm_info": ([vp, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "frcnn_comm_query": ([vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "frcnn_allreduce_f32": ([vp, vp, C.c_longlong, vp], C.c_int),
    "frcnn_allreduce_f64": ([vp, vp, C.c_longlong, vp], C.c_int),
    "frcnn_broadcast_f32": ([vp, vp, C.c_longlong, C.c_int, vp], C.c_int),
    "frcnn_image_rgb2yuv": ([vp, vp, C.c_int, C.c_int, vp], C.c_int),
    "frcnn_image_rgb2hsv": ([vp, vp, C.c_int, C.c_int, vp], C.c_int),
    "frcnn_image_rgb2lab": ([vp, vp, C.c_int, C.c_int, vp], C.c_int),
    "frcnn_image_scale": ([vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp], C.c_int),
    "frcnn_image_scale_u8": ([vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp], C.c_int),
    "frcnn_image_crop_flip": ([vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp], C.c_int),
    "frcnn_image_normalize_workspace_bytes": ([C.c_int], C.c_size_t),
    "frcnn_image_normalize": ([vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp], C.c_int),
    "frcnn_image_contrastive_norm": ([vp, C.c_int, C.c_int, vp, C.c_int, C.c_float, vp, vp, vp], C.c_int),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS.keys())


def load():
    """Load libfrcnn_hip.so; raises FrcnnError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FrcnnError(
            "libfrcnn_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)" % SO_PATH)
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).
    # Importing torch first makes the dynamic loader satisfy our NEEDED libamdhip64.so.7 with that copy,
    # so device pointers, streams and RCCL communicators are shared with torch.  (A host without torch,
    # e.g. the LuaJIT shim, simply gets the system runtime.)  Loading in the other order creates two
    # runtimes in one process and the second one fails with "no ROCm-capable device is detected".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    for name, (args, res) in _SIGS.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise FrcnnError("libfrcnn_hip: %s (code %d)" % (load().frcnn_last_error().decode("utf-8", "replace"), rc))


def call(name, *args):
    check(getattr(load(), name)(*args))
