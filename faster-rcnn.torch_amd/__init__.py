"""faster-rcnn.torch_amd -- MI355X-native Faster R-CNN detection/training hot path behind the
reference's own surface (Rect, Localizer, Anchors, nms, Detector, create_objective, model factories).
All arithmetic runs in libfrcnn_hip.so (hand-written HIP for gfx950, C ABI in include/frcnn_hip.h);
this package is the host-side mirror of the Lua interface.  There is no CPU fallback."""
from . import _lib
from ._lib import FrcnnError
from .Anchors import Anchors, MT19937, manualSeed
from .BatchIterator import BatchIterator, decode_image, find_target_size, gaussian1D
from . import comm
from .comm import Comm
from .config import duplo_cfg, imgnet_cfg
from .Detector import Detector
from . import evaluation
from .evaluation import evaluate_detections, mean_average_precision, validation_losses
from .Localizer import Localizer
from .model_utilities import create_model
from .nms import nms
from .objective import allreduce_begin, allreduce_begin_rest, allreduce_gradient_and_stats, create_objective, extract_roi_pooling_input, roi_window, roi_windows
from .Rect import Rect
from . import t7, traindata
from .t7 import load_obj, restore_weights, save_model, save_obj
from .synthetic import Roi, SyntheticBatchIterator, assemble_examples, clean_examples, output_map_sizes, synthetic_image, synthetic_rois
from .tensor import DeviceTensor, ptr, stream_ptr, to_device
from .utilities import combine_and_flatten_parameters, rmsprop
from .vgg_large import vgg_large
from .vgg_small import vgg_small

__all__ = ["Comm", "comm", "evaluation", "evaluate_detections", "mean_average_precision", "validation_losses", "decode_image", "traindata", "t7", "load_obj", "restore_weights", "save_model", "save_obj", "BatchIterator", "find_target_size", "gaussian1D", "allreduce_begin", "allreduce_begin_rest", "Anchors", "Detector", "DeviceTensor", "FrcnnError", "Localizer", "MT19937", "Rect", "combine_and_flatten_parameters",
           "create_model", "create_objective", "duplo_cfg", "extract_roi_pooling_input", "imgnet_cfg", "manualSeed", "nms",
           "rmsprop", "roi_window", "roi_windows", "vgg_large", "vgg_small"]
