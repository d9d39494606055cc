"""Import shim: the product package lives in the directory `faster-rcnn.torch_amd/`, whose name
is not a valid Python identifier.  `import frcnn_amd` loads that directory as the package
`frcnn_amd` (sub-modules import as `frcnn_amd.Anchors`, ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "faster-rcnn.torch_amd")
_spec = importlib.util.spec_from_file_location(
    "frcnn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["frcnn_amd"] = _mod
_spec.loader.exec_module(_mod)
