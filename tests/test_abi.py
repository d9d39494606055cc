"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/frcnn_hip.h
declares; the ctypes table of the host mirror covers exactly that set."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(F):
    names = _header_symbols()
    assert len(names) >= 45
    lib = ctypes.CDLL(F._lib.SO_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_matches_header(F):
    assert sorted(F._lib.exported_symbols()) == _header_symbols()


def test_host_only_entry_points(F):
    lib = F._lib.load()
    assert lib.frcnn_version() >= 100
    assert lib.frcnn_nms_workspace_bytes(300) > 300 * 12
    m = F.vgg_small(dict(F.duplo_cfg))
    nat = m["native"]
    assert (nat.total_params, nat.pnet_params) == (26784106, 12089683)
    assert [len(nat.localizer_layers(i)) for i in range(1, 6)] == [10, 13, 13, 13, 11]
    large = F.vgg_large(dict(F.imgnet_cfg))
    # hand count (models/vgg_large.lua:5-22, imagenet.lua:2,9): backbone 7 635 274 + anchor nets 11 488 332 = 19 123 606,
    # cnet 18432*1024+1024 + 2*1024 + 1 + 1024*512+512 + 1 + 512*4+4 + 512*201+201 = 19 507 407
    assert (large["native"].total_params, large["native"].pnet_params) == (38631013, 19123606)
    c7 = dict(F.imgnet_cfg); c7["roi_pooling"] = dict(kw=7, kh=7)      # README.md:19 experiment
    assert F.vgg_large(c7)["native"].total_params == 38631013 + 512 * 13 * 1024
    # errors surface as exceptions carrying frcnn_last_error()
    import pytest
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_model_localizer_layers", nat.h, 99, None, 0, ctypes.byref(ctypes.c_int()))


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "faster-rcnn.torch_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "pyoracle" not in txt and "frcnn_oracle" not in txt and "naive_np" not in txt and "orc_image" not in txt, fn
    # tools/ and the entry points: only bench.py (cpu_baseline leg) and __graft_entry__ (build / smoke) may touch oracle/
    for fn in sorted(os.listdir(os.path.join(ROOT, "tools"))):
        if fn.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", fn)).read()
            assert "pyoracle" not in txt and "orc_image" not in txt and "naive_np" not in txt and "/oracle" not in txt, fn


def test_lua_binding_in_sync_with_header():
    """bindings/frcnn_hip.lua (the LuaJIT ffi binding a maintainer of the reference would `require`) carries a
    generated ffi.cdef: it must declare exactly the functions of include/frcnn_hip.h."""
    import re
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_lua_binding.py"), "--check"]) == 0, \
        "bindings/frcnn_hip.lua is stale: run python tools/gen_lua_binding.py"
    lua = open(os.path.join(ROOT, "bindings", "frcnn_hip.lua")).read()
    cdef = lua[lua.index("ffi.cdef[["):lua.index("]]")]
    declared = set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", cdef))
    assert declared == set(_header_symbols())
    used = set(re.findall(r"C\.(frcnn_[a-z0-9_]+)", lua))
    assert used <= declared, used - declared


def test_kernel_class_table_matches_header():
    """frcnn_prof_collect fills FRCNN_KC_COUNT entries: the header's constants and the Python name table must agree."""
    import re
    from frcnn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    consts = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"#define FRCNN_KC_(\w+) (\d+)", hdr))
    count = consts.pop("COUNT")
    assert count == len(_lib.KC_NAMES) == len(consts)
    for name, idx in consts.items():
        assert _lib.KC_NAMES[idx].upper() == name, (name, idx, _lib.KC_NAMES[idx])
