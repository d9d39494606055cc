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


def _lua_code(txt):
    """Lua source with comments and string literals blanked (enough for keyword counting)."""
    import re
    txt = re.sub(r"--\[\[.*?\]\]", " ", txt, flags=re.S)
    txt = re.sub(r"\[\[.*?\]\]", " S ", txt, flags=re.S)
    out = []
    for line in txt.splitlines():
        res, i, n = [], 0, len(line)
        while i < n:
            c = line[i]
            if c == "-" and line[i:i + 2] == "--":
                break
            if c in "'\"":
                j = i + 1
                while j < n and line[j] != c:
                    j += 2 if line[j] == "\\" else 1
                res.append(" S ")
                i = j + 1
                continue
            res.append(c)
            i += 1
        out.append("".join(res))
    return "\n".join(out)


def test_lua_drop_in_files_are_consistent_with_the_abi():
    """bindings/objective_hip.lua / Detector_hip.lua / frcnn_hip.lua cannot be executed here (no Lua runtime in the
    image).  What can be checked: every C.frcnn_* they call is declared by the generated cdef (= the header), blocks are
    balanced (function / do / then / repeat vs end / until), the globals the reference's call sites need are defined, and
    pnet.outnode.children[i] is built as the node chain Localizer.lua:8-36 walks (.data.module leaf, .children[1])."""
    import re
    lua = open(os.path.join(ROOT, "bindings", "frcnn_hip.lua")).read()
    cdef = lua[lua.index("ffi.cdef[["):lua.index("]]")]
    declared = set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", cdef))
    used_all = set()
    for fn in ("frcnn_hip.lua", "objective_hip.lua", "Detector_hip.lua", "frcnn_nn.lua", "cunn.lua", "nms.lua", "objective.lua", "Detector.lua"):
        txt = open(os.path.join(ROOT, "bindings", fn)).read()
        code = _lua_code(txt[txt.index("]]") + 2:] if fn == "frcnn_hip.lua" else txt)
        used = set(re.findall(r"\bC\.(frcnn_[a-z0-9_]+)", code))
        assert used <= declared, (fn, used - declared)
        used_all |= used
        words = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", code)
        opens = words.count("function") + words.count("do") + words.count("then") - words.count("elseif") + words.count("repeat")
        closes = words.count("end") + words.count("until")
        assert opens == closes, "%s: %d block openers vs %d closers" % (fn, opens, closes)
        assert code.count("(") == code.count(")") and code.count("{") == code.count("}") and code.count("[") == code.count("]"), fn
    # the batched entry points of the training step and of detect, and the exchange step, are all reached from Lua
    for name in ("frcnn_pnet_forward_async_heads", "frcnn_pnet_anchor_loss_begin", "frcnn_pnet_anchor_loss_wait",
                 "frcnn_pnet_set_sparse_deltas", "frcnn_roi_pool_forward", "frcnn_roi_pool_backward", "frcnn_cnet_losses",
                 "frcnn_cnet_forward", "frcnn_cnet_backward", "frcnn_pnet_backward", "frcnn_rpn_scan", "frcnn_nms_device_n",
                 "frcnn_roi_windows", "frcnn_detect_post", "frcnn_detect_gather",
                 "frcnn_cnet_decode", "frcnn_allreduce_f32", "frcnn_allreduce_f64", "frcnn_comm_init_rank_file",
                 "frcnn_broadcast_f32", "frcnn_rmsprop", "frcnn_add"):
        assert name in used_all, name
    obj = open(os.path.join(ROOT, "bindings", "objective_hip.lua")).read()
    det = open(os.path.join(ROOT, "bindings", "Detector_hip.lua")).read()
    assert re.search(r"^function extract_roi_pooling_input\(input_rect, localizer, feature_layer_output\)", obj, re.M)
    assert re.search(r"^function create_objective\(model, weights, gradient, batch_iterator, stats\)", obj, re.M)
    assert "torch.class('Detector')" in det and "function Detector:detect(input)" in det and "function Detector:__init(model)" in det
    shim = open(os.path.join(ROOT, "bindings", "frcnn_shims.lua.in")).read()
    assert "node = { data = { module = leaf }, children = { node } }" in shim
    for g in ("create_model =", "combine_and_flatten_parameters =", "nms =", "cutorch.setDevice =", "optim.rmsprop =", "save_model ="):
        assert g in shim, g
    # zero-diff main.lua: every module name main.lua:1-14 requires that this repository replaces exists under that very name
    # in bindings/ (package.path precedence swaps them), and they chain to the binding / the batched drop-ins
    B = lambda fn: open(os.path.join(ROOT, "bindings", fn)).read()
    assert "require 'frcnn_hip'" in B("cunn.lua") and "require 'frcnn_nn'" in B("cunn.lua")
    assert re.search(r"^nms = hip\.nms", B("nms.lua"), re.M)
    assert "return require 'objective_hip'" in B("objective.lua") and "return require 'Detector_hip'" in B("Detector.lua")
    main = os.path.join("/root/reference", "main.lua")
    if os.path.exists(main):   # (build container only) every `require` of the hot path is served without touching main.lua
        req = re.findall(r"^require '([A-Za-z_]+)'", open(main).read(), re.M)
        assert req[:14] == ["torch", "pl", "optim", "image", "nngraph", "cunn", "nms", "gnuplot", "utilities", "Anchors", "BatchIterator",
                            "objective", "Detector"][:len(req[:14])]
        for name in ("cunn", "nms", "objective", "Detector"):
            assert os.path.exists(os.path.join(ROOT, "bindings", name + ".lua")), name
    # the stand-alone nn modules of SURVEY 8b (the reference's own objective.lua:24-30 / Detector.lua:13-14 construct them)
    nnl = B("frcnn_nn.lua")
    for cls in ("nn.SpatialAdaptiveMaxPooling", "nn.LogSoftMax", "nn.CrossEntropyCriterion", "nn.ClassNLLCriterion", "nn.SmoothL1Criterion"):
        assert "device_twin(%s," % cls in nnl, cls
    for need in ("self.indices =", "function amp:backward(input, gradOutput)", "sizeAverage", "frcnn_roi_pool_forward", "frcnn_roi_pool_backward"):
        assert need in nnl, need


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
