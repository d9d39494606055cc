"""Decision injection of the oracle (oracle/frcnn_oracle.h orc_set_decisions; CPU only, narrow model).
Recording a run's own decisions and injecting them back must reproduce the run bit for bit; injecting a changed
max-pool winner / PReLU branch must move the gradient the way the choice says; nothing else may depend on it."""
import numpy as np

from test_dp_gloo import _rank_inputs


def _shapes(O, m, img, R):
    """decision arrays (zeros) of the right shapes for model m on img, from the model description"""
    import math
    _, h, w = img.shape
    d = dict(pool_idx=[], conv_pos=[], head_pos=[], cnet_pos=[], roi_idx=None)
    hw = []
    for b in range(m.nblocks):
        for _ in range(m.conv_steps[b]):
            h = h + 2 * m.pad[b] - m.ksize[b] + 1; w = w + 2 * m.pad[b] - m.ksize[b] + 1
            d["conv_pos"].append(np.zeros((m.filters[b], h, w), np.uint8))
        h, w = int(math.ceil((h - 2) / 2.0)) + 1, int(math.ceil((w - 2) / 2.0)) + 1
        d["pool_idx"].append(np.zeros((m.filters[b], h, w), np.int32))
        hw.append((h, w))
    for i in range(m.nheads):
        bh, bw = hw[m.head_input[i] - 1]
        k = m.head_k[i]
        d["head_pos"].append(np.zeros((m.head_n[i], bh - k + 1, bw - k + 1), np.uint8))
    for l in range(m.ncls):
        d["cnet_pos"].append(np.zeros((R, m.cls_n[l]), np.uint8))
    d["roi_idx"] = np.zeros((R, m.kh * m.kw * m.filters[m.nblocks - 1]), np.int32)
    return d


def _run(O, k=0, inject=None, record=None):
    m, w, img, pidx, prect, rois, rcls, nidx, nrect, pm, cm = _rank_inputs(k)
    g = np.zeros_like(w); acc = np.zeros(8)
    bn = np.concatenate([np.zeros(48, np.float32), np.ones(48, np.float32)])
    with O.decisions(inject=inject, record=record):
        O.train_image(m, w, g, img, pidx, prect, rois, rcls, nidx, nrect, pm, cm, bn, acc)
    return g, acc, (m, img, len(pidx) + len(nidx))


def test_recorded_decisions_injected_back_reproduce_the_run(O):
    g0, acc0, (m, img, R) = _run(O)
    rec = _shapes(O, m, img, R)
    g1, acc1, _ = _run(O, record=rec)
    assert np.array_equal(g0, g1) and np.array_equal(acc0, acc1)          # recording changes nothing
    assert all(a.any() for a in rec["conv_pos"]) and rec["roi_idx"].any() and all(a.max() > 0 for a in rec["pool_idx"])
    g2, acc2, _ = _run(O, inject=rec)
    assert np.array_equal(g0, g2) and np.array_equal(acc0, acc2)          # own decisions injected: bit-identical
    # injecting + recording at once: the record still holds the run's OWN choices
    rec2 = _shapes(O, m, img, R)
    g3, _, _ = _run(O, inject=rec, record=rec2)
    assert np.array_equal(g0, g3)
    for kname in ("pool_idx", "conv_pos", "head_pos", "cnet_pos"):
        assert all(np.array_equal(a, b) for a, b in zip(rec[kname], rec2[kname])), kname
    assert np.array_equal(rec["roi_idx"], rec2["roi_idx"])


def test_an_injected_choice_re_routes_the_gradient(O):
    g0, _, (m, img, R) = _run(O)
    rec = _shapes(O, m, img, R)
    _run(O, record=rec)
    # 1) every last-block pooling window takes its top-left entry instead of its maximum: the loss changes, the run stays finite
    alt = {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in rec.items()}
    b = m.nblocks - 1
    C_, hp, wp = alt["pool_idx"][b].shape
    in_w = rec["conv_pos"][-1].shape[2]
    oy = np.arange(hp)[None, :, None]; ox = np.arange(wp)[None, None, :]
    alt["pool_idx"][b][:] = (2 * oy) * in_w + 2 * ox
    alt["roi_idx"] = None     # (cells now hold other values: let the oracle choose the cell winners itself)
    g1, acc1, _ = _run(O, inject=alt)
    assert np.isfinite(g1).all() and not np.array_equal(g0, g1)
    # 2) all PReLU branches of the first anchor net forced negative: its slope gradient becomes sum(x * gy) over ALL entries,
    #    the tensors of the other anchor nets do not move
    alt = {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in rec.items()}
    alt["head_pos"][0][:] = 0
    g2, _, _ = _run(O, inject=alt)
    assert not np.array_equal(g0, g2)
    n, pn = O.param_count(m)
    assert np.array_equal(g0[pn:], g2[pn:])      # the classification net sits above every pnet decision of the backward path
