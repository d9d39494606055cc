"""Host-side mirror of the reference's Lua classes (Rect / Localizer / Anchors / MT19937 / ROI
windows) against the CPU oracle: anchor indices and tables bit-exact."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st


@pytest.fixture(scope="module")
def env(F, O):
    from util import oracle_model
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    om = oracle_model(O, cfg)
    return dict(cfg=cfg, model=model, om=om, A=F.Anchors(model["pnet"], cfg["scales"]), OA=O.Anchors(om))


def test_rect_semantics(F):
    R = F.Rect
    a, b = R(0, 0, 10, 10), R(5, 5, 20, 30)
    assert R.IoU(a, b) == 25.0 / (100 + 375 - 25)
    assert R.intersect(a, R(10.5, 0, 12, 3)).unpack() == (0, 0, 0, 0)      # disjoint -> empty()
    assert a.overlaps(b) and not a.overlaps(R(10, 0, 20, 10))              # strict
    assert R(-3, 2, 50, 7).clip(R(0, 0, 10, 5)).unpack() == (0, 2, 10, 5)
    assert R(0.2, 1.7, 3.1, 4.0).snapToInt().unpack() == (0, 1, 4, 4)
    assert R.fromCenterWidthHeight(10, 10, 4, 6).unpack() == (8, 7, 12, 13)


def test_tables_and_localizers_bit_exact(env, O):
    A, OA = env["A"], env["OA"]
    assert np.array_equal(A.w, OA.w_table) and np.array_equal(A.h, OA.h_table)
    for i in range(5):
        want = O.model_localizer_layers(env["om"], i + 1)
        got = env["model"]["pnet"].outnode.children[i].layers
        assert np.array_equal(got, want)


@settings(max_examples=100, deadline=None)
@given(st.floats(-50, 820), st.floats(-50, 470), st.floats(0.5, 400), st.floats(0.5, 300))
def test_roi_window_matches_oracle(env, F, O, x, y, w, h):
    loc = F.Localizer(env["model"]["pnet"].outnode.children[4])
    fl = O.model_localizer_layers(env["om"], 5)
    r = F.Rect(x, y, x + w, y + h)
    assert list(F.roi_window(r, loc, 29, 50)) == O.extract_roi_window(fl, r.unpack(), 29, 50).tolist()
    fr = loc.inputToFeatureRect(r)
    assert list(fr.unpack()) == O.loc_input_to_feature(fl, r.unpack()).tolist()


def test_find_positive_sample_negative_find_nearby(env, F, O):
    A, OA, cfg = env["A"], env["OA"], env["cfg"]
    img = F.Rect(0, 0, 800, 450)
    rng = np.random.RandomState(21)
    for trial in range(6):
        rois = F.synthetic_rois(cfg, 800, 450, 4, 7, trial)
        rr = [r.rect.unpack() for r in rois]
        pos = A.findPositive(rois, img, cfg["positive_threshold"], cfg["negative_threshold"], trial % 2 == 0)
        idx, rc = OA.find_positive(rr, img.unpack(), cfg["positive_threshold"], cfg["negative_threshold"], trial % 2 == 0)
        got = [[a.layer, a.aspect, a.index[1], a.index[2], rois.index(r) + 1] for a, r in pos]
        assert got == idx.tolist()
        assert [list(a.unpack()) for a, _ in pos] == rc.tolist()
        neg = A.sampleNegative(img, rois, cfg["negative_threshold"], 16, F.MT19937(100 + trial))
        nidx, _ = OA.sample_negative(img.unpack(), rr, cfg["negative_threshold"], 16, O.MT(100 + trial))
        assert [[e[0].layer, e[0].aspect, e[0].index[1], e[0].index[2]] for e in neg] == nidx.tolist()
        cx, cy = rng.uniform(0, 800), rng.uniform(0, 450)
        near = A.findNearby(cx, cy)
        oidx, orc = OA.find_nearby(cx, cy)
        assert [[a.layer, a.aspect, a.index[1], a.index[2]] for a in near] == oidx.tolist()
    rg = A.findRangesXY(img, img)
    org = OA.find_ranges_xy(img.unpack(), img.unpack())
    assert [[r["layer"], r["aspect"], r["lx"], r["ly"], r["ux"], r["uy"]] for r in rg] == org.tolist()


def test_bbox_parameterisation_roundtrip(F, O):
    a = F.Rect(24, -8, 280, 248); r = F.Rect(40.5, 3.25, 200.0, 180.75)
    t = F.Anchors.inputToAnchor(a, r)
    assert t.dtype == np.float32 and np.array_equal(t, O.input_to_anchor(a.unpack(), r.unpack()))
    back = F.Anchors.anchorToInput(a, t)
    assert np.allclose(back.unpack(), O.anchor_to_input(a.unpack(), t), rtol=0, atol=1e-12)
    assert np.allclose(back.unpack(), r.unpack(), atol=1e-3)   # top-left relative parameterisation inverts


def test_mt19937_matches_oracle(F, O):
    a, b = F.MT19937(7), O.MT(7)
    assert [a.random() for _ in range(1300)] == [b.random() for _ in range(1300)]


def test_nms_key_dispatch(F):
    from frcnn_amd.nms import _key
    import numpy as np
    assert _key(None) == (0, 0) and _key(np.zeros(3)) == (0, 0) and _key("area") == (1, 0) and _key(5) == (2, 5)
    assert _key("score") == (0, 0)   # any other string also falls through to y2 (nms.lua:41-43)


def test_batch_rule_of_the_reference(F):
    """BatchIterator.lua:166-268: nextTraining(count) keeps adding images until `count` examples are on board
    (default cfg.batch_size = 256); the benchmark configuration pins one image per call instead."""
    cfg = dict(F.duplo_cfg)
    model = F.vgg_small(cfg)
    it = F.SyntheticBatchIterator(model, H=200, W=320, images_per_batch=None, pool=3, device_images=False)
    per_image = [len(x["positive"]) + len(x["negative"]) for x in it.pool]
    assert all(n > 0 for n in per_image)
    batch = it.nextTraining()
    total = sum(len(x["positive"]) + len(x["negative"]) for x in batch)
    assert total >= cfg["batch_size"]
    assert total - (len(batch[-1]["positive"]) + len(batch[-1]["negative"])) < cfg["batch_size"]   # no image too many
    assert len(it.nextTraining(count=1)) == 1
    one = F.SyntheticBatchIterator(model, H=200, W=320, images_per_batch=1, pool=2, device_images=False)
    assert len(one.nextTraining()) == 1


def test_native_example_assembly_equals_python_mirror(F, small_cfg):
    """frcnn_anchors_assemble (anchors.cpp) against synthetic.assemble_examples' Python path: identical positives,
    negatives (tags, rects, ROI links) and an identical MT19937 state afterwards, over many images, ROI layouts,
    thresholds and both configs of the best-match / nearby-aversion switches."""
    model = F.vgg_small(small_cfg)
    anchors = F.Anchors(model["pnet"], small_cfg["scales"])
    rng_np = np.random.RandomState(4)
    for trial in range(24):
        cfg = dict(small_cfg)
        cfg["best_match"] = bool(trial % 2)
        cfg["nearby_aversion"] = bool((trial // 2) % 2)
        if trial % 5 == 4:
            cfg["positive_threshold"], cfg["negative_threshold"] = 0.6, 0.3
        W, H = [(800, 450), (450, 800), (640, 480), (1000, 600)][trial % 4]
        n = trial % 5     # incl. images without any ROI
        rois = []
        for _ in range(n):
            w = rng_np.uniform(20, 0.8 * W); h = rng_np.uniform(20, 0.8 * H)
            x = rng_np.uniform(0, W - w); y = rng_np.uniform(0, H - h)
            rois.append(F.Roi(F.Rect(x, y, x + w, y + h), int(rng_np.randint(1, 17))))
        a, b = F.MT19937(100 + trial), F.MT19937(100 + trial)
        for _ in range(trial):            # (start somewhere inside the stream, incl. across a state regeneration)
            a.random(); b.random()
        if trial == 7:
            for _ in range(700):
                a.random(); b.random()
        p1, n1 = F.assemble_examples(anchors, cfg, rois, W, H, a, negatives=16, native=False)
        p2, n2 = F.assemble_examples(anchors, cfg, rois, W, H, b, negatives=16, native=True)
        key = lambda r: (r.layer, r.aspect, r.index, r.minX, r.minY, r.maxX, r.maxY)
        assert [key(e[0]) for e in p1] == [key(e[0]) for e in p2], trial
        assert [e[1] is f[1] for e, f in zip(p1, p2)] == [True] * len(p1)
        assert [key(e[0]) for e in n1] == [key(e[0]) for e in n2], trial
        assert a.idx == b.idx and np.array_equal(a.state, b.state)
        assert a.random() == b.random()
