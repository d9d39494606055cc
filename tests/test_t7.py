"""Torch7 snapshot serialisation (frcnn_amd/t7.py): hand-derived byte strings of torch's ASCII and
binary modes (File.lua writeObject, Tensor.c / Storage.c write, DiskFile.c formats -- restated, no real Torch7 file
was available), round trips, and the save_model / restore pair of utilities.lua:126-134 / main.lua:94-98."""
import io
import struct

import numpy as np
import pytest

from frcnn_amd import t7


def _ascii(obj):
    f = io.BytesIO(); t7.Writer(f, True).object(obj); return f.getvalue()


def _binary(obj):
    f = io.BytesIO(); t7.Writer(f, False).object(obj); return f.getvalue()


def test_ascii_known_answers():
    assert _ascii(1.5) == b"1\n1.5\n"                       # TYPE_NUMBER, %.17g
    assert _ascii(0.1) == b"1\n0.10000000000000001\n"
    assert _ascii(None) == b"0\n"
    assert _ascii(True) == b"5\n1\n"
    assert _ascii("abc") == b"2\n3\nabc\n"
    assert _ascii("") == b"2\n0\n"                          # no newline after zero raw chars
    assert _ascii({"a": True}) == b"3\n1\n1\n2\n1\na\n5\n1\n"   # TYPE_TABLE, index 1, one pair
    assert _ascii([7]) == b"3\n1\n1\n1\n1\n1\n7\n"            # {7}: key 1 (number) -> value 7
    t = np.array([1.5, 2.0], np.float32)
    assert _ascii(t) == (b"4\n1\n3\nV 1\n17\ntorch.FloatTensor\n1\n2\n1\n1\n"
                         b"4\n2\n3\nV 1\n18\ntorch.FloatStorage\n2\n1.5 2\n")
    m = np.arange(6, dtype=np.float64).reshape(2, 3)
    assert _ascii(m).startswith(b"4\n1\n3\nV 1\n18\ntorch.DoubleTensor\n2\n2 3\n3 1\n1\n4\n2\n3\nV 1\n19\ntorch.DoubleStorage\n6\n0 1 2 3 4 5\n")


def test_binary_known_answers():
    assert _binary(1.5) == struct.pack("<id", 1, 1.5)
    assert _binary("ab") == struct.pack("<ii", 2, 2) + b"ab"
    t = np.array([1.5, 2.0], np.float32)
    want = (struct.pack("<ii", 4, 1) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 17) + b"torch.FloatTensor" +
            struct.pack("<iqqq", 1, 2, 1, 1) + struct.pack("<ii", 4, 2) + struct.pack("<i", 3) + b"V 1" +
            struct.pack("<i", 18) + b"torch.FloatStorage" + struct.pack("<q", 2) + struct.pack("<ff", 1.5, 2.0))
    assert _binary(t) == want


def test_cuda_tensor_snapshot_reads_as_float():
    """The reference's snapshots hold `weights` as a torch.CudaTensor over a torch.CudaStorage (the nets are :cuda()
    before they are flattened, main.lua:86-92; utilities.lua:126-134): cutorch writes them like the Float classes."""
    raw = (b"4\n1\n3\nV 1\n16\ntorch.CudaTensor\n1\n3\n1\n1\n"
           b"4\n2\n3\nV 1\n17\ntorch.CudaStorage\n3\n1.5 2 -0.25\n")
    t = t7.Reader(io.BytesIO(raw), True).object()
    assert t.dtype == np.float32 and t.tolist() == [1.5, 2.0, -0.25]
    rawb = (struct.pack("<ii", 4, 1) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 16) + b"torch.CudaTensor" +
            struct.pack("<iqqq", 1, 2, 1, 1) + struct.pack("<ii", 4, 2) + struct.pack("<i", 3) + b"V 1" +
            struct.pack("<i", 17) + b"torch.CudaStorage" + struct.pack("<q", 2) + struct.pack("<ff", 1.5, 2.0))
    t = t7.Reader(io.BytesIO(rawb), False).object()
    assert t.dtype == np.float32 and t.tolist() == [1.5, 2.0]
    # a whole {version, weights, options, stats} table with device-class weights restores into the flat vector
    snap = (b"3\n1\n2\n2\n7\nversion\n1\n0\n2\n7\nweights\n" + raw.replace(b"4\n1\n3", b"4\n2\n3", 1).replace(b"4\n2\n3\nV 1\n17", b"4\n3\n3\nV 1\n17"))
    obj = t7.Reader(io.BytesIO(snap), True).object()
    assert obj["version"] == 0 and obj["weights"].tolist() == [1.5, 2.0, -0.25]


@pytest.mark.parametrize("ascii_mode", [True, False])
def test_round_trip(tmp_path, ascii_mode):
    rng = np.random.RandomState(0)
    shared = {"x": 1, "s": "two words"}
    obj = dict(version=0, weights=rng.randn(1000).astype(np.float32), options=dict(lr=1e-4, name="duplo", plot=False, opt=shared),
               stats=dict(pcls=[0.5, 0.25, 1e-9], preg=[], dcls=[3.0], dreg=[1e300]), again=shared,
               m=rng.randn(3, 4, 5), idx=np.arange(7, dtype=np.int64), empty=np.zeros(0, np.float32), none_inside=[1, "a", True])
    fn = str(tmp_path / "snap.t7")
    t7.save_obj(fn, obj, ascii_mode)
    back = t7.load_obj(fn, ascii_mode)
    assert back["version"] == 0 and back["options"]["name"] == "duplo" and back["options"]["plot"] is False
    assert np.array_equal(back["weights"], obj["weights"]) and back["weights"].dtype == np.float32   # %.9g round-trips fp32
    assert np.array_equal(back["m"], obj["m"]) and back["m"].shape == (3, 4, 5)                      # %.17g round-trips fp64
    assert np.array_equal(back["idx"], obj["idx"]) and back["empty"].size == 0
    assert back["stats"]["pcls"] == [0.5, 0.25, 1e-9] and back["stats"]["preg"] == {} and back["stats"]["dreg"] == [1e300]
    assert back["again"] is back["options"]["opt"]          # shared tables keep their identity (object indices)
    assert back["none_inside"] == [1, "a", True]


def test_save_model_and_restore(tmp_path):
    rng = np.random.RandomState(1)
    w = rng.randn(4321).astype(np.float32)
    fn = str(tmp_path / "model.t7")
    t7.save_model(fn, w, dict(model="models/vgg_small.lua", lr=1e-4), dict(pcls=[1.0], preg=[2.0], dcls=[3.0], dreg=[4.0]))
    head = open(fn, "rb").read(16)
    assert head.startswith(b"3\n1\n4\n")     # an ASCII table with four pairs, as torch.DiskFile(fn, 'w') writes it
    target = np.zeros(4321, np.float32)
    stored = t7.restore_weights(fn, target)
    assert np.array_equal(target, w) and stored["version"] == 0 and stored["stats"]["dreg"] == [4.0]
    with pytest.raises(ValueError):
        t7.restore_weights(fn, np.zeros(10, np.float32))


def test_strided_tensor_and_errors(tmp_path):
    # a transposed tensor as torch would write it: sizes 3x2, strides 1x3 over a 6-element storage
    txt = (b"4\n1\n3\nV 1\n17\ntorch.FloatTensor\n2\n3 2\n1 3\n1\n4\n2\n3\nV 1\n18\ntorch.FloatStorage\n6\n0 1 2 3 4 5\n")
    a = t7.Reader(io.BytesIO(txt), True).object()
    assert np.array_equal(a, np.arange(6, dtype=np.float32).reshape(2, 3).T)
    with pytest.raises(EOFError):
        t7.Reader(io.BytesIO(b"4\n1\n3\nV 1\n17\ntorch.Float"), True).object()
    with pytest.raises(ValueError):
        t7.Reader(io.BytesIO(b"9\n"), True).object()
    with pytest.raises(TypeError):
        t7.Writer(io.BytesIO(), True).object(object())


@pytest.mark.gpu
def test_snapshot_of_device_weights(F, small_cfg, tmp_path):
    """save_model on the flat device weight vector, restore into a perturbed copy: bit-identical weights."""
    model = F.vgg_small(small_cfg)
    w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=3)
    ref = w.cpu().numpy().copy()
    fn = str(tmp_path / "snap.t7")
    F.save_model(fn, w[:200000], dict(name="unit"), dict(pcls=[], preg=[], dcls=[], dreg=[]), ascii=False)
    part = w[:200000]
    part.mul_(0.5)
    stored = F.restore_weights(fn, part, ascii=False)
    assert np.array_equal(w.cpu().numpy(), ref) and stored["options"]["name"] == "unit"
