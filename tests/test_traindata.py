"""Dataset preparation + training-data files (frcnn_amd/traindata.py; create-duplo-traindata.lua,
create-imagenet-traindata.lua): CSV / ILSVRC XML -> ground-truth table -> torch object file -> back, incl. the
torch.class 'Rect' objects inside it."""
import io
import os

import numpy as np

from frcnn_amd import Rect, t7, traindata


CSV = '''"img1.png", 0, 573, 59, 701, "DuploBrick_2x2", 2, "DuploBrightGreen", 11
"img2.png", 10, 20, 110, 220, "DuploBrick_2x4", 3, "DuploRed", 4
"img1.png", 5, 6, 50, 60, "DuploBrick_2x4", 3, "DuploRed", 4
"img3.png", 1, 2, 3, 4, "DuploBrick_2x2", 2, "DuploRed", 4
"img4.png", 7, 8, 9, 10, "DuploFigure", 9, "DuploRed", 4
"img5.png", 7, 8, 9, 10, "DuploFigure", 9, "DuploRed", 4
'''


def test_rect_is_written_as_a_torch_class_object():
    f = io.BytesIO(); t7.Writer(f, True).object(Rect(1, 2, 30, 40))
    txt = f.getvalue()
    assert txt.startswith(b"4\n1\n3\nV 1\n4\nRect\n3\n2\n4\n")     # TYPE_TORCH, index 1, "V 1", "Rect", then the field table
    back = t7.Reader(io.BytesIO(txt), True).object()
    assert isinstance(back, Rect) and (back.minX, back.minY, back.maxX, back.maxY) == (1, 2, 30, 40)
    # an unknown torch.class comes back as its field table
    other = txt.replace(b"4\nRect\n", b"5\nOther\n")
    o = t7.Reader(io.BytesIO(other), True).object()
    assert isinstance(o, t7.T7Object) and o.torch_class == "Other" and o["maxY"] == 40


def test_csv_to_training_data_and_back(tmp_path):
    csv = tmp_path / "boxes.csv"; csv.write_text(CSV)
    bg = tmp_path / "bg"; bg.mkdir(); (bg / "b1.png").write_bytes(b"x"); (bg / "b0.png").write_bytes(b"x"); (bg / "sub").mkdir()
    out = str(tmp_path / "duplo.t7")
    data = traindata.create_training_data("duplo-bricks", str(csv), str(bg), out, seed=3)
    # class table in order of first appearance; names keep the blank + quotes (the reference does not trim: split(',')
    # leaves ' "DuploBrick_2x2"', which '^"(.*)"$' does not match), file names are unquoted
    assert data["class_names"] == [' "DuploBrick_2x2"', ' "DuploBrick_2x4"', ' "DuploFigure"']
    assert sorted(data["ground_truth"]) == ["img1.png", "img2.png", "img3.png", "img4.png", "img5.png"]
    r = data["ground_truth"]["img1.png"]["rois"]
    assert len(r) == 2 and (r[0].rect.minX, r[0].rect.maxY, r[0].class_index) == (0, 701, 1) and r[1].class_index == 2
    # 80:20 split of the shuffled names: ceil(5 * 0.2) = 1 validation image, disjoint, complete
    assert len(data["validation_set"]) == 1 and len(data["training_set"]) == 4
    assert sorted(data["validation_set"] + data["training_set"]) == sorted(data["ground_truth"])
    assert data["background_files"] == ["b0.png", "b1.png"]
    assert traindata.create_training_data("d", str(csv), None, None, seed=3)["training_set"] == data["training_set"]
    back = traindata.load_training_data(out)
    assert back["dataset_name"] == "duplo-bricks" and back["training_set"] == data["training_set"]
    assert back["class_index"][' "DuploFigure"'] == 3 and back["background_files"] == ["b0.png", "b1.png"]
    rb = back["ground_truth"]["img1.png"]["rois"]
    assert isinstance(rb[0].rect, Rect) and (rb[1].rect.minX, rb[1].rect.minY, rb[1].rect.maxX, rb[1].rect.maxY) == (5, 6, 50, 60)
    assert open(out, "rb").read(8).startswith(b"3\n1\n7\n")    # ASCII table with the seven fields of the Lua script


XML = '''<annotation><folder>n01</folder><filename>a_{i}</filename><source><database>ILSVRC_2015</database></source>
<size><width>500</width><height>375</height></size>
<object><name>n0{i}</name><bndbox><xmin>1{i}</xmin><xmax>20{i}</xmax><ymin>5</ymin><ymax>17{i}</ymax></bndbox></object>
<object><name>n09</name><bndbox><xmin>3</xmin><xmax>40</xmax><ymin>5</ymin><ymax>60</ymax></bndbox></object></annotation>'''


def test_imagenet_annotations(tmp_path):
    base = tmp_path
    for split, n in (("Annotations/train/n01", 2), ("Annotations/val", 1)):
        d = base / split; d.mkdir(parents=True)
        for i in range(n):
            (d / ("f%d.xml" % i)).write_text(XML.format(i=i))
    (base / "bgdir").mkdir(); (base / "bgdir" / "x.JPEG").write_bytes(b"x"); (base / "bgdir" / "y.png").write_bytes(b"x")
    data = traindata.create_ground_truth_file("ILSVRC2015", str(base), "Annotations/train", "Annotations/val", "Data/train",
                                              "Data/val", ["bgdir"], str(base / "imagenet.t7"))
    p0 = os.path.join(str(base), "Data/train", "n01", "f0.JPEG")
    assert p0 in data["ground_truth"] and len(data["ground_truth"][p0]["rois"]) == 2
    assert data["training_set"].count(p0) == 2                   # one entry per <object>, as in the reference
    assert len(data["validation_set"]) == 2 and data["class_names"] == ["n00", "n09", "n01"]
    r = data["ground_truth"][p0]["rois"][0]
    assert (r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY, r.class_index) == (10, 5, 200, 170, 1)
    assert data["background_files"] == [os.path.join(str(base), "bgdir", "x.JPEG")]
    back = traindata.load_training_data(str(base / "imagenet.t7"))
    assert back["ground_truth"][p0]["rois"][1].class_name == "n09" and back["class_index"]["n01"] == 3
