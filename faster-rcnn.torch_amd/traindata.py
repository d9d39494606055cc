"""Dataset preparation: create-duplo-traindata.lua:7-81 (CSV of boxes) and
create-imagenet-traindata.lua:13-122 (ILSVRC annotation XML) -> the training-data table that main.lua loads with
`load_obj(opt.train)` and BatchIterator consumes: {dataset_name, ground_truth[file] = {image_file_name, rois =
{{rect, class_name, class_index}}}, training_set, validation_set, class_names, class_index, background_files},
written with t7.save_obj in torch's ASCII object format (Rect objects as torch.class 'Rect').
`math.random` (the 80:20 shuffle, utilities.lua:44-53) is replaced by the MT19937 stream, as everywhere else."""
import math
import os
import xml.etree.ElementTree as ET

from . import t7
from .Anchors import MT19937
from .Rect import Rect


class RoiEntry(object):
    """One ground-truth box: roi.rect / roi.class_name / roi.class_index (the table rows of the Lua scripts)."""

    def __init__(self, rect, class_name, class_index):
        self.rect, self.class_name, self.class_index = rect, class_name, class_index

    def as_table(self):
        return dict(rect=self.rect, class_name=self.class_name, class_index=self.class_index)


def _remove_quotes(s):  # utilities.lua remove_quotes: only a fully quoted value is unquoted
    return s[1:-1] if len(s) >= 2 and s[0] == '"' and s[-1] == '"' else s


def read_csv_file(fn):  # create-duplo-traindata.lua:7-47
    filemap, class_names, class_index = {}, [], {}
    with open(fn, "r") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            v = line.split(",")   # (a trivial csv file without ',' in string values)
            class_name = _remove_quotes(v[5])   # (no trimming, as in the reference: ' "name"' keeps its blank and quotes)
            if class_name not in class_index:
                class_names.append(class_name)
                class_index[class_name] = len(class_names)
            image_file_name = _remove_quotes(v[0])
            entry = filemap.setdefault(image_file_name, dict(image_file_name=image_file_name, rois=[]))
            entry["rois"].append(RoiEntry(Rect(float(v[1]), float(v[2]), float(v[3]), float(v[4])), class_name,
                                          class_index[class_name]))
    return filemap, class_names, class_index


def _shuffle(array, rng):  # utilities.lua:44-53 (math.random(n) in 1..n)
    for n in range(len(array), 1, -1):
        i = rng.random() % n + 1
        array[n - 1], array[i - 1] = array[i - 1], array[n - 1]
    return array


def _table(ground_truth, training_set, validation_set, class_names, class_index, background_files, dataset_name):
    return dict(dataset_name=dataset_name,
                ground_truth={k: dict(image_file_name=v["image_file_name"], rois=[r.as_table() for r in v["rois"]])
                              for k, v in ground_truth.items()},
                training_set=training_set, validation_set=validation_set, class_names=class_names,
                class_index=class_index, background_files=background_files)


def create_training_data(dataset_name, csv_file_name, background_dir, output_fn=None, validation_size=None, seed=5489):
    """create-duplo-traindata.lua:50-79.  Returns the table (ROIs as RoiEntry objects, ready for BatchIterator) and,
    when output_fn is given, writes it as a torch object file."""
    ground_truth, class_names, class_index = read_csv_file(csv_file_name)
    file_names = list(ground_truth.keys())
    validation_size = 0.2 if validation_size is None else validation_size   # 80:20 split
    if 0 <= validation_size < 1:
        validation_size = int(math.ceil(len(file_names) * validation_size))
    _shuffle(file_names, MT19937(seed))
    validation_size = int(validation_size)
    validation_set = file_names[len(file_names) - validation_size:] if validation_size else []   # remove_tail
    training_set = file_names[:len(file_names) - validation_size]
    background_files = []
    if background_dir:   # list_files(dir, nil, false): plain names of the regular files
        background_files = [fn for fn in sorted(os.listdir(background_dir)) if os.path.isfile(os.path.join(background_dir, fn))]
    data = dict(dataset_name=dataset_name, ground_truth=ground_truth, training_set=training_set, validation_set=validation_set,
                class_names=class_names, class_index=class_index, background_files=background_files)
    if output_fn:
        t7.save_obj(output_fn, _table(ground_truth, training_set, validation_set, class_names, class_index, background_files,
                                      dataset_name))
    return data


def import_annotation_file(anno_base, data_base, fn, name_table, ground_truth, class_names, class_index):
    """create-imagenet-traindata.lua:13-63: one ILSVRC annotation XML; every <object> appends the image path to
    name_table (so an image with n objects is listed n times, as in the reference)."""
    a = ET.parse(fn).getroot()
    if a.tag != "annotation":
        a = a.find("annotation")
    for obj in a.findall("object"):
        name = obj.find("name").text
        bb = obj.find("bndbox")
        xmin, xmax = float(bb.find("xmin").text), float(bb.find("xmax").text)
        ymin, ymax = float(bb.find("ymin").text), float(bb.find("ymax").text)
        if name not in class_index:
            class_names.append(name)
            class_index[name] = len(class_names)
        image_path = os.path.join(data_base, os.path.relpath(fn, anno_base))
        image_path = image_path[:-3] + "JPEG"   # replace the 'xml' ending
        name_table.append(image_path)
        entry = ground_truth.setdefault(image_path, dict(image_file_name=image_path, rois=[]))
        entry["rois"].append(RoiEntry(Rect(xmin, ymin, xmax, ymax), name, class_index[name]))


def create_ground_truth_file(dataset_name, base_dir, train_annotation_dir, val_annotation_dir, train_data_dir, val_data_dir,
                             background_dirs, output_fn=None):
    """create-imagenet-traindata.lua:82-122 (recursive walk of the annotation directories, sorted for reproducibility)."""
    ground_truth, class_names, class_index = {}, [], {}
    expand = lambda p: os.path.join(base_dir, p)

    def walk(anno_base, data_base, names):
        for root, dirs, files in os.walk(anno_base):
            dirs.sort()
            for fn in sorted(files):
                if fn[-4:].lower() == ".xml":
                    import_annotation_file(anno_base, data_base, os.path.join(root, fn), names, ground_truth, class_names, class_index)
    training_set, validation_set = [], []
    walk(expand(train_annotation_dir), expand(train_data_dir), training_set)
    walk(expand(val_annotation_dir), expand(val_data_dir), validation_set)
    background_files = []
    for d in background_dirs:
        d = expand(d)
        background_files += [os.path.join(d, fn) for fn in sorted(os.listdir(d))
                             if os.path.isfile(os.path.join(d, fn)) and fn[-5:].lower() == ".jpeg"]
    data = dict(dataset_name=dataset_name, ground_truth=ground_truth, training_set=training_set, validation_set=validation_set,
                class_names=class_names, class_index=class_index, background_files=background_files)
    if output_fn:
        t7.save_obj(output_fn, _table(ground_truth, training_set, validation_set, class_names, class_index, background_files,
                                      dataset_name))
    return data


def load_training_data(file_name, ascii=True):
    """main.lua `load_obj(opt.train)`: a training-data file (written by this module or by the Lua scripts) -> the
    table with RoiEntry rows that BatchIterator takes."""
    d = t7.load_obj(file_name, ascii)
    gt = {}
    for k, v in d["ground_truth"].items():
        rois = v["rois"] if isinstance(v["rois"], list) else [v["rois"][i] for i in sorted(v["rois"])]
        gt[k] = dict(image_file_name=v["image_file_name"],
                     rois=[RoiEntry(r["rect"], r.get("class_name"), int(r["class_index"])) for r in rois])
    as_list = lambda x: x if isinstance(x, list) else [x[i] for i in sorted(x)] if x else []
    return dict(dataset_name=d.get("dataset_name"), ground_truth=gt, training_set=as_list(d["training_set"]),
                validation_set=as_list(d["validation_set"]), class_names=as_list(d.get("class_names")),
                class_index=d.get("class_index") or {}, background_files=as_list(d.get("background_files")))
