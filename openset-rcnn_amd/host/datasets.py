"""Dataset catalog + the VOC-COCO open-set splits (the data format either side of the hot path).

Mirrors /root/reference/openset_rcnn/data/voc_coco.py (category list :5-29, register_voc_coco :32-42) and data/custom.py
(register_opendet_voc_coco :32-51) on a minimal catalog of our own, plus [d2] load_voc_instances: one dict per image with
file_name, image_id, height, width and annotations [{category_id, bbox XYXY with xmin/ymin - 1}]."""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence

# the 20 VOC classes, the 60 remaining COCO classes in the order of the 20-40 / 40-60 / 60-80 splits, then "unknown"
VOC_COCO_CATEGORIES = [
    "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse", "motorbike",
    "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor",
    "truck", "traffic light", "fire hydrant", "stop sign", "parking meter", "bench", "elephant", "bear", "zebra", "giraffe", "backpack",
    "umbrella", "handbag", "tie", "suitcase", "microwave", "oven", "toaster", "sink", "refrigerator",
    "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard", "tennis racket",
    "banana", "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake",
    "bed", "toilet", "laptop", "mouse", "remote", "keyboard", "cell phone", "book", "clock", "vase", "scissors", "teddy bear", "hair drier",
    "toothbrush", "wine glass", "cup", "fork", "knife", "spoon", "bowl",
    "unknown",
]

VOC_COCO_SPLITS = ("voc_coco_20_40_test", "voc_coco_20_60_test", "voc_coco_20_80_test", "voc_coco_2500_test", "voc_coco_5000_test",
                   "voc_coco_10000_test", "voc_coco_20000_test", "voc_coco_val")


class _Catalog(dict):
    def register(self, name: str, value) -> None:
        assert name not in self, f"dataset '{name}' is already registered"
        self[name] = value


DatasetCatalog: Dict[str, Callable[[], List[dict]]] = _Catalog()
_METADATA: Dict[str, SimpleNamespace] = {}


class MetadataCatalog:
    @staticmethod
    def get(name: str) -> SimpleNamespace:
        if name not in _METADATA:
            _METADATA[name] = SimpleNamespace(name=name)
        return _METADATA[name]


def load_voc_instances(dirname: str, split: str, class_names: Sequence[str]) -> List[dict]:
    """[d2] detectron2.data.datasets.pascal_voc.load_voc_instances."""
    with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
        fileids = [x.strip() for x in f.readlines() if x.strip()]
    dicts = []
    for fid in fileids:
        tree = ET.parse(os.path.join(dirname, "Annotations", fid + ".xml"))
        rec = dict(file_name=os.path.join(dirname, "JPEGImages", fid + ".jpg"), image_id=fid,
                   height=int(tree.findall("./size/height")[0].text), width=int(tree.findall("./size/width")[0].text), annotations=[])
        for obj in tree.findall("object"):
            bb = obj.find("bndbox")
            box = [float(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]
            box[0] -= 1.0  # VOC pixel indices are 1-based; boxes become 0-based half-open (the evaluator adds the 1 back)
            box[1] -= 1.0
            rec["annotations"].append(dict(category_id=class_names.index(obj.find("name").text), bbox=box, bbox_mode="XYXY_ABS"))
        dicts.append(rec)
    return dicts


def register_voc_coco(name: str, dirname: str, split: str, year: int) -> None:
    class_names = VOC_COCO_CATEGORIES
    DatasetCatalog.register(name, lambda: load_voc_instances(dirname, split, class_names))
    meta = MetadataCatalog.get(name)
    meta.thing_classes, meta.dirname, meta.year, meta.split = list(class_names), dirname, year, split
    meta.thing_dataset_id_to_contiguous_id = {i: i for i in range(len(class_names))}
    meta.evaluator_type = "pascal_voc"


def register_opendet_voc_coco(root: str) -> None:
    for split in VOC_COCO_SPLITS:
        register_voc_coco(split, os.path.join(root, "voc_coco"), split, 2007 if "2007" in split else 2012)


def register_builtin_pascal_voc(root: str) -> None:
    """The PASCAL VOC names the VOC-COCO yaml trains on and tests first ('voc_2007_train', 'voc_2012_trainval', 'voc_2007_test'):
    detectron2 pre-registers them ([d2] data/datasets/builtin.py register_all_pascal_voc) as <root>/VOC{2007,2012} with the 20
    VOC class names, which are the first 20 entries of VOC_COCO_CATEGORIES."""
    names = list(VOC_COCO_CATEGORIES[:20])
    for year, splits in ((2007, ("trainval", "train", "val", "test")), (2012, ("trainval", "train", "val"))):
        for split in splits:
            name, dirname = f"voc_{year}_{split}", os.path.join(root, f"VOC{year}")
            DatasetCatalog.register(name, lambda d=dirname, s=split: load_voc_instances(d, s, names))
            meta = MetadataCatalog.get(name)
            meta.thing_classes, meta.dirname, meta.year, meta.split, meta.evaluator_type = list(names), dirname, year, split, "pascal_voc"
            meta.thing_dataset_id_to_contiguous_id = {i: i for i in range(len(names))}


# ---------------------------------------------------------------------------------------------------------------
# GraspNet open-set splits (COCO-format json): openset_rcnn/data/graspnet.py, graspnet_meta.py, custom.py:9-30
# ---------------------------------------------------------------------------------------------------------------
# the 28 known object categories of the benchmark (graspnet_meta.py:92-98); the full 88-entry category table is read from the
# annotation json itself
GRASPNET_KNOWN_CATEGORIES = [
    "cracker_box", "tomato_soup_can", "banana", "mug", "power_drill", "scissors", "strawberry", "peach", "plum", "knife", "flat_screwdriver",
    "racquetball", "b_cups", "d_toy_airplane", "f_toy_airplane", "i_toy_airplane", "j_toy_airplane", "dabao_sod", "darlie_toothpaste", "camel",
    "large_elephant", "rhinocero", "darlie_box", "black_mouse", "dabao_facewash", "pantene", "head_shoulders_supreme", "head_shoulders_care",
]
GRASPNET_SPLITS = {"graspnet_train": "graspnet_os_train.json", **{f"graspnet_test_{i}": f"graspnet_os_test_{i}.json" for i in range(1, 7)}}


def load_coco_json(json_file: str, image_root: str, dataset_name: Optional[str] = None) -> List[dict]:
    """[d2] load_coco_json without pycocotools: one dict per image with file_name, height, width, image_id and annotations
    [{bbox XYWH_ABS, category_id (contiguous), iscrowd}]; records the category tables in the dataset's metadata."""
    import json
    with open(json_file) as f:
        data = json.load(f)
    cats = sorted(data["categories"], key=lambda c: c["id"])
    id_map = {c["id"]: i for i, c in enumerate(cats)}
    if dataset_name is not None:
        meta = MetadataCatalog.get(dataset_name)
        meta.thing_classes = [c["name"] for c in cats]
        meta.thing_dataset_id_to_contiguous_id = id_map
    by_img: Dict[int, List[dict]] = {}
    for a in data["annotations"]:
        by_img.setdefault(a["image_id"], []).append(a)
    out = []
    for im in sorted(data["images"], key=lambda r: r["id"]):
        rec = dict(file_name=os.path.join(image_root, im["file_name"]), height=im["height"], width=im["width"], image_id=im["id"], annotations=[])
        for a in by_img.get(im["id"], []):
            if a.get("ignore", 0):
                continue
            x, y, w, h = a["bbox"]
            rec["annotations"].append(dict(bbox=[x, y, x + w, y + h], bbox_mode="XYXY_ABS", category_id=id_map[a["category_id"]],
                                           iscrowd=a.get("iscrowd", 0)))
        out.append(rec)
    return out


def register_graspnet_instances(name: str, json_file: str, image_root: str) -> None:
    DatasetCatalog.register(name, lambda: load_coco_json(json_file, image_root, name))
    meta = MetadataCatalog.get(name)
    meta.json_file, meta.image_root, meta.evaluator_type = json_file, image_root, "coco"


def register_graspnet_os(root: str) -> None:
    for name, jf in GRASPNET_SPLITS.items():
        register_graspnet_instances(name, os.path.join(root, "graspnet_os", "annotations", jf), os.path.join(root, "graspnet_os", "images"))


def graspnet_class_map(thing_classes: Sequence[str]):
    """Sorted contiguous ids of the known categories: the `class_id` tensor PLN / SoftMaxClassifier build
    (prototype_learning_network.py:80-86); pass it as `class_id` to build_model / OpensetRCNNEngine(class_map=...)."""
    import torch
    return torch.tensor(sorted(thing_classes.index(n) for n in GRASPNET_KNOWN_CATEGORIES if n in thing_classes), dtype=torch.int64)


def get_evaluator(cfg, dataset_name: str, output_folder=None):
    """train.py:57-78: "pascal_voc" -> open-set VOC evaluator, "coco" -> open-set COCO-style evaluator (GraspNet)."""
    from .evaluation import PascalVOCDetectionEvaluator
    meta = MetadataCatalog.get(dataset_name)
    if getattr(meta, "evaluator_type", None) == "coco":
        from .os_coco_evaluation import OpensetCOCOEvaluator
        if not hasattr(meta, "thing_dataset_id_to_contiguous_id"):
            DatasetCatalog[dataset_name]()  # loading the json fills the category tables
        rev = {v: k for k, v in meta.thing_dataset_id_to_contiguous_id.items()}
        return OpensetCOCOEvaluator(meta.json_file, GRASPNET_KNOWN_CATEGORIES, rev, max_dets_per_image=[10, 20, 30, 50, 100], output_dir=output_folder)
    if getattr(meta, "evaluator_type", None) != "pascal_voc":
        raise NotImplementedError(f"no Evaluator for the dataset {dataset_name} with the type {getattr(meta, 'evaluator_type', None)}")
    return PascalVOCDetectionEvaluator(meta.dirname, meta.split, meta.thing_classes, cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES,
                                       output_folder if output_folder is not None else getattr(cfg, "OUTPUT_DIR", None))
