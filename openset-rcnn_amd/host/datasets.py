"""Dataset catalog + the VOC-COCO open-set splits (the data format either side of the hot path).

Mirrors /root/reference/openset_rcnn/data/voc_coco.py (category list :5-29, register_voc_coco :32-42) and data/custom.py
(register_opendet_voc_coco :32-51) on a minimal catalog of our own, plus [d2] load_voc_instances: one dict per image with
file_name, image_id, height, width and annotations [{category_id, bbox XYXY with xmin/ymin - 1}]."""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from types import SimpleNamespace
from typing import Callable, Dict, List, Sequence

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


def get_evaluator(cfg, dataset_name: str, output_folder=None):
    """train.py:57-78 for the pascal_voc evaluator type."""
    from .evaluation import PascalVOCDetectionEvaluator
    meta = MetadataCatalog.get(dataset_name)
    if getattr(meta, "evaluator_type", None) != "pascal_voc":
        raise NotImplementedError(f"no Evaluator for the dataset {dataset_name} with the type {getattr(meta, 'evaluator_type', None)}")
    return PascalVOCDetectionEvaluator(meta.dirname, meta.split, meta.thing_classes, cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES,
                                       output_folder if output_folder is not None else getattr(cfg, "OUTPUT_DIR", None))
