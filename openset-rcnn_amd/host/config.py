"""yacs-style configuration for the hot path: `get_cfg()` (the detectron2 defaults the reference's yaml files touch),
`add_openset_rcnn_config()` (the keys added by /root/reference/openset_rcnn/config/config.py:6-43) and a CfgNode that
loads the reference's `configs/*.yaml` unchanged (`_BASE_` inheritance, `KEY VALUE` list overrides, freeze)."""
from __future__ import annotations

import ast
import copy
import os
from typing import Any, List

import yaml


class CfgNode(dict):
    """Nested attribute dict. New keys may only be introduced in code (defaults), not by merged files."""

    def __init__(self, init: dict = None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError(f"Attempted to set {name} on a frozen CfgNode")
        self[name] = value

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def is_frozen(self):
        return self._frozen

    def clone(self):
        c = CfgNode()
        for k, v in self.items():
            c[k] = v.clone() if isinstance(v, CfgNode) else copy.deepcopy(v)
        return c

    @staticmethod
    def _decode(v: Any) -> Any:
        """yaml gives tuples like ("a",) and (1, 2) as strings: evaluate literals the way yacs does."""
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @staticmethod
    def _coerce(new, old, key):
        if old is None or isinstance(new, type(old)):
            return new
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        raise ValueError(f"Type mismatch for config key {key}: {type(old).__name__} vs {type(new).__name__} ({new!r})")

    def _merge(self, other: dict, path: str):
        for k, v in other.items():
            full = f"{path}.{k}" if path else k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"Config key {full} is a section")
                self[k]._merge(v, full)
            else:
                self[k] = self._coerce(self._decode(v), self[k], full)

    @staticmethod
    def load_yaml_with_base(path: str) -> dict:
        with open(path) as f:
            cfg = yaml.safe_load(f) or {}
        base = cfg.pop("_BASE_", None)
        if base is not None:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(path), base)
            merged = CfgNode.load_yaml_with_base(base)

            def rec(a, b):
                for k, v in b.items():
                    if isinstance(v, dict) and isinstance(a.get(k), dict):
                        rec(a[k], v)
                    else:
                        a[k] = v
            rec(merged, cfg)
            return merged
        return cfg

    def merge_from_file(self, path: str):
        if self._frozen:
            raise AttributeError("cfg is frozen")
        self._merge(self.load_yaml_with_base(path), "")

    def merge_from_list(self, opts: List[Any]):
        if self._frozen:
            raise AttributeError("cfg is frozen")
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            node[parts[-1]] = self._coerce(self._decode(val), node[parts[-1]], key)


CN = CfgNode


def get_cfg() -> CfgNode:
    """The subset of detectron2 v0.6 defaults ([d2]) that the hot path and the reference's yaml files read."""
    c = CN()
    c.VERSION = 2
    c.OUTPUT_DIR = "./output"
    c.SEED = -1
    c.MODEL = CN()
    m = c.MODEL
    m.META_ARCHITECTURE = "GeneralizedRCNN"
    m.DEVICE = "cuda"
    m.WEIGHTS = ""
    m.MASK_ON = False
    m.KEYPOINT_ON = False
    m.LOAD_PROPOSALS = False
    m.PIXEL_MEAN = [103.530, 116.280, 123.675]
    m.PIXEL_STD = [1.0, 1.0, 1.0]
    m.BACKBONE = CN({"NAME": "build_resnet_backbone", "FREEZE_AT": 2})
    m.RESNETS = CN({"DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN", "WIDTH_PER_GROUP": 64,
                    "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64})
    m.FPN = CN({"IN_FEATURES": [], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"})
    m.ANCHOR_GENERATOR = CN({"NAME": "DefaultAnchorGenerator", "SIZES": [[32, 64, 128, 256, 512]],
                             "ASPECT_RATIOS": [[0.5, 1.0, 2.0]], "ANGLES": [[-90, 0, 90]], "OFFSET": 0.0})
    m.PROPOSAL_GENERATOR = CN({"NAME": "RPN", "MIN_SIZE": 0})
    m.RPN = CN({"HEAD_NAME": "StandardRPNHead", "IN_FEATURES": ["res4"], "BOUNDARY_THRESH": -1, "IOU_THRESHOLDS": [0.3, 0.7],
                "IOU_LABELS": [0, -1, 1], "BATCH_SIZE_PER_IMAGE": 256, "POSITIVE_FRACTION": 0.5, "BBOX_REG_LOSS_TYPE": "smooth_l1",
                "BBOX_REG_LOSS_WEIGHT": 1.0, "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0), "SMOOTH_L1_BETA": 0.0, "LOSS_WEIGHT": 1.0,
                "PRE_NMS_TOPK_TRAIN": 12000, "PRE_NMS_TOPK_TEST": 6000, "POST_NMS_TOPK_TRAIN": 2000, "POST_NMS_TOPK_TEST": 1000,
                "NMS_THRESH": 0.7, "CONV_DIMS": [-1]})
    m.ROI_HEADS = CN({"NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["res4"], "IOU_THRESHOLDS": [0.5], "IOU_LABELS": [0, 1],
                      "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25, "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5,
                      "PROPOSAL_APPEND_GT": True})
    m.ROI_BOX_HEAD = CN({"NAME": "", "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
                         "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "SMOOTH_L1_BETA": 0.0, "POOLER_RESOLUTION": 14,
                         "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "NUM_FC": 0, "FC_DIM": 1024, "NUM_CONV": 0,
                         "CONV_DIM": 256, "NORM": "", "CLS_AGNOSTIC_BBOX_REG": False, "TRAIN_ON_PRED_BOXES": False})
    m.ROI_MASK_HEAD = CN({"NAME": "MaskRCNNConvUpsampleHead", "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "NUM_CONV": 0,
                          "CONV_DIM": 256, "NORM": "", "CLS_AGNOSTIC_MASK": False, "POOLER_TYPE": "ROIAlignV2"})
    c.INPUT = CN({"MIN_SIZE_TRAIN": (800,), "MIN_SIZE_TRAIN_SAMPLING": "choice", "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800,
                  "MAX_SIZE_TEST": 1333, "FORMAT": "BGR", "RANDOM_FLIP": "horizontal"})
    c.DATASETS = CN({"TRAIN": (), "TEST": (), "PROPOSAL_FILES_TRAIN": (), "PROPOSAL_FILES_TEST": ()})
    c.DATALOADER = CN({"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True, "SAMPLER_TRAIN": "TrainingSampler", "FILTER_EMPTY_ANNOTATIONS": True})
    c.SOLVER = CN({"LR_SCHEDULER_NAME": "WarmupMultiStepLR", "MAX_ITER": 40000, "BASE_LR": 0.001, "MOMENTUM": 0.9, "NESTEROV": False,
                   "WEIGHT_DECAY": 0.0001, "WEIGHT_DECAY_NORM": 0.0, "GAMMA": 0.1, "STEPS": (30000,), "WARMUP_FACTOR": 1.0 / 1000,
                   "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear", "CHECKPOINT_PERIOD": 5000, "IMS_PER_BATCH": 16,
                   "BIAS_LR_FACTOR": 1.0, "WEIGHT_DECAY_BIAS": 0.0001})
    c.TEST = CN({"EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100})
    return c


def add_openset_rcnn_config(cfg: CfgNode) -> None:
    """Keys the reference adds on top of detectron2 (/root/reference/openset_rcnn/config/config.py:6-43), same defaults."""
    cfg.OPENDET_BENCHMARK = False
    rpn = cfg.MODEL.RPN
    rpn.CTR_REG_LOSS_WEIGHT = 1.0
    rpn.CTR_REG_LOSS_TYPE = "smooth_l1"
    rpn.CTR_SMOOTH_L1_BETA = 0.0
    rpn.IOU_THRESHOLDS_OBJECTNESS = [0.1, 0.3]
    rpn.POSITIVE_FRACTION_OBJECTNESS = 1.0
    rpn.NMS_THRESH_TEST = 1.0
    bh = cfg.MODEL.ROI_BOX_HEAD
    bh.IOU_REG_LOSS_WEIGHT = 1.0
    bh.IOU_REG_LOSS_TYPE = "smooth_l1"
    bh.IOU_SMOOTH_L1_BETA = 0.0
    bh.CLS_LOSS_WEIGHT = 1.0
    rh = cfg.MODEL.ROI_HEADS
    rh.MEAN_TYPE = "geometric"
    rh.OBJ_SCORE_THRESH_TEST = 0.05
    rh.NUM_KNOWN_CLASSES = 20
    rh.KNOWN_SCORE_THRESH = 0.05
    rh.KNOWN_NMS_THRESH = 0.5
    rh.KNOWN_TOPK = 1000
    rh.UNKNOWN_SCORE_THRESH = 0.05
    rh.UNKNOWN_NMS_THRESH = 0.5
    rh.UNKNOWN_TOPK = 1000
    rh.UNKNOWN_ID = 1000
    cfg.MODEL.PLN = CN({"EMD_DIM": 256, "DISTANCE_TYPE": "COS", "REPS_PER_CLASS": 1, "ALPHA": 0.1, "BETA": 0.9,
                        "IOU_THRESHOLD": 0.5, "UNK_THR": 0.4, "LOSS_WEIGHT": 2.0})
