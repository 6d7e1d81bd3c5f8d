"""Open-set PASCAL-VOC evaluator -- the `mAP_k` half of the headline metric.

Own numpy implementation of the protocol in /root/reference/openset_rcnn/evaluation/pascal_voc_evaluation.py:
  * process (:53-70): per detection one text record "image score xmin ymin xmax ymax" with xmin/ymin + 1 (the inverse of
    the VOC loader's -1), score rounded to 3 decimals and coordinates to 1 decimal BEFORE scoring -- kept, because the
    rounding changes ranks and overlaps;
  * per class (:130-163, voc_eval :264-379): GT of the class from the XML annotations with every category outside the
    known set renamed "unknown"; detections sorted by descending confidence; greedy matching at IoU > 0.5 with the
    inclusive "+1" pixel box convention (:247-261); "difficult" GT neither counts nor penalises; a second match of the same
    GT is a false positive; precision/recall curves; VOC-2012 style AP (area under the monotone precision envelope,
    `_is_2007 = False` :41);
  * open-set extras: A-OSE = known-class detections that overlap an unknown GT at IoU > 0.5 (:350-377), WI = wilderness
    impact at recall 0.8 (:72-100): mean over known classes of open-set false positives / mean of closed-set TP+FP, each
    taken at the detection whose recall is closest to 0.8;
  * summary keys (:165-215): mAP (over ALL class names, empty classes count as 0), WI (x100), AOSE, AP@K / P@K / R@K over the
    known classes, AP@U / P@U / R@U for the last class ("unknown"), rounded to 2 decimals.
Multi-process runs gather the per-rank records on rank 0 ([d2] comm.gather :106) through host.parallel.

Tie order: like the reference, detections are ordered with np.argsort(-confidence) (not a stable sort); with scores rounded
to 3 decimals ties exist, and their order may differ between numpy builds. AP@K moves by < 0.01 in practice."""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import parallel

UNKNOWN_NAME = "unknown"


def voc_ap(rec: np.ndarray, prec: np.ndarray, use_07_metric: bool = False) -> float:
    """[d2] detectron2.evaluation.pascal_voc_evaluation.voc_ap: 11-point average (VOC07) or the exact area under the
    precision envelope (VOC10+)."""
    rec, prec = np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = float(np.max(prec[rec >= t])) if np.sum(rec >= t) > 0 else 0.0
            ap += p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]  # envelope: running max from the right
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def parse_voc_xml(path: str, known_classes: Sequence[str]) -> List[dict]:
    """Objects of one annotation file; categories outside `known_classes` become "unknown" (:230-232)."""
    known = set(known_classes)
    out = []
    for obj in ET.parse(path).findall("object"):
        name = obj.find("name").text
        bb = obj.find("bndbox")
        out.append(dict(name=name if name in known else UNKNOWN_NAME, difficult=int(obj.find("difficult").text),
                        bbox=[int(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]))
    return out


def _overlaps(gt: np.ndarray, bb: np.ndarray) -> np.ndarray:
    """IoU of one detection against (G,4) GT boxes with the inclusive-pixel (+1) convention (:247-261)."""
    iw = np.maximum(np.minimum(gt[:, 2], bb[2]) - np.maximum(gt[:, 0], bb[0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(gt[:, 3], bb[3]) - np.maximum(gt[:, 1], bb[1]) + 1.0, 0.0)
    inter = iw * ih
    union = (bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (gt[:, 2] - gt[:, 0] + 1.0) * (gt[:, 3] - gt[:, 1] + 1.0) - inter
    return inter / union


def _class_gt(annos: Dict[str, List[dict]], image_ids: Sequence[str], name: str):
    per_image, npos = {}, 0
    for im in image_ids:
        objs = [o for o in annos[im] if o["name"] == name]
        bbox = np.array([o["bbox"] for o in objs], dtype=np.float64).reshape(-1, 4)
        difficult = np.array([o["difficult"] for o in objs], dtype=bool)
        npos += int(np.sum(~difficult))
        per_image[im] = (bbox, difficult, np.zeros(len(objs), dtype=bool))
    return per_image, npos


def voc_eval(records: Sequence[str], annos: Dict[str, List[dict]], image_ids: Sequence[str], classname: str, ovthresh: float = 0.5,
             use_07_metric: bool = False):
    """One class (:264-379). records: "image score xmin ymin xmax ymax" strings. Returns (rec, prec, ap, unknown dets taken
    as this class, number of unknown GT, closed-set TP+FP curve, open-set FP curve); the last two are None for "unknown"."""
    gt, npos = _class_gt(annos, image_ids, classname)
    rows = [r.strip().split(" ") for r in records if r.strip()]
    det_images = [r[0] for r in rows]
    conf = np.array([float(r[1]) for r in rows], dtype=np.float64)
    bb = np.array([[float(z) for z in r[2:]] for r in rows], dtype=np.float64).reshape(-1, 4)
    order = np.argsort(-conf)
    bb = bb[order]
    det_images = [det_images[i] for i in order]
    nd = len(det_images)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        boxes, difficult, taken = gt[det_images[d]]
        ovmax, jmax = -np.inf, -1
        if boxes.size:
            ov = _overlaps(boxes, bb[d])
            jmax = int(np.argmax(ov))
            ovmax = ov[jmax]
        if ovmax > ovthresh:
            if not difficult[jmax]:
                if not taken[jmax]:
                    tp[d] = 1.0
                    taken[jmax] = True
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    with np.errstate(divide="ignore", invalid="ignore"):
        rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = voc_ap(rec, prec, use_07_metric)
    unk, n_unk = _class_gt(annos, image_ids, UNKNOWN_NAME)
    if classname == UNKNOWN_NAME:
        return rec, prec, ap, 0, n_unk, None, None
    is_unk = np.zeros(nd)
    for d in range(nd):
        boxes = unk[det_images[d]][0]
        if boxes.size and np.max(_overlaps(boxes, bb[d])) > ovthresh:
            is_unk[d] = 1.0
    return rec, prec, ap, float(np.sum(is_unk)), n_unk, tp + fp, np.cumsum(is_unk)


def wilderness_impact(recalls: List[np.ndarray], tp_plus_fp: List[Optional[np.ndarray]], fp_open: List[Optional[np.ndarray]],
                      num_known: int, recall_level: float = 0.8) -> float:
    """compute_WI_at_a_recall_level (:84-100) at one IoU threshold."""
    tpfp, fps = [], []
    for cls_id, rec in enumerate(recalls):
        if cls_id < num_known and len(rec) > 0:
            idx = int(np.argmin(np.abs(np.asarray(rec) - recall_level)))  # first closest, as min(range, key=...)
            tpfp.append(tp_plus_fp[cls_id][idx])
            fps.append(fp_open[cls_id][idx])
    return float(np.mean(fps) / np.mean(tpfp)) if tpfp else 0.0


class PascalVOCDetectionEvaluator:
    """DatasetEvaluator-shaped (reset / process / evaluate) open-set VOC evaluator (pascal_voc_evaluation.py:21-215).

    dirname holds Annotations/<id>.xml and ImageSets/Main/<split>.txt; class_names are the dataset's thing classes with
    "unknown" last (openset_rcnn/data/voc_coco.py:5-29); predicted class ids index into class_names (unknown id =
    len(class_names) - 1 = 80 under --opendet-benchmark)."""

    def __init__(self, dirname: str, split: str, class_names: Sequence[str], num_known_classes: int, output_dir: Optional[str] = None,
                 annotations: Optional[Dict[str, List[dict]]] = None, image_ids: Optional[Sequence[str]] = None):
        self._class_names = list(class_names)
        self.num_known_classes = int(num_known_classes)
        self.known_classes = self._class_names[: self.num_known_classes]
        self.total_num_class = len(self._class_names)
        self.unknown_class_index = self.total_num_class - 1
        self.output_dir = output_dir
        self._is_2007 = False
        if annotations is None:
            with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
                image_ids = [x.strip() for x in f.readlines()]
            annotations = {im: parse_voc_xml(os.path.join(dirname, "Annotations", im + ".xml"), self.known_classes) for im in image_ids}
        self._annos = annotations
        self._image_ids = list(image_ids if image_ids is not None else annotations.keys())
        self.reset()

    def reset(self) -> None:
        self._predictions: Dict[int, List[str]] = defaultdict(list)

    def process(self, inputs: Sequence[dict], outputs: Sequence[dict]) -> None:
        for inp, out in zip(inputs, outputs):
            inst = out["instances"]
            boxes = inst.pred_boxes.tensor.detach().cpu().numpy()
            scores = inst.scores.detach().cpu().tolist()
            classes = inst.pred_classes.detach().cpu().tolist()
            for box, score, cls in zip(boxes, scores, classes):
                xmin, ymin, xmax, ymax = box
                self._predictions[int(cls)].append(f"{inp['image_id']} {score:.3f} {xmin + 1:.1f} {ymin + 1:.1f} {xmax:.1f} {ymax:.1f}")

    def evaluate(self) -> Optional[Dict[str, float]]:
        gathered = parallel.gather_to_rank0(dict(self._predictions))
        if gathered is None:
            return None
        predictions: Dict[int, List[str]] = defaultdict(list)
        for per_rank in gathered:
            for cls_id, lines in per_rank.items():
                predictions[cls_id].extend(lines)
        if self.output_dir is not None:  # the reference leaves one <class>.txt per class behind (:121-136)
            d = os.path.join(self.output_dir, "pascal_voc_eval")
            os.makedirs(d, exist_ok=True)
            for cls_id, name in enumerate(self._class_names):
                with open(os.path.join(d, name + ".txt"), "w") as f:
                    f.write("\n".join(predictions.get(cls_id, [""])))
        aps, recs, precs, all_recs, aose, tpfp, fpo = [], [], [], [], [], [], []
        for cls_id, name in enumerate(self._class_names):
            rec, prec, ap, unk_as_known, _, tp_plus_fp, fp_open = voc_eval(predictions.get(cls_id, []), self._annos, self._image_ids, name,
                                                                          0.5, self._is_2007)
            aps.append(ap * 100)
            aose.append(unk_as_known)
            all_recs.append(rec)
            tpfp.append(tp_plus_fp)
            fpo.append(fp_open)
            recs.append(rec[-1] * 100 if len(rec) else 0)
            precs.append(prec[-1] * 100 if len(prec) else 0)
        k = self.num_known_classes
        res = {"mAP": np.mean(aps), "WI": wilderness_impact(all_recs, tpfp, fpo, k, 0.8) * 100, "AOSE": np.sum(aose),
               "AP@K": np.mean(aps[:k]), "P@K": np.mean(precs[:k]), "R@K": np.mean(recs[:k]),
               "AP@U": aps[-1], "P@U": precs[-1], "R@U": recs[-1]}
        return {key: round(float(v), 2) for key, v in res.items()}


def inference_on_dataset(model, batches, evaluator, rank: Optional[int] = None, world: Optional[int] = None):
    """[d2] inference_on_dataset as train.py:96 drives it: every rank runs the model over its share of the batches and feeds
    the evaluator; evaluate() gathers on rank 0 (None elsewhere). `batches` is a sequence of list[dict] inputs (each dict with
    "image", "image_id", optionally "height"/"width"); batch i goes to rank i % world -- images are independent, there is no
    data-path collective."""
    if rank is None or world is None:
        rank, world = parallel.world_info()
    evaluator.reset()
    for i, batch in enumerate(batches):
        if i % world != rank:
            continue
        evaluator.process(batch, model(batch))
    parallel.barrier()
    return evaluator.evaluate()
