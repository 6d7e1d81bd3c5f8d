"""Open-set COCO-style evaluator of the GraspNet benchmark, bounding boxes only.

Own numpy implementation of the protocol in /root/reference/openset_rcnn/evaluation/os_cocoeval.py (OpensetCOCOEval, a subclass
of pycocotools' COCOeval -- not installed here, so the inherited parts are restated too) and of the result derivation in
os_coco_evaluation.py:336-440:

  * ground truth: every annotation whose category is not one of the known ids becomes category 1000 ("unknown",
    os_coco_evaluation.py:603-605); detections carry dataset category ids, 1000 for the unknown class;
  * per image and known category c, detections of c (score-descending, stable, at most maxDets[-1]) are matched greedily,
    COCO style (a detection takes the best still-free GT with IoU >= t; ignored GT sorted last; crowd GT can be matched
    repeatedly), three times: against the GT of c (true positives), against the GT of the OTHER known categories and against the
    unknown GT (os_cocoeval.py:242-424). Unknown detections are matched against all known GT and against the unknown GT
    (:426-555). IoU as pycocotools' maskUtils.iou on xywh boxes (intersection / union, or / detection area for crowd GT);
  * accumulation (:557-787): per IoU threshold, category, area range and maxDets the precision envelope sampled at 101 recall
    thresholds, recall, and the open-set counters: known detections matched to unknown GT (A-OSE), to other-known GT, unknown
    detections matched to known GT, and closed-set TP+FP / open-set FP at each recall threshold (wilderness impact);
  * summary (:789-972): 30 numbers -- known AP / AP50 / AP75 / APs / APm / APl, AR@10/20/30/50/100, ARs / ARm / ARl, WI at
    recall 0.8 and IoU 0.5, A-OSE at IoU 0.5, then the same 14 AP/AR numbers for the unknown class.
"""
from __future__ import annotations

import copy
import json
import os
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import parallel

UNKNOWN_CAT = 1000


def box_iou_xywh(d: np.ndarray, g: np.ndarray, iscrowd: Sequence[int]) -> np.ndarray:
    """pycocotools maskUtils.iou for boxes: (D,4) x (G,4) xywh -> (D,G); crowd columns use the detection's area as denominator."""
    d, g = np.asarray(d, dtype=np.float64).reshape(-1, 4), np.asarray(g, dtype=np.float64).reshape(-1, 4)
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    dx2, dy2, gx2, gy2 = d[:, 0] + d[:, 2], d[:, 1] + d[:, 3], g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    w = np.clip(np.minimum(dx2[:, None], gx2[None]) - np.maximum(d[:, 0][:, None], g[:, 0][None]), 0, None)
    h = np.clip(np.minimum(dy2[:, None], gy2[None]) - np.maximum(d[:, 1][:, None], g[:, 1][None]), 0, None)
    inter = w * h
    da, ga = (d[:, 2] * d[:, 3])[:, None], (g[:, 2] * g[:, 3])[None]
    crowd = np.asarray(iscrowd, dtype=bool)[None]
    union = np.where(crowd, da, da + ga - inter)
    return np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)


def _greedy_match(ious: np.ndarray, dt_ids, gt_ids, gt_ignore: np.ndarray, gt_crowd, iou_thrs) -> tuple:
    """COCOeval.evaluateImg inner loops: returns (dtm (T,D) matched gt id or 0, gtm (T,G), dtIg (T,D))."""
    T, D, G = len(iou_thrs), len(dt_ids), len(gt_ids)
    dtm, gtm, dtig = np.zeros((T, D)), np.zeros((T, G)), np.zeros((T, D))
    if ious.size == 0:
        return dtm, gtm, dtig
    for ti, t in enumerate(iou_thrs):
        for di in range(D):
            iou, m = min(t, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[ti, gi] > 0 and not gt_crowd[gi]:
                    continue
                if m > -1 and gt_ignore[m] == 0 and gt_ignore[gi] == 1:
                    break
                if ious[di, gi] < iou:
                    continue
                iou, m = ious[di, gi], gi
            if m == -1:
                continue
            dtig[ti, di] = gt_ignore[m]
            dtm[ti, di] = gt_ids[m]
            gtm[ti, m] = dt_ids[di]
    return dtm, gtm, dtig


class OpensetCOCOEval:
    def __init__(self, gt_dataset: dict, detections: List[dict], known_cat_ids: Sequence[int], max_dets: Sequence[int] = (10, 20, 30, 50, 100),
                 img_ids: Optional[Sequence] = None):
        self.cat_ids = sorted(int(c) for c in known_cat_ids)
        known = set(self.cat_ids)
        self.max_dets = sorted(max_dets)
        self.iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.area_rng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.area_lbl = ["all", "small", "medium", "large"]
        self.img_ids = sorted(set(img_ids)) if img_ids is not None else sorted({im["id"] for im in gt_dataset["images"]})
        self._k_gts, self._ok_gts, self._unk_gts = defaultdict(list), defaultdict(list), defaultdict(list)
        self._k_dts, self._unk_dts = defaultdict(list), defaultdict(list)
        imgs = set(self.img_ids)
        k_gts = []
        for a in gt_dataset["annotations"]:
            if a["image_id"] not in imgs:
                continue
            a = dict(a)
            a["category_id"] = a["category_id"] if a["category_id"] in known else UNKNOWN_CAT
            a["iscrowd"] = int(a.get("iscrowd", 0))
            a["ignore"] = a["iscrowd"]
            a.setdefault("area", a["bbox"][2] * a["bbox"][3])
            if a["category_id"] == UNKNOWN_CAT:
                self._unk_gts[a["image_id"]].append(a)
            else:
                self._k_gts[a["image_id"], a["category_id"]].append(a)
                k_gts.append(a)
        for c in self.cat_ids:
            for a in k_gts:
                if a["category_id"] != c:
                    self._ok_gts[a["image_id"], c].append(a)
        for i, d in enumerate(detections):  # COCO.loadRes for box results: area = w*h, id = running index, iscrowd = 0
            if d["image_id"] not in imgs:
                continue
            d = dict(d)
            d["area"] = d["bbox"][2] * d["bbox"][3]
            d["id"] = i + 1
            d["iscrowd"] = 0
            if d["category_id"] == UNKNOWN_CAT:
                self._unk_dts[d["image_id"]].append(d)
            elif d["category_id"] in known:
                self._k_dts[d["image_id"], d["category_id"]].append(d)
        self.eval_kdt: dict = {}
        self.eval_unkdt: dict = {}
        self.stats = None

    # ---- per image ----------------------------------------------------------------------------------------
    @staticmethod
    def _sorted_dets(dt, max_det):
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")
        return [dt[i] for i in order[:max_det]]

    def _ious(self, dt, gt):
        if len(gt) == 0 and len(dt) == 0:
            return np.zeros((0, 0))
        dt = self._sorted_dets(dt, self.max_dets[-1])
        return box_iou_xywh([d["bbox"] for d in dt], [g["bbox"] for g in gt], [g["iscrowd"] for g in gt])

    def _one(self, dt_sorted, gt, ious_full, a_rng):
        """Match one detection list against one GT list for an area range: (dtm, dtIg incl. the out-of-range rule, gtIg)."""
        ig = np.array([1 if (g["ignore"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gt], dtype=np.int64)
        order = np.argsort(ig, kind="mergesort")
        gt = [gt[i] for i in order]
        ig = ig[order] if len(ig) else ig
        ious = ious_full[:, order] if ious_full.size else ious_full
        dtm, gtm, dtig = _greedy_match(ious[: len(dt_sorted)] if ious.size else ious, [d["id"] for d in dt_sorted], [g["id"] for g in gt], ig,
                                       [g["iscrowd"] for g in gt], self.iou_thrs)
        out = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt_sorted]).reshape(1, len(dt_sorted))
        dtig = np.logical_or(dtig, np.logical_and(dtm == 0, np.repeat(out, len(self.iou_thrs), 0)))
        return dtm, dtig, ig

    def evaluate(self):
        max_det = self.max_dets[-1]
        self.eval_imgs_kdt, self.eval_imgs_unkdt = [], []
        for c in self.cat_ids:
            for a_rng in self.area_rng:
                for im in self.img_ids:
                    k_gt, ok_gt, u_gt, k_dt = self._k_gts[im, c], self._ok_gts[im, c], self._unk_gts[im], self._k_dts[im, c]
                    dts = self._sorted_dets(k_dt, max_det)
                    m_k, ig_k, gig_k = self._one(dts, k_gt, self._ious(k_dt, k_gt), a_rng)
                    m_ok, ig_ok, _ = self._one(dts, ok_gt, self._ious(k_dt, ok_gt), a_rng)
                    m_u, ig_u, _ = self._one(dts, u_gt, self._ious(k_dt, u_gt), a_rng)
                    self.eval_imgs_kdt.append(dict(scores=[d["score"] for d in dts], m_k=m_k, m_ok=m_ok, m_u=m_u, ig_k=ig_k, ig_ok=ig_ok, ig_u=ig_u,
                                                   gt_ig=gig_k))
        for a_rng in self.area_rng:
            for im in self.img_ids:
                k_gt = [g for c in self.cat_ids for g in self._k_gts[im, c]]
                u_gt, u_dt = self._unk_gts[im], self._unk_dts[im]
                if len(u_gt) == 0 and len(u_dt) == 0:
                    self.eval_imgs_unkdt.append(None)
                    continue
                dts = self._sorted_dets(u_dt, max_det)
                m_k, ig_k, _ = self._one(dts, k_gt, self._ious(u_dt, k_gt), a_rng)
                m_u, ig_u, gig_u = self._one(dts, u_gt, self._ious(u_dt, u_gt), a_rng)
                self.eval_imgs_unkdt.append(dict(scores=[d["score"] for d in dts], m_k=m_k, m_u=m_u, ig_k=ig_k, ig_u=ig_u, gt_ig=gig_u))

    # ---- accumulation -------------------------------------------------------------------------------------
    def _pr(self, tp, fp, npig, scores_sorted):
        """Precision envelope sampled at the recall thresholds (COCOeval.accumulate inner block)."""
        R = len(self.rec_thrs)
        nd = len(tp)
        rc = tp / npig
        pr = (tp / (fp + tp + np.spacing(1))).tolist()
        q, ss = [0.0] * R, [0.0] * R
        for i in range(nd - 1, 0, -1):
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        inds = np.searchsorted(rc, self.rec_thrs, side="left")
        for ri, pi in enumerate(inds):
            if pi >= nd:
                break
            q[ri] = pr[pi]
            ss[ri] = scores_sorted[pi]
        return rc, np.array(q), np.array(ss), inds

    def accumulate(self):
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(self.area_rng), len(self.max_dets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        ok_as_k, unk_as_k = np.zeros((T, K, A, M)), np.zeros((T, K, A, M))
        fp_os, tp_plus_fp_cs = np.zeros((T, R, K, A, M)), np.zeros((T, R, K, A, M))
        I = len(self.img_ids)
        for k in range(K):
            for a in range(A):
                E = self.eval_imgs_kdt[(k * A + a) * I:(k * A + a + 1) * I]
                for m, max_det in enumerate(self.max_dets):
                    sc = np.concatenate([np.asarray(e["scores"][:max_det], dtype=np.float64) for e in E]) if E else np.zeros(0)
                    inds = np.argsort(-sc, kind="mergesort")
                    scs = sc[inds]
                    cat = lambda key: np.concatenate([e[key][:, :max_det] for e in E], axis=1)[:, inds]  # noqa: E731
                    m_k, m_ok, m_u, ig_k, ig_ok, ig_u = cat("m_k"), cat("m_ok"), cat("m_u"), cat("ig_k"), cat("ig_ok"), cat("ig_u")
                    npig = int(np.count_nonzero(np.concatenate([e["gt_ig"] for e in E]) == 0))
                    if npig == 0:
                        continue
                    tps = np.logical_and(m_k, np.logical_not(ig_k))
                    fps = np.logical_and(np.logical_not(m_k), np.logical_not(ig_k))
                    okfps = np.logical_and(m_ok, np.logical_not(ig_ok))
                    ufps = np.logical_and(m_u, np.logical_not(ig_u))
                    tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float)
                    ufp_sum = np.cumsum(ufps, axis=1).astype(float)
                    for t in range(T):
                        tp, fp, ufp = tp_sum[t], fp_sum[t], ufp_sum[t]
                        if len(ufp):
                            unk_as_k[t, k, a, m] = ufp[-1]
                        ok_as_k[t, k, a, m] = float(np.sum(okfps[t]))
                        rc, q, ss, pinds = self._pr(tp, fp, npig, scs)
                        recall[t, k, a, m] = rc[-1] if len(tp) else 0
                        precision[t, :, k, a, m], scores[t, :, k, a, m] = q, ss
                        n = len(tp)
                        if n:
                            tp_fp = tp + fp
                            for ri, pi in enumerate(pinds):
                                pi = pi - 1 if pi == n else pi
                                tp_plus_fp_cs[t, ri, k, a, m] = tp_fp[pi]
                                fp_os[t, ri, k, a, m] = ufp[pi]
        self.eval_kdt = dict(precision=precision, recall=recall, scores=scores, ok_det_as_known=ok_as_k, unk_det_as_known=unk_as_k,
                             tp_plus_fp_cs=tp_plus_fp_cs, fp_os=fp_os)
        precision, recall, scores = -np.ones((T, R, A, M)), -np.ones((T, A, M)), -np.ones((T, R, A, M))
        k_as_unk = np.zeros((T, A, M))
        for a in range(A):
            E = [e for e in self.eval_imgs_unkdt[a * I:(a + 1) * I] if e is not None]
            if not E:
                continue
            for m, max_det in enumerate(self.max_dets):
                sc = np.concatenate([np.asarray(e["scores"][:max_det], dtype=np.float64) for e in E])
                inds = np.argsort(-sc, kind="mergesort")
                scs = sc[inds]
                cat = lambda key: np.concatenate([e[key][:, :max_det] for e in E], axis=1)[:, inds]  # noqa: E731
                m_k, m_u, ig_k, ig_u = cat("m_k"), cat("m_u"), cat("ig_k"), cat("ig_u")
                npig = int(np.count_nonzero(np.concatenate([e["gt_ig"] for e in E]) == 0))
                if npig == 0:
                    continue
                tps = np.logical_and(m_u, np.logical_not(ig_u))
                fps = np.logical_and(np.logical_not(m_u), np.logical_not(ig_u))
                kfps = np.logical_and(m_k, np.logical_not(ig_k))
                tp_sum, fp_sum, kfp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float), np.cumsum(kfps, axis=1).astype(float)
                for t in range(T):
                    if kfp_sum.shape[1]:
                        k_as_unk[t, a, m] = kfp_sum[t][-1]
                    rc, q, ss, _ = self._pr(tp_sum[t], fp_sum[t], npig, scs)
                    recall[t, a, m] = rc[-1] if len(tp_sum[t]) else 0
                    precision[t, :, a, m], scores[t, :, a, m] = q, ss
        self.eval_unkdt = dict(precision=precision, recall=recall, scores=scores, k_det_as_unk=k_as_unk)

    # ---- summary ------------------------------------------------------------------------------------------
    def summarize(self) -> np.ndarray:
        ai = {lbl: i for i, lbl in enumerate(self.area_lbl)}
        mi = {m: i for i, m in enumerate(self.max_dets)}

        def mean_valid(s):
            return -1 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

        def summ(ev, unknown, ap, iou=None, area="all", max_dets=100):
            s = ev["precision"] if ap else ev["recall"]
            if iou is not None:
                s = s[np.where(np.isclose(self.iou_thrs, iou))[0]]
            s = s[..., ai[area], mi[max_dets]]
            return mean_valid(s)

        last = self.max_dets[-1]
        st = np.zeros(30)
        for base, ev, unk in ((0, self.eval_kdt, False), (16, self.eval_unkdt, True)):
            st[base + 0] = summ(ev, unk, 1, max_dets=100 if 100 in mi else last)
            st[base + 1] = summ(ev, unk, 1, iou=.5, max_dets=last)
            st[base + 2] = summ(ev, unk, 1, iou=.75, max_dets=last)
            for j, area in enumerate(("small", "medium", "large")):
                st[base + 3 + j] = summ(ev, unk, 1, area=area, max_dets=last)
                st[base + 11 + j] = summ(ev, unk, 0, area=area, max_dets=last)
            for j in range(5):
                st[base + 6 + j] = summ(ev, unk, 0, max_dets=self.max_dets[j]) if j < len(self.max_dets) else -1
        t5, r8 = int(np.where(np.isclose(self.iou_thrs, .5))[0][0]), int(np.where(np.isclose(self.rec_thrs, .8))[0][0])
        a_all, m100 = ai["all"], mi.get(100, len(self.max_dets) - 1)
        tpfp, fpo = self.eval_kdt["tp_plus_fp_cs"][t5, r8, :, a_all, m100], self.eval_kdt["fp_os"][t5, r8, :, a_all, m100]
        with np.errstate(divide="ignore", invalid="ignore"):
            st[14] = np.mean(fpo) / np.mean(tpfp)
        st[15] = float(np.sum(self.eval_kdt["unk_det_as_known"][t5, :, a_all, m100]))
        self.stats = st
        self.k_det_as_unk = float(self.eval_unkdt["k_det_as_unk"][t5, a_all, m100])
        return st


METRICS = ["AP", "AP50", "AP75", "APs", "APm", "APl", "AR10", "AR20", "AR30", "AR50", "AR100", "ARs", "ARm", "ARl"]


def derive_results(stats: np.ndarray) -> Dict[str, Dict[str, float]]:
    """os_coco_evaluation.py:336-440 for eval_type "openset": percentages for known and unknown, WI and A-OSE as raw numbers.
    (The reference tests stats[idx] >= 0 -- the KNOWN entry -- when it formats the unknown entry idx + 16; kept.)"""
    known = {m: float(stats[i] * 100 if stats[i] >= 0 else "nan") for i, m in enumerate(METRICS)}
    known["WI"], known["AOSE"] = float(stats[14]), float(stats[15])
    unknown = {m: float(stats[i + 16] * 100 if stats[i] >= 0 else "nan") for i, m in enumerate(METRICS)}
    return {"bbox": known, "bbox_unknown": unknown}


def instances_to_coco_json(instances, img_id, reverse_id_map: Optional[Dict[int, int]] = None) -> List[dict]:
    """[d2] instances_to_coco_json for boxes: XYXY -> XYWH, contiguous class id -> dataset id (1000 stays 1000)."""
    boxes = instances.pred_boxes.tensor.detach().cpu().numpy().astype(np.float64)
    scores = instances.scores.detach().cpu().tolist()
    classes = instances.pred_classes.detach().cpu().tolist()
    out = []
    for b, s, c in zip(boxes, scores, classes):
        c = int(c)
        if reverse_id_map is not None and c != UNKNOWN_CAT:
            c = reverse_id_map[c]
        out.append(dict(image_id=img_id, category_id=c, bbox=[float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])], score=float(s)))
    return out


class OpensetCOCOEvaluator:
    """DatasetEvaluator-shaped wrapper (os_coco_evaluation.py:31-300): reset / process / evaluate on a COCO-format ground-truth
    json; known_names select the known categories, everything else is "unknown"."""

    def __init__(self, gt_json, known_names: Sequence[str], contiguous_to_dataset_id: Optional[Dict[int, int]] = None,
                 max_dets_per_image: Sequence[int] = (10, 20, 30, 50, 100), output_dir: Optional[str] = None):
        if isinstance(gt_json, str):
            with open(gt_json) as f:
                gt_json = json.load(f)
        self.gt = gt_json
        names = set(known_names)
        self.known_ids = sorted(c["id"] for c in gt_json["categories"] if c["name"] in names)
        self.reverse_id_map = contiguous_to_dataset_id
        self.max_dets = list(max_dets_per_image)
        self.output_dir = output_dir
        self.reset()

    def reset(self):
        self._predictions: List[dict] = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            self._predictions += instances_to_coco_json(out["instances"], inp["image_id"], self.reverse_id_map)

    RESULTS_FILE = "coco_instances_results.json"

    def evaluate(self, img_ids=None, resume: bool = False):
        """Rank 0 gathers every rank's detections, writes them to <output_dir>/coco_instances_results.json
        (os_coco_evaluation.py:259-263) and scores them; `resume=True` scores that file instead of new detections
        (train.py --resume_test, os_coco_evaluation.py:156-190). The known / unknown precision and recall arrays are saved
        next to it (:428-431)."""
        if resume:
            if self.output_dir is None:
                raise ValueError("resume=True needs output_dir (where a previous run wrote its detections)")
            if parallel.world_info()[0] != 0:
                return None
            with open(os.path.join(self.output_dir, self.RESULTS_FILE)) as f:
                dets = json.load(f)
        else:
            gathered = parallel.gather_to_rank0(self._predictions)
            if gathered is None:
                return None
            dets = [d for part in gathered for d in part]
            if self.output_dir:
                os.makedirs(self.output_dir, exist_ok=True)
                with open(os.path.join(self.output_dir, self.RESULTS_FILE), "w") as f:
                    json.dump(dets, f)
        if not dets:
            return {"bbox": {m: float("nan") for m in METRICS}}
        ev = OpensetCOCOEval(copy.deepcopy(self.gt), dets, self.known_ids, self.max_dets, img_ids)
        ev.evaluate()
        ev.accumulate()
        if self.output_dir:
            for tag, e in (("known", ev.eval_kdt), ("unknown", ev.eval_unkdt)):
                np.save(os.path.join(self.output_dir, f"{tag}_precision_bbox.npy"), e["precision"])
                np.save(os.path.join(self.output_dir, f"{tag}_recall_bbox.npy"), e["recall"])
        return derive_results(ev.summarize())
