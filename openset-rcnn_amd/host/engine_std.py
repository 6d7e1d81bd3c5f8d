"""BASELINE config 1 on the HIP path: the stock detectron2 Faster R-CNN that /root/reference/configs/Base-RCNN-FPN.yaml describes
on its own -- PROPOSAL_GENERATOR "RPN" with "StandardRPNHead" (three aspect ratios per cell, objectness logits, Box2BoxTransform
deltas, per-level NMS at 0.7, post-NMS top-k) and ROI_HEADS "StandardROIHeads" with FastRCNNOutputLayers (softmax over 80 + 1
classes, class-specific box deltas, per-class NMS at 0.5, 100 detections per image). None of that code lives in /root/reference
(it is detectron2's); the reference only selects it by name (Base-RCNN-FPN.yaml:2-33), and SURVEY.md 8b lists the names as part of
the drop-in surface. Same kernels as the open-set path: the MFMA convolutions, osr_gemm_f32 for the small output layers,
osr_rpn_select_ex (decode mode 1), osr_nms_topk, osr_roi_align_fwd, plus osr_fastrcnn_candidates."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops
from .engine import OpensetRCNNEngine
from .weights import pack_fc1_weight

STD_DEFAULT_CFG = dict(
    anchor_ratios=(0.5, 1.0, 2.0), post_nms_topk_test=1000, rpn_nms_thresh=0.7, rpn_bbox_reg_weights=(1.0, 1.0, 1.0, 1.0),
    std_num_classes=80, score_thresh_test=0.05, std_nms_thresh_test=0.5, std_detections_per_image=100, cls_agnostic_bbox_reg=False,
)


def cell_anchor_table(sizes, ratios) -> torch.Tensor:
    """[d2] DefaultAnchorGenerator.generate_cell_anchors per level: for (size, ratio): area = size^2, w = sqrt(area / ratio),
    h = ratio * w, anchor [-w/2, -h/2, w/2, h/2], computed in double and rounded to fp32 as torch.tensor() does. (L, A, 4)."""
    rows = []
    for z in sizes:
        cells = []
        for r in ratios:
            w = math.sqrt(float(z) ** 2 / r)
            h = r * w
            cells.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        rows.append(cells)
    return torch.tensor(rows, dtype=torch.float32)


class StandardRCNNEngine(OpensetRCNNEngine):
    def __init__(self, params: Dict[str, torch.Tensor], cfg: Optional[dict] = None, dtype: torch.dtype = torch.float16, device: str = "cuda"):
        full = dict(STD_DEFAULT_CFG)
        if cfg:
            full.update(cfg)
        super().__init__(params, full, dtype, device)

    def _init_rpn(self, params) -> None:
        dev, c = self.device, self.cfg
        f32 = lambda k: params[k].float().contiguous().to(dev)  # noqa: E731
        self.has_rpn = "proposal_generator.rpn_head.objectness_logits.weight" in params
        if not self.has_rpn:
            return
        self.num_anchors = len(c["anchor_ratios"])
        self.rpn_wo = f32("proposal_generator.rpn_head.objectness_logits.weight").view(self.num_anchors, -1)
        self.rpn_bo = f32("proposal_generator.rpn_head.objectness_logits.bias")
        self.rpn_wd = f32("proposal_generator.rpn_head.anchor_deltas.weight").view(self.num_anchors * 4, -1)
        self.rpn_bd = f32("proposal_generator.rpn_head.anchor_deltas.bias")
        self.cell_anchors = cell_anchor_table(c["anchor_sizes"], c["anchor_ratios"]).to(dev)
        self.fuse_rpn_head = False

    def _init_roi_heads(self, params) -> None:
        dev, c, dtype = self.device, self.cfg, self.dtype
        f32 = lambda k: params[k].float().contiguous().to(dev)  # noqa: E731
        self.has_roi = "roi_heads.box_predictor.cls_score.weight" in params
        if not self.has_roi:
            return
        self.fc1_w = pack_fc1_weight(params["roi_heads.box_head.fc1.weight"], 256, c["pooler_resolution"], dtype).to(dev)
        self.fc1_b = params["roi_heads.box_head.fc1.bias"].float().to(dev)
        self.fc2_w = params["roi_heads.box_head.fc2.weight"].to(dtype).contiguous().to(dev)
        self.fc2_b = params["roi_heads.box_head.fc2.bias"].float().to(dev)
        self.cls_w, self.cls_b = f32("roi_heads.box_predictor.cls_score.weight"), f32("roi_heads.box_predictor.cls_score.bias")
        self.box_w, self.box_b = f32("roi_heads.box_predictor.bbox_pred.weight"), f32("roi_heads.box_predictor.bbox_pred.bias")
        k = c["std_num_classes"]
        assert self.cls_w.shape[0] == k + 1 and self.box_w.shape[0] in (4, 4 * k), "cls_score: K+1 rows; bbox_pred: 4 or 4K rows"

    def _levels(self, shapes, n):
        key = (tuple(shapes), n)
        if key not in self._lv_cache:
            self._lv_cache[key] = ops.make_rpn_levels(shapes, self.cfg["fpn_strides"], n, self.num_anchors)
        return self._lv_cache[key]

    # ---- [d2] RPN.forward (inference) -------------------------------------------------------------------------------------------
    def _rpn(self, feats, image_hw, keep=None, topk=None):
        c = self.cfg
        fl = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        n = fl[0].shape[0]
        shapes = [(f.shape[1], f.shape[2]) for f in fl]
        rows = [n * h * w for h, w in shapes]
        a = self.num_anchors
        # StandardRPNHead: 3x3 conv + ReLU on the MFMA kernel with an fp32 hidden state, then the two 1x1 convs as exact-fp32 GEMMs.
        # With NHWC rows the (N,A,H,W)->(N,H*W*A) / (N,A*4,H,W)->(N,H*W*A,4) flattening of [d2] RPN.forward is a no-op: row-major
        # (pixel, anchor) is exactly the memory order of the (rows, A) and (rows, A*4) GEMM outputs.
        t_all = torch.empty((sum(rows), 256), dtype=torch.float32, device=self.device)
        off = 0
        for f, r in zip(fl, rows):
            self._conv(f, "proposal_generator.rpn_head.conv", 1, 1, relu=True, out=t_all[off:off + r], out_dtype=torch.float32)
            off += r
        logits = ops.gemm_f32(t_all, self.rpn_wo, self.rpn_bo).view(-1)
        deltas = ops.gemm_f32(t_all, self.rpn_wd, self.rpn_bd).view(-1, 4)
        # level offsets of make_rpn_levels count anchors (pixels * A): the GEMM outputs above are laid out exactly so
        k = c["pre_nms_topk_test"] if topk is None else topk
        sel = ops.rpn_select(self._levels(shapes, n), self.cell_anchors, logits, deltas, n, image_hw, k, c["min_box_size"],
                             b2b_weights=c["rpn_bbox_reg_weights"])
        # [d2] find_top_rpn_proposals: batched NMS with the level as category, then the first POST_NMS_TOPK of the keep list
        post = c["post_nms_topk_test"]
        pk, pcnt = ops.nms_topk(sel["boxes"], sel["scores"], sel["level"], None, n, sel["cap"], sel["counts"], c["rpn_nms_thresh"], post)
        boxes = ops.gather_rows(sel["boxes"].view(-1, 4), sel["cap"], pk, pcnt)
        scores = ops.gather_rows(sel["scores"].view(-1), sel["cap"], pk, pcnt).view(n, post)
        ar = torch.arange(post, device=self.device, dtype=torch.int32)[None, :]
        bidx = torch.where(ar < pcnt[:, None], torch.arange(n, device=self.device, dtype=torch.int32)[:, None], torch.full((1, 1), -1, device=self.device, dtype=torch.int32))
        out = dict(boxes=boxes, scores=scores, counts=pcnt, batch_idx=bidx.reshape(-1).contiguous(), cap=post, pre=sel, keep_idx=pk)
        if keep is not None:
            keep.update(rpn_t=t_all, rpn_logits=logits, rpn_deltas=deltas, rpn_shapes=shapes, rpn_pre=sel, rpn_keep=pk)
        return out

    # ---- [d2] StandardROIHeads._forward_box (inference) -----------------------------------------------------------------------
    def _roi_heads(self, feats, sel, image_hw, keep=None):
        c = self.cfg
        n, cap = sel["boxes"].shape[0], sel["cap"]
        boxes = sel["boxes"].view(-1, 4)
        pooled = ops.roi_align([feats[k] for k in ("p2", "p3", "p4", "p5")], c["pooler_scales"], boxes, sel["batch_idx"], c["pooler_resolution"],
                               self.dtype, c["canonical_level"], c["canonical_size"], 2)
        m = pooled.shape[0]
        h1 = self._linear(pooled.view(m, -1), self.fc1_w, self.fc1_b, True, name="roi_heads.box_head.fc1")
        box_feats = self._linear(h1, self.fc2_w, self.fc2_b, True, torch.float32, name="roi_heads.box_head.fc2")
        logits = ops.gemm_f32(box_feats, self.cls_w, self.cls_b)
        deltas = ops.gemm_f32(box_feats, self.box_w, self.box_b)
        k = c["std_num_classes"]
        cands = ops.fastrcnn_candidates(logits, deltas, sel["boxes"], sel["counts"], image_hw, k, c["bbox_reg_weights"], c["score_thresh_test"])
        topk = c["std_detections_per_image"]
        dk, dcnt = ops.nms_topk(cands["boxes"], cands["scores"], cands["cls"], None, n, cands["cap"], cands["count"], c["std_nms_thresh_test"], topk)
        ob = ops.gather_rows(cands["boxes"].view(-1, 4), cands["cap"], dk, dcnt)
        osc = ops.gather_rows(cands["scores"].view(-1), cands["cap"], dk, dcnt).view(n, topk)
        # (class ids travel through the fp32 row gather as bit patterns: it only copies)
        ocl = ops.gather_rows(cands["cls"].view(-1).view(torch.float32), cands["cap"], dk, dcnt).view(n, topk).view(torch.int32).to(torch.int64)
        ocl = torch.where(torch.arange(topk, device=self.device)[None, :] < dcnt[:, None], ocl, torch.full_like(ocl, -1))
        if keep is not None:
            keep.update(pooled=pooled, h1=h1, box_feats=box_feats, logits=logits, deltas=deltas, cands=cands, det_keep=dk, det_count=dcnt)
        return ob, osc, ocl, dcnt
