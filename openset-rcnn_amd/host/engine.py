"""Batched inference engine for the Openset R-CNN hot path on one MI355X.

Everything the reference does in per-image / per-level Python loops with host syncs runs here as
one stream of HIP launches over all N images with on-device counts; the only device->host transfer is the
final (N, 100) detection block. Mirrors, stage by stage:

  GeneralizedRCNN.inference [d2]                         -> OpensetRCNNEngine.forward
    preprocess_image + ImageList.from_tensors [d2]       -> ops.preprocess
    build_resnet_fpn_backbone [d2]                       -> _backbone (stem view conv, bottlenecks, FPN, p6)
    ClsFreeRPN.forward (classification_free_rpn.py:493)  -> _rpn
    OpensetROIHeads._forward_box (osrcnn_roi_heads.py:282) -> _roi_heads
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .weights import R50_BLOCKS, pack_conv_weight, pack_fc1_weight, pack_stem_weight

DEFAULT_CFG = dict(
    pixel_mean=(103.53, 116.28, 123.675), pixel_std=(1.0, 1.0, 1.0), size_divisibility=32,
    fpn_strides=(4, 8, 16, 32, 64), anchor_sizes=(32, 64, 128, 256, 512),
    pre_nms_topk_test=1000, min_box_size=0.0,
    pooler_resolution=7, pooler_scales=(0.25, 0.125, 0.0625, 0.03125), canonical_level=4, canonical_size=224,
    bbox_reg_weights=(10.0, 10.0, 5.0, 5.0), mean_type="geometric",
    obj_score_thresh=0.05, nms_thresh_test=1.0, detections_per_image=1000,
    known_score_thresh=0.05, known_nms_thresh=0.5, known_topk=50,
    unknown_score_thresh=0.0, unknown_nms_thresh=0.5, unknown_topk=50,
    num_classes=81, num_known=20, reps_per_class=1, pln_distance="COS", unknown_id=80, unk_thr=0.23,
    # training step (configs/VOC-COCO/openset_rcnn_R50_FPN_128k.yaml + [d2] defaults)
    pre_nms_topk_train=2000, rpn_batch_size=256, rpn_positive_fraction=0.5, rpn_positive_fraction_objectness=1.0,
    rpn_iou_thresholds=(0.3, 0.7), rpn_iou_thresholds_objectness=(0.1, 0.3), rpn_loc_weight=0.5, rpn_ctr_weight=0.5,
    roi_batch_size=512, roi_positive_fraction=0.25, roi_iou_threshold=0.5, box_reg_weight=0.5, iou_reg_weight=0.5,
    pln_alpha=0.1, pln_beta=0.9, pln_iou_threshold=0.5, pln_loss_weight=0.5, cls_loss_weight=0.9,
)


DEFAULT_LOSS_TYPES = dict(rpn_box=("iou", 0.0), rpn_ctr=("smooth_l1", 0.0), roi_box=("smooth_l1", 0.0), roi_iou=("smooth_l1", 0.0))
BOX_LOSS_NAMES = ("smooth_l1", "iou", "giou", "diou", "ciou")


def loss_types_of(cfg: dict) -> dict:
    """(type, smooth-L1 beta) of the four regression losses: cfg['loss_types'] (modeling.engine_cfg_from fills it from MODEL.RPN.* /
    MODEL.ROI_BOX_HEAD.*) over the defaults both Openset yaml files select."""
    lt = dict(DEFAULT_LOSS_TYPES)
    lt.update(cfg.get("loss_types") or {})
    return lt


def check_supported_losses(cfg: dict) -> None:
    """What the loss kernels implement: every box regression loss box_regression_w_iou.py:13-85 names ("smooth_l1" with its beta,
    "iou", "giou", "diou", "ciou") for the CF-RPN and the RoI box head, and smooth L1 with any beta for the centerness and IoU
    regressions (the only type classification_free_rpn.py:475-481 / osrcnn_fast_rcnn.py:368 implement). Anything else is refused
    here instead of silently training another loss."""
    lt = loss_types_of(cfg)
    for key, name in (("rpn_box", "MODEL.RPN.BBOX_REG_LOSS_TYPE"), ("roi_box", "MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE")):
        if lt[key][0] not in BOX_LOSS_NAMES:
            raise NotImplementedError(f"{name} '{lt[key][0]}': one of {BOX_LOSS_NAMES}")
    for key, name in (("rpn_ctr", "MODEL.RPN.CTR_REG_LOSS_TYPE"), ("roi_iou", "MODEL.ROI_BOX_HEAD.IOU_REG_LOSS_TYPE")):
        if lt[key][0] != "smooth_l1":
            raise NotImplementedError(f"{name} '{lt[key][0]}': the reference implements \"smooth_l1\" only")
    for key in lt:
        if lt[key][1] < 0.0:
            raise ValueError(f"loss_types['{key}']: negative smooth-L1 beta")


class OpensetRCNNEngine:
    """dtype: storage type of activations and MFMA operands. torch.float16 / torch.bfloat16 = the fast path (fp32 accumulation,
    fp32 heads from the box features on). torch.float32 = PARITY MODE: every tensor and every product in fp32 (osr_conv_f32.hip,
    1/16 of the fp16 matrix rate), the arithmetic the reference itself runs in -- boxes, scores and embeddings then agree with the
    fp32 oracle to summation order (tests/test_e2e_parity.py)."""

    FP32_POINTS = ("backbone", "rpn_hidden", "pooled", "h1")

    def __init__(self, params: Dict[str, torch.Tensor], cfg: Optional[dict] = None, dtype: torch.dtype = torch.float16,
                 device: str = "cuda", class_map: Optional[torch.Tensor] = None, fp32_points: Sequence[str] = ()):
        """fp32_points (diagnostic; fast mode only): storage points of the fp16 path kept in fp32 instead, to measure what each one
        costs in agreement with the fp32 reference (tests/test_e2e_parity.py): "backbone" (stem .. FPN outputs computed by the
        fp32 kernels, the pyramid handed on in fp16), "rpn_hidden" (the CF-RPN hidden state: un-fused head, fp32 t), "pooled"
        (RoIAlign output and FC1 in fp32), "h1" (FC1 output and FC2 in fp32). The layers behind such a point run on the fp32
        kernels (1/16 of the matrix rate): a measurement aid, not a product configuration."""
        self.cfg = dict(DEFAULT_CFG)
        if cfg:
            self.cfg.update(cfg)
        self.dtype = dtype
        self.device = torch.device(device)
        dev = self.device
        self.fp32_points = frozenset(fp32_points) if dtype != torch.float32 else frozenset()
        assert self.fp32_points <= set(self.FP32_POINTS), self.fp32_points
        self.w = self._pack_convs(params)
        if "backbone" in self.fp32_points:
            self._bb32 = OpensetRCNNEngine({k: v for k, v in params.items() if k.startswith("backbone.")}, cfg, torch.float32, device)
        c = self.cfg
        self.has_backbone = "backbone.bottom_up.stem.conv1.weight" in params
        self.class_map = None if class_map is None else class_map.to(torch.int64).to(dev)
        # training-side id_map of the GraspNet configuration (prototype_learning_network.py:80-95, softmax_classifier.py:214-229):
        # dataset class id -> index in the sorted known list, background (NUM_CLASSES) -> NUM_KNOWN, anything else -> -1
        self.id_map = None
        if class_map is not None:
            nc = self.cfg["num_classes"]
            idm = torch.full((nc + 2,), -1, dtype=torch.int64)
            idm[class_map.to(torch.int64)] = torch.arange(len(class_map))
            idm[nc] = self.cfg["num_known"]
            self.id_map = idm.to(dev)  # the extra last slot (-1) is where padding rows (class -1) index
        self._lv_cache = {}
        self._streams = []
        self.profile = None  # set to a list to collect (name, algorithmic flops, start event, end event) per MFMA launch
        self.profile_hbm = None  # set to a list to collect (name, algorithmic bytes, start event, end event, info) of the HBM-group kernels
        # the box head's FC layers skip the tiles that hold only padding rows of the per-image proposal lists (fp16 / bf16 kernels;
        # the fp32 parity kernel computes every row)
        self.skip_padding_tiles = dtype != torch.float32
        # the three (four) convolutions of a res2 block run as ONE launch (osr_bottleneck_fwd): fp16 / bf16 storage only
        self.fuse_res2 = dtype != torch.float32
        # conv2 -> conv3 + shortcut of a res3 block run as ONE launch (osr_conv2d_chain_fwd: conv2's output stays in LDS)
        self.chain_res3 = dtype != torch.float32
        # the stem's convolution, ReLU and max pool run as ONE launch (osr_stem_maxpool_fwd): fp16 / bf16 storage only
        self.fuse_stem = dtype != torch.float32
        # the FPN's four output convs run as ONE launch, and so does the CF-RPN head over p2..p6 (osr_conv2d_fwd_levels,
        # osr_cfrpn_head_fwd_levels: a level table in the 256 x 256 kernel): fp16 / bf16 storage only
        self.fuse_levels = dtype != torch.float32
        self._init_rpn(params)
        self._init_roi_heads(params)

    def _pack_convs(self, params) -> Dict[str, torch.Tensor]:
        """Every 4-d conv weight (backbone, FPN, RPN 3x3) repacked to [cout][kh][kw][cin] in the storage dtype + its fp32 bias; the
        1x1 output convs of the RPN heads stay fp32 matrices (handled by _init_rpn)."""
        w: Dict[str, torch.Tensor] = {}
        dev, dtype = self.device, self.dtype
        for k, v in params.items():
            if not k.endswith(".weight") or v.dim() != 4 or (k.startswith("proposal_generator.rpn_head.") and not k.startswith("proposal_generator.rpn_head.conv.")):
                continue
            pre = k[: -len(".weight")]
            if pre == "backbone.bottom_up.stem.conv1":
                w[pre + ".w"] = pack_stem_weight(v, dtype).to(dev)
            else:
                w[pre + ".w"] = pack_conv_weight(v, dtype).to(dev)
            w[pre + ".b"] = params[pre + ".bias"].float().contiguous().to(dev)
        return w

    def _init_rpn(self, params) -> None:
        dev, c = self.device, self.cfg
        f32 = lambda k: params[k].float().contiguous().to(dev)  # noqa: E731
        self.has_rpn = "proposal_generator.rpn_head.centerness.weight" in params
        if self.has_rpn:
            self.rpn_wd = f32("proposal_generator.rpn_head.anchor_deltas.weight").view(-1, 256)
            self.rpn_bd = f32("proposal_generator.rpn_head.anchor_deltas.bias")
            self.rpn_wc = f32("proposal_generator.rpn_head.centerness.weight").view(1, 256)
            self.rpn_bc = f32("proposal_generator.rpn_head.centerness.bias")
            self.rpn_wtail = torch.cat((self.rpn_wd, self.rpn_wc)).contiguous()
            self.rpn_btail = torch.cat((self.rpn_bd, self.rpn_bc)).contiguous()
            # (the fused head kernel is an fp16/bf16 MFMA kernel that parks the hidden state in the storage dtype; fp32 = parity mode)
            self.fuse_rpn_head = self.dtype != torch.float32 and "rpn_hidden" not in self.fp32_points
            self.rpn_keep_hidden = False
            sizes = c["anchor_sizes"]
            self.cell_anchors = torch.tensor([[[-s / 2.0, -s / 2.0, s / 2.0, s / 2.0]] for s in sizes], dtype=torch.float32, device=dev)

    def _init_roi_heads(self, params) -> None:
        dev, c, dtype = self.device, self.cfg, self.dtype
        f32 = lambda k: params[k].float().contiguous().to(dev)  # noqa: E731
        self.has_roi = "roi_heads.box_predictor.iou_pred.weight" in params
        if not self.has_roi:
            return
        fc1_dt = torch.float32 if "pooled" in self.fp32_points else dtype
        fc2_dt = torch.float32 if "h1" in self.fp32_points else dtype
        self.fc1_w = pack_fc1_weight(params["roi_heads.box_head.fc1.weight"], 256, c["pooler_resolution"], fc1_dt).to(dev)
        self.fc1_b = params["roi_heads.box_head.fc1.bias"].float().to(dev)
        self.fc2_w = params["roi_heads.box_head.fc2.weight"].to(fc2_dt).contiguous().to(dev)
        self.fc2_b = params["roi_heads.box_head.fc2.bias"].float().to(dev)
        self.pred_w = torch.cat((f32("roi_heads.box_predictor.bbox_pred.weight"), f32("roi_heads.box_predictor.iou_pred.weight"))).contiguous()
        self.pred_b = torch.cat((f32("roi_heads.box_predictor.bbox_pred.bias"), f32("roi_heads.box_predictor.iou_pred.bias"))).contiguous()
        self.enc_w, self.enc_b = f32("roi_heads.dml.encoder.weight"), f32("roi_heads.dml.encoder.bias")
        self.dec_w, self.dec_b = f32("roi_heads.dml.decoder.weight"), f32("roi_heads.dml.decoder.bias")
        self.protos = ops.l2_normalize_rows(f32("roi_heads.dml.representatives"))  # prototype_learning_network.py:199
        self.cls_w, self.cls_b = f32("roi_heads.softmaxcls.cls_score.weight"), f32("roi_heads.softmaxcls.cls_score.bias")

    # ---- backbone -------------------------------------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=0, relu=False, residual=None, res_mode=0, out=None, out_dtype=None):
        w = self.w[name + ".w"]
        if self.profile is None:
            return ops.conv2d(x, w, self.w[name + ".b"], stride, pad, relu, residual, res_mode, out_dtype, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = ops.conv2d(x, w, self.w[name + ".b"], stride, pad, relu, residual, res_mode, out_dtype, out)
        e1.record()
        rows = y.numel() // w.shape[0]
        nbytes = x.numel() * x.element_size() + w.numel() * w.element_size() + y.numel() * y.element_size() + \
            (residual.numel() * residual.element_size() if residual is not None else 0)
        flops = 2.0 * rows * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        self.profile.append((name, flops, e0, e1, nbytes, flops))
        return y

    def _bottleneck(self, x, pre: str, first: bool, stride: int = 1):
        """One [d2] BottleneckBlock (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + shortcut, ReLU after each). res2's blocks run as one
        fused launch when the engine allows it (the intermediates never reach HBM); everything else as separate launches."""
        w = self.w
        if self.fuse_res2 and stride == 1 and w[pre + ".conv1.w"].shape[0] == 64:
            if self.profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            y = ops.bottleneck(x, w[pre + ".conv1.w"], w[pre + ".conv1.b"], w[pre + ".conv2.w"], w[pre + ".conv2.b"], w[pre + ".conv3.w"],
                               w[pre + ".conv3.b"], w[pre + ".shortcut.w"] if first else None, w[pre + ".shortcut.b"] if first else None)
            if y is not None:
                if self.profile is not None:
                    e1.record()
                    names = ["conv1", "conv2", "conv3"] + (["shortcut"] if first else [])
                    px = y.numel() // y.shape[-1]
                    flops = sum(2.0 * px * w[f"{pre}.{c}.w"].numel() for c in names)
                    nbytes = x.numel() * x.element_size() + y.numel() * y.element_size() + sum(w[f"{pre}.{c}.w"].numel() for c in names) * x.element_size()
                    self.profile.append((pre + " (fused block)", flops, e0, e1, nbytes, flops))
                return y
        pair = self._shortcut_conv1_one_launch(x, pre, stride) if first and self.fuse_levels else None
        if pair is not None:
            sc, o = pair
        else:
            sc = self._conv(x, pre + ".shortcut", stride) if first else x
            o = self._conv(x, pre + ".conv1", stride, relu=True)
        if self.chain_res3 and w[pre + ".conv2.w"].shape[0] == 128 and w[pre + ".conv3.w"].shape[0] == 512:
            if self.profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            y = ops.conv2d_chain(o, w[pre + ".conv2.w"], w[pre + ".conv2.b"], w[pre + ".conv3.w"], w[pre + ".conv3.b"], sc, 1, 1)
            if y is not None:
                if self.profile is not None:
                    e1.record()
                    px, es = y.numel() // y.shape[-1], y.element_size()
                    flops = 2.0 * px * (w[pre + ".conv2.w"].numel() + w[pre + ".conv3.w"].numel())
                    nbytes = (o.numel() + sc.numel() + y.numel() + w[pre + ".conv2.w"].numel() + w[pre + ".conv3.w"].numel()) * es
                    self.profile.append((pre + ".conv2+conv3 (chained)", flops, e0, e1, nbytes, flops))
                return y
        o = self._conv(o, pre + ".conv2", 1, 1, relu=True)
        return self._conv(o, pre + ".conv3", relu=True, residual=sc, res_mode=1)

    def _linear(self, x, w, b, relu, out_dtype=None, name="fc", row_seg=None, real_rows=None):
        """real_rows (profiling only): how many of x's rows carry data. The padding rows of the per-image proposal lists are not
        algorithmic work: FLOPs and bytes are credited for the real rows only, the nominal figure (all rows of the
        fixed-capacity list) is kept beside it."""
        if self.profile is None:
            return ops.linear(x, w, b, relu=relu, out_dtype=out_dtype, row_seg=row_seg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = ops.linear(x, w, b, relu=relu, out_dtype=out_dtype, row_seg=row_seg)
        e1.record()
        # (real_rows may be a callable that reads a device count: it is resolved by resolve_profile() AFTER the pass, so that no host
        # sync sits between the launches of the attribution pass -- a sync in front of a kernel makes it start on an idle, down-clocked GPU)
        rows_of = (lambda: x.shape[0]) if real_rows is None else (real_rows if callable(real_rows) else (lambda: int(real_rows)))
        kx, ky, es_x, es_y, wbytes, kn = x.shape[1], y.shape[1], x.element_size(), y.element_size(), w.numel() * w.element_size(), w.shape[0] * w.shape[1]
        self.profile.append((name, lambda: 2.0 * rows_of() * kn, e0, e1, lambda: rows_of() * kx * es_x + wbytes + rows_of() * ky * es_y,
                             2.0 * x.shape[0] * kn))
        return y

    def _backbone(self, images: torch.Tensor, hp: int, wp: int, keep: Optional[dict] = None, normalized: bool = False) -> Dict[str, torch.Tensor]:
        c = self.cfg
        if "backbone" in self.fp32_points:  # diagnostic: the fp32 kernels compute the pyramid, the heads get it in the storage dtype
            return {k: v.to(self.dtype) for k, v in self._bb32._backbone(images, hp, wp, None, normalized).items()}
        mean, std = ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)) if normalized else (c["pixel_mean"], c["pixel_std"])
        fused = self.fuse_stem and keep is None
        xpad = None if fused else ops.preprocess(images, hp, wp, mean, std, self.dtype)
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        sw, sb = self.w["backbone.bottom_up.stem.conv1.w"], self.w["backbone.bottom_up.stem.conv1.b"]
        if fused:  # normalise + pad + conv1 + ReLU + max pool in one launch: neither the padded batch nor the stem output reaches HBM (keep wants the latter)
            x = ops.stem_maxpool_raw(images, hp, wp, mean, std, sw, sb)
            stem_px = images.shape[0] * (hp // 2) * (wp // 2) * 64
            if self.profile is not None:
                e1.record()
                self.profile.append(("backbone.bottom_up.stem (preprocess + conv1 + max pool, fused)", 2.0 * stem_px * 147, e0, e1,
                                     images.numel() * images.element_size() + x.numel() * 2, 2.0 * stem_px * 147))  # 7*7*3 real taps of every stem pixel
        else:
            x = ops.stem_conv(xpad, sw, sb, hp, wp, relu=True)
            if self.profile is not None:
                e1.record()
                self.profile.append(("backbone.bottom_up.stem.conv1", 2.0 * x.numel() * 147, e0, e1,
                                     xpad.numel() * 2 + x.numel() * 2, 2.0 * x.numel() * 147))  # 7*7*3 real taps
            if keep is not None:
                keep["stem"] = x
            x = ops.maxpool3x3s2(x)
        feats = {}
        for si, nb in enumerate(R50_BLOCKS):
            for b in range(nb):
                pre = f"backbone.bottom_up.res{si + 2}.{b}"
                stride = 2 if (b == 0 and si > 0) else 1  # MSRA: stride in the first 1x1
                x = self._bottleneck(x, pre, b == 0, stride)
            feats[f"res{si + 2}"] = x
        out = {}

        # [d2] FPN.forward: the lateral 1x1 convs + top-down sums form a chain (p5 -> p2); the four 3x3 output convs only read its results, so
        # they run behind it as ONE launch over the four levels (ops.conv2d_levels: one partial last dispatch round instead of four, and the
        # p4 / p5 convs no longer leave most of the chip idle), bit-identical to the per-level launches
        lat = {5: self._conv(feats["res5"], "backbone.fpn_lateral5")}
        for lvl in (4, 3, 2):
            lat[lvl] = self._conv(feats[f"res{lvl}"], f"backbone.fpn_lateral{lvl}", residual=lat[lvl + 1], res_mode=2)
        outs = self._fpn_outputs_one_launch([lat[l] for l in (2, 3, 4, 5)]) if self.fuse_levels else None
        for i, lvl in enumerate((2, 3, 4, 5)):
            out[f"p{lvl}"] = outs[i] if outs is not None else self._conv(lat[lvl], f"backbone.fpn_output{lvl}", 1, 1)
        out["p6"] = ops.subsample2(out["p5"])
        if keep is not None:
            keep.update(feats)
        return out

    def pool_rois(self, feats: Dict[str, torch.Tensor], boxes: torch.Tensor, batch_idx: torch.Tensor, out_dtype=None, fill_padding: bool = True) -> torch.Tensor:
        """[d2] ROIPooler + torchvision roi_align on p2..p5 (osrcnn_roi_heads.py:306) -> (m, 49 * 256) rows in (ph, pw, c) order, the K
        order self.fc1_w is packed in."""
        c = self.cfg
        fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
        pooled = ops.roi_align(fl, c["pooler_scales"], boxes, batch_idx, c["pooler_resolution"], out_dtype or self.dtype,
                               c["canonical_level"], c["canonical_size"], 2, fill_padding=fill_padding)
        return pooled.view(pooled.shape[0], -1)

    def pooled_bin_major(self, pooled: torch.Tensor) -> torch.Tensor:
        """pool_rois' rows as (m, 7, 7, 256): what `keep` hands to tests and diagnostics."""
        m, P = pooled.shape[0], self.cfg["pooler_resolution"]
        return pooled.view(m, P, P, 256)

    def _shortcut_conv1_one_launch(self, x, pre, stride):
        """A stage's first bottleneck reads its input twice ([d2] BottleneckBlock.forward: self.shortcut(x), self.conv1(x)): both 1x1 layers
        as ONE launch (ops.conv2d_pair, bit-identical to the two). Returns (shortcut output, relu(conv1 output)) or None."""
        w = self.w
        if self.profile is None:
            return ops.conv2d_pair(x, w[pre + ".shortcut.w"], w[pre + ".shortcut.b"], False, w[pre + ".conv1.w"], w[pre + ".conv1.b"], True, stride, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.conv2d_pair(x, w[pre + ".shortcut.w"], w[pre + ".shortcut.b"], False, w[pre + ".conv1.w"], w[pre + ".conv1.b"], True, stride, 0)
        e1.record()
        if out is not None:
            px = out[0].numel() // out[0].shape[-1]
            flops = 2.0 * px * (w[pre + ".shortcut.w"].numel() + w[pre + ".conv1.w"].numel())
            nbytes = (x.numel() // (stride * stride) + out[0].numel() + out[1].numel() + w[pre + ".shortcut.w"].numel() + w[pre + ".conv1.w"].numel()) * x.element_size()
            self.profile.append((pre + ".shortcut+conv1 (one launch)", flops, e0, e1, nbytes, flops))
        return out

    def _fpn_outputs_one_launch(self, lats):
        ws = [self.w[f"backbone.fpn_output{l}.w"] for l in (2, 3, 4, 5)]
        bs = [self.w[f"backbone.fpn_output{l}.b"] for l in (2, 3, 4, 5)]
        if self.profile is None:
            return ops.conv2d_levels(lats, ws, bs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = ops.conv2d_levels(lats, ws, bs)
        e1.record()
        if outs is not None:
            flops = sum(2.0 * (x.numel() // x.shape[-1]) * w_.numel() for x, w_ in zip(lats, ws))
            nbytes = sum(x.numel() + o.numel() + w_.numel() for x, o, w_ in zip(lats, outs, ws)) * lats[0].element_size()
            self.profile.append(("backbone.fpn_output2-5 (four levels, one launch)", flops, e0, e1, nbytes, flops))
        return outs

    def _rpn_levels_fused(self, fl, deltas, ctrs, hiddens):
        w, b = self.w["proposal_generator.rpn_head.conv.w"], self.w["proposal_generator.rpn_head.conv.b"]
        if self.profile is None:
            return ops.cfrpn_head_fused_levels(fl, w, b, self.rpn_wtail, self.rpn_btail, deltas, ctrs, hiddens)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ok = ops.cfrpn_head_fused_levels(fl, w, b, self.rpn_wtail, self.rpn_btail, deltas, ctrs, hiddens)
        e1.record()
        if ok:
            rows = sum(d.shape[0] for d in deltas)
            flops = 2.0 * rows * 256 * (2304 + 5)
            self.profile.append(("proposal_generator.rpn_head.conv+tail (p2-p6, one launch)", flops, e0, e1,
                                 sum(f.numel() for f in fl) * 2 + w.numel() * 2 + rows * 20, flops))
        return ok

    def _rpn_level_fused(self, f, deltas, ctr, hidden=None):
        w, b = self.w["proposal_generator.rpn_head.conv.w"], self.w["proposal_generator.rpn_head.conv.b"]
        if self.profile is None:
            return ops.cfrpn_head_fused(f, w, b, self.rpn_wtail, self.rpn_btail, deltas, ctr, hidden)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.cfrpn_head_fused(f, w, b, self.rpn_wtail, self.rpn_btail, deltas, ctr, hidden)
        e1.record()
        rows = deltas.shape[0]
        self.profile.append(("proposal_generator.rpn_head.conv+tail", 2.0 * rows * 256 * (2304 + 5), e0, e1,
                             f.numel() * 2 + w.numel() * 2 + rows * 20, 2.0 * rows * 256 * (2304 + 5)))

    def _hbm(self, name, fn, nbytes, info=None):
        """Run fn(); with profile_hbm set, bracket it with HIP events on the launch stream and record its ALGORITHMIC bytes
        (unique bytes read + bytes written, no credit for re-reads)."""
        if self.profile_hbm is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.profile_hbm.append((name, nbytes() if callable(nbytes) else nbytes, e0, e1, info(out) if info else None))
        return out

    def resolve_profile(self):
        """After the profiled pass has been synchronised: evaluate the entries that were left as callables (they read device-side
        counts; evaluating them inside the pass would put a host sync between its launches). Returns (profile, profile_hbm) with
        plain numbers."""
        val = lambda v: v() if callable(v) else v  # noqa: E731
        prof = [(n, float(val(f)), e0, e1, float(val(nb)), float(nom)) for n, f, e0, e1, nb, nom in (self.profile or [])]
        hbm = [(n, float(val(nb)), e0, e1, val(info)) for n, nb, e0, e1, info in (self.profile_hbm or [])]
        return prof, hbm

    # ---- CF-RPN ---------------------------------------------------------------------------------------------
    def _levels(self, shapes, n):
        key = (tuple(shapes), n)
        if key not in self._lv_cache:
            self._lv_cache[key] = ops.make_rpn_levels(shapes, self.cfg["fpn_strides"], n, 1)
        return self._lv_cache[key]

    def _rpn(self, feats: Dict[str, torch.Tensor], image_hw: torch.Tensor, keep: Optional[dict] = None, topk: Optional[int] = None):
        fl = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        n = fl[0].shape[0]
        shapes = [(f.shape[1], f.shape[2]) for f in fl]
        rows = [n * h * w for h, w in shapes]
        if self.fuse_rpn_head:
            # one launch per level: 3x3 conv + ReLU + normalise + both 1x1 + sigmoid, hidden state never leaves the chip
            deltas = torch.empty((sum(rows), 4), dtype=torch.float32, device=self.device)
            ctr = torch.empty((sum(rows),), dtype=torch.float32, device=self.device)
            # (rpn_keep_hidden: the trainer's engine also wants the hidden state, for the head's backward)
            t_all = torch.empty((sum(rows), 256), dtype=self.dtype, device=self.device) if self.rpn_keep_hidden else None
            offs = [sum(rows[:i]) for i in range(len(rows))]
            # ClsFreeRPNHead.forward's `for x in features` (classification_free_rpn.py:157-161) as ONE launch over the five levels
            if not (self.fuse_levels and self._rpn_levels_fused(fl, [deltas[o:o + r] for o, r in zip(offs, rows)], [ctr[o:o + r] for o, r in zip(offs, rows)],
                                                                None if t_all is None else [t_all[o:o + r] for o, r in zip(offs, rows)])):
                for f, o, r in zip(fl, offs, rows):
                    self._rpn_level_fused(f, deltas[o:o + r], ctr[o:o + r], None if t_all is None else t_all[o:o + r])
        else:
            t_dt = torch.float32 if "rpn_hidden" in self.fp32_points else self.dtype
            t_all = torch.empty((sum(rows), 256), dtype=t_dt, device=self.device)
            off = 0
            for f, r in zip(fl, rows):
                self._conv(f, "proposal_generator.rpn_head.conv", 1, 1, relu=True, out=t_all[off:off + r], out_dtype=t_dt)
                off += r
            deltas, ctr = ops.cfrpn_head_tail(t_all, self.rpn_wd, self.rpn_bd, self.rpn_wc, self.rpn_bc)
        k = self.cfg["pre_nms_topk_test"] if topk is None else topk
        # algorithmic bytes: every centerness score once + the k selected anchors' deltas + the padded outputs
        sel = self._hbm("rpn_select", lambda: ops.rpn_select(self._levels(shapes, n), self.cell_anchors, ctr, deltas, n, image_hw, k, self.cfg["min_box_size"]),
                        lambda: ctr.numel() * 4 + n * sum(min(k, h * w) for h, w in shapes) * (16 + 16 + 4 + 4 + 4))
        sel.update(pred_deltas=deltas, pred_ctr=ctr, levels=self._levels(shapes, n))
        if keep is not None:
            keep.update(rpn_t=t_all, rpn_deltas=deltas, rpn_ctr=ctr, rpn_shapes=shapes)
        return sel

    # ---- RoI heads ------------------------------------------------------------------------------------------
    def _roi_heads(self, feats, sel, image_hw, keep: Optional[dict] = None):
        c = self.cfg
        n, cap = sel["boxes"].shape[0], sel["cap"]
        boxes = sel["boxes"].view(-1, 4)
        fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
        es = fl[0].element_size()
        # algorithmic bytes: the pyramid once + the pooled rows of the REAL RoIs once + their boxes (the padding rows of
        # the fixed-capacity lists are zero-filled, not algorithmic output); `info` carries (real rows, nominal bytes)
        profiling = self.profile is not None or self.profile_hbm is not None
        _real = []

        def real():  # the number of proposals that exist: read from the device once, when first asked (after the pass: resolve_profile)
            if not _real:
                _real.append(int(sel["counts"].sum()))
            return _real[0]
        row_b = c["pooler_resolution"] ** 2 * 256 * es + 20
        pooled_dt = torch.float32 if "pooled" in self.fp32_points else self.dtype
        h1_dt = torch.float32 if self.fp32_points & {"pooled", "h1"} else None  # (the fp32 kernel writes fp32 only)
        # (the padding rows of the per-image lists are not zero-filled unless `keep` hands the pooled rows out: the box head skips the
        # tiles that hold only padding, computes row by row in the others, and nothing downstream reads a padding row)
        pooled = self._hbm("roi_align", lambda: self.pool_rois(feats, boxes, sel["batch_idx"], pooled_dt, fill_padding=keep is not None),
                           lambda: (lambda: sum(f.numel() for f in fl) * es + real() * row_b),
                           lambda o: (lambda: dict(real_rois=real(), list_rows=boxes.shape[0], nominal_bytes=sum(f.numel() for f in fl) * es + boxes.shape[0] * row_b)))
        m = pooled.shape[0]
        # each image's list is [its proposals ..., padding]: the FC tiles that hold only padding rows are skipped (their rows of h1 /
        # box_feats stay unwritten; nothing downstream reads past an image's count)
        seg = (sel["counts"], cap) if self.skip_padding_tiles and not self.fp32_points & {"pooled", "h1"} else None
        h1 = self._linear(pooled, self.fc1_w, self.fc1_b, True, h1_dt, name="roi_heads.box_head.fc1", row_seg=seg, real_rows=real if profiling else None)
        if "pooled" in self.fp32_points and "h1" not in self.fp32_points:
            h1 = h1.to(self.dtype)  # (diagnostic configuration: FC1 ran in fp32, h1 is stored in the fast path's dtype again)
        box_feats = self._linear(h1, self.fc2_w, self.fc2_b, True, torch.float32, name="roi_heads.box_head.fc2", row_seg=seg, real_rows=real if profiling else None)
        pt = ops.box_predictor_tail(box_feats, self.pred_w, self.pred_b, boxes, sel["scores"].view(-1), sel["batch_idx"], image_hw,
                                    c["bbox_reg_weights"], 0 if c["mean_type"] == "geometric" else 1, c["obj_score_thresh"])
        topk1 = c["detections_per_image"]
        keep1, cnt1 = self._hbm("nms_topk(first stage: sort, thr 1.0)", lambda: ops.nms_topk(pt["boxes"], pt["score"], None, pt["cand"], n, cap, sel["counts"],
                                                                                              c["nms_thresh_test"], topk1),
                                n * cap * 24, lambda o: sel["counts"])
        det_boxes = ops.gather_rows(pt["boxes"], cap, keep1, cnt1)
        det_scores = ops.gather_rows(pt["score"], cap, keep1, cnt1)
        det_feats = ops.gather_rows(box_feats, cap, keep1, cnt1)
        emb = ops.gemm_f32(det_feats.view(n * topk1, -1), self.enc_w, self.enc_b)
        rec = ops.gemm_f32(emb, self.dec_w, self.dec_b)
        pcls, mind = ops.pln_tail(emb, self.protos, c["num_known"], c["reps_per_class"], c["unk_thr"], c["unknown_id"], self.class_map,
                                  cnt1, topk1, distance=c["pln_distance"])
        logits = ops.gemm_f32(rec, self.cls_w, self.cls_b)
        # class_map (GraspNet) remaps known ids at the very end; the known/unknown split uses the un-mapped unknown id
        cands = ops.softmax_candidates(logits, c["num_known"], det_boxes.view(-1, 4), det_scores.view(-1), pcls, cnt1, n, topk1,
                                       c["unknown_id"], c["known_score_thresh"], c["unknown_score_thresh"])
        kk, kc = self._hbm("nms_topk(known, per class)", lambda: ops.nms_topk(cands["k_boxes"], cands["k_scores"], cands["k_cls"], None, n, topk1 * c["num_known"],
                                                                              cands["k_count"], c["known_nms_thresh"], c["known_topk"]),
                           lambda: (lambda: int(cands["k_count"].sum()) * 24), lambda o: cands["k_count"])
        uk, uc = self._hbm("nms_topk(unknown, class-agnostic)", lambda: ops.nms_topk(cands["u_boxes"], cands["u_scores"], None, None, n, topk1, cands["u_count"],
                                                                                     c["unknown_nms_thresh"], c["unknown_topk"]),
                           lambda: (lambda: int(cands["u_count"].sum()) * 20), lambda o: cands["u_count"])
        ob, osc, ocl, on = ops.assemble_detections(cands, kk, kc, uk, uc, n, c["unknown_id"], self.class_map)
        if keep is not None:
            keep.update(pooled=self.pooled_bin_major(pooled), h1=h1, box_feats=box_feats, pred=pt, keep1=keep1, cnt1=cnt1, det_boxes=det_boxes,
                        det_scores=det_scores, det_feats=det_feats, emb=emb, rec=rec, pln_class=pcls, min_dist=mind, logits=logits,
                        cands=cands, k_keep=kk, k_keep_count=kc, u_keep=uk, u_keep_count=uc)
        return ob, osc, ocl, on

    # ---- whole path -----------------------------------------------------------------------------------------
    def forward(self, images: torch.Tensor, image_sizes: Optional[Sequence[Tuple[int, int]]] = None, keep: Optional[dict] = None):
        """images: (n,3,h,w) uint8 or float32 BGR on the GPU (one common size; ragged batches are padded by the
        caller). Returns padded (boxes (n,100,4), scores, classes int64, counts) on the GPU."""
        n, _, h, w = images.shape
        d = self.cfg["size_divisibility"]
        hp, wp = (h + d - 1) // d * d, (w + d - 1) // d * d
        if image_sizes is None:
            image_sizes = [(h, w)] * n
        image_hw = torch.tensor(image_sizes, dtype=torch.int32).to(self.device, non_blocking=True)
        return self.forward_device(images, image_hw, hp, wp, keep)

    def forward_device(self, images, image_hw, hp, wp, keep=None):
        feats = self._backbone(images, hp, wp, keep)
        sel = self._rpn(feats, image_hw, keep)
        if keep is not None:
            keep.update(feats=feats, sel=sel)
        return self._roi_heads(feats, sel, image_hw, keep)

    def forward_device_streams(self, images, image_hw, hp, wp, nstreams: int = 4):
        """Same result as forward_device, with the batch split into `nstreams` contiguous micro-batches that run the
        whole path concurrently on separate HIP streams. Images are independent, so this is data parallelism inside
        one GPU: while one micro-batch's kernel drains its last, sparsely occupied wave of workgroups, the other
        streams' kernels fill the idle CUs (the layers of res4/res5/FPN have too few tiles to cover 256 CUs evenly)."""
        n = images.shape[0]
        ns = max(1, min(nstreams, n))
        if ns == 1:
            return self.forward_device(images, image_hw, hp, wp)
        while len(self._streams) < ns:
            self._streams.append(torch.cuda.Stream(device=self.device))
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        outs = []
        base, extra = divmod(n, ns)
        sizes = [base + (1 if i < extra else 0) for i in range(ns)]
        lo = 0
        for i in range(ns):
            hi = lo + sizes[i]
            st = self._streams[i]
            st.wait_event(start)
            with torch.cuda.stream(st), ops.concurrent_streams(ns):
                o = self.forward_device(images[lo:hi], image_hw[lo:hi], hp, wp)
                done = torch.cuda.Event()
                done.record(st)
            if not torch.cuda.is_current_stream_capturing():
                for t in o:
                    t.record_stream(cur)
            cur.wait_event(done)
            outs.append(o)
            lo = hi
        return tuple(torch.cat([o[k] for o in outs]) for k in range(4))

    def capture(self, images, image_hw, hp, wp, nstreams: int = 1):
        """Capture one whole pass (all micro-batch streams, ~110 launches each) into a hipGraph. Returns (graph, outputs):
        `graph.replay()` re-runs the pass on the same input buffers and refreshes `outputs` in place. The path has no
        host syncs, no host-side shape decisions and fixed-capacity outputs, so it is capturable as is; replay removes
        the per-launch host cost, which otherwise bounds the step once several streams are in play."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside capture: lazy one-time work (func attributes, allocator pools)
                self.forward_device_streams(images, image_hw, hp, wp, nstreams)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.forward_device_streams(images, image_hw, hp, wp, nstreams)
        return graph, out

    def known_class_targets(self, cls: torch.Tensor):
        """Classes as the PLN / classifier losses see them: (classes, background id). VOC-COCO: unchanged, background =
        NUM_CLASSES. GraspNet: id_map[cls] (index in the sorted known list, background -> NUM_KNOWN, others and padding -> -1)."""
        if self.id_map is None:
            return cls, self.cfg["num_classes"]
        return self.id_map[cls].contiguous(), self.cfg["num_known"]  # a gather: cls == -1 reads the last slot (-1)

    # ---- training step, forward half ----------------------------------------------------------------------
    @staticmethod
    def pyramid_shapes(hp: int, wp: int):
        """(h, w) of p2..p6 for a padded hp x wp batch: stem 7x7/2 pad 3, max pool 3x3/2 pad 1, stride-2 1x1 convs, p6 = p5 subsampled."""
        down = lambda v: (v - 1) // 2 + 1  # noqa: E731
        h, w = (hp + 6 - 7) // 2 + 1, (wp + 6 - 7) // 2 + 1
        shapes = []
        for _ in range(5):
            h, w = down(h), down(w)
            shapes.append((h, w))
        return shapes

    def rpn_targets_forward(self, lv, n: int, gt_boxes: torch.Tensor, gt_count: torch.Tensor, keys: Dict[str, torch.Tensor],
                            keep: Optional[dict] = None) -> dict:
        """ClsFreeRPN.label_and_sample_anchors (classification_free_rpn.py:320-402): anchor labels, sampling and regression /
        centerness targets. A function of the ground truth, the anchor grid and the sampling keys only -- not of the network's
        outputs -- so the trainer runs it on a side stream beside the backbone."""
        c = self.cfg
        midx, miou, lab, olab = ops.rpn_match_anchors(lv, self.cell_anchors, n, gt_boxes, gt_count, c["rpn_iou_thresholds"],
                                                      c["rpn_iou_thresholds_objectness"])
        if keep is not None:
            keep.update(matched_idx=midx, matched_iou=miou, labels_pre=lab.clone(), obj_labels_pre=olab.clone())
        ops.subsample_labels_(lab, keys["rpn_reg"], c["rpn_batch_size"], c["rpn_positive_fraction"])
        ops.subsample_labels_(olab, keys["rpn_obj"], c["rpn_batch_size"], c["rpn_positive_fraction_objectness"])
        mboxes, ctr_t = ops.rpn_anchor_targets(lv, self.cell_anchors, n, gt_boxes, gt_count, midx, olab)
        return dict(labels=lab, obj_labels=olab, matched_boxes=mboxes, ctr_target=ctr_t)

    def rpn_losses_forward(self, sel: dict, n: int, gt_boxes: torch.Tensor, gt_count: torch.Tensor, keys: Dict[str, torch.Tensor],
                           keep: Optional[dict] = None, targets: Optional[dict] = None):
        """ClsFreeRPN.label_and_sample_anchors + .losses (classification_free_rpn.py:320-491) on the head outputs `sel` carries
        (level-major pred_deltas / pred_ctr). Returns (6 floats: loss_rpn_loc, loss_rpn_ctr, 4 anchor counts; state dict that the
        backward needs: labels, obj_labels, matched_boxes, ctr_target). targets: the result of rpn_targets_forward when the
        caller has already computed it."""
        check_supported_losses(self.cfg)
        c, lv = self.cfg, sel["levels"]
        tg = targets if targets is not None else self.rpn_targets_forward(lv, n, gt_boxes, gt_count, keys, keep)
        lt = loss_types_of(c)
        rpn = ops.rpn_losses_fwd(lv, self.cell_anchors, n, sel["pred_deltas"], sel["pred_ctr"], tg["labels"], tg["obj_labels"], tg["matched_boxes"],
                                 tg["ctr_target"], c["rpn_loc_weight"], c["rpn_ctr_weight"], c["rpn_batch_size"], box_loss=lt["rpn_box"],
                                 ctr_beta=lt["rpn_ctr"][1])
        return rpn, dict(tg)

    def roi_losses_forward(self, feats: Dict[str, torch.Tensor], prop_boxes, prop_scores, prop_counts, gt_boxes, gt_classes, gt_count,
                           keys_roi: torch.Tensor):
        """OpensetROIHeads.label_and_sample_proposals + _forward_box in training mode (osrcnn_roi_heads.py:137-230,282-318):
        proposals are fixed inputs (predict_proposals runs under no_grad, classification_free_rpn.py:575). Returns (losses of the
        four heads as a dict of GPU scalars + 'roi_counts', state dict with every activation the backward reads)."""
        check_supported_losses(self.cfg)
        c = self.cfg
        smp = ops.roi_match_and_sample(prop_boxes, prop_scores, prop_counts, gt_boxes, gt_classes, gt_count, keys_roi,
                                       c["num_classes"], c["roi_batch_size"], c["roi_positive_fraction"], c["roi_iou_threshold"])
        boxes = smp["boxes"].view(-1, 4)
        pooled = self.pool_rois(feats, boxes, smp["batch_idx"])
        m = pooled.shape[0]
        h1 = self._linear(pooled, self.fc1_w, self.fc1_b, True, name="roi_heads.box_head.fc1")
        box_feats = self._linear(h1, self.fc2_w, self.fc2_b, True, torch.float32, name="roi_heads.box_head.fc2")
        pred = ops.gemm_f32(box_feats, self.pred_w, self.pred_b)  # (m,5): 4 deltas + IoU logit
        cls, ious = smp["gt_classes"].view(-1), smp["ious"].view(-1)
        lt = loss_types_of(c)
        box = ops.roi_box_losses_fwd(pred[:, :4], pred[:, 4], boxes, smp["gt_boxes"].view(-1, 4), cls, ious, c["num_classes"],
                                     c["bbox_reg_weights"], c["box_reg_weight"], c["iou_reg_weight"], iou_is_logit=True, box_loss=lt["roi_box"],
                                     iou_beta=lt["roi_iou"][1])
        cls_k, nck = self.known_class_targets(cls)
        emb = ops.gemm_f32(box_feats, self.enc_w, self.enc_b)
        rec = ops.gemm_f32(emb, self.dec_w, self.dec_b)
        dml = ops.pln_loss_fwd(emb, self.protos, cls_k, ious, c["pln_iou_threshold"], c["pln_alpha"], c["pln_beta"], c["pln_loss_weight"],
                               reps=c["reps_per_class"], distance=c["pln_distance"])
        logits = ops.gemm_f32(rec, self.cls_w, self.cls_b)
        ce = ops.softmax_ce_loss_fwd(logits, cls_k, nck, c["cls_loss_weight"])
        state = dict(smp=smp, sampled=smp, boxes=boxes, pooled=self.pooled_bin_major(pooled), h1=h1, box_feats=box_feats, pred=pred, emb=emb, rec=rec, logits=logits,
                     cls=cls, ious=ious, cls_k=cls_k, nck=nck)
        return dict(loss_box_reg=box[0], loss_iou=box[1], loss_dml=dml[0], loss_cls=ce[0], roi_counts=smp["counts"]), state

    def forward_losses(self, images: torch.Tensor, image_hw: torch.Tensor, hp: int, wp: int, gt_boxes: torch.Tensor,
                       gt_classes: torch.Tensor, gt_count: torch.Tensor, keys: Dict[str, torch.Tensor], keep: Optional[dict] = None):
        """GeneralizedRCNN.forward in training mode, forward values only (no gradients yet): the loss dict of
        ClsFreeRPN.forward (classification_free_rpn.py:493-547) and OpensetROIHeads._forward_box (osrcnn_roi_heads.py:282-
        318) for a batch, every tensor staying on the GPU. gt_boxes (n,gmax,4) fp32 / gt_classes (n,gmax) int64 padded,
        gt_count (n) int32. keys: uniform fp32 sampling keys 'rpn_reg' (n,R), 'rpn_obj' (n,R), 'roi' (n, cap+gmax) -- the
        randomness [d2] subsample_labels draws with torch.randperm (see include/osr.h). Returns a dict of 0-d / small GPU
        tensors named as the reference names its losses."""
        c = self.cfg
        n = images.shape[0]
        feats = self._backbone(images, hp, wp, keep)
        sel = self._rpn(feats, image_hw, keep, topk=c["pre_nms_topk_train"])
        if keep is not None:
            keep.update(feats=feats, sel=sel)
        rpn, rpn_state = self.rpn_losses_forward(sel, n, gt_boxes, gt_count, keys, keep)
        roi, roi_state = self.roi_losses_forward(feats, sel["boxes"], sel["scores"], sel["counts"], gt_boxes, gt_classes, gt_count, keys["roi"])
        if keep is not None:
            keep.update(rpn_state)
            keep.update({k: roi_state[k] for k in ("sampled", "pooled", "box_feats", "pred", "emb", "rec", "logits")})
        return dict(loss_rpn_loc=rpn[0], loss_rpn_ctr=rpn[1], rpn_anchor_counts=rpn[2:], **roi)

    @staticmethod
    def to_instances(result, n: int) -> List[dict]:
        """One D2H copy; list of {'pred_boxes','scores','pred_classes'} per image (the fields the evaluators read,
        pascal_voc_evaluation.py:58-61)."""
        ob, osc, ocl, on = [t.cpu() for t in result]
        out = []
        for i in range(n):
            c = int(on[i])
            out.append(dict(pred_boxes=ob[i, :c], scores=osc[i, :c], pred_classes=ocl[i, :c]))
        return out
