"""Host-side mirror of the reference's plug-in surface: the detectron2-style registries and the
registered names the yaml files select --

    META_ARCHITECTURE "GeneralizedRCNN", BACKBONE.NAME "build_resnet_fpn_backbone",
    PROPOSAL_GENERATOR.NAME "ClsFreeRPN"  (classification_free_rpn.py:165), RPN.HEAD_NAME "ClsFreeRPNHead" (:50),
    ROI_HEADS.NAME "OpensetROIHeads" (osrcnn_roi_heads.py:26), ROI_BOX_HEAD.NAME "FastRCNNConvFCHead"

-- with the same constructor configuration keys, forward signatures, output field names and state-dict key names.
The modules own fp32 parameters under detectron2's names (checkpoint compatible); their inference forwards run on
the HIP library through `OpensetRCNNEngine` (no eager path). Training: `model(batched_inputs)` in training mode returns the six
losses as GPU scalars whose sum's `.backward()` runs the explicit HIP backward (no autograd graph is recorded: one
torch.autograd.Function stands for the whole step), so the loop body of train.py:135-146 runs as written with
`solver.build_optimizer` / `build_lr_scheduler`; `GeneralizedRCNN.make_trainer()` returns the object that does the work
(`trainer.step()` = forward + backward + all-reduce + SGD in one call). `ClsFreeRPN.forward(..., gt_instances)` and
`OpensetROIHeads.forward(..., targets)` return `(proposals, losses)` with the reference's keys -- loss VALUES: gradients exist
only through the model-level call, which fuses both modules' backward passes into one explicit sequence of launches.
Feature maps cross these signatures as logical (N,C,H,W) tensors in channels_last memory (= the kernels' NHWC)."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops, parallel
from .config import CfgNode
from .engine import DEFAULT_CFG, OpensetRCNNEngine
from .engine_std import StandardRCNNEngine
from .structures import Boxes, ImageList, Instances, ShapeSpec
from .weights import R50_BLOCKS, R50_MID, fold_frozen_bn


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._map: Dict[str, object] = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do(o.__name__, o)
                return o
            return deco
        self._do(obj.__name__, obj)
        return obj

    def _do(self, name, obj):
        assert name not in self._map, f"'{name}' already registered in '{self._name}'"
        self._map[name] = obj

    def get(self, name: str):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry (have: {sorted(self._map)})")
        return self._map[name]

    def __contains__(self, name):
        return name in self._map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")



def pad_ground_truth(instances: List["Instances"]):
    """list[Instances{gt_boxes, gt_classes}] -> padded CPU tensors (n,gmax,4) fp32, (n,gmax) int64, (n) int32 counts."""
    n = len(instances)
    gmax = max(1, max(len(x) for x in instances))
    gt = torch.zeros((n, gmax, 4), dtype=torch.float32)
    gcls = torch.zeros((n, gmax), dtype=torch.int64)
    gcnt = torch.zeros((n,), dtype=torch.int32)
    for i, x in enumerate(instances):
        k = len(x)
        gcnt[i] = k
        if k:
            gt[i, :k] = x.gt_boxes.tensor.float().cpu()
            gcls[i, :k] = x.gt_classes.cpu()
    return gt, gcls, gcnt


class _ExplicitBackward(torch.autograd.Function):
    """Stands for the whole training step in torch's autograd: forward hands out the six loss values the HIP kernels computed,
    backward runs OpensetRCNNTrainer._backward (the explicit sequence of data-gradient / weight-gradient launches) when
    `losses.backward()` reaches it. The gradients land in the trainer's flat fp32 buffer, where the optimizer mirror reads them."""

    @staticmethod
    def forward(ctx, hook, trainer, saved, n, *values):
        ctx.trainer, ctx.saved, ctx.n = trainer, saved, n
        return tuple(v.detach().clone() for v in values)

    @staticmethod
    def backward(ctx, *grads):
        if ctx.saved is None:
            raise RuntimeError("the HIP backward of this iteration has already run (its saved activations are released)")
        vals = [float(g) if g is not None else 0.0 for g in grads]  # one small D2H read; train.py:138 syncs on the losses anyway
        if min(vals) != max(vals) or vals[0] <= 0.0:
            raise NotImplementedError(f"the HIP backward differentiates a uniformly weighted sum of the six losses (train.py:136); got d total / d loss = {vals}")
        ctx.trainer._backward(ctx.saved, ctx.n, grad_scale=vals[0])
        ctx.trainer.grads_ready = True
        ctx.saved = None
        return (None,) * (4 + len(grads))



def _to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Logical NCHW -> contiguous NHWC view/copy of the kernel dtype (free when already channels_last + dtype)."""
    return x.to(dtype).permute(0, 2, 3, 1).contiguous()


class _EngineOwner(nn.Module):
    """Lazily packs this module's parameters (under canonical names) into an OpensetRCNNEngine."""
    _prefix = ""
    _engine_cls = OpensetRCNNEngine

    def __init__(self):
        super().__init__()
        self._eng: Optional[OpensetRCNNEngine] = None
        self._shared: Optional[OpensetRCNNEngine] = None
        self._eng_cfg: dict = {}
        self._class_map = None
        self.kernel_dtype = torch.float16

    def engine(self) -> OpensetRCNNEngine:
        if self._shared is not None:
            return self._shared
        if self._eng is None:
            sd = {self._prefix + k: v.detach() for k, v in self.state_dict().items()}
            dev = next(iter(sd.values())).device
            if dev.type != "cuda":
                raise ops.OsrError("the model must be on the GPU (model.to('cuda')): the HIP path has no CPU fallback")
            sd = fold_frozen_bn({k: v.cpu() for k, v in sd.items()})
            if self._engine_cls is OpensetRCNNEngine:
                self._eng = OpensetRCNNEngine(sd, self._eng_cfg, self.kernel_dtype, str(dev), self._class_map)
            else:
                self._eng = self._engine_cls(sd, self._eng_cfg, self.kernel_dtype, str(dev))
        return self._eng

    def refresh(self):
        self._eng = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.refresh()
        return r


# ---------------------------------------------------------------------------------------------------------------
# backbone
# ---------------------------------------------------------------------------------------------------------------
class FrozenBatchNorm2d(nn.Module):
    def __init__(self, c: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c) - eps)


class _ConvBN(nn.Module):
    def __init__(self, cin, cout, k, gain=1.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k) * gain * math.sqrt(2.0 / (cout * k * k)))
        self.norm = FrozenBatchNorm2d(cout)


class _ConvBias(nn.Module):
    def __init__(self, cin, cout, k, gain=1.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k) * gain * math.sqrt(2.0 / (cout * k * k)))
        self.bias = nn.Parameter(torch.zeros(cout))


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, first):
        super().__init__()
        if first:
            self.shortcut = _ConvBN(cin, cout, 1, 0.7)
        self.conv1 = _ConvBN(cin, mid, 1)
        self.conv2 = _ConvBN(mid, mid, 3)
        self.conv3 = _ConvBN(mid, cout, 1, 0.5)


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = _ConvBN(3, 64, 7, 0.02)


class _BottomUp(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = _Stem()
        cin = 64
        for si, (nb, mid) in enumerate(zip(R50_BLOCKS, R50_MID)):
            blocks = []
            for b in range(nb):
                blocks.append(_Bottleneck(cin, mid, mid * 4, b == 0))
                cin = mid * 4
            setattr(self, f"res{si + 2}", nn.Sequential(*blocks))


class ResNetFPN(_EngineOwner):
    """[d2] build_resnet_fpn_backbone for R-50 (Base-RCNN-FPN.yaml:3-8): parameters under backbone.bottom_up.* /
    backbone.fpn_{lateral,output}{2..5}.*; forward(x) takes the normalised, padded (N,3,H,W) fp32 batch."""
    _prefix = "backbone."

    def __init__(self, cfg: CfgNode):
        super().__init__()
        assert cfg.MODEL.RESNETS.DEPTH == 50, "only R-50 is on the hot path (VOC-COCO / GraspNet yaml: DEPTH 50)"
        assert cfg.MODEL.RESNETS.STRIDE_IN_1X1 and cfg.MODEL.RESNETS.NORM == "FrozenBN"
        self.bottom_up = _BottomUp()
        for lvl, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
            setattr(self, f"fpn_lateral{lvl}", _ConvBias(c, 256, 1, 0.7))
            setattr(self, f"fpn_output{lvl}", _ConvBias(256, 256, 3, 0.7))
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]
        self.size_divisibility = 32

    def output_shape(self) -> Dict[str, ShapeSpec]:
        return {f"p{l}": ShapeSpec(channels=256, stride=2 ** l) for l in range(2, 7)}

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        n, _, h, w = x.shape
        assert h % 32 == 0 and w % 32 == 0, "pad the batch to a multiple of 32 (ImageList.from_tensors(..., 32))"
        feats = self.engine()._backbone(x.float().contiguous(), h, w, normalized=True)
        return {k: v.permute(0, 3, 1, 2) for k, v in feats.items()}  # logical NCHW, channels_last memory


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg: CfgNode, input_shape=None):
    return ResNetFPN(cfg)


# ---------------------------------------------------------------------------------------------------------------
# CF-RPN
# ---------------------------------------------------------------------------------------------------------------
class _Conv3x3ReLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))


@RPN_HEAD_REGISTRY.register()
class ClsFreeRPNHead(_EngineOwner):
    """classification_free_rpn.py:50-162. forward(features) -> (list[(N,A*4,Hi,Wi)], list[(N,A,Hi,Wi)] sigmoid-ed)."""
    _prefix = "proposal_generator.rpn_head."

    def __init__(self, cfg: CfgNode = None, input_shape: List[ShapeSpec] = None, *, in_channels: int = 256, num_anchors: int = 1,
                 box_dim: int = 4):
        super().__init__()
        if cfg is not None:
            in_channels = input_shape[0].channels
            assert len(set(s.channels for s in input_shape)) == 1, "Each level must have the same channel!"
            assert list(cfg.MODEL.RPN.CONV_DIMS) == [-1], "only the single-conv head is on the hot path"
            num_anchors = len(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0]) * len(cfg.MODEL.ANCHOR_GENERATOR.SIZES[0])
        assert in_channels == 256 and num_anchors == 1 and box_dim == 4, "HIP CF-RPN head: 256 channels, A=1 (VOC-COCO/GraspNet yaml)"
        self.conv = _Conv3x3ReLU(in_channels, in_channels)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, kernel_size=1)
        self.centerness = nn.Conv2d(in_channels, num_anchors, kernel_size=1)
        for m in (self.conv, self.anchor_deltas, self.centerness):  # classification_free_rpn.py:105-108
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, features: List[torch.Tensor]):
        eng = self.engine()
        deltas, ctrs = [], []
        for x in features:
            n, _, h, w = x.shape
            d, c = ops.cfrpn_head_fused(_to_nhwc(x, eng.dtype), eng.w["proposal_generator.rpn_head.conv.w"],
                                        eng.w["proposal_generator.rpn_head.conv.b"], eng.rpn_wtail, eng.rpn_btail)
            deltas.append(d.view(n, h, w, 4).permute(0, 3, 1, 2))
            ctrs.append(c.view(n, h, w, 1).permute(0, 3, 1, 2))
        return deltas, ctrs


@PROPOSAL_GENERATOR_REGISTRY.register()
class ClsFreeRPN(nn.Module):
    """classification_free_rpn.py:165-610. forward(images, features, gt_instances=None) ->
    (list[Instances{proposal_boxes, objectness_logits}], losses dict)."""

    def __init__(self, cfg: CfgNode, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        self.in_features = list(cfg.MODEL.RPN.IN_FEATURES)
        shapes = [input_shape[f] for f in self.in_features]
        self.rpn_head = RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, shapes)
        self.strides = [s.stride for s in shapes]
        self.anchor_sizes = [float(s[0]) for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES]
        assert len(self.anchor_sizes) == len(self.in_features)
        self.pre_nms_topk = {True: cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, False: cfg.MODEL.RPN.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, False: cfg.MODEL.RPN.POST_NMS_TOPK_TEST}  # accepted, unused (F1)
        self.nms_thresh = {True: cfg.MODEL.RPN.NMS_THRESH, False: cfg.MODEL.RPN.NMS_THRESH_TEST}                  # idem
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.loss_weight = {"loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_ctr": cfg.MODEL.RPN.CTR_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT}
        self.rpn_head._eng_cfg = engine_cfg_from(cfg)  # a stand-alone ClsFreeRPN trains / selects with the yaml's hyper-parameters
        self.sampler_generator = torch.Generator().manual_seed(max(int(cfg.SEED), 0))  # uniform keys replacing torch.randperm (H6)

    def select(self, features: Dict[str, torch.Tensor], image_sizes: List[Tuple[int, int]]):
        """Device-resident result of predict_proposals (padded arrays + counts), as the engine consumes it."""
        feats = [features[f] for f in self.in_features]
        deltas, ctrs = self.rpn_head(feats)
        n = feats[0].shape[0]
        shapes = [(f.shape[2], f.shape[3]) for f in feats]
        # (N,4,H,W)/(N,1,H,W) views of NHWC memory -> level-major (N*H*W, 4) / (N*H*W)
        d_cat = torch.cat([d.permute(0, 2, 3, 1).reshape(-1, 4) for d in deltas]).contiguous()
        c_cat = torch.cat([c.permute(0, 2, 3, 1).reshape(-1) for c in ctrs]).contiguous()
        dev = d_cat.device
        lv = ops.make_rpn_levels(shapes, self.strides, n, 1)
        cell = torch.tensor([[[-s / 2, -s / 2, s / 2, s / 2]] for s in self.anchor_sizes], dtype=torch.float32, device=dev)
        hw = torch.tensor(image_sizes, dtype=torch.int32, device=dev)
        sel = ops.rpn_select(lv, cell, c_cat, d_cat, n, hw, int(self.pre_nms_topk[self.training]), self.min_box_size)
        sel.update(pred_deltas=d_cat, pred_ctr=c_cat, levels=lv)
        return sel, hw

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], gt_instances: Optional[List[Instances]] = None):
        sel, _ = self.select(features, images.image_sizes)
        losses = {}
        if self.training:
            # classification_free_rpn.py:531-547: targets + both losses on the head outputs. Loss VALUES (GPU scalars): the
            # gradients of a training step come from the model-level call (see the module docstring).
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            n, dev = len(gt_instances), sel["boxes"].device
            gt, _, gcnt = pad_ground_truth(gt_instances)
            r = sel["pred_ctr"].numel() // n
            keys = {k: torch.rand((n, r), generator=self.sampler_generator).to(dev) for k in ("rpn_reg", "rpn_obj")}
            with torch.no_grad():
                rpn, _ = self.rpn_head.engine().rpn_losses_forward(sel, n, gt.to(dev), gcnt.to(dev), keys)
            losses = {"loss_rpn_loc": rpn[0], "loss_rpn_ctr": rpn[1]}
        # the list-of-Instances API needs the lengths on the host: one D2H copy brings the selection's status word with them
        host = torch.cat((sel["counts"], sel["status_flags"])).cpu().tolist()
        counts = host[:-1]
        if self.training and host[-1] != 0:  # find_top_proposals.py:96-101 (at test time the rows are dropped silently: the kernel did)
            raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
        if self.training:  # the scalars classification_free_rpn.py:459-463,553-554 puts into EventStorage
            pos_reg, neg_reg, pos_obj, neg_obj = [float(v) for v in rpn[2:6].tolist()]
            self.storage = {"rpn/num_pos_anchors": pos_reg / n, "rpn/num_neg_anchors": neg_reg / n, "rpn/obj_num_pos_anchors": pos_obj / n,
                            "rpn/obj_num_neg_anchors": neg_obj / n, "rpn/num_proposals": sum(counts) / max(n, 1)}
        out = []
        for i, size in enumerate(images.image_sizes):
            r = Instances(size)
            r.proposal_boxes = Boxes(sel["boxes"][i, : counts[i]])
            r.objectness_logits = sel["scores"][i, : counts[i]]
            out.append(r)
        return out, losses


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(_EngineOwner):
    """[d2] StandardRPNHead (RPN.HEAD_NAME default; Base-RCNN-FPN.yaml leaves it): 3x3 conv + ReLU, `objectness_logits` 1x1 (A) and
    `anchor_deltas` 1x1 (A*4); init N(0, 0.01), zero bias. forward(features) -> (list[(N,A,Hi,Wi)] logits, list[(N,A*4,Hi,Wi)] deltas)
    -- [d2]'s order: logits first."""
    _prefix = "proposal_generator.rpn_head."
    _engine_cls = StandardRCNNEngine

    def __init__(self, cfg: CfgNode, input_shape: List[ShapeSpec]):
        super().__init__()
        in_channels = input_shape[0].channels
        assert len(set(s.channels for s in input_shape)) == 1, "Each level must have the same channel!"
        assert list(cfg.MODEL.RPN.CONV_DIMS) == [-1] and in_channels == 256
        self.num_anchors = len(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0]) * len(cfg.MODEL.ANCHOR_GENERATOR.SIZES[0])
        self.conv = _Conv3x3ReLU(in_channels, in_channels)
        self.objectness_logits = nn.Conv2d(in_channels, self.num_anchors, kernel_size=1)
        self.anchor_deltas = nn.Conv2d(in_channels, self.num_anchors * 4, kernel_size=1)
        for m in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, features: List[torch.Tensor]):
        eng = self.engine()
        logits, deltas = [], []
        for x in features:
            n, _, h, w = x.shape
            t = ops.conv2d(_to_nhwc(x, eng.dtype), eng.w["proposal_generator.rpn_head.conv.w"], eng.w["proposal_generator.rpn_head.conv.b"], 1, 1, True,
                           out_dtype=torch.float32).view(-1, 256)
            logits.append(ops.gemm_f32(t, eng.rpn_wo, eng.rpn_bo).view(n, h, w, -1).permute(0, 3, 1, 2))
            deltas.append(ops.gemm_f32(t, eng.rpn_wd, eng.rpn_bd).view(n, h, w, -1).permute(0, 3, 1, 2))
        return logits, deltas


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    """[d2] RPN (PROPOSAL_GENERATOR.NAME default, Base-RCNN-FPN.yaml:9-21), inference branch: anchors with the yaml's aspect
    ratios, StandardRPNHead, Box2BoxTransform(RPN.BBOX_REG_WEIGHTS) decode, per-level top PRE_NMS_TOPK, NMS at RPN.NMS_THRESH per
    level, first POST_NMS_TOPK. forward(images, features, gt_instances=None) -> (list[Instances{proposal_boxes,
    objectness_logits}], {}). Training of the stock RPN (BCE objectness + smooth-L1 deltas) is not on the Openset hot path."""

    def __init__(self, cfg: CfgNode, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        self.in_features = list(cfg.MODEL.RPN.IN_FEATURES)
        shapes = [input_shape[f] for f in self.in_features]
        self.rpn_head = RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, shapes)
        self.rpn_head._eng_cfg = engine_cfg_from(cfg)

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], gt_instances: Optional[List[Instances]] = None):
        if self.training:
            raise NotImplementedError("the stock RPN's training branch is outside the Openset hot path (SURVEY.md 8a): train ClsFreeRPN configs")
        eng = self.rpn_head.engine()
        feats = {k: _to_nhwc(features[k], eng.dtype) for k in self.in_features}
        hw = torch.tensor(images.image_sizes, dtype=torch.int32, device=feats[self.in_features[0]].device)
        sel = eng._rpn(feats, hw)
        counts = sel["counts"].cpu().tolist()
        out = []
        for i, size in enumerate(images.image_sizes):
            r = Instances(size)
            r.proposal_boxes = Boxes(sel["boxes"][i, : counts[i]])
            r.objectness_logits = sel["scores"][i, : counts[i]]
            out.append(r)
        return out, {}


# ---------------------------------------------------------------------------------------------------------------
# RoI heads
# ---------------------------------------------------------------------------------------------------------------
@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    """[d2] box head with NUM_FC=2 (Base-RCNN-FPN.yaml:24-27): fc1 (c*7*7 -> 1024), fc2 (1024 -> 1024), ReLU each."""

    def __init__(self, cfg: CfgNode, input_shape: ShapeSpec):
        super().__init__()
        assert cfg.MODEL.ROI_BOX_HEAD.NUM_CONV == 0 and cfg.MODEL.ROI_BOX_HEAD.NUM_FC == 2
        fc = cfg.MODEL.ROI_BOX_HEAD.FC_DIM
        self.fc1 = nn.Linear(input_shape.channels * input_shape.height * input_shape.width, fc)
        self.fc2 = nn.Linear(fc, fc)
        self.output_shape = ShapeSpec(channels=fc)


class OpensetFastRCNNOutputLayers(nn.Module):
    """osrcnn_fast_rcnn.py:148-264: bbox_pred (class-agnostic 4) and iou_pred (1, sigmoid)."""

    def __init__(self, cfg: CfgNode, input_shape: ShapeSpec):
        super().__init__()
        assert cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG, "the Openset yaml files use class-agnostic box regression"
        self.bbox_pred = nn.Linear(input_shape.channels, 4)
        self.iou_pred = nn.Linear(input_shape.channels, 1)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.normal_(self.iou_pred.weight, std=0.01)
        nn.init.constant_(self.bbox_pred.bias, 0)
        nn.init.constant_(self.iou_pred.bias, 0)


class PLN(nn.Module):
    """prototype_learning_network.py:17-97: encoder, decoder, `representatives` (num_known*reps, emd)."""

    def __init__(self, cfg: CfgNode):
        super().__init__()
        fd, ed = cfg.MODEL.ROI_BOX_HEAD.FC_DIM, cfg.MODEL.PLN.EMD_DIM
        if cfg.MODEL.PLN.DISTANCE_TYPE not in ("COS", "L1", "L2"):
            raise ValueError(f"MODEL.PLN.DISTANCE_TYPE '{cfg.MODEL.PLN.DISTANCE_TYPE}': one of COS, L1, L2 (prototype_learning_network.py:155-160)")
        self.encoder = nn.Linear(fd, ed)
        self.decoder = nn.Linear(ed, fd)
        for l in (self.encoder, self.decoder):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        self.representatives = nn.Parameter(torch.randn(cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES * cfg.MODEL.PLN.REPS_PER_CLASS, ed))


class SoftMaxClassifier(nn.Module):
    """softmax_classifier.py:170-245: cls_score (FC_DIM -> num_known+1)."""

    def __init__(self, cfg: CfgNode):
        super().__init__()
        self.cls_score = nn.Linear(cfg.MODEL.ROI_BOX_HEAD.FC_DIM, cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES + 1)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.constant_(self.cls_score.bias, 0)


def engine_cfg_from(cfg: CfgNode) -> dict:
    """yaml keys -> engine hyper-parameters."""
    rh, bh, rpn = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD, cfg.MODEL.RPN
    return dict(
        pixel_mean=tuple(cfg.MODEL.PIXEL_MEAN), pixel_std=tuple(cfg.MODEL.PIXEL_STD),
        anchor_sizes=tuple(float(s[0]) for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES),
        pre_nms_topk_test=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, min_box_size=float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE),
        pooler_resolution=bh.POOLER_RESOLUTION, bbox_reg_weights=tuple(bh.BBOX_REG_WEIGHTS), mean_type=rh.MEAN_TYPE,
        obj_score_thresh=rh.OBJ_SCORE_THRESH_TEST, nms_thresh_test=rh.NMS_THRESH_TEST, detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
        known_score_thresh=rh.KNOWN_SCORE_THRESH, known_nms_thresh=rh.KNOWN_NMS_THRESH, known_topk=rh.KNOWN_TOPK,
        unknown_score_thresh=rh.UNKNOWN_SCORE_THRESH, unknown_nms_thresh=rh.UNKNOWN_NMS_THRESH, unknown_topk=rh.UNKNOWN_TOPK,
        num_classes=rh.NUM_CLASSES, num_known=rh.NUM_KNOWN_CLASSES, reps_per_class=cfg.MODEL.PLN.REPS_PER_CLASS, pln_distance=cfg.MODEL.PLN.DISTANCE_TYPE,
        # the reference hard-codes the unknown id: 80 with --opendet-benchmark, else 1000
        unknown_id=80 if cfg.OPENDET_BENCHMARK else 1000, unk_thr=cfg.MODEL.PLN.UNK_THR,
        # training step
        pre_nms_topk_train=cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, rpn_batch_size=cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
        rpn_positive_fraction=cfg.MODEL.RPN.POSITIVE_FRACTION, rpn_positive_fraction_objectness=cfg.MODEL.RPN.POSITIVE_FRACTION_OBJECTNESS,
        rpn_iou_thresholds=tuple(cfg.MODEL.RPN.IOU_THRESHOLDS), rpn_iou_thresholds_objectness=tuple(cfg.MODEL.RPN.IOU_THRESHOLDS_OBJECTNESS),
        rpn_loc_weight=rpn.BBOX_REG_LOSS_WEIGHT * rpn.LOSS_WEIGHT, rpn_ctr_weight=rpn.CTR_REG_LOSS_WEIGHT * rpn.LOSS_WEIGHT,
        roi_batch_size=rh.BATCH_SIZE_PER_IMAGE, roi_positive_fraction=rh.POSITIVE_FRACTION, roi_iou_threshold=float(rh.IOU_THRESHOLDS[0]),
        box_reg_weight=bh.BBOX_REG_LOSS_WEIGHT, iou_reg_weight=bh.IOU_REG_LOSS_WEIGHT, pln_alpha=cfg.MODEL.PLN.ALPHA, pln_beta=cfg.MODEL.PLN.BETA,
        pln_iou_threshold=cfg.MODEL.PLN.IOU_THRESHOLD, pln_loss_weight=cfg.MODEL.PLN.LOSS_WEIGHT, cls_loss_weight=bh.CLS_LOSS_WEIGHT,
        # the stock detectron2 modules of Base-RCNN-FPN.yaml (BASELINE config 1; engine_std.StandardRCNNEngine)
        anchor_ratios=tuple(float(r) for r in cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0]), post_nms_topk_test=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
        rpn_nms_thresh=cfg.MODEL.RPN.NMS_THRESH, rpn_bbox_reg_weights=tuple(cfg.MODEL.RPN.BBOX_REG_WEIGHTS), std_num_classes=rh.NUM_CLASSES,
        score_thresh_test=rh.SCORE_THRESH_TEST, std_nms_thresh_test=rh.NMS_THRESH_TEST, std_detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
        cls_agnostic_bbox_reg=bool(bh.CLS_AGNOSTIC_BBOX_REG),
        # checked by engine.check_supported_losses when a loss is first computed (inference never needs them)
        loss_types=dict(rpn_box=(rpn.BBOX_REG_LOSS_TYPE, float(rpn.SMOOTH_L1_BETA)), rpn_ctr=(rpn.CTR_REG_LOSS_TYPE, float(rpn.CTR_SMOOTH_L1_BETA)),
                        roi_box=(bh.BBOX_REG_LOSS_TYPE, float(bh.SMOOTH_L1_BETA)), roi_iou=(bh.IOU_REG_LOSS_TYPE, float(bh.IOU_SMOOTH_L1_BETA))),
    )


@ROI_HEADS_REGISTRY.register()
class OpensetROIHeads(_EngineOwner):
    """osrcnn_roi_heads.py:26-329. forward(images, features, proposals, targets=None) ->
    (list[Instances{pred_boxes, scores, pred_classes}], {}) at inference."""
    _prefix = "roi_heads."

    def __init__(self, cfg: CfgNode, input_shape: Dict[str, ShapeSpec], class_id: Optional[torch.Tensor] = None):
        super().__init__()
        self.in_features = self.box_in_features = list(cfg.MODEL.ROI_HEADS.IN_FEATURES)
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        assert cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE == "ROIAlignV2" and cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO == 0
        self.pooler_scales = tuple(1.0 / input_shape[k].stride for k in self.in_features)
        ch = input_shape[self.in_features[0]].channels
        self.box_head = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, ShapeSpec(channels=ch, height=res, width=res))
        self.box_predictor = OpensetFastRCNNOutputLayers(cfg, self.box_head.output_shape)
        self.dml = PLN(cfg)
        self.softmaxcls = SoftMaxClassifier(cfg)
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self._eng_cfg = engine_cfg_from(cfg)
        self._eng_cfg["pooler_scales"] = self.pooler_scales
        # GraspNet: sorted contiguous ids of the known categories (prototype_learning_network.py:80-95); VOC-COCO: identity
        self._class_map = class_id
        self.sampler_generator = torch.Generator().manual_seed(max(int(cfg.SEED), 0))  # uniform keys replacing torch.randperm (H6)

    def forward_device(self, features: Dict[str, torch.Tensor], sel: dict, image_hw: torch.Tensor):
        eng = self.engine()
        feats = {k: _to_nhwc(features[k], eng.dtype) for k in self.in_features}
        return eng._roi_heads(feats, sel, image_hw)

    @staticmethod
    def _pack_proposals(proposals: List[Instances]):
        n = len(proposals)
        dev = proposals[0].proposal_boxes.tensor.device
        counts = [len(p) for p in proposals]
        cap = max(max(counts), 1)
        boxes = torch.zeros((n, cap, 4), dtype=torch.float32, device=dev)
        scores = torch.zeros((n, cap), dtype=torch.float32, device=dev)
        bidx = torch.full((n, cap), -1, dtype=torch.int32, device=dev)
        for i, p in enumerate(proposals):
            boxes[i, : counts[i]] = p.proposal_boxes.tensor
            scores[i, : counts[i]] = p.objectness_logits
            bidx[i, : counts[i]] = i
        return boxes, scores, bidx, torch.tensor(counts, dtype=torch.int32, device=dev), cap

    def _forward_train(self, features: Dict[str, torch.Tensor], proposals: List[Instances], targets: List[Instances]):
        """osrcnn_roi_heads.py:268-277: label_and_sample_proposals + the four head losses. Returns the sampled proposals (with
        gt_classes / gt_boxes / ious attached, as :203-216 does) and the loss VALUES (see the module docstring for gradients)."""
        eng = self.engine()
        boxes, scores, _, counts, cap = self._pack_proposals(proposals)
        n, dev = len(proposals), boxes.device
        gt, gcls, gcnt = pad_ground_truth(targets)
        keys = torch.rand((n, cap + gt.shape[1]), generator=self.sampler_generator).to(dev)
        feats = {k: _to_nhwc(features[k], eng.dtype) for k in self.in_features}
        with torch.no_grad():
            losses, st = eng.roi_losses_forward(feats, boxes, scores, counts, gt.to(dev), gcls.to(dev), gcnt.to(dev), keys)
        smp = st["smp"]
        nsmp = smp["counts"][:, 0].cpu().tolist()  # rows actually sampled per image (<= BATCH_SIZE_PER_IMAGE)
        out = []
        for i, p in enumerate(proposals):
            k = int(nsmp[i])
            r = Instances(p.image_size)
            r.proposal_boxes = Boxes(smp["boxes"][i, :k])
            r.objectness_logits = smp["logits"][i, :k]
            r.gt_classes = smp["gt_classes"][i, :k]
            r.gt_boxes = Boxes(smp["gt_boxes"][i, :k])
            r.ious = smp["ious"][i, :k]
            out.append(r)
        return out, {k: v for k, v in losses.items() if k.startswith("loss_")}

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], proposals: List[Instances],
                targets: Optional[List[Instances]] = None):
        del images
        if self.training:
            assert targets, "'targets' argument is required during training"
            return self._forward_train(features, proposals, targets)
        n = len(proposals)
        boxes, scores, bidx, counts, cap = self._pack_proposals(proposals)
        dev = boxes.device
        sel = dict(boxes=boxes, scores=scores, batch_idx=bidx.view(-1), counts=counts, cap=cap)
        hw = torch.tensor([p.image_size for p in proposals], dtype=torch.int32, device=dev)
        res = OpensetRCNNEngine.to_instances(self.forward_device(features, sel, hw), n)
        out = []
        for r, p in zip(res, proposals):
            inst = Instances(p.image_size)
            inst.pred_boxes = Boxes(r["pred_boxes"])
            inst.scores = r["scores"]
            inst.pred_classes = r["pred_classes"]
            out.append(inst)
        return out, {}


class FastRCNNOutputLayers(nn.Module):
    """[d2] FastRCNNOutputLayers: cls_score (K+1) and bbox_pred (4, or 4K when not class-agnostic); init std 0.01 / 0.001."""

    def __init__(self, cfg: CfgNode, input_shape: ShapeSpec):
        super().__init__()
        k = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.cls_score = nn.Linear(input_shape.channels, k + 1)
        self.bbox_pred = nn.Linear(input_shape.channels, 4 if cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG else 4 * k)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in (self.cls_score, self.bbox_pred):
            nn.init.constant_(l.bias, 0)


@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(_EngineOwner):
    """[d2] StandardROIHeads (Base-RCNN-FPN.yaml:22-28), box branch, inference: ROIPooler(7x7, ROIAlignV2) -> FastRCNNConvFCHead ->
    FastRCNNOutputLayers.inference (softmax, Box2BoxTransform(10,10,5,5), SCORE_THRESH_TEST, per-class NMS_THRESH_TEST,
    TEST.DETECTIONS_PER_IMAGE). forward(images, features, proposals, targets=None) -> (list[Instances{pred_boxes, scores,
    pred_classes}], {})."""
    _prefix = "roi_heads."
    _engine_cls = StandardRCNNEngine

    def __init__(self, cfg: CfgNode, input_shape: Dict[str, ShapeSpec], class_id=None):
        super().__init__()
        assert not cfg.MODEL.MASK_ON and not cfg.MODEL.KEYPOINT_ON, "box branch only (the Openset path has no mask / keypoint heads)"
        self.in_features = self.box_in_features = list(cfg.MODEL.ROI_HEADS.IN_FEATURES)
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        assert cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE == "ROIAlignV2" and cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO == 0
        ch = input_shape[self.in_features[0]].channels
        self.box_head = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, ShapeSpec(channels=ch, height=res, width=res))
        self.box_predictor = FastRCNNOutputLayers(cfg, self.box_head.output_shape)
        self._eng_cfg = engine_cfg_from(cfg)
        self._eng_cfg["pooler_scales"] = tuple(1.0 / input_shape[k].stride for k in self.in_features)

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], proposals: List[Instances], targets=None):
        del images
        if self.training:
            raise NotImplementedError("the stock ROI heads' training branch is outside the Openset hot path (SURVEY.md 8a): train OpensetROIHeads configs")
        eng = self.engine()
        n = len(proposals)
        boxes, scores, bidx, counts, cap = OpensetROIHeads._pack_proposals(proposals)
        sel = dict(boxes=boxes, scores=scores, batch_idx=bidx.view(-1), counts=counts, cap=cap)
        hw = torch.tensor([p.image_size for p in proposals], dtype=torch.int32, device=boxes.device)
        feats = {k: _to_nhwc(features[k], eng.dtype) for k in self.in_features}
        res = OpensetRCNNEngine.to_instances(eng._roi_heads(feats, sel, hw), n)
        out = []
        for r, p in zip(res, proposals):
            out.append(Instances(p.image_size, pred_boxes=Boxes(r["pred_boxes"]), scores=r["scores"], pred_classes=r["pred_classes"]))
        return out, {}


# ---------------------------------------------------------------------------------------------------------------
# meta architecture
# ---------------------------------------------------------------------------------------------------------------
def detector_postprocess(results: Instances, output_height: int, output_width: int) -> Instances:
    """[d2] rescale to the requested output resolution, clip, drop empty boxes."""
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    b = out.pred_boxes.clone()
    b.scale(sx, sy)
    b.clip(out.image_size)
    out.pred_boxes = b
    return out[b.nonempty()]


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(_EngineOwner):
    """[d2] GeneralizedRCNN as the reference builds it (train.py:189): backbone + proposal_generator + roi_heads.
    model(list[dict{image,height,width}]) -> list[{"instances": Instances}] in eval mode (train.py:96)."""
    _prefix = ""

    def __init__(self, cfg: CfgNode, class_id: Optional[torch.Tensor] = None):
        super().__init__()
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg)
        shapes = self.backbone.output_shape()
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, shapes)
        self.roi_heads = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, shapes, class_id)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self._eng_cfg = engine_cfg_from(cfg)
        self._class_map = class_id
        self._engine_cls = StandardRCNNEngine if isinstance(self.roi_heads, StandardROIHeads) else OpensetRCNNEngine
        if self._engine_cls is StandardRCNNEngine:
            self._eng_cfg["pooler_scales"] = self.roi_heads._eng_cfg["pooler_scales"]
        self._trainer = None
        self._hook: Optional[torch.Tensor] = None
        self.sampler_generator = torch.Generator().manual_seed(max(int(cfg.SEED), 0))  # uniform keys replacing torch.randperm (H6)

    @property
    def device(self):
        return self.pixel_mean.device

    def engine(self) -> OpensetRCNNEngine:
        eng = super().engine()
        for child in (self.backbone, self.proposal_generator.rpn_head, self.roi_heads):
            child._shared = eng  # one packed copy of the weights for the whole model
        return eng

    def preprocess_image(self, batched_inputs: List[dict]) -> ImageList:
        images = [(x["image"].to(self.device).float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        return ImageList.from_tensors(images, self.backbone.size_divisibility)

    def forward(self, batched_inputs: List[dict]):
        """[d2] GeneralizedRCNN.forward. Eval mode: list[{"instances": Instances}]. Training mode (train.py:135): the dict of the six
        losses, GPU scalars whose sum's .backward() runs the explicit HIP backward of this iteration into the trainer's gradient
        buffer (`solver.build_optimizer(cfg, model).step()` then all-reduces and applies it)."""
        if not self.training:
            return self.inference(batched_inputs)
        trainer = self.trainer()
        tensors = self._train_tensors(batched_inputs, self.sampler_generator)
        with torch.no_grad():
            # (several ranks: all apply the same verdicts -- those of the updates up to two iterations back -- at the same iteration)
            trainer.poll_overflow(wait=parallel.is_dist(), lag=trainer.MULTI_RANK_LAG if parallel.is_dist() else 0)
            losses, saved = trainer._forward(*tensors)
        names = list(losses)
        outs = _ExplicitBackward.apply(self._grad_hook(), trainer, saved, tensors[0].shape[0], *[losses[k] for k in names])
        trainer.grads_ready = False
        return dict(zip(names, outs))

    def event_scalars(self) -> Dict[str, float]:
        """The ten scalars the reference's training forward puts into EventStorage (rpn/num_{pos,neg}_anchors, rpn/obj_num_{pos,neg}_anchors,
        rpn/num_proposals, roi_head/num_{fg,bg}_samples, softmax_classifier/{cls_accuracy,fg_cls_accuracy,false_negative}) for the last
        training-mode forward of this model (OpensetRCNNTrainer.event_scalars)."""
        return self._trainer.event_scalars() if self._trainer is not None else {}

    def _grad_hook(self) -> torch.Tensor:
        """A 0-d leaf that requires grad: what makes torch call _ExplicitBackward.backward (it receives no gradient itself)."""
        if self._hook is None or self._hook.device != self.device:
            self._hook = torch.zeros((), device=self.device, requires_grad=True)
        return self._hook

    def trainer(self):
        """The trainer behind training-mode forwards, built on first use with the module's current parameters
        (`solver.build_optimizer` sets its learning rate, momentum and weight decay from cfg.SOLVER)."""
        if self._trainer is None:
            self._trainer = self.make_trainer()
        return self._trainer

    def train(self, mode: bool = True):
        """Leaving training mode writes the trained masters back into the module's parameters, so that eval-mode forwards,
        state_dict() and checkpoints see them ([d2] trains the module's parameters in place)."""
        if not mode and self.training and self._trainer is not None:
            self.load_trainer_state(self._trainer, keep_trainer=True)
        return super().train(mode)

    def make_trainer(self, lr: float = 0.005, momentum: float = 0.9, weight_decay: float = 1e-4, loss_scale: float = 1024.0):
        """The training loop body of train.py:132-148 as one object: `losses = trainer.step(...)` replaces
        `loss_dict = model(data); losses.backward(); optimizer.step()` (there is no autograd graph on the HIP path). The trainer
        owns fp32 master copies of this model's trainable parameters (res3+ weights un-folded from their FrozenBN, FPN, heads);
        `load_trainer_state(trainer)` writes them back into the module for evaluation / checkpointing."""
        from .train import OpensetRCNNTrainer
        if self.device.type != "cuda":
            raise ops.OsrError("the model must be on the GPU (model.to('cuda')): the HIP path has no CPU fallback")
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        bn = {}
        for k in sd:
            if k.endswith(".norm.weight"):
                pre = k[: -len(".norm.weight")]
                scale = sd[pre + ".norm.weight"] * (sd[pre + ".norm.running_var"] + 1e-5).rsqrt()
                bn[pre] = (sd[pre + ".weight"], scale)
        # class_map: the GraspNet id_map of the PLN / classifier losses (prototype_learning_network.py:80-95) -- without it the
        # trainer would treat dataset ids 0..NUM_KNOWN-1 as the known classes
        return OpensetRCNNTrainer(fold_frozen_bn(sd), self._eng_cfg, self.kernel_dtype, str(self.device), lr=lr, momentum=momentum,
                                  weight_decay=weight_decay, loss_scale=loss_scale, frozen_bn=bn, class_map=self._class_map)

    def load_trainer_state(self, trainer, keep_trainer: bool = False) -> None:
        sd = dict(self.state_dict())
        for k, v in trainer.export_state_dict().items():
            assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
            sd[k] = v
        kept = self._trainer if keep_trainer else None
        self.load_state_dict(sd)
        self._trainer = kept  # (load_state_dict drops a cached trainer: its masters would no longer match the module)

    def refresh(self):
        super().refresh()
        self._trainer = None
        for child in (self.backbone, self.proposal_generator.rpn_head, self.roi_heads):
            child._shared = None
            child.refresh()

    def _stack_images(self, imgs: List[torch.Tensor]):
        """One (N,3,H,W) device tensor for a list of CHW images: same-size uint8/float32 images are stacked as they are (the
        fused normalise+pad kernel takes both); a ragged batch is padded bottom/right with the pixel mean, so that the
        normalised padding is exactly zero, as ImageList.from_tensors does after normalisation ([d2])."""
        sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in imgs]
        if len(set(sizes)) == 1 and len(set(i.dtype for i in imgs)) == 1 and imgs[0].dtype in (torch.uint8, torch.float32):
            return torch.stack([i.to(self.device) for i in imgs]), sizes
        hm, wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
        batch = self.pixel_mean.to(self.device).float().expand(3, hm, wm).unsqueeze(0).repeat(len(imgs), 1, 1, 1)
        for k, im in enumerate(imgs):
            batch[k, :, : sizes[k][0], : sizes[k][1]] = im.to(self.device).float()
        return batch, sizes

    def _train_tensors(self, batched_inputs: List[dict], generator: Optional[torch.Generator] = None):
        """The tensors a training iteration consumes: stacked images, their true sizes, the padded size, ground truth padded to
        the batch maximum, and the uniform keys that replace torch.randperm in the two samplers (seeded by `generator`)."""
        ecfg = {**DEFAULT_CFG, **self._eng_cfg}
        batch, sizes = self._stack_images([x["image"] for x in batched_inputs])
        n, dev = len(sizes), self.device
        gt, gcls, gcnt = pad_ground_truth([x["instances"] for x in batched_inputs])
        gmax = gt.shape[1]
        d = ecfg["size_divisibility"]
        hm, wm = int(batch.shape[-2]), int(batch.shape[-1])
        hp, wp = (hm + d - 1) // d * d, (wm + d - 1) // d * d
        shapes = [((hp // s), (wp // s)) for s in ecfg["fpn_strides"][:4]]
        shapes.append(((shapes[-1][0] - 1) // 2 + 1, (shapes[-1][1] - 1) // 2 + 1))
        r = sum(a * b for a, b in shapes)
        cap = sum(min(ecfg["pre_nms_topk_train"], a * b) for a, b in shapes)
        keys = {k: torch.rand(shape, generator=generator).to(dev) for k, shape in
                (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + gmax)))}
        hw = torch.tensor(sizes, dtype=torch.int32, device=dev)
        return batch, hw, hp, wp, gt.to(dev), gcls.to(dev), gcnt.to(dev), keys

    @torch.no_grad()
    def losses_forward(self, batched_inputs: List[dict], generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """The loss dict GeneralizedRCNN.forward returns in training mode ([d2]; train.py:189 trainer loop), forward values
        only: loss_rpn_loc, loss_rpn_ctr, loss_box_reg, loss_iou, loss_dml, loss_cls. Each input dict carries "image" and
        "instances" (gt_boxes: Boxes, gt_classes); a ragged batch is padded to its largest image. `generator` seeds the
        uniform keys that replace torch.randperm in the two samplers."""
        out = self.engine().forward_losses(*self._train_tensors(batched_inputs, generator))
        return {k: v for k, v in out.items() if k.startswith("loss_")}

    @torch.no_grad()
    def train_step(self, trainer, batched_inputs: List[dict], generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """One iteration of train.py:132-148 on the trainer made by make_trainer(): forward, explicit backward, gradient
        all-reduce over the ranks and the SGD update; returns the loss dict (GPU scalars, this rank's values)."""
        out = trainer.step(*self._train_tensors(batched_inputs, generator))
        return {k: v for k, v in out.items() if k.startswith("loss_")}

    @torch.no_grad()
    def inference(self, batched_inputs: List[dict], do_postprocess: bool = True):
        eng = self.engine()
        batch, sizes = self._stack_images([x["image"] for x in batched_inputs])
        res = eng.forward(batch, sizes)
        if do_postprocess:  # [d2] detector_postprocess on the device: rescale, clip, drop empties
            outs = [(int(inp.get("height", s[0])), int(inp.get("width", s[1]))) for inp, s in zip(batched_inputs, sizes)]
            scale = torch.tensor([[ow / s[1], oh / s[0]] for (oh, ow), s in zip(outs, sizes)], dtype=torch.float32, device=self.device)
            res = ops.detector_postprocess(res[0], res[1], res[2], res[3], scale, torch.tensor(outs, dtype=torch.int32, device=self.device))
        else:
            outs = sizes
        insts = OpensetRCNNEngine.to_instances(res, len(sizes))
        return [{"instances": Instances(o, pred_boxes=Boxes(r["pred_boxes"]), scores=r["scores"], pred_classes=r["pred_classes"])}
                for r, o in zip(insts, outs)]


def build_model(cfg: CfgNode, class_id: Optional[torch.Tensor] = None) -> nn.Module:
    """[d2] detectron2.modeling.build_model (train.py:189)."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg, class_id)
    return model.to(torch.device(cfg.MODEL.DEVICE))
