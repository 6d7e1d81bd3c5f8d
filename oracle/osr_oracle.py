"""CPU oracle for the Openset R-CNN per-image detection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path (``openset-rcnn_amd``) never
imports this module and fails loudly when its HIP library is missing.

PARITY UNPINNED.  The reference (/root/reference, Yifei-Y/Openset-RCNN) ships no tests,
golden vectors or fixtures, and it cannot be imported in the build container
(``ModuleNotFoundError: fvcore`` / ``detectron2``).  This file is
a from-scratch restatement, in plain torch-CPU / numpy ops, of
  * the reference's own arithmetic (cited ``file:line`` relative to /root/reference), and
  * the published algorithms of its un-vendored dependencies, detectron2 v0.6 /
    torchvision 0.11 / fvcore (marked [d2-mem]; pinned by README.md:20-33 of the
    reference) at the reference's call sites.
It is pinned only by the analytic known-answer tests in ``tests/test_oracle_kat.py``.

Tie-breaking that the reference stack leaves undefined (``torch.topk`` / ``sort`` /
``nms`` order among equal scores, ``argmin`` among equal distances) is defined here as
*stable: the lower original index wins*; the HIP kernels implement the same rule.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Boxes ([d2-mem] detectron2.structures.Boxes / pairwise_iou)
# --------------------------------------------------------------------------------------


def box_area(b: torch.Tensor) -> torch.Tensor:
    """[d2-mem] Boxes.area: (x2-x1)*(y2-y1)."""
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_clip(b: torch.Tensor, image_size: Tuple[int, int]) -> torch.Tensor:
    """[d2-mem] Boxes.clip((h, w)): clamp x to [0,w], y to [0,h]. Returns a new tensor."""
    h, w = image_size
    x1 = b[:, 0].clamp(min=0, max=w)
    y1 = b[:, 1].clamp(min=0, max=h)
    x2 = b[:, 2].clamp(min=0, max=w)
    y2 = b[:, 3].clamp(min=0, max=h)
    return torch.stack((x1, y1, x2, y2), dim=-1)


def box_nonempty(b: torch.Tensor, threshold: float = 0.0) -> torch.Tensor:
    """[d2-mem] Boxes.nonempty: (w > thr) & (h > thr)."""
    return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)


def pairwise_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """[d2-mem] detectron2.structures.pairwise_iou: inter/(a1+a2-inter), 0 where inter==0.
    Call sites: classification_free_rpn.py:365, osrcnn_roi_heads.py:187."""
    a1, a2 = box_area(b1), box_area(b2)
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def elementwise_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """diag(pairwise_iou(b1, b2)) without the PxP matrix (SURVEY F9;
    box_regression_w_iou.py:56)."""
    a1, a2 = box_area(b1), box_area(b2)
    wh = (torch.min(b1[:, 2:], b2[:, 2:]) - torch.max(b1[:, :2], b2[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    return torch.where(inter > 0, inter / (a1 + a2 - inter), torch.zeros(1, dtype=inter.dtype))


# --------------------------------------------------------------------------------------
# Anchors ([d2-mem] DefaultAnchorGenerator; classification_free_rpn.py:289,514)
# --------------------------------------------------------------------------------------

FPN_STRIDES = (4, 8, 16, 32, 64)
ANCHOR_SIZES = (32, 64, 128, 256, 512)


def level_shapes(h: int, w: int, strides: Sequence[int] = FPN_STRIDES) -> List[Tuple[int, int]]:
    """Feature-map sizes of p2..p6 for a (h, w) padded input (h, w divisible by 32).
    p6 = max_pool2d(p5, k=1, s=2) -> floor((x-1)/2)+1."""
    shapes = []
    for s in strides[:4]:
        shapes.append((h // s, w // s))
    h5, w5 = shapes[-1]
    shapes.append(((h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1))
    return shapes[: len(strides)]


def anchor_grid(shapes: Sequence[Tuple[int, int]], strides: Sequence[int] = FPN_STRIDES,
                sizes: Sequence[float] = ANCHOR_SIZES, ratios: Sequence[float] = (1.0,),
                offset: float = 0.0) -> List[torch.Tensor]:
    """[d2-mem] DefaultAnchorGenerator: per level, cell anchors for (size, ratio):
    area=size^2, w=sqrt(area/ratio), h=ratio*w, box=[-w/2,-h/2,w/2,h/2]; grid shifts
    x=j*stride+offset*stride, y=i*stride+...; row-major (y outer, x inner), A innermost."""
    out = []
    for (hh, ww), s, z in zip(shapes, strides, sizes):
        cells = []
        for r in ratios:
            area = float(z) ** 2
            w_ = math.sqrt(area / r)
            h_ = r * w_
            cells.append([-w_ / 2.0, -h_ / 2.0, w_ / 2.0, h_ / 2.0])
        cell = torch.tensor(cells, dtype=torch.float32)
        sx = torch.arange(offset * s, ww * s, step=s, dtype=torch.float32)
        sy = torch.arange(offset * s, hh * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
    return out


# --------------------------------------------------------------------------------------
# Box transforms ([d2-mem] Box2BoxTransformLinear / Box2BoxTransform)
# --------------------------------------------------------------------------------------


def ltrb_apply_deltas(deltas: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    """[d2-mem] Box2BoxTransformLinear(normalize_by_size=True).apply_deltas, used at
    classification_free_rpn.py:607: d=relu(delta)*[w,h,w,h]; box=[cx-l, cy-t, cx+r, cy+b]."""
    d = F.relu(deltas)
    cx = 0.5 * (anchors[:, 0] + anchors[:, 2])
    cy = 0.5 * (anchors[:, 1] + anchors[:, 3])
    sw = anchors[:, 2] - anchors[:, 0]
    sh = anchors[:, 3] - anchors[:, 1]
    d = d * torch.stack((sw, sh, sw, sh), dim=1)
    return torch.stack((cx - d[:, 0], cy - d[:, 1], cx + d[:, 2], cy + d[:, 3]), dim=1)


def ltrb_get_deltas(src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """[d2-mem] Box2BoxTransformLinear(normalize_by_size=True).get_deltas
    (classification_free_rpn.py:393)."""
    cx = 0.5 * (src[:, 0] + src[:, 2])
    cy = 0.5 * (src[:, 1] + src[:, 3])
    d = torch.stack((cx - tgt[:, 0], cy - tgt[:, 1], tgt[:, 2] - cx, tgt[:, 3] - cy), dim=1)
    sw = src[:, 2] - src[:, 0]
    sh = src[:, 3] - src[:, 1]
    return d / torch.stack((sw, sh, sw, sh), dim=1)


SCALE_CLAMP = math.log(1000.0 / 16)


def b2b_apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor,
                     weights=(10.0, 10.0, 5.0, 5.0)) -> torch.Tensor:
    """[d2-mem] Box2BoxTransform.apply_deltas (osrcnn_fast_rcnn.py:231,423)."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx = deltas[:, 0] / wx
    dy = deltas[:, 1] / wy
    dw = torch.clamp(deltas[:, 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3] / wh, max=SCALE_CLAMP)
    pcx = dx * w + cx
    pcy = dy * h + cy
    pw = torch.exp(dw) * w
    ph = torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def b2b_get_deltas(src: torch.Tensor, tgt: torch.Tensor, weights=(10.0, 10.0, 5.0, 5.0)) -> torch.Tensor:
    """[d2-mem] Box2BoxTransform.get_deltas (targets of osrcnn_fast_rcnn.py:330)."""
    sw = src[:, 2] - src[:, 0]
    sh = src[:, 3] - src[:, 1]
    scx = src[:, 0] + 0.5 * sw
    scy = src[:, 1] + 0.5 * sh
    tw = tgt[:, 2] - tgt[:, 0]
    th = tgt[:, 3] - tgt[:, 1]
    tcx = tgt[:, 0] + 0.5 * tw
    tcy = tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tcx - scx) / sw, wy * (tcy - scy) / sh,
                        ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


# --------------------------------------------------------------------------------------
# Backbone: ResNet-50 + FPN ([d2-mem] build_resnet_fpn_backbone; Base-RCNN-FPN.yaml:3-8)
# --------------------------------------------------------------------------------------

R50_BLOCKS = (3, 4, 6, 3)


def frozen_bn_fold(conv_w: torch.Tensor, bn_w, bn_b, bn_mean, bn_var, eps: float = 1e-5):
    """[d2-mem] FrozenBatchNorm2d: y = x*scale + shift, scale = w*rsqrt(var+eps),
    shift = b - mean*scale. Folded into the preceding bias-free conv."""
    scale = bn_w * (bn_var + eps).rsqrt()
    shift = bn_b - bn_mean * scale
    return conv_w * scale.view(-1, 1, 1, 1), shift


def make_r50_fpn_params(seed: int = 0, res_gain: float = 0.5) -> Dict[str, torch.Tensor]:
    """Random, already BN-folded ResNet-50+FPN parameters with detectron2's state-dict
    names (checkpoint row). He-normal fan_out init; the last conv of each
    bottleneck is scaled by ``res_gain`` so activations of the synthetic net stay in fp16
    range (no checkpoint is reachable offline)."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, gain=1.0, bias_std=0.05):
        std = gain * math.sqrt(2.0 / (cout * k * k))
        p[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std
        p[name + ".bias"] = torch.randn(cout, generator=g) * bias_std

    conv("backbone.bottom_up.stem.conv1", 64, 3, 7, gain=0.02)  # pixels are O(100)
    cin = 64
    for si, (nb, mid) in enumerate(zip(R50_BLOCKS, (64, 128, 256, 512))):
        cout = mid * 4
        for b in range(nb):
            pre = f"backbone.bottom_up.res{si + 2}.{b}"
            if b == 0:
                conv(pre + ".shortcut", cout, cin, 1, gain=0.7)
            conv(pre + ".conv1", mid, cin, 1)
            conv(pre + ".conv2", mid, mid, 3)
            conv(pre + ".conv3", cout, mid, 1, gain=res_gain)
            cin = cout
    for lvl, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
        conv(f"backbone.fpn_lateral{lvl}", 256, c, 1, gain=0.7)
        conv(f"backbone.fpn_output{lvl}", 256, 256, 3, gain=0.7)
    return p


def resnet_fpn_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], quant=None) -> Dict[str, torch.Tensor]:
    """[d2-mem] ResNet-50 (MSRA: stride in the 1x1) + FPN + LastLevelMaxPool on BN-folded
    parameters. ``quant`` (optional) is applied to every stored activation and mirrors the
    HIP path's fp16/bf16 storage points."""
    q = quant if quant is not None else (lambda t: t)

    def cv(t, name, stride=1, pad=0):
        return F.conv2d(t, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=pad)

    t = q(F.relu(cv(x, "backbone.bottom_up.stem.conv1", 2, 3)))
    t = F.max_pool2d(t, kernel_size=3, stride=2, padding=1)
    feats = {}
    for si, nb in enumerate(R50_BLOCKS):
        for b in range(nb):
            pre = f"backbone.bottom_up.res{si + 2}.{b}"
            stride = 2 if (b == 0 and si > 0) else 1
            sc = q(cv(t, pre + ".shortcut", stride)) if b == 0 else t
            o = q(F.relu(cv(t, pre + ".conv1", stride)))
            o = q(F.relu(cv(o, pre + ".conv2", 1, 1)))
            t = q(F.relu(cv(o, pre + ".conv3") + sc))
        feats[f"res{si + 2}"] = t
    out = {}
    prev = q(cv(feats["res5"], "backbone.fpn_lateral5"))
    out["p5"] = q(cv(prev, "backbone.fpn_output5", 1, 1))
    for lvl in (4, 3, 2):
        td = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = q(cv(feats[f"res{lvl}"], f"backbone.fpn_lateral{lvl}") + td)
        out[f"p{lvl}"] = q(cv(prev, f"backbone.fpn_output{lvl}", 1, 1))
    out["p6"] = F.max_pool2d(out["p5"], kernel_size=1, stride=2, padding=0)
    return out


def preprocess_images(images: Sequence[torch.Tensor], pixel_mean=(103.53, 116.28, 123.675),
                      pixel_std=(1.0, 1.0, 1.0), size_divisibility: int = 32):
    """[d2-mem] GeneralizedRCNN.preprocess_image + ImageList.from_tensors:
    (x-mean)/std per channel, zero-pad bottom/right to a multiple of 32."""
    mean = torch.tensor(pixel_mean, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(pixel_std, dtype=torch.float32).view(3, 1, 1)
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    hm = max(s[0] for s in sizes)
    wm = max(s[1] for s in sizes)
    d = size_divisibility
    hm, wm = (hm + d - 1) // d * d, (wm + d - 1) // d * d
    batch = torch.zeros(len(images), 3, hm, wm, dtype=torch.float32)
    for i, im in enumerate(images):
        batch[i, :, : im.shape[-2], : im.shape[-1]] = (im.float() - mean) / std
    return batch, sizes


# --------------------------------------------------------------------------------------
# CF-RPN head + proposal selection
# --------------------------------------------------------------------------------------


def cfrpn_head(feat: torch.Tensor, p: Dict[str, torch.Tensor], prefix="proposal_generator.rpn_head"):
    """ClsFreeRPNHead.forward for one level (classification_free_rpn.py:157-161):
    t=relu(conv3x3(x)); t=t/max(||t||_2 over C, 1e-12); deltas=conv1x1(t);
    ctr=sigmoid(conv1x1(t)). Returns (N,4A,H,W), (N,A,H,W)."""
    t = F.relu(F.conv2d(feat, p[prefix + ".conv.weight"], p[prefix + ".conv.bias"], padding=1))
    t = F.normalize(t, p=2, dim=1)
    d = F.conv2d(t, p[prefix + ".anchor_deltas.weight"], p[prefix + ".anchor_deltas.bias"])
    c = F.conv2d(t, p[prefix + ".centerness.weight"], p[prefix + ".centerness.bias"]).sigmoid()
    return d, c


def cfrpn_head_tail(t: torch.Tensor, p: Dict[str, torch.Tensor], prefix="proposal_generator.rpn_head"):
    """The part of ClsFreeRPNHead.forward after the 3x3 conv+ReLU (:159-161), on a
    channels-last hidden state t of shape (P, C). Returns deltas (P,4), ctr (P,)."""
    tn = F.normalize(t, p=2, dim=1)
    wd = p[prefix + ".anchor_deltas.weight"].view(4, -1)
    wc = p[prefix + ".centerness.weight"].view(1, -1)
    d = tn @ wd.t() + p[prefix + ".anchor_deltas.bias"]
    c = (tn @ wc.t() + p[prefix + ".centerness.bias"]).sigmoid().view(-1)
    return d, c


def flatten_head_outputs(deltas: List[torch.Tensor], ctrs: List[torch.Tensor], box_dim: int = 4):
    """ClsFreeRPN.forward reshapes (classification_free_rpn.py:518-529):
    (N,A*B,H,W)->(N,H*W*A,B) and (N,A,H,W)->(N,H*W*A)."""
    d = [x.view(x.shape[0], -1, box_dim, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2)
         for x in deltas]
    c = [x.permute(0, 2, 3, 1).flatten(1) for x in ctrs]
    return d, c


def stable_topk(scores: torch.Tensor, k: int):
    """torch.topk with the tie rule made explicit: value-descending, lower index first
    among equals (-0.0 == +0.0)."""
    vals, idx = torch.sort(scores, dim=-1, descending=True, stable=True)
    return vals[..., :k], idx[..., :k]


def find_top_rpn_proposals(proposals: List[torch.Tensor], scores: List[torch.Tensor],
                           image_sizes: Sequence[Tuple[int, int]], pre_nms_topk: int,
                           min_box_size: float = 0.0, training: bool = False):
    """find_top_rpn_proposals (find_top_proposals.py:60-127). NMS and post_nms_topk are
    commented out in the reference (:112-126): output = per-level top-k,
    level-major, score-descending inside a level, minus non-finite (:96-104) and empty
    (:108-110) boxes, clipped (:105). Returns per image (boxes (R,4), scores (R,),
    src_index (R,) int64 into the concatenated per-image anchor list)."""
    n_img = len(image_sizes)
    tk_s, tk_b, tk_i = [], [], []
    off = 0
    for prop_l, sc_l in zip(proposals, scores):
        k = min(sc_l.shape[1], pre_nms_topk)
        v, idx = stable_topk(sc_l, k)
        tk_s.append(v)
        tk_b.append(prop_l[torch.arange(n_img)[:, None], idx])
        tk_i.append(idx + off)
        off += sc_l.shape[1]
    tk_s, tk_b, tk_i = torch.cat(tk_s, 1), torch.cat(tk_b, 1), torch.cat(tk_i, 1)
    results = []
    for n, size in enumerate(image_sizes):
        b, s, i = tk_b[n], tk_s[n], tk_i[n]
        valid = torch.isfinite(b).all(dim=1) & torch.isfinite(s)
        if not bool(valid.all()):
            if training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            b, s, i = b[valid], s[valid], i[valid]
        b = box_clip(b, size)
        keep = box_nonempty(b, min_box_size)
        results.append((b[keep], s[keep], i[keep]))
    return results


# --------------------------------------------------------------------------------------
# RoI pooling ([d2-mem] ROIPooler + torchvision.ops.roi_align aligned=True)
# --------------------------------------------------------------------------------------


def assign_levels(boxes: torch.Tensor, min_level=2, max_level=5, canonical_size=224, canonical_level=4):
    """[d2-mem] detectron2.modeling.poolers.assign_boxes_to_levels:
    floor(canonical_level + log2(sqrt(area)/canonical_size + 1e-8)) clamped; returns
    0-based level index (osrcnn_roi_heads.py:108-113,306)."""
    sizes = torch.sqrt(box_area(boxes))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_align_ref(feat: np.ndarray, rois: np.ndarray, scale: float, out_size: int = 7,
                  sampling_ratio: int = 0, aligned: bool = True) -> np.ndarray:
    """[d2-mem] torchvision 0.11 roi_align CPU kernel semantics (fp32), per RoI loops.
    feat (N,C,H,W) float32; rois (K,5) [batch,x1,y1,x2,y2] -> (K,C,out,out).
    Slow reference implementation for small cases; the C oracle restates the same
    algorithm for the large ones and is cross-checked against this."""
    f32 = np.float32
    n, c, height, width = feat.shape
    k = rois.shape[0]
    out = np.zeros((k, c, out_size, out_size), dtype=f32)
    off = f32(0.5) if aligned else f32(0.0)
    for r in range(k):
        b = int(rois[r, 0])
        sw_ = f32(rois[r, 1]) * f32(scale) - off
        sh_ = f32(rois[r, 2]) * f32(scale) - off
        ew_ = f32(rois[r, 3]) * f32(scale) - off
        eh_ = f32(rois[r, 4]) * f32(scale) - off
        rw = f32(ew_ - sw_)
        rh = f32(eh_ - sh_)
        if not aligned:
            rw = max(rw, f32(1.0))
            rh = max(rh, f32(1.0))
        bh = f32(rh / f32(out_size))
        bw = f32(rw / f32(out_size))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rh) / out_size))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rw) / out_size))
        count = f32(max(gh * gw, 1))
        for ph in range(out_size):
            for pw in range(out_size):
                acc = np.zeros((c,), dtype=f32)
                for iy in range(gh):
                    y = f32(sh_ + f32(ph) * bh + f32(f32(iy) + f32(0.5)) * bh / f32(gh))
                    for ix in range(gw):
                        x = f32(sw_ + f32(pw) * bw + f32(f32(ix) + f32(0.5)) * bw / f32(gw))
                        if y < -1.0 or y > height or x < -1.0 or x > width:
                            continue
                        yy = max(y, f32(0.0))
                        xx = max(x, f32(0.0))
                        yl, xl = int(yy), int(xx)
                        if yl >= height - 1:
                            yh = yl = height - 1
                            yy = f32(yl)
                        else:
                            yh = yl + 1
                        if xl >= width - 1:
                            xh = xl = width - 1
                            xx = f32(xl)
                        else:
                            xh = xl + 1
                        ly = f32(yy - f32(yl))
                        lx = f32(xx - f32(xl))
                        hy = f32(f32(1.0) - ly)
                        hx = f32(f32(1.0) - lx)
                        acc = acc + (f32(hy * hx) * feat[b, :, yl, xl] + f32(hy * lx) * feat[b, :, yl, xh]
                                     + f32(ly * hx) * feat[b, :, yh, xl] + f32(ly * lx) * feat[b, :, yh, xh])
                out[r, :, ph, pw] = acc / count
    return out



def roi_align_torch(feat: torch.Tensor, rois: torch.Tensor, scale: float, out_size: int = 7) -> torch.Tensor:
    """The same algorithm as roi_align_ref (aligned=True, sampling_ratio=0) in differentiable torch ops, so that autograd
    provides the reference for osr_roi_align_bwd. The sample positions do not depend on the features; only the gathers
    and the weighted sums are traced. Small cases only (python loops)."""
    n, c, height, width = feat.shape
    outs = []
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        f32 = np.float32
        sw_ = f32(rois[r, 1]) * f32(scale) - f32(0.5)
        sh_ = f32(rois[r, 2]) * f32(scale) - f32(0.5)
        ew_ = f32(rois[r, 3]) * f32(scale) - f32(0.5)
        eh_ = f32(rois[r, 4]) * f32(scale) - f32(0.5)
        rw, rh = f32(ew_ - sw_), f32(eh_ - sh_)
        bh, bw = f32(rh / f32(out_size)), f32(rw / f32(out_size))
        gh, gw = int(math.ceil(float(rh) / out_size)), int(math.ceil(float(rw) / out_size))
        count = float(max(gh * gw, 1))
        ys, xs, ws, bins = [], [], [], []
        for ph in range(out_size):
            for pw in range(out_size):
                for iy in range(gh):
                    y = f32(sh_ + f32(ph) * bh + f32(f32(iy) + f32(0.5)) * bh / f32(gh))
                    for ix in range(gw):
                        x = f32(sw_ + f32(pw) * bw + f32(f32(ix) + f32(0.5)) * bw / f32(gw))
                        if y < -1.0 or y > height or x < -1.0 or x > width:
                            continue
                        yy, xx = max(y, f32(0.0)), max(x, f32(0.0))
                        yl, xl = int(yy), int(xx)
                        if yl >= height - 1:
                            yh = yl = height - 1
                            yy = f32(yl)
                        else:
                            yh = yl + 1
                        if xl >= width - 1:
                            xh = xl = width - 1
                            xx = f32(xl)
                        else:
                            xh = xl + 1
                        ly, lx = f32(yy - f32(yl)), f32(xx - f32(xl))
                        hy, hx = f32(f32(1.0) - ly), f32(f32(1.0) - lx)
                        for (py_, px_, wgt) in ((yl, xl, hy * hx), (yl, xh, hy * lx), (yh, xl, ly * hx), (yh, xh, ly * lx)):
                            ys.append(py_); xs.append(px_); ws.append(float(wgt) / count); bins.append(ph * out_size + pw)
        o = feat.new_zeros((out_size * out_size, c))
        if ys:
            vals = feat[b][:, torch.tensor(ys), torch.tensor(xs)].t() * torch.tensor(ws, dtype=feat.dtype).unsqueeze(1)  # (taps, c)
            o = o.index_add(0, torch.tensor(bins), vals)
        outs.append(o.t().reshape(c, out_size, out_size))
    return torch.stack(outs) if outs else feat.new_zeros((0, c, out_size, out_size))


def roi_pooler_ref(feats: List[torch.Tensor], boxes_per_image: List[torch.Tensor],
                   scales=(0.25, 0.125, 0.0625, 0.03125), out_size=7, roi_align_fn=None) -> torch.Tensor:
    """[d2-mem] ROIPooler.forward (ROIAlignV2, sampling_ratio 0): per-level RoIAlign,
    scattered back in the original RoI order. Returns (M,C,7,7)."""
    fn = roi_align_fn or (lambda f, r, s: torch.from_numpy(roi_align_ref(f.numpy(), r.numpy(), s, out_size)))
    rois = torch.cat([torch.cat((torch.full((len(b), 1), i, dtype=b.dtype), b), dim=1)
                      for i, b in enumerate(boxes_per_image)], dim=0)
    lv = assign_levels(rois[:, 1:])
    out = torch.zeros(rois.shape[0], feats[0].shape[1], out_size, out_size, dtype=feats[0].dtype)
    for l, (f, s) in enumerate(zip(feats, scales)):
        inds = torch.nonzero(lv == l).squeeze(1)
        if len(inds):
            out[inds] = fn(f, rois[inds], s)
    return out


# --------------------------------------------------------------------------------------
# Box head, predictor, first-stage filtering
# --------------------------------------------------------------------------------------


def box_head(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix="roi_heads.box_head"):
    """[d2-mem] FastRCNNConvFCHead with NUM_FC=2, FC_DIM=1024 (Base-RCNN-FPN.yaml:24-27):
    flatten (C,7,7) -> fc1 -> relu -> fc2 -> relu (osrcnn_roi_heads.py:308)."""
    x = torch.flatten(x, start_dim=1)
    x = F.relu(F.linear(x, p[prefix + ".fc1.weight"], p[prefix + ".fc1.bias"]))
    return F.relu(F.linear(x, p[prefix + ".fc2.weight"], p[prefix + ".fc2.bias"]))


def box_predictor(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix="roi_heads.box_predictor"):
    """OpensetFastRCNNOutputLayers.forward (osrcnn_fast_rcnn.py:248-264)."""
    d = F.linear(x, p[prefix + ".bbox_pred.weight"], p[prefix + ".bbox_pred.bias"])
    iou = F.linear(x, p[prefix + ".iou_pred.weight"], p[prefix + ".iou_pred.bias"]).sigmoid()
    return d, iou


def objectness_score(iou: torch.Tensor, ctr: torch.Tensor, mean_type="geometric") -> torch.Tensor:
    """predict_ious (osrcnn_fast_rcnn.py:443-450): sqrt(iou*ctr) or (iou+ctr)/2."""
    if mean_type == "geometric":
        return torch.sqrt(iou * ctr)
    return (iou + ctr) / 2.0


def nms_ref(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """[d2-mem] torchvision.ops.nms CPU kernel: sort by score descending (stable: lower
    index first among equals), greedy; suppress j if inter/(a_i+a_j-inter) > thr, areas
    from raw coordinates, fp32 arithmetic. Returns kept indices, score-descending."""
    f32 = np.float32
    boxes = boxes.astype(f32, copy=False)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores.astype(f32), kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(f32)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = f32(thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(f32(0), xx2 - xx1)
        h = np.maximum(f32(0), yy2 - yy1)
        inter = (w * h).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / ((areas[i] + areas[rest]).astype(f32) - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms_ref(boxes: np.ndarray, scores: np.ndarray, idxs: np.ndarray, thr: float) -> np.ndarray:
    """[d2-mem] detectron2.layers.batched_nms -> torchvision batched_nms, *vanilla*
    (per-class) semantics: NMS inside each class, kept indices of all classes merged and
    ordered by score descending (stable). The coordinate-trick branch torchvision takes for
    small inputs perturbs fp32 IoUs by ulps; the build defines the segment-wise result
    as the contract."""
    keep_mask = np.zeros(boxes.shape[0], dtype=bool)
    for c in np.unique(idxs):
        ids = np.nonzero(idxs == c)[0]
        k = nms_ref(boxes[ids], scores[ids], thr)
        keep_mask[ids[k]] = True
    kept = np.nonzero(keep_mask)[0]
    order = np.argsort(-scores[kept].astype(np.float32), kind="stable")
    return kept[order]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, feats, score_thresh, nms_thresh, topk):
    """fast_rcnn_inference_single_image (osrcnn_fast_rcnn.py:89-145) for class-agnostic
    boxes (R,4) and objectness scores (R,1). Returns (boxes, scores, feats, kept_index)."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    src = torch.arange(boxes.shape[0])
    if not bool(valid.all()):
        boxes, scores, feats, src = boxes[valid], scores[valid], feats[valid], src[valid]
    boxes = box_clip(boxes, image_shape)
    mask = scores > score_thresh
    inds = mask.nonzero()
    boxes = boxes[inds[:, 0]]
    sc = scores[mask]
    feats = feats[inds[:, 0]]
    src = src[inds[:, 0]]
    keep = torch.from_numpy(batched_nms_ref(boxes.numpy(), sc.numpy(), inds[:, 1].numpy(), nms_thresh))
    if topk >= 0:
        keep = keep[:topk]
    return boxes[keep], sc[keep], feats[keep], src[keep]


# --------------------------------------------------------------------------------------
# PLN + softmax classifier (inference)
# --------------------------------------------------------------------------------------


def pln_distance(a: torch.Tensor, b: torch.Tensor, kind: str = "COS") -> torch.Tensor:
    """MODEL.PLN.DISTANCE_TYPE between rows of a and rows of b (prototype_learning_network.py:155-160, 213-218)."""
    if kind == "L1":
        return torch.cdist(a, b, p=1.0)
    if kind == "L2":
        return torch.cdist(a, b)
    if kind == "COS":
        return 1.0 - a @ b.t()
    raise ValueError(f"MODEL.PLN.DISTANCE_TYPE '{kind}'")


def pln_inference(feats: torch.Tensor, p: Dict[str, torch.Tensor], unk_thr: float, unknown_id: int = 80,
                  num_known: int = 20, reps: int = 1, class_id: Optional[torch.Tensor] = None,
                  prefix="roi_heads.dml", distance: str = "COS"):
    """PLN.inference (prototype_learning_network.py:199-226). Returns
    (pred_classes int64, rec_features, min_dist, emb)."""
    rep = F.normalize(p[prefix + ".representatives"])
    emb = F.linear(feats, p[prefix + ".encoder.weight"], p[prefix + ".encoder.bias"])
    rec = F.linear(emb, p[prefix + ".decoder.weight"], p[prefix + ".decoder.bias"])
    new = F.normalize(emb)
    dist = pln_distance(new, rep, distance)
    md, _ = torch.min(dist.reshape(-1, num_known, reps), dim=2)
    md, mi = torch.min(md, dim=1)
    unknown = md > unk_thr
    if class_id is not None:
        mi = class_id[mi]
    mi = mi.clone()
    mi[unknown] = unknown_id
    return mi, rec, md, emb


def softmax_known_inference(boxes, probs, image_shape, score_thresh, nms_thresh, topk):
    """fast_rcnn_inference_single_image_known (softmax_classifier.py:47-104)."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
    if not bool(valid.all()):
        boxes, probs = boxes[valid], probs[valid]
    probs = probs[:, :-1]
    boxes = box_clip(boxes, image_shape)
    mask = probs > score_thresh
    inds = mask.nonzero()
    b = boxes[inds[:, 0]]
    s = probs[mask]
    keep = torch.from_numpy(batched_nms_ref(b.numpy(), s.numpy(), inds[:, 1].numpy(), nms_thresh))
    if topk >= 0:
        keep = keep[:topk]
    return b[keep], s[keep], inds[keep, 1]


def softmax_unknown_inference(boxes, scores, image_shape, score_thresh, nms_thresh, topk, unknown_id):
    """fast_rcnn_inference_single_image_unknown (softmax_classifier.py:106-168)."""
    scores = scores.unsqueeze(1)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not bool(valid.all()):
        boxes, scores = boxes[valid], scores[valid]
    boxes = box_clip(boxes, image_shape)
    mask = scores > score_thresh
    inds = mask.nonzero()
    b = boxes[inds[:, 0]]
    s = scores[mask]
    keep = torch.from_numpy(batched_nms_ref(b.numpy(), s.numpy(), inds[:, 1].numpy(), nms_thresh))
    if topk >= 0:
        keep = keep[:topk]
    return b[keep], s[keep], torch.full((len(keep),), unknown_id, dtype=torch.int64)


def softmax_classifier_inference(boxes, scores, pred_classes, rec_feats, image_shape, p, cfg,
                                 class_id: Optional[torch.Tensor] = None, prefix="roi_heads.softmaxcls"):
    """SoftMaxClassifier.inference for one image (softmax_classifier.py:299-344).
    Output order [unknown..., known...] (:328-334)."""
    unk = cfg["unknown_id"]
    known = pred_classes != unk
    logits = F.linear(rec_feats[known], p[prefix + ".cls_score.weight"], p[prefix + ".cls_score.bias"])
    probs = F.softmax(logits, dim=-1)
    kb, ks, kc = softmax_known_inference(boxes[known], probs, image_shape, cfg["known_score_thresh"],
                                         cfg["known_nms_thresh"], cfg["known_topk"])
    if class_id is not None:
        kc = class_id[kc]
    if not bool(known.all()):
        ub, us, uc = softmax_unknown_inference(boxes[~known], scores[~known], image_shape,
                                               cfg["unknown_score_thresh"], cfg["unknown_nms_thresh"],
                                               cfg["unknown_topk"], unk)
        return torch.cat((ub, kb)), torch.cat((us, ks)), torch.cat((uc, kc))
    return kb, ks, kc


VOC_COCO_CFG = dict(
    pre_nms_topk_test=1000, pre_nms_topk_train=2000, obj_score_thresh=0.05, nms_thresh_test=1.0,
    detections_per_image=1000, known_score_thresh=0.05, known_nms_thresh=0.5, known_topk=50,
    unknown_score_thresh=0.0, unknown_nms_thresh=0.5, unknown_topk=50, num_classes=81, num_known=20,
    unknown_id=80, unk_thr=0.23, emd_dim=256, alpha=0.1, beta=0.9, mean_type="geometric",
)


def make_head_params(seed: int = 0, num_known: int = 20, fc_dim: int = 1024, emd: int = 256,
                     in_dim: int = 256 * 49, spread: bool = True) -> Dict[str, torch.Tensor]:
    """Random head parameters with the reference's names and initialisers
    (classification_free_rpn.py:105-108; osrcnn_fast_rcnn.py:209-212;
    prototype_learning_network.py:68-78; softmax_classifier.py:210-211). With
    ``spread`` the tiny init stds are widened so synthetic scores are not all ~0.5."""
    g = torch.Generator().manual_seed(seed + 1000)
    m = 30.0 if spread else 1.0
    p = {}

    def lin(name, o, i, std, bstd=0.0):
        p[name + ".weight"] = torch.randn(o, i, generator=g) * std
        p[name + ".bias"] = torch.randn(o, generator=g) * bstd if bstd else torch.zeros(o)

    p["proposal_generator.rpn_head.conv.weight"] = torch.randn(256, 256, 3, 3, generator=g) * 0.01 * (2 if spread else 1)
    p["proposal_generator.rpn_head.conv.bias"] = torch.zeros(256)
    p["proposal_generator.rpn_head.anchor_deltas.weight"] = torch.randn(4, 256, 1, 1, generator=g) * 0.01 * m * 3
    p["proposal_generator.rpn_head.anchor_deltas.bias"] = torch.full((4,), 0.5 if spread else 0.0)
    p["proposal_generator.rpn_head.centerness.weight"] = torch.randn(1, 256, 1, 1, generator=g) * 0.01 * m * 10
    p["proposal_generator.rpn_head.centerness.bias"] = torch.zeros(1)
    lin("roi_heads.box_head.fc1", fc_dim, in_dim, math.sqrt(2.0 / in_dim), 0.02)
    lin("roi_heads.box_head.fc2", fc_dim, fc_dim, math.sqrt(2.0 / fc_dim), 0.02)
    lin("roi_heads.box_predictor.bbox_pred", 4, fc_dim, 0.001 * m)
    lin("roi_heads.box_predictor.iou_pred", 1, fc_dim, 0.01 * (m / 3))
    lin("roi_heads.dml.encoder", emd, fc_dim, 0.01 * (m / 6))
    lin("roi_heads.dml.decoder", fc_dim, emd, 0.01 * (m / 6))
    p["roi_heads.dml.representatives"] = torch.randn(num_known, emd, generator=g)
    lin("roi_heads.softmaxcls.cls_score", num_known + 1, fc_dim, 0.01 * (m / 3))
    return p


def roi_heads_inference(feats: Dict[str, torch.Tensor], proposals, image_sizes, p, cfg=VOC_COCO_CFG,
                        roi_align_fn=None, quant=None):
    """OpensetROIHeads._forward_box, inference branch (osrcnn_roi_heads.py:304-329).
    ``proposals`` = per image (boxes, ctr_scores). Returns per image (boxes, scores, classes)
    plus a dict of intermediates for stage-wise parity tests."""
    q = quant if quant is not None else (lambda t: t)
    fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
    pooled = q(roi_pooler_ref(fl, [b for b, _ in proposals], roi_align_fn=roi_align_fn))
    x = box_head(pooled, p)
    deltas, iou = box_predictor(x, p)
    all_boxes = b2b_apply_deltas(deltas, torch.cat([b for b, _ in proposals]))
    ctr = torch.cat([s for _, s in proposals]).unsqueeze(1)
    score = objectness_score(iou, ctr, cfg["mean_type"])
    counts = [len(b) for b, _ in proposals]
    results, inter = [], dict(pooled=pooled, box_features=x, deltas=deltas, iou=iou, boxes=all_boxes, score=score)
    stage1 = []
    for bx, sc, ft, size in zip(all_boxes.split(counts), score.split(counts), x.split(counts), image_sizes):
        b1, s1, f1, k1 = fast_rcnn_inference_single_image(bx, sc, size, ft, cfg["obj_score_thresh"],
                                                          cfg["nms_thresh_test"], cfg["detections_per_image"])
        cls, rec, md, emb = pln_inference(f1, p, cfg["unk_thr"], cfg["unknown_id"], cfg["num_known"])
        stage1.append(dict(boxes=b1, scores=s1, kept=k1, pln_classes=cls, min_dist=md, emb=emb, rec=rec))
        results.append(softmax_classifier_inference(b1, s1, cls, rec, size, p, cfg))
    inter["stage1"] = stage1
    return results, inter


def rpn_inference(feats: Dict[str, torch.Tensor], image_sizes, p, pre_nms_topk=1000, quant=None):
    """ClsFreeRPN.forward, inference branch (classification_free_rpn.py:513-547)."""
    fl = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    anchors = anchor_grid([tuple(f.shape[-2:]) for f in fl])
    ds, cs = [], []
    for f in fl:
        d, c = cfrpn_head(f, p)
        ds.append(d)
        cs.append(c)
    ds, cs = flatten_head_outputs(ds, cs)
    n = ds[0].shape[0]
    props = [ltrb_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)).view(n, -1, 4)
             for d, a in zip(ds, anchors)]
    res = find_top_rpn_proposals(props, cs, image_sizes, pre_nms_topk)
    return res, dict(deltas=ds, ctr=cs, anchors=anchors, decoded=props)


def detector_inference(images: Sequence[torch.Tensor], bb_params, head_params, cfg=VOC_COCO_CFG,
                       roi_align_fn=None, quant=None):
    """GeneralizedRCNN.inference ([d2-mem]) without the final rescale (the
    synthetic inputs are already at network resolution)."""
    batch, sizes = preprocess_images(images)
    feats = resnet_fpn_forward(batch, bb_params, quant=quant)
    props, _ = rpn_inference(feats, sizes, head_params, cfg["pre_nms_topk_test"])
    res, _ = roi_heads_inference(feats, [(b, s) for b, s, _ in props], sizes, head_params, cfg, roi_align_fn)
    return res


# --------------------------------------------------------------------------------------
# Training-side pieces (targets + losses); their gradients are checked through torch autograd over these functions
# --------------------------------------------------------------------------------------


def matcher(quality: torch.Tensor, thresholds: Sequence[float], labels: Sequence[int], low_quality: bool):
    """[d2-mem] detectron2.modeling.matcher.Matcher.__call__ on a (G, P) quality matrix."""
    if quality.numel() == 0:
        m = quality.new_full((quality.size(1),), 0, dtype=torch.int64)
        return m, quality.new_full((quality.size(1),), labels[0], dtype=torch.int8)
    vals, matches = quality.max(dim=0)
    lab = matches.new_full(matches.size(), 1, dtype=torch.int8)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        lab[(vals >= lo) & (vals < hi)] = l
    if low_quality:
        best_per_gt, _ = quality.max(dim=1)
        _, pred_idx = torch.nonzero(quality == best_per_gt[:, None], as_tuple=True)
        lab[pred_idx] = 1
    return matches, lab


def centerness_target(anchors: torch.Tensor, matched_gt: torch.Tensor, obj_labels: torch.Tensor) -> torch.Tensor:
    """classification_free_rpn.py:393-402."""
    d = ltrb_get_deltas(anchors, matched_gt)[:, [0, 2, 1, 3]]
    inside = (d >= 0).all(dim=1)
    d = d.clone()
    d[~inside, :] = 0
    lr, tb = d[:, 0:2], d[:, 2:4]
    c = torch.sqrt((lr.min(-1)[0] / (lr.max(-1)[0] + 1e-12)) * (tb.min(-1)[0] / (tb.max(-1)[0] + 1e-12)))
    c[obj_labels == 0] = 0.0
    return c


def pln_loss_terms(new, rep, gt_classes, ious, alpha, beta, num_known=20, iou_thr=0.5, reps=1, distance="COS"):
    """The three hinge sums of PLN.loss on normalised embeddings `new` and normalised prototypes `rep` (num_known * reps rows,
    class-major): prototype_learning_network.py:149-185."""
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_known) & (ious > iou_thr)).squeeze(1)
    dist = pln_distance(new[fg], rep, distance)
    md = dist.reshape(-1, num_known, reps).min(dim=2)[0]
    ar = torch.arange(md.shape[0])
    intra = md[ar, gt_classes[fg]]
    d2 = md.clone()
    d2[ar, gt_classes[fg]] = 1000
    inter = d2.min(dim=1)[0] if d2.shape[0] else d2.new_zeros(0)
    cd = pln_distance(rep, rep, distance).clone()
    for i in range(num_known):
        cd[i * reps:(i + 1) * reps, i * reps:(i + 1) * reps] = 1000
    cdist = cd.min(dim=1)[0]
    return (torch.clamp(intra - alpha, min=0).sum() + torch.clamp(beta - inter, min=0).sum()
            + torch.clamp(beta + alpha - cdist, min=0).sum())


def pln_loss(feats, gt_classes, ious, p, alpha, beta, loss_weight, num_known=20, iou_thr=0.5, prefix="roi_heads.dml", reps=1, distance="COS"):
    """PLN.loss (prototype_learning_network.py:133-187); both yaml files: COS distance, one prototype per class."""
    emb = F.linear(feats, p[prefix + ".encoder.weight"], p[prefix + ".encoder.bias"])
    new = F.normalize(emb)
    rec = F.linear(emb, p[prefix + ".decoder.weight"], p[prefix + ".decoder.bias"])
    rep = F.normalize(p[prefix + ".representatives"])
    loss = pln_loss_terms(new, rep, gt_classes, ious, alpha, beta, num_known, iou_thr, reps, distance)
    return emb, rec, loss * loss_weight / max(gt_classes.numel(), 1.0)


# --------------------------------------------------------------------------------------
# Training targets / losses, continued (forward only). Random sampling: the reference draws
# torch.randperm inside [d2] subsample_labels (RNG-stream dependent). Oracle and HIP
# kernels instead take caller-supplied uniform keys and keep the k smallest keys of each class
# (ties: lower index) -- the same distribution, reproducible, and the selected lists are ordered
# by key, which plays the role of the randperm order.
# --------------------------------------------------------------------------------------


def _k_smallest(keys: torch.Tensor, mask: torch.Tensor, k: int) -> torch.Tensor:
    idx = torch.nonzero(mask).squeeze(1)
    if k <= 0 or idx.numel() == 0:
        return idx[:0]
    order = torch.sort(keys[idx], stable=True)[1]
    return idx[order[:k]]


def subsample_by_keys(labels: torch.Tensor, keys: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """[d2] subsample_labels(labels, num_samples, positive_fraction, bg_label) with keys instead of randperm.
    Returns (pos_idx, neg_idx), each ordered by key."""
    pos = (labels != -1) & (labels != bg_label)
    neg = labels == bg_label
    num_pos = min(int(num_samples * positive_fraction), int(pos.sum()))
    num_neg = min(num_samples - num_pos, int(neg.sum()))
    return _k_smallest(keys, pos, num_pos), _k_smallest(keys, neg, num_neg)


def rpn_label_and_sample(anchors: torch.Tensor, gt_boxes: torch.Tensor, keys_reg: torch.Tensor, keys_obj: torch.Tensor,
                         reg_thr=(0.3, 0.7), obj_thr=(0.1, 0.3), batch_size=256, pos_frac=0.5, obj_pos_frac=1.0):
    """ClsFreeRPN.label_and_sample_anchors for one image (classification_free_rpn.py:359-409)."""
    q = pairwise_iou(gt_boxes, anchors)
    midx, lab = matcher(q, list(reg_thr), [0, -1, 1], True)
    _, olab = matcher(q, list(obj_thr), [0, -1, 1], True)
    miou = q.max(dim=0)[0] if q.numel() else torch.zeros(anchors.shape[0])

    def sub(l, keys, frac):
        p, n = subsample_by_keys(l, keys, batch_size, frac, 0)
        out = torch.full_like(l, -1)
        out[p] = 1
        out[n] = 0
        return out

    lab_s, olab_s = sub(lab, keys_reg, pos_frac), sub(olab, keys_obj, obj_pos_frac)
    if len(gt_boxes) == 0:
        mboxes = torch.zeros_like(anchors)
        ctr = torch.zeros(anchors.shape[0])
    else:
        mboxes = gt_boxes[midx]
        ctr = centerness_target(anchors, gt_boxes[midx], olab_s)
    return dict(matched_idx=midx, matched_iou=miou, labels_pre=lab, obj_labels_pre=olab, labels=lab_s, obj_labels=olab_s,
                matched_boxes=mboxes, ctr_target=ctr)


def smooth_l1(x: torch.Tensor, beta: float) -> torch.Tensor:
    """fvcore.nn.smooth_l1_loss elementwise (third-party, not vendored in /root/reference; its published definition):
    beta < 1e-5 -> |x|, else 0.5 x^2 / beta below beta and |x| - 0.5 beta above."""
    ax = x.abs()
    if beta < 1e-5:
        return ax
    return torch.where(ax < beta, 0.5 * x * x / beta, ax - 0.5 * beta)


def box_pair_losses(pred: torch.Tensor, gt: torch.Tensor, kind: str) -> torch.Tensor:
    """Per-pair loss of decoded boxes against their targets as box_regression_w_iou.py:49-82 selects it:
    "iou": 1 - diag(pairwise_iou(pred, gt)).clamp(min=1e-6) (:49-61); "giou": fvcore.nn.giou_loss; "diou" / "ciou":
    detectron2.layers.diou_loss / ciou_loss (third-party, absent from /root/reference; published definitions, eps = 1e-7;
    alpha of CIoU is computed without gradient, as there)."""
    if kind == "iou":
        return 1 - elementwise_iou(pred, gt).clamp(min=1e-6)
    eps = 1e-7
    x1, y1, x2, y2 = pred.unbind(-1)
    x1g, y1g, x2g, y2g = gt.unbind(-1)
    xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    hit = (yk2 > yk1) & (xk2 > xk1)
    inter = torch.where(hit, (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    xc1, yc1, xc2, yc2 = torch.min(x1, x1g), torch.min(y1, y1g), torch.max(x2, x2g), torch.max(y2, y2g)
    if kind == "giou":
        area_c = (xc2 - xc1) * (yc2 - yc1)
        return 1 - (iou - (area_c - union) / (area_c + eps))
    diag = (xc2 - xc1) ** 2 + (yc2 - yc1) ** 2 + eps
    dist = ((x2 + x1) / 2 - (x2g + x1g) / 2) ** 2 + ((y2 + y1) / 2 - (y2g + y1g) / 2) ** 2
    if kind == "diou":
        return 1 - iou + dist / diag
    if kind == "ciou":
        v = (4 / math.pi ** 2) * (torch.atan((x2g - x1g) / (y2g - y1g)) - torch.atan((x2 - x1) / (y2 - y1))) ** 2
        with torch.no_grad():
            alpha = v / (1 - iou + v + eps)
        return 1 - iou + dist / diag + alpha * v
    raise ValueError(f"Invalid dense box regression loss type '{kind}'")


def rpn_losses(anchors: torch.Tensor, pred_deltas: torch.Tensor, pred_ctr: torch.Tensor, labels: torch.Tensor, obj_labels: torch.Tensor,
               matched_boxes: torch.Tensor, ctr_target: torch.Tensor, batch_size=256, w_loc=0.5, w_ctr=0.5, box_loss=("iou", 0.0), ctr_beta=0.0):
    """ClsFreeRPN.losses (classification_free_rpn.py:446-490) with BBOX_REG_LOSS_TYPE box_loss[0] (box_regression_w_iou.py:13-85;
    both yaml files: "iou", :49-61) and the smooth-L1 centerness loss (:475-481). All inputs stacked over images: pred_deltas
    (N,R,4), pred_ctr (N,R), labels (N,R) ..."""
    n = labels.shape[0]
    pos = labels == 1
    if box_loss[0] == "smooth_l1":
        tgt = torch.stack([ltrb_get_deltas(anchors, matched_boxes[i]) for i in range(n)])
        loss_loc = smooth_l1(pred_deltas[pos] - tgt[pos], box_loss[1]).sum()
    else:
        pb = torch.stack([ltrb_apply_deltas(pred_deltas[i], anchors) for i in range(n)])
        loss_loc = box_pair_losses(pb[pos], matched_boxes[pos], box_loss[0]).sum()
    om = obj_labels != -1
    loss_ctr = smooth_l1(pred_ctr[om] - ctr_target[om], ctr_beta).sum()
    norm = batch_size * n
    return dict(loss_rpn_loc=loss_loc / norm * w_loc, loss_rpn_ctr=loss_ctr / norm * w_ctr,
                num_pos=int(pos.sum()), num_neg=int((labels == 0).sum()),
                obj_num_pos=int((obj_labels == 1).sum()), obj_num_neg=int((obj_labels == 0).sum()))


GT_PROPOSAL_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))


def roi_label_and_sample(prop_boxes: torch.Tensor, prop_logits: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor,
                         keys: torch.Tensor, num_classes=81, batch_size=512, pos_frac=0.25, iou_thr=0.5):
    """OpensetROIHeads.label_and_sample_proposals for one image (osrcnn_roi_heads.py:177-216):
    append GT ([d2] add_ground_truth_to_proposals), IoU, Matcher([0.5],[0,1]), matched_iou, sample 512 (<=25% fg).
    keys has one entry per (proposal..., gt...) candidate. Output rows are [fg by key..., bg by key...]."""
    boxes = torch.cat((prop_boxes, gt_boxes))
    logits = torch.cat((prop_logits, torch.full((len(gt_boxes),), GT_PROPOSAL_LOGIT)))
    q = pairwise_iou(gt_boxes, boxes)
    midx, mlab = matcher(q, [iou_thr], [0, 1], False)
    if len(gt_boxes):
        miou = q[midx, torch.arange(q.shape[1])]
        cls = gt_classes[midx].clone()
        cls[mlab == 0] = num_classes
    else:
        miou = torch.zeros(len(boxes))
        cls = torch.full((len(boxes),), num_classes, dtype=torch.int64)
    fg, bg = subsample_by_keys(cls, keys, batch_size, pos_frac, num_classes)
    sidx = torch.cat((fg, bg))
    gtb = gt_boxes[midx[sidx]] if len(gt_boxes) else torch.zeros(len(sidx), 4)
    return dict(sampled_idx=sidx, boxes=boxes[sidx], logits=logits[sidx], gt_classes=cls[sidx], ious=miou[sidx], gt_boxes=gtb,
                num_fg=len(fg), num_bg=len(bg))


def roi_box_losses(pred_deltas, pred_iou, proposal_boxes, gt_boxes, gt_classes, gt_iou, num_classes=81, w_box=0.5, w_iou=0.5,
                   box_loss=("smooth_l1", 0.0), iou_beta=0.0):
    """OpensetFastRCNNOutputLayers.losses (osrcnn_fast_rcnn.py:266-370): the box regression loss box_loss[0] names
    (box_regression_w_iou.py:13-85: smooth L1 on Box2BoxTransform deltas, or a box loss on the decoded boxes) and smooth L1 on the
    predicted IoU over foreground rows, both divided by the total number of rows."""
    fg = (gt_classes >= 0) & (gt_classes < num_classes)
    if box_loss[0] == "smooth_l1":
        lb = smooth_l1(pred_deltas[fg] - b2b_get_deltas(proposal_boxes[fg], gt_boxes[fg]), box_loss[1]).sum()
    else:
        lb = box_pair_losses(b2b_apply_deltas(pred_deltas[fg], proposal_boxes[fg]), gt_boxes[fg], box_loss[0]).sum()
    li = smooth_l1(pred_iou[fg] - gt_iou[fg], iou_beta).sum()
    r = max(gt_classes.numel(), 1.0)
    return lb / r * w_box, li / r * w_iou


def softmax_ce_loss(logits, gt_classes, num_classes=81, num_known=20, weight=0.9):
    """SoftMaxClassifier.loss (softmax_classifier.py:266-285): id_map (known i -> i, background -> K, anything else -> -1),
    mean cross entropy. Targets of -1 cannot occur with the VOC training data; they are ignored here."""
    id_map = torch.full((num_classes + 1,), -1, dtype=torch.int64)
    id_map[:num_known] = torch.arange(num_known)
    id_map[num_classes] = num_known
    t = id_map[gt_classes]
    keep = t >= 0
    return weight * F.cross_entropy(logits[keep], t[keep], reduction="mean") if bool(keep.any()) else logits.sum() * 0


# --------------------------------------------------------------------------------------
# BASELINE config 1: the stock detectron2 modules /root/reference/configs/Base-RCNN-FPN.yaml names on its own
# (PROPOSAL_GENERATOR "RPN" + "StandardRPNHead", ROI_HEADS "StandardROIHeads" + FastRCNNOutputLayers). None of their code is in
# /root/reference: everything below restates detectron2 v0.6's published algorithms [d2-mem] at the yaml's call sites
# (Base-RCNN-FPN.yaml:9-33), with the same tie rule as the rest of this file.
# --------------------------------------------------------------------------------------

BASE_RCNN_CFG = dict(
    anchor_sizes=(32, 64, 128, 256, 512), anchor_ratios=(0.5, 1.0, 2.0), pre_nms_topk_test=1000, post_nms_topk_test=1000,
    rpn_nms_thresh=0.7, rpn_bbox_reg_weights=(1.0, 1.0, 1.0, 1.0), min_box_size=0.0, num_classes=80, bbox_reg_weights=(10.0, 10.0, 5.0, 5.0),
    score_thresh_test=0.05, nms_thresh_test=0.5, detections_per_image=100,
)


def b2b_apply_deltas_multi(deltas: torch.Tensor, boxes: torch.Tensor, weights=(10.0, 10.0, 5.0, 5.0)) -> torch.Tensor:
    """[d2-mem] Box2BoxTransform.apply_deltas for (R, k*4) deltas (class-specific regression): the same arithmetic as
    b2b_apply_deltas for each of the k groups of 4, output (R, k*4)."""
    r, k4 = deltas.shape
    out = b2b_apply_deltas(deltas.reshape(-1, 4), boxes.unsqueeze(1).expand(r, k4 // 4, 4).reshape(-1, 4), weights)
    return out.view(r, k4)


def standard_rpn_head(feat: torch.Tensor, p: Dict[str, torch.Tensor], prefix="proposal_generator.rpn_head"):
    """[d2-mem] StandardRPNHead.forward for one level: t = relu(conv3x3(x)); objectness_logits = conv1x1(t) (A channels),
    anchor_deltas = conv1x1(t) (A*4 channels). Returns (deltas (N,4A,H,W), logits (N,A,H,W))."""
    t = F.relu(F.conv2d(feat, p[prefix + ".conv.weight"], p[prefix + ".conv.bias"], padding=1))
    return (F.conv2d(t, p[prefix + ".anchor_deltas.weight"], p[prefix + ".anchor_deltas.bias"]),
            F.conv2d(t, p[prefix + ".objectness_logits.weight"], p[prefix + ".objectness_logits.bias"]))


def standard_find_top_rpn_proposals(proposals: List[torch.Tensor], logits: List[torch.Tensor], image_sizes, nms_thresh: float,
                                    pre_nms_topk: int, post_nms_topk: int, min_box_size: float = 0.0, training: bool = False):
    """[d2-mem] detectron2.modeling.proposal_generator.proposal_utils.find_top_rpn_proposals: per level top-k of the objectness
    logits, concatenation with level ids, per image: finite filter, clip, drop empty boxes, batched NMS with the LEVEL as the
    category (thr 0.7), first post_nms_topk of the score-descending keep list. Returns per image (boxes, logits, level ids)."""
    n_img = len(image_sizes)
    tk_s, tk_b, lvl = [], [], []
    for level_id, (prop_l, sc_l) in enumerate(zip(proposals, logits)):
        k = min(sc_l.shape[1], pre_nms_topk)
        v, idx = stable_topk(sc_l, k)
        tk_s.append(v)
        tk_b.append(prop_l[torch.arange(n_img)[:, None], idx])
        lvl.append(torch.full((k,), level_id, dtype=torch.int64))
    tk_s, tk_b, lvl = torch.cat(tk_s, 1), torch.cat(tk_b, 1), torch.cat(lvl)
    results = []
    for n, size in enumerate(image_sizes):
        b, s, lv = tk_b[n], tk_s[n], lvl
        valid = torch.isfinite(b).all(dim=1) & torch.isfinite(s)
        if not bool(valid.all()):
            if training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            b, s, lv = b[valid], s[valid], lv[valid]
        b = box_clip(b, size)
        keep = box_nonempty(b, min_box_size)
        b, s, lv = b[keep], s[keep], lv[keep]
        kept = torch.from_numpy(batched_nms_ref(b.numpy(), s.numpy(), lv.numpy(), nms_thresh))[:post_nms_topk]
        results.append((b[kept], s[kept], lv[kept]))
    return results


def standard_rpn_inference(feats: Dict[str, torch.Tensor], image_sizes, p, cfg=BASE_RCNN_CFG):
    """[d2-mem] RPN.forward, inference branch: anchors (3 aspect ratios per cell), StandardRPNHead, (N,A*4,H,W)->(N,H*W*A,4) /
    (N,A,H,W)->(N,H*W*A) flattening, Box2BoxTransform(weights 1,1,1,1) decode, find_top_rpn_proposals."""
    fl = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    anchors = anchor_grid([tuple(f.shape[-2:]) for f in fl], sizes=cfg["anchor_sizes"], ratios=cfg["anchor_ratios"])
    ds, ls = [], []
    for f in fl:
        d, l = standard_rpn_head(f, p)
        ds.append(d)
        ls.append(l)
    ds, ls = flatten_head_outputs(ds, ls)
    n = ds[0].shape[0]
    props = [b2b_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4), cfg["rpn_bbox_reg_weights"]).view(n, -1, 4)
             for d, a in zip(ds, anchors)]
    res = standard_find_top_rpn_proposals(props, ls, image_sizes, cfg["rpn_nms_thresh"], cfg["pre_nms_topk_test"], cfg["post_nms_topk_test"],
                                          cfg["min_box_size"])
    return res, dict(deltas=ds, logits=ls, anchors=anchors, decoded=props)


def fast_rcnn_output_inference(x: torch.Tensor, proposals: torch.Tensor, image_size, p, cfg=BASE_RCNN_CFG, prefix="roi_heads.box_predictor"):
    """[d2-mem] FastRCNNOutputLayers.inference for one image: scores = cls_score(x) (K+1 logits), deltas = bbox_pred(x) (K*4,
    class-specific), predict_probs = softmax, predict_boxes = Box2BoxTransform(10,10,5,5).apply_deltas, then
    fast_rcnn_inference_single_image: rows with a non-finite box or score dropped, background column dropped, boxes clipped,
    (row, class) pairs with score > SCORE_THRESH_TEST kept, per-class NMS (0.5), first DETECTIONS_PER_IMAGE of the keep list.
    Returns (boxes, scores, classes, (row, class) of every kept detection)."""
    logits = F.linear(x, p[prefix + ".cls_score.weight"], p[prefix + ".cls_score.bias"])
    deltas = F.linear(x, p[prefix + ".bbox_pred.weight"], p[prefix + ".bbox_pred.bias"])
    probs = F.softmax(logits, dim=-1)
    boxes = b2b_apply_deltas_multi(deltas, proposals, cfg["bbox_reg_weights"])
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
    rows = torch.arange(boxes.shape[0])
    if not bool(valid.all()):
        boxes, probs, rows = boxes[valid], probs[valid], rows[valid]
    probs = probs[:, :-1]
    k = boxes.shape[1] // 4
    boxes = box_clip(boxes.reshape(-1, 4), image_size).view(-1, k, 4)
    mask = probs > cfg["score_thresh_test"]
    inds = mask.nonzero()
    b = boxes[inds[:, 0], inds[:, 1]] if k > 1 else boxes[inds[:, 0], 0]
    s = probs[mask]
    keep = torch.from_numpy(batched_nms_ref(b.numpy(), s.numpy(), inds[:, 1].numpy(), cfg["nms_thresh_test"]))[: cfg["detections_per_image"]]
    return b[keep], s[keep], inds[keep, 1], torch.stack((rows[inds[keep, 0]], inds[keep, 1]), dim=1)


def standard_roi_heads_inference(feats, proposals, image_sizes, p, cfg=BASE_RCNN_CFG, roi_align_fn=None):
    """[d2-mem] StandardROIHeads._forward_box, inference branch: ROIPooler (7x7, ROIAlignV2) -> FastRCNNConvFCHead ->
    FastRCNNOutputLayers.inference. ``proposals``: per image (boxes, logits, ...)."""
    fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
    x = box_head(roi_pooler_ref(fl, [pr[0] for pr in proposals], roi_align_fn=roi_align_fn), p)
    counts = [len(pr[0]) for pr in proposals]
    return [fast_rcnn_output_inference(xi, pr[0], size, p, cfg) for xi, pr, size in zip(x.split(counts), proposals, image_sizes)], x


def standard_detector_inference(images: Sequence[torch.Tensor], params, cfg=BASE_RCNN_CFG, roi_align_fn=None, image_sizes=None):
    """[d2-mem] GeneralizedRCNN.inference for Base-RCNN-FPN.yaml (no final rescale: inputs are at network resolution)."""
    batch, sizes = preprocess_images(images)
    sizes = image_sizes or sizes
    feats = resnet_fpn_forward(batch, params)
    props, _ = standard_rpn_inference(feats, sizes, params, cfg)
    res, _ = standard_roi_heads_inference(feats, props, sizes, params, cfg, roi_align_fn)
    return res, props
