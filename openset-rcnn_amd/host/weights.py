"""Parameter handling for the HIP path: detectron2-compatible state-dict names (checkpoint row),
FrozenBatchNorm folding, and the one-time repacking of weights into the layouts the kernels read.

No checkpoint is reachable offline, so ``random_params`` builds a seeded synthetic parameter set with the
reference's parameter names and shapes (He-style init scaled so that activations stay inside fp16 range)."""
from __future__ import annotations

import math
from typing import Dict

import torch

R50_BLOCKS = (3, 4, 6, 3)
R50_MID = (64, 128, 256, 512)


def fold_frozen_bn(state: Dict[str, torch.Tensor], eps: float = 1e-5) -> Dict[str, torch.Tensor]:
    """[d2] FrozenBatchNorm2d (y = x*w*rsqrt(var+eps) + (b - mean*w*rsqrt(var+eps))) folded into the preceding
    bias-free conv: '<conv>.norm.{weight,bias,running_mean,running_var}' keys are consumed and '<conv>.bias' produced."""
    out = dict(state)
    for k in list(state.keys()):
        if k.endswith(".norm.weight"):
            pre = k[: -len(".norm.weight")]
            w, b = state[pre + ".norm.weight"], state[pre + ".norm.bias"]
            mean, var = state[pre + ".norm.running_mean"], state[pre + ".norm.running_var"]
            scale = w * (var + eps).rsqrt()
            out[pre + ".weight"] = state[pre + ".weight"] * scale.view(-1, 1, 1, 1)
            out[pre + ".bias"] = b - mean * scale
            for s in ("weight", "bias", "running_mean", "running_var"):
                out.pop(pre + ".norm." + s, None)
    return out


def random_params(seed: int = 0, num_known: int = 20, fc_dim: int = 1024, emd: int = 256, res_gain: float = 0.5,
                  spread: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded synthetic parameters (already BN-folded) for R50-FPN + CF-RPN head + Openset RoI heads."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, gain=1.0, bias_std=0.05):
        std = gain * math.sqrt(2.0 / (cout * k * k))
        p[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std
        p[name + ".bias"] = torch.randn(cout, generator=g) * bias_std

    conv("backbone.bottom_up.stem.conv1", 64, 3, 7, gain=0.02)
    cin = 64
    for si, (nb, mid) in enumerate(zip(R50_BLOCKS, R50_MID)):
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

    g2 = torch.Generator().manual_seed(seed + 1000)
    m = 30.0 if spread else 1.0

    def lin(name, o, i, std, bstd=0.0):
        p[name + ".weight"] = torch.randn(o, i, generator=g2) * std
        p[name + ".bias"] = torch.randn(o, generator=g2) * bstd if bstd else torch.zeros(o)

    in_dim = 256 * 49
    p["proposal_generator.rpn_head.conv.weight"] = torch.randn(256, 256, 3, 3, generator=g2) * 0.01 * (2 if spread else 1)
    p["proposal_generator.rpn_head.conv.bias"] = torch.zeros(256)
    p["proposal_generator.rpn_head.anchor_deltas.weight"] = torch.randn(4, 256, 1, 1, generator=g2) * 0.01 * m * 3
    p["proposal_generator.rpn_head.anchor_deltas.bias"] = torch.full((4,), 0.5 if spread else 0.0)
    p["proposal_generator.rpn_head.centerness.weight"] = torch.randn(1, 256, 1, 1, generator=g2) * 0.01 * m * 10
    p["proposal_generator.rpn_head.centerness.bias"] = torch.zeros(1)
    lin("roi_heads.box_head.fc1", fc_dim, in_dim, math.sqrt(2.0 / in_dim), 0.02)
    lin("roi_heads.box_head.fc2", fc_dim, fc_dim, math.sqrt(2.0 / fc_dim), 0.02)
    lin("roi_heads.box_predictor.bbox_pred", 4, fc_dim, 0.001 * m)
    lin("roi_heads.box_predictor.iou_pred", 1, fc_dim, 0.01 * (m / 3))
    lin("roi_heads.dml.encoder", emd, fc_dim, 0.01 * (m / 6))
    lin("roi_heads.dml.decoder", fc_dim, emd, 0.01 * (m / 6))
    p["roi_heads.dml.representatives"] = torch.randn(num_known, emd, generator=g2)
    lin("roi_heads.softmaxcls.cls_score", num_known + 1, fc_dim, 0.01 * (m / 3))
    return p


def pack_conv_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(cout,cin,kh,kw) -> (cout,kh,kw,cin): the GEMM K axis becomes contiguous per tap."""
    return w.permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_stem_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(64,3,7,7) -> stem view (64,8,1,32): K slice kh holds 8 taps x 4 channels of one image row; the 8th tap,
    the 4th channel and the whole 8th row are zero (the row pads K to a multiple of 64 for the BK=64 kernel)."""
    cout = w.shape[0]
    v = torch.zeros(cout, 8, 8, 4, dtype=torch.float32)
    v[:, :7, :7, :3] = w.float().permute(0, 2, 3, 1)
    return v.view(cout, 8, 1, 32).contiguous().to(dtype)


def pack_fc1_weight(w: torch.Tensor, channels: int, pooled: int, dtype: torch.dtype) -> torch.Tensor:
    """fc1 (out, c*p*p) with the reference's (c,ph,pw) flatten order -> (out, p*p*c) matching RoIAlign's (ph,pw,c) output."""
    o = w.shape[0]
    return w.view(o, channels, pooled, pooled).permute(0, 2, 3, 1).reshape(o, pooled * pooled * channels).contiguous().to(dtype)


def pack_dgrad_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(cout,cin,kh,kw) -> (cin,kh,kw,cout), spatially flipped: the backward-data pass of a stride-1 convolution is the
    forward convolution of dy with these weights and padding k-1-pad; for a 1x1 layer it is the transposed matrix."""
    return w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(dtype)


def with_known_unknown_mix(params: Dict[str, torch.Tensor], embeddings: torch.Tensor, unk_thr: float = 0.23, known_fraction: float = 0.5,
                           proto: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic-weights helper (benchmarks / tests; no checkpoint is reachable offline). Random prototypes put every embedding
    ~0.85-1.0 away from all of them, so PLN.inference (prototype_learning_network.py:213-223) calls every detection "unknown" and
    the known-class leg (softmax over <= 20 000 candidates, per-class NMS) idles. This returns a copy of `params` whose PLN
    encoder bias is c * P_hat[proto]: with emb = W f + b, cos(emb, P_hat) = c / sqrt(c^2 + |W f|^2) (W f is nearly orthogonal to
    one fixed direction of 256), so the detections whose |W f| is below c * sqrt(1 / (1 - unk_thr)^2 - 1) become known. `c` is
    set from the `known_fraction` quantile of |embeddings| (rows = embeddings of a run with zero encoder bias, valid rows only).
    UNK_THR itself stays the yaml's value."""
    out = dict(params)
    p = params["roi_heads.dml.representatives"].float()
    ph = p[proto] / p[proto].norm().clamp(min=1e-12)
    norms = embeddings.detach().float().cpu().norm(dim=1)
    q = float(torch.quantile(norms, known_fraction))
    cos_thr = 1.0 - unk_thr
    c = q / math.sqrt(1.0 / (cos_thr * cos_thr) - 1.0)
    out["roi_heads.dml.encoder.bias"] = (c * ph).contiguous()
    return out


def random_standard_params(seed: int = 0, num_classes: int = 80, num_anchors: int = 3, cls_agnostic: bool = True, fc_dim: int = 1024) -> Dict[str, torch.Tensor]:
    """Seeded synthetic parameters (BN-folded backbone of random_params + the stock detectron2 heads of Base-RCNN-FPN.yaml):
    StandardRPNHead (conv, objectness_logits (A), anchor_deltas (4A)) and FastRCNNOutputLayers (cls_score (K+1), bbox_pred (4 or
    4K)), scaled so that logits and deltas spread (the [d2] initialisers, std 0.01 / 0.001, give near-constant outputs and
    degenerate top-k / NMS ties)."""
    p = {k: v for k, v in random_params(seed).items() if k.startswith("backbone.")}
    g = torch.Generator().manual_seed(seed + 2000)
    p["proposal_generator.rpn_head.conv.weight"] = torch.randn(256, 256, 3, 3, generator=g) * 0.02
    p["proposal_generator.rpn_head.conv.bias"] = torch.zeros(256)
    p["proposal_generator.rpn_head.objectness_logits.weight"] = torch.randn(num_anchors, 256, 1, 1, generator=g) * 0.5
    p["proposal_generator.rpn_head.objectness_logits.bias"] = torch.zeros(num_anchors)
    p["proposal_generator.rpn_head.anchor_deltas.weight"] = torch.randn(num_anchors * 4, 256, 1, 1, generator=g) * 0.1
    p["proposal_generator.rpn_head.anchor_deltas.bias"] = torch.zeros(num_anchors * 4)
    in_dim = 256 * 49
    p["roi_heads.box_head.fc1.weight"] = torch.randn(fc_dim, in_dim, generator=g) * math.sqrt(2.0 / in_dim)
    p["roi_heads.box_head.fc1.bias"] = torch.randn(fc_dim, generator=g) * 0.02
    p["roi_heads.box_head.fc2.weight"] = torch.randn(fc_dim, fc_dim, generator=g) * math.sqrt(2.0 / fc_dim)
    p["roi_heads.box_head.fc2.bias"] = torch.randn(fc_dim, generator=g) * 0.02
    p["roi_heads.box_predictor.cls_score.weight"] = torch.randn(num_classes + 1, fc_dim, generator=g) * 0.15
    p["roi_heads.box_predictor.cls_score.bias"] = torch.zeros(num_classes + 1)
    p["roi_heads.box_predictor.bbox_pred.weight"] = torch.randn(4 if cls_agnostic else 4 * num_classes, fc_dim, generator=g) * 0.03
    p["roi_heads.box_predictor.bbox_pred.bias"] = torch.zeros(4 if cls_agnostic else 4 * num_classes)
    return p
