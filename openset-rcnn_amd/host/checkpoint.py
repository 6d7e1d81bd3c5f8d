"""Checkpoint import for the HIP path: the two on-disk formats the reference loads --

  * `.pth` written by [d2] DetectionCheckpointer (train.py:113-118, :195-197): torch.save({"model": state_dict, ...}) with
    detectron2 parameter names (the names host/modeling.py's modules own), FrozenBN statistics included;
  * `detectron2://ImageNetPretrained/MSRA/R-50.pkl` (VOC-COCO yaml:3): a pickled {blob name: ndarray} dict in Caffe2 / MSRA
    naming (conv1_w, res_conv1_bn_s, res2_0_branch2a_w, res2_0_branch2a_bn_b, ..., fc1000_w), BN already reduced to a
    scale/shift pair. [d2] loads it through name heuristics; the mapping is closed-form for R-50 and written out below.

Nothing here reads from the network: paths are local files. The result is a flat {d2 name: fp32 tensor} dict, ready for
`model.load_state_dict` (host/modeling.py) or, after `weights.fold_frozen_bn`, for `OpensetRCNNEngine`."""
from __future__ import annotations

import pickle
import re
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

_BRANCH = {"branch1": "shortcut", "branch2a": "conv1", "branch2b": "conv2", "branch2c": "conv3"}
_BN = {"s": "norm.weight", "b": "norm.bias", "rm": "norm.running_mean", "riv": "norm.running_var"}


def convert_msra_name(name: str) -> str:
    """One Caffe2/MSRA blob name -> the detectron2 backbone parameter name ("" for blobs the detector does not use)."""
    if name.startswith("fc1000") or name.endswith("_momentum"):
        return ""
    m = re.fullmatch(r"conv1_(w|b)", name)
    if m:
        return "backbone.bottom_up.stem.conv1." + ("weight" if m.group(1) == "w" else "bias")
    m = re.fullmatch(r"res_conv1_bn_(s|b|rm|riv)", name)
    if m:
        return "backbone.bottom_up.stem.conv1." + _BN[m.group(1)]
    m = re.fullmatch(r"res(\d)_(\d+)_(branch1|branch2a|branch2b|branch2c)_(w|b|bn_s|bn_b|bn_rm|bn_riv)", name)
    if m:
        stage, blk, br, kind = m.groups()
        leaf = {"w": "weight", "b": "bias"}.get(kind) or _BN[kind[3:]]
        return f"backbone.bottom_up.res{stage}.{int(blk)}.{_BRANCH[br]}.{leaf}"
    raise KeyError(f"unrecognised MSRA/Caffe2 blob name '{name}'")


def convert_msra_state(blobs: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    """Whole MSRA R-50 dict -> d2-named tensors. BN blobs hold scale/shift only; FrozenBN's running_mean = 0 and
    running_var = 1 - eps make `scale * rsqrt(var + eps)` reproduce the stored scale exactly ([d2] FrozenBatchNorm2d init)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in blobs.items():
        name = convert_msra_name(k)
        if name:
            out[name] = torch.from_numpy(np.ascontiguousarray(v)).float()
    for k in [k for k in out if k.endswith(".norm.weight")]:
        pre = k[: -len("weight")]
        c = out[k].numel()
        out.setdefault(pre + "running_mean", torch.zeros(c))
        out.setdefault(pre + "running_var", torch.ones(c) - 1e-5)
    return out


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Read a `.pth` or `.pkl` checkpoint into a flat {d2 name: tensor} dict (fp32, CPU)."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if isinstance(data, dict) and "model" in data and "__author__" in data:  # already converted by detectron2's tools
            model = data["model"]
            if not data.get("matching_heuristics", False):
                return {k: torch.as_tensor(np.asarray(v)).float() for k, v in model.items()}
            data = model
        if isinstance(data, dict) and "blobs" in data:
            data = data["blobs"]
        return convert_msra_state({k: np.asarray(v) for k, v in data.items()})
    data = torch.load(path, map_location="cpu", weights_only=False)
    state = data["model"] if isinstance(data, dict) and "model" in data else data
    out = {}
    for k, v in state.items():
        k = k[len("module."):] if k.startswith("module.") else k  # DDP-wrapped saves (train.py:201-205)
        out[k] = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
        if out[k].is_floating_point():
            out[k] = out[k].float()
    return out


def load_into(model: torch.nn.Module, state: Dict[str, torch.Tensor], strict: bool = False) -> Tuple[List[str], List[str]]:
    """[d2] Checkpointer semantics: load matching names, report (missing, unexpected); shape mismatches raise. A backbone-only
    checkpoint (MSRA R-50) leaves FPN / RPN / RoI-head parameters at their initial values, as in the reference's training
    start."""
    own = model.state_dict()
    for k, v in state.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f"shape mismatch for '{k}': checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
    missing = [k for k in own if k not in state]
    unexpected = [k for k in state if k not in own]
    if strict and (missing or unexpected):
        raise KeyError(f"missing {missing[:5]}... unexpected {unexpected[:5]}...")
    merged = dict(own)
    merged.update({k: v for k, v in state.items() if k in own})
    model.load_state_dict(merged)
    return missing, unexpected


def msra_names_for(d2_names: Iterable[str]) -> Dict[str, str]:
    """Inverse mapping (d2 backbone name -> MSRA blob name); used by the tests to fabricate an R-50.pkl look-alike."""
    inv_branch = {v: k for k, v in _BRANCH.items()}
    inv_bn = {v: k for k, v in _BN.items()}
    out = {}
    for n in d2_names:
        m = re.fullmatch(r"backbone\.bottom_up\.stem\.conv1\.(.+)", n)
        if m:
            leaf = m.group(1)
            out[n] = "conv1_w" if leaf == "weight" else ("conv1_b" if leaf == "bias" else "res_conv1_bn_" + inv_bn[leaf])
            continue
        m = re.fullmatch(r"backbone\.bottom_up\.res(\d)\.(\d+)\.(shortcut|conv1|conv2|conv3)\.(.+)", n)
        if m:
            stage, blk, br, leaf = m.groups()
            kind = {"weight": "w", "bias": "b"}.get(leaf) or "bn_" + inv_bn[leaf]
            out[n] = f"res{stage}_{blk}_{inv_branch[br]}_{kind}"
    return out
