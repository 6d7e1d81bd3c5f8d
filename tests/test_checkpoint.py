"""Checkpoint import (host/checkpoint.py): MSRA/Caffe2 R-50.pkl naming and DetectionCheckpointer-style .pth files."""
import pickle

import numpy as np
import pytest
import torch


def test_msra_name_mapping(osr):
    from openset_rcnn_amd.host.checkpoint import convert_msra_name
    assert convert_msra_name("conv1_w") == "backbone.bottom_up.stem.conv1.weight"
    assert convert_msra_name("res_conv1_bn_s") == "backbone.bottom_up.stem.conv1.norm.weight"
    assert convert_msra_name("res_conv1_bn_b") == "backbone.bottom_up.stem.conv1.norm.bias"
    assert convert_msra_name("res2_0_branch1_w") == "backbone.bottom_up.res2.0.shortcut.weight"
    assert convert_msra_name("res2_0_branch1_bn_s") == "backbone.bottom_up.res2.0.shortcut.norm.weight"
    assert convert_msra_name("res4_5_branch2b_w") == "backbone.bottom_up.res4.5.conv2.weight"
    assert convert_msra_name("res5_2_branch2c_bn_b") == "backbone.bottom_up.res5.2.conv3.norm.bias"
    assert convert_msra_name("fc1000_w") == "" and convert_msra_name("res2_0_branch2a_w_momentum") == ""
    with pytest.raises(KeyError):
        convert_msra_name("totally_unknown_blob")


def _cfg(osr, tmp_path):
    from openset_rcnn_amd.host import config as Cfg
    cfg = Cfg.get_cfg()
    Cfg.add_openset_rcnn_config(cfg)
    y = tmp_path / "m.yaml"
    y.write_text("MODEL:\n  META_ARCHITECTURE: GeneralizedRCNN\n  DEVICE: cpu\n  BACKBONE:\n    NAME: build_resnet_fpn_backbone\n"
                 "  RESNETS:\n    OUT_FEATURES: [res2, res3, res4, res5]\n  FPN:\n    IN_FEATURES: [res2, res3, res4, res5]\n"
                 "  ANCHOR_GENERATOR:\n    SIZES: [[32], [64], [128], [256], [512]]\n    ASPECT_RATIOS: [[1.0]]\n"
                 "  PROPOSAL_GENERATOR:\n    NAME: ClsFreeRPN\n  RPN:\n    HEAD_NAME: ClsFreeRPNHead\n    IN_FEATURES: [p2, p3, p4, p5, p6]\n"
                 "    PRE_NMS_TOPK_TRAIN: 2000\n    PRE_NMS_TOPK_TEST: 1000\n"
                 "  ROI_HEADS:\n    NAME: OpensetROIHeads\n    IN_FEATURES: [p2, p3, p4, p5]\n    NUM_CLASSES: 81\n"
                 "  ROI_BOX_HEAD:\n    NAME: FastRCNNConvFCHead\n    NUM_FC: 2\n    POOLER_RESOLUTION: 7\n    CLS_AGNOSTIC_BBOX_REG: True\n")
    cfg.merge_from_file(str(y))
    return cfg


def test_msra_pkl_and_pth_roundtrip(osr, tmp_path):
    from openset_rcnn_amd.host import checkpoint as CK
    from openset_rcnn_amd.host import modeling as M
    model = M.build_model(_cfg(osr, tmp_path))
    sd = model.state_dict()
    g = torch.Generator().manual_seed(0)
    # fabricate an MSRA-style R-50.pkl: backbone weights + BN scale/shift blobs, plus blobs the detector must ignore
    names = CK.msra_names_for([k for k in sd if k.startswith("backbone.bottom_up.")])
    blobs = {}
    for d2n, msra in names.items():
        if d2n.endswith("running_mean") or d2n.endswith("running_var"):
            continue  # the MSRA file stores scale/shift only
        blobs[msra] = torch.randn(sd[d2n].shape, generator=g).numpy()
    blobs["fc1000_w"] = np.zeros((1000, 2048), np.float32)
    blobs["fc1000_b"] = np.zeros((1000,), np.float32)
    assert len(blobs) == 53 * 3 + 2  # 53 convs x (w, bn_s, bn_b) + the classifier
    pkl = tmp_path / "R-50.pkl"
    with open(pkl, "wb") as f:
        pickle.dump(blobs, f)
    state = CK.load_checkpoint(str(pkl))
    missing, unexpected = CK.load_into(model, state)
    assert unexpected == []
    assert all(not k.startswith("backbone.bottom_up.") for k in missing) and any(k.startswith("roi_heads.") for k in missing)
    sd2 = model.state_dict()
    w = "backbone.bottom_up.res3.1.conv2"
    assert torch.equal(sd2[w + ".weight"], torch.from_numpy(blobs["res3_1_branch2b_w"]))
    assert torch.equal(sd2[w + ".norm.weight"], torch.from_numpy(blobs["res3_1_branch2b_bn_s"]))
    # FrozenBN statistics defaulted so that folding reproduces the stored scale / shift
    from openset_rcnn_amd.host.weights import fold_frozen_bn
    folded = fold_frozen_bn({k: v for k, v in sd2.items()})
    assert torch.allclose(folded[w + ".weight"], sd2[w + ".weight"] * sd2[w + ".norm.weight"].view(-1, 1, 1, 1), rtol=1e-6)
    assert torch.allclose(folded[w + ".bias"], sd2[w + ".norm.bias"])
    # .pth in DetectionCheckpointer layout, saved from a DDP-wrapped model ("module." prefix)
    pth = tmp_path / "model_final.pth"
    torch.save({"model": {"module." + k: v for k, v in sd2.items()}, "iteration": 7}, pth)
    state2 = CK.load_checkpoint(str(pth))
    model2 = M.build_model(_cfg(osr, tmp_path))
    assert CK.load_into(model2, state2, strict=True) == ([], [])
    for k, v in model2.state_dict().items():
        assert torch.equal(v, sd2[k]), k
    # shape mismatches are errors, not silent skips
    bad = dict(state2)
    bad["roi_heads.box_predictor.bbox_pred.weight"] = torch.zeros(8, 1024)
    with pytest.raises(ValueError):
        CK.load_into(model2, bad)
