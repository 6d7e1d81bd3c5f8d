"""The host-side mirror of the reference's plug-in surface: registries, config keys, yaml loading, state-dict names,
signatures. CPU tests cover construction/configuration; GPU tests run the mirrored modules on the HIP path."""
import os

import pytest
import torch

REF = "/root/reference/configs"

YAML_BASE = """
MODEL:
  META_ARCHITECTURE: "GeneralizedRCNN"
  BACKBONE: {NAME: "build_resnet_fpn_backbone"}
  RESNETS: {OUT_FEATURES: ["res2", "res3", "res4", "res5"]}
  FPN: {IN_FEATURES: ["res2", "res3", "res4", "res5"]}
  ANCHOR_GENERATOR: {SIZES: [[32], [64], [128], [256], [512]], ASPECT_RATIOS: [[0.5, 1.0, 2.0]]}
  RPN: {IN_FEATURES: ["p2", "p3", "p4", "p5", "p6"], PRE_NMS_TOPK_TRAIN: 2000, PRE_NMS_TOPK_TEST: 1000}
  ROI_HEADS: {NAME: "StandardROIHeads", IN_FEATURES: ["p2", "p3", "p4", "p5"]}
  ROI_BOX_HEAD: {NAME: "FastRCNNConvFCHead", NUM_FC: 2, POOLER_RESOLUTION: 7, CLS_AGNOSTIC_BBOX_REG: true}
INPUT:
  MIN_SIZE_TRAIN: (640, 672, 704, 736, 768, 800)
DATASETS:
  TRAIN: ("coco_2017_train",)
VERSION: 2
"""
YAML_CHILD = """
_BASE_: "base.yaml"
MODEL:
  ANCHOR_GENERATOR: {ASPECT_RATIOS: [[1.0]]}
  PROPOSAL_GENERATOR: {NAME: "ClsFreeRPN"}
  RPN: {HEAD_NAME: "ClsFreeRPNHead", BBOX_REG_LOSS_TYPE: "iou", NMS_THRESH_TEST: 1.0, IOU_THRESHOLDS_OBJECTNESS: [0.1, 0.3]}
  ROI_HEADS: {NAME: "OpensetROIHeads", NUM_CLASSES: 81, NUM_KNOWN_CLASSES: 20, NMS_THRESH_TEST: 1.0, KNOWN_TOPK: 50, UNKNOWN_TOPK: 50,
              UNKNOWN_SCORE_THRESH: 0.0}
  PLN: {UNK_THR: 0.23}
TEST: {DETECTIONS_PER_IMAGE: 1000}
"""


def _cfg(osr, tmp_path):
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    (tmp_path / "base.yaml").write_text(YAML_BASE)
    (tmp_path / "child.yaml").write_text(YAML_CHILD)
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    cfg.merge_from_file(str(tmp_path / "child.yaml"))
    cfg.merge_from_list(["MODEL.PLN.ALPHA", "0.2", "OPENDET_BENCHMARK", "True"])
    return cfg


def test_config_base_inheritance_overrides_and_freeze(osr, tmp_path):
    cfg = _cfg(osr, tmp_path)
    assert cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS == [[1.0]] and cfg.MODEL.RPN.PRE_NMS_TOPK_TEST == 1000
    assert cfg.MODEL.PLN.ALPHA == 0.2 and cfg.OPENDET_BENCHMARK is True and cfg.MODEL.PLN.UNK_THR == 0.23
    assert cfg.INPUT.MIN_SIZE_TRAIN == (640, 672, 704, 736, 768, 800) and cfg.DATASETS.TRAIN == ("coco_2017_train",)
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cpu"
    with pytest.raises(KeyError):
        c2 = cfg.clone()
        c2.merge_from_list(["MODEL.NOT_A_KEY", 1])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name,known,unk", [("VOC-COCO/openset_rcnn_R50_FPN_128k.yaml", 20, 0.23), ("GraspNet/openset_rcnn_R50_FPN_128k.yaml", 28, 0.09),
                                            ("Base-RCNN-FPN.yaml", 20, 0.4)])
def test_reference_yaml_files_load_unchanged(osr, name, known, unk):
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    cfg.merge_from_file(os.path.join(REF, name))
    assert cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES == known and cfg.MODEL.PLN.UNK_THR == unk
    assert cfg.MODEL.RPN.IN_FEATURES == ["p2", "p3", "p4", "p5", "p6"]


def test_registry_names_and_state_dict_keys(osr, tmp_path):
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr, tmp_path)
    cfg.MODEL.DEVICE = "cpu"
    for reg, name in ((M.META_ARCH_REGISTRY, "GeneralizedRCNN"), (M.BACKBONE_REGISTRY, "build_resnet_fpn_backbone"),
                      (M.PROPOSAL_GENERATOR_REGISTRY, "ClsFreeRPN"), (M.RPN_HEAD_REGISTRY, "ClsFreeRPNHead"),
                      (M.ROI_HEADS_REGISTRY, "OpensetROIHeads"), (M.ROI_BOX_HEAD_REGISTRY, "FastRCNNConvFCHead")):
        assert name in reg
    model = M.build_model(cfg)
    keys = set(model.state_dict().keys())
    for k in ("proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.anchor_deltas.bias",
              "proposal_generator.rpn_head.centerness.weight", "roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc2.bias",
              "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.box_predictor.iou_pred.bias", "roi_heads.dml.encoder.weight",
              "roi_heads.dml.decoder.bias", "roi_heads.dml.representatives", "roi_heads.softmaxcls.cls_score.weight",
              "backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.stem.conv1.norm.running_var",
              "backbone.bottom_up.res2.0.shortcut.weight", "backbone.bottom_up.res5.2.conv3.norm.bias",
              "backbone.fpn_lateral5.weight", "backbone.fpn_output2.bias"):
        assert k in keys, k
    n_train = sum(p.numel() for n, p in model.named_parameters() if not n.startswith("backbone.bottom_up.stem") and not n.startswith("backbone.bottom_up.res2"))
    assert n_train == 41_621_279  # trainable parameters with FREEZE_AT=2
    assert model.state_dict()["roi_heads.dml.representatives"].shape == (20, 256)
    assert model.state_dict()["roi_heads.softmaxcls.cls_score.weight"].shape == (21, 1024)
    assert model.roi_heads._eng_cfg["unknown_id"] == 80 and model.roi_heads._eng_cfg["unk_thr"] == 0.23
    model.eval()
    with pytest.raises(osr.OsrError):  # CPU model: refused, no eager fallback
        model([{"image": torch.zeros(3, 64, 64, dtype=torch.uint8), "height": 64, "width": 64}])
    model.train()
    with pytest.raises(osr.OsrError):  # training mode builds the trainer: refused on the CPU as well
        model([{"image": torch.zeros(3, 64, 64, dtype=torch.uint8)}])


def test_fold_frozen_bn_matches_definition(osr):
    from openset_rcnn_amd.host.weights import fold_frozen_bn
    g = torch.Generator().manual_seed(0)
    sd = {"c.weight": torch.randn(8, 4, 3, 3, generator=g), "c.norm.weight": torch.rand(8, generator=g) + 0.5,
          "c.norm.bias": torch.randn(8, generator=g), "c.norm.running_mean": torch.randn(8, generator=g),
          "c.norm.running_var": torch.rand(8, generator=g) + 0.1}
    x = torch.randn(2, 4, 9, 9, generator=g)
    y = torch.nn.functional.conv2d(x, sd["c.weight"], padding=1)
    scale = sd["c.norm.weight"] * (sd["c.norm.running_var"] + 1e-5).rsqrt()
    ref = y * scale.view(1, -1, 1, 1) + (sd["c.norm.bias"] - sd["c.norm.running_mean"] * scale).view(1, -1, 1, 1)
    f = fold_frozen_bn(sd)
    assert set(f) == {"c.weight", "c.bias"}
    assert torch.allclose(torch.nn.functional.conv2d(x, f["c.weight"], f["c.bias"], padding=1), ref, atol=1e-5)


def test_structures(osr):
    from openset_rcnn_amd.host.structures import Boxes, ImageList, Instances
    b = Boxes(torch.tensor([[-5.0, 2.0, 30.0, 50.0], [10.0, 10.0, 10.0, 20.0]]))
    b.clip((40, 25))
    assert b.tensor.tolist() == [[0.0, 2.0, 25.0, 40.0], [10.0, 10.0, 10.0, 20.0]]
    assert b.nonempty().tolist() == [True, False] and b.area().tolist() == [25.0 * 38.0, 0.0]
    i = Instances((40, 25), pred_boxes=b, scores=torch.tensor([0.9, 0.1]))
    assert len(i) == 2 and len(i[i.scores > 0.5]) == 1 and i.has("scores") and i.image_size == (40, 25)
    il = ImageList.from_tensors([torch.ones(3, 5, 7), torch.ones(3, 6, 4)], 32)
    assert il.tensor.shape == (2, 3, 32, 32) and il.image_sizes == [(5, 7), (6, 4)] and float(il.tensor[0, 0, 5:].sum()) == 0.0


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_mirrored_modules_run_on_the_hip_path(osr, tmp_path):
    from openset_rcnn_amd.host import modeling as M
    from openset_rcnn_amd.host.structures import ImageList
    from openset_rcnn_amd.host.weights import random_params
    cfg = _cfg(osr, tmp_path)
    model = M.build_model(cfg).eval()
    # load seeded BN-folded parameters through the state-dict surface (identity FrozenBN statistics)
    p = random_params(0)
    sd = model.state_dict()
    for k, v in p.items():
        if k in sd:
            sd[k] = v
        elif k.endswith(".bias") and k[:-5] + ".norm.bias" in sd:
            sd[k[:-5] + ".norm.bias"] = v
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randint(0, 256, (3, 200, 300), generator=g, dtype=torch.uint8) for _ in range(2)]
    out = model([{"image": im, "height": 400, "width": 600} for im in imgs])
    assert len(out) == 2 and {"pred_boxes", "scores", "pred_classes"} <= set(out[0]["instances"].get_fields())
    # same result as the engine fed the same parameters directly
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    eng = OpensetRCNNEngine(p, M.engine_cfg_from(cfg), device="cuda:0")
    res = eng.to_instances(eng.forward(torch.stack(imgs).cuda()), 2)
    for r, o in zip(res, out):
        inst = o["instances"]
        assert inst.image_size == (400, 600)
        assert torch.allclose(inst.pred_boxes.tensor, r["pred_boxes"] * 2.0) and torch.equal(inst.pred_classes, r["pred_classes"])
    # stage-level signatures: backbone -> proposal generator -> roi heads, chained through the mirrored API
    il = model.preprocess_image([{"image": im} for im in imgs])
    feats = model.backbone(il.tensor)
    assert feats["p2"].shape == (2, 256, 56, 80) and feats["p6"].shape == (2, 256, 4, 5)
    d, c = model.proposal_generator.rpn_head([feats["p4"]])
    assert d[0].shape == (2, 4, 14, 20) and c[0].shape == (2, 1, 14, 20) and float(c[0].min()) > 0 and float(c[0].max()) < 1
    props, losses = model.proposal_generator(il, feats)
    assert losses == {} and props[0].has("proposal_boxes") and props[0].has("objectness_logits")
    dets, _ = model.roi_heads(il, feats, props)
    for r, dd in zip(res, dets):
        assert torch.equal(dd.pred_classes, r["pred_classes"]) and torch.allclose(dd.pred_boxes.tensor, r["pred_boxes"])
    # ragged batch path
    out2 = model([{"image": imgs[0]}, {"image": imgs[1][:, :180, :250]}])
    assert len(out2) == 2


@pytest.mark.gpu
def test_losses_forward_through_the_mirror(osr, tmp_path):
    from openset_rcnn_amd.host import modeling as M
    from openset_rcnn_amd.host.structures import Boxes, Instances
    from openset_rcnn_amd.host.weights import random_params
    cfg = _cfg(osr, tmp_path)
    model = M.build_model(cfg)
    p = random_params(0)
    sd = model.state_dict()
    for k, v in p.items():
        if k in sd:
            sd[k] = v
        elif k.endswith(".bias") and k[:-5] + ".norm.bias" in sd:
            sd[k[:-5] + ".norm.bias"] = v
    model.load_state_dict(sd)
    model.train()
    g = torch.Generator().manual_seed(5)
    inputs = []
    for i in range(2):
        inst = Instances((200, 300))
        inst.gt_boxes = Boxes(torch.tensor([[20.0, 30.0, 120.0, 150.0], [150.0, 40.0, 280.0, 190.0]])[: 2 - i])
        inst.gt_classes = torch.tensor([3, 17])[: 2 - i]
        inputs.append({"image": torch.randint(0, 256, (3, 200, 300), generator=g, dtype=torch.uint8), "instances": inst})
    model.sampler_generator.manual_seed(9)
    ld = model(inputs)  # training mode: the loss dict of train.py:135; its sum's .backward() runs the explicit HIP backward
    assert all(v.grad_fn is not None for v in ld.values())
    l1 = model.losses_forward(inputs, torch.Generator().manual_seed(9))
    l2 = model.losses_forward(inputs, torch.Generator().manual_seed(9))
    assert set(l1) == {"loss_rpn_loc", "loss_rpn_ctr", "loss_box_reg", "loss_iou", "loss_dml", "loss_cls"} == set(ld)
    for k in l1:  # same keys, same sampled sets (the trainer's unfused RPN head gives the fused head's values)
        assert float(ld[k]) == pytest.approx(float(l1[k]), rel=1e-5, abs=1e-7), k
    for k in l1:
        assert torch.isfinite(l1[k]).all() and float(l1[k]) > 0 and torch.equal(l1[k], l2[k]), k  # reproducible bit for bit


@pytest.mark.gpu
def test_trainer_from_the_mirror_roundtrip(osr, tmp_path):
    """model.make_trainer(): FrozenBN-aware masters, one SGD step, parameters written back into the module."""
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr, tmp_path)
    model = M.build_model(cfg)
    g = torch.Generator().manual_seed(11)
    sd = model.state_dict()
    for k in sd:  # non-trivial FrozenBN scales on a trained layer
        if k.endswith("res4.1.conv2.norm.weight"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    model.load_state_dict(sd)
    tr = model.make_trainer(lr=1e-5, loss_scale=256.0)
    before = {k: v.clone() for k, v in tr.export_state_dict().items()}
    name = "backbone.bottom_up.res4.1.conv2"
    assert torch.allclose(before[name + ".weight"], model.state_dict()[name + ".weight"].cpu())  # the master is the UN-folded weight
    assert name + ".w" in tr.row_scale
    n, h, w = 1, 128, 160
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).cuda()
    hw = torch.tensor([(h, w)], dtype=torch.int32).cuda()
    gt = torch.tensor([[[20.0, 30.0, 90.0, 100.0], [70.0, 20.0, 150.0, 110.0]]]).cuda()
    gcls = torch.tensor([[3, 7]]).cuda()
    gcnt = torch.tensor([2], dtype=torch.int32).cuda()
    shapes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    r = sum(a * b for a, b in shapes)
    cap = sum(min(cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, a * b) for a, b in shapes)
    keys = {k: torch.rand(s, generator=g).cuda() for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + 2)))}
    losses = tr.step(images, hw, h, w, gt, gcls, gcnt, keys)
    assert all(torch.isfinite(v).all() for v in losses.values())
    after = tr.export_state_dict()
    assert not torch.equal(after[name + ".weight"], before[name + ".weight"])
    model.load_trainer_state(tr)
    assert torch.equal(model.state_dict()[name + ".weight"].cpu(), after[name + ".weight"])
    assert torch.equal(model.state_dict()["roi_heads.box_head.fc1.weight"].cpu(), after["roi_heads.box_head.fc1.weight"])


def test_warmup_multistep_lr(osr):
    from openset_rcnn_amd.host.train import warmup_multistep_lr as lr
    kw = dict(base_lr=0.005, steps=(84000, 116000), gamma=0.1, warmup_iters=400, warmup_factor=0.001)
    assert lr(0, **kw) == pytest.approx(0.005 * 0.001)
    assert lr(200, **kw) == pytest.approx(0.005 * (0.001 * 0.5 + 0.5))
    assert lr(400, **kw) == pytest.approx(0.005) and lr(83999, **kw) == pytest.approx(0.005)
    assert lr(84000, **kw) == pytest.approx(0.0005) and lr(116000, **kw) == pytest.approx(0.00005)
