"""Training through the reference's own call signatures, on the GPU:
  * the loop body of /root/reference/train.py:135-146 executed line for line against the mirror (model(data) -> loss dict ->
    .backward() -> optimizer.step() -> scheduler.step()) equals OpensetRCNNTrainer.step;
  * ClsFreeRPN.forward(images, features, gt_instances) and OpensetROIHeads.forward(images, features, proposals, targets) return
    (proposals, loss dict) in training mode (classification_free_rpn.py:531-556, osrcnn_roi_heads.py:268-277);
  * the GraspNet id_map reaches the trainer (make_trainer passes class_map);
  * the overflow guard: an iteration with non-finite gradients changes neither parameters nor momentum."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
LOSS_KEYS = {"loss_rpn_loc", "loss_rpn_ctr", "loss_box_reg", "loss_iou", "loss_dml", "loss_cls"}


def _cfg(osr, yaml="voc_coco.yaml", extra=()):
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", yaml))
    cfg.merge_from_list(["MODEL.DEVICE", DEV, "SOLVER.BASE_LR", "0.0001", "SOLVER.WARMUP_ITERS", "0", "OPENDET_BENCHMARK", "True"] + list(extra))
    return cfg


def _data(classes, seed=3, n=2, h=128, w=160):
    from openset_rcnn_amd.host.structures import Boxes, Instances
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        k = 2 + i
        ctr = torch.rand(k, 2, generator=g) * torch.tensor([w * 0.6, h * 0.6]) + 24
        size = torch.rand(k, 2, generator=g) * 50 + 24
        b = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
        b[:, 0::2].clamp_(0, w)
        b[:, 1::2].clamp_(0, h)
        inst = Instances((h, w), gt_boxes=Boxes(b), gt_classes=torch.tensor([classes[(i + j) % len(classes)] for j in range(k)], dtype=torch.int64))
        out.append({"image": torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8), "height": h, "width": w, "instances": inst})
    return out


def _twin_models(osr, cfg, class_id=None):
    from openset_rcnn_amd.host import modeling as M
    torch.manual_seed(0)
    a = M.build_model(cfg, class_id)
    b = M.build_model(cfg, class_id)
    b.load_state_dict(a.state_dict())
    return a, b


def test_reference_loop_body_runs_verbatim_and_matches_trainer_step(osr):
    from openset_rcnn_amd.host import parallel as comm
    from openset_rcnn_amd.host.solver import build_lr_scheduler, build_optimizer
    cfg = _cfg(osr)
    ref_model, model = _twin_models(osr, cfg)
    data = _data(list(range(20)))
    # --- the one-call trainer of round 1
    tr = ref_model.make_trainer(lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    ref_model.sampler_generator.manual_seed(9)
    want = ref_model.train_step(tr, data, ref_model.sampler_generator)
    snap = {k: v.clone() for k, v in tr.master.items()}  # parameters after ONE update
    # --- /root/reference/train.py:109-111,135-146, as written there
    model.train()
    optimizer = build_optimizer(cfg, model)
    scheduler = build_lr_scheduler(cfg, optimizer)
    model.sampler_generator.manual_seed(9)
    with pytest.raises(RuntimeError):
        optimizer.step()  # no backward yet
    for iteration in range(2):
        loss_dict = model(data)
        losses = sum(loss_dict.values())
        assert torch.isfinite(losses).all(), loss_dict

        loss_dict_reduced = {k: v.item() for k, v in comm.reduce_dict(loss_dict).items()}
        losses_reduced = sum(loss for loss in loss_dict_reduced.values())

        optimizer.zero_grad()
        losses.backward()
        optimizer.step()
        scheduler.step()
        if iteration == 0:
            assert set(loss_dict) == LOSS_KEYS and all(v.is_cuda and v.dim() == 0 for v in loss_dict.values())
            for k in LOSS_KEYS:
                assert torch.equal(loss_dict[k].detach(), want[k]), k  # the forward is bitwise deterministic
            assert losses_reduced == pytest.approx(sum(float(v) for v in want.values()), rel=1e-6)
            first = losses_reduced
            # after ONE update the two paths hold bit-identical parameters wherever the gradient does not pass through RoIAlign's
            # atomic scatter (its fp32 summation order is the step's only run-to-run freedom)
            mine = model.trainer().master
            for k in ("fc1.w", "fc2.w", "pred.w", "enc.w", "dec.w", "cls.w", "protos", "rpn_tail.w", "proposal_generator.rpn_head.conv.w"):
                assert torch.equal(mine[k], snap[k]), k
            for k, v in snap.items():
                assert torch.allclose(mine[k], v, rtol=1e-4, atol=1e-7), k
    assert losses_reduced == losses_reduced and abs(losses_reduced) < 10 * abs(first)
    # one more iteration on the round-1 trainer, then compare the trained parameters
    want2 = ref_model.train_step(tr, data, ref_model.sampler_generator)
    assert all(torch.isfinite(v) for v in want2.values())
    model.eval()  # leaving training mode writes the masters back into the module
    got, exp = model.state_dict(), tr.export_state_dict()
    # (after the second iteration "identical" means to fp16 rounding of the working weights: the second forward reads backbone / FPN
    # weights whose first update carried the scatter's summation-order noise, and a master that lands on the other side of an fp16
    # rounding boundary moves its working copy by one ulp)
    for k, v in exp.items():  # everything downstream of RoIAlign's atomic scatter: equal up to its fp32 summation order
        assert torch.allclose(got[k].cpu(), v, rtol=1e-3, atol=1e-6), k
    assert not torch.equal(got["backbone.fpn_output2.weight"].cpu(), ref_model.state_dict()["backbone.fpn_output2.weight"].cpu())  # it did train
    out = model([{k: v for k, v in d.items() if k != "instances"} for d in data])  # eval-mode forward on the trained weights
    assert len(out) == 2 and out[0]["instances"].has("pred_boxes")
    # a non-uniform weighting of the six losses is refused (the explicit backward differentiates their plain sum)
    model.train()
    ld = model(data)
    with pytest.raises(NotImplementedError):
        (ld["loss_cls"] * 2 + ld["loss_dml"]).backward()


def test_module_level_training_forwards_return_proposals_and_losses(osr):
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr)
    torch.manual_seed(0)
    model = M.build_model(cfg)
    data = _data(list(range(20)))
    model.train()
    images = model.preprocess_image(data)
    gt = [d["instances"] for d in data]
    features = model.backbone(images.tensor)
    model.proposal_generator.sampler_generator.manual_seed(4)
    proposals, rpn_losses = model.proposal_generator(images, features, gt)
    assert set(rpn_losses) == {"loss_rpn_loc", "loss_rpn_ctr"} and all(float(v) > 0 and torch.isfinite(v) for v in rpn_losses.values())
    assert len(proposals) == 2 and all(p.has("proposal_boxes") and p.has("objectness_logits") for p in proposals)
    cap = sum(min(2000, s) for s in (32 * 40, 16 * 20, 8 * 10, 4 * 5, 2 * 3))
    assert all(0 < len(p) <= cap for p in proposals)  # training-time selection: PRE_NMS_TOPK_TRAIN per level
    # the same keys through the model-level path give the same CF-RPN losses (fused head, same selection)
    model.sampler_generator.manual_seed(4)
    whole = model.losses_forward(data, model.sampler_generator)
    for k in rpn_losses:
        assert float(rpn_losses[k]) == pytest.approx(float(whole[k]), rel=1e-4), k
    with pytest.raises(AssertionError):
        model.proposal_generator(images, features, None)  # classification_free_rpn.py:532
    sampled, roi_losses = model.roi_heads(images, features, proposals, gt)
    assert set(roi_losses) == {"loss_box_reg", "loss_iou", "loss_dml", "loss_cls"} and all(torch.isfinite(v) for v in roi_losses.values())
    assert float(roi_losses["loss_cls"]) > 0
    for s, g in zip(sampled, gt):
        assert 0 < len(s) <= cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE
        assert s.has("gt_classes") and s.has("gt_boxes") and s.has("ious") and s.has("proposal_boxes")
        fg = (s.gt_classes >= 0) & (s.gt_classes < cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        assert int(fg.sum()) >= len(g)  # the appended ground-truth boxes are foreground samples (osrcnn_roi_heads.py:177)
        assert float(s.ious[fg].min()) >= cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS[0]
    with pytest.raises(AssertionError):
        model.roi_heads(images, features, proposals, None)  # osrcnn_roi_heads.py:269
    model.eval()
    props, none = model.proposal_generator(images, features)
    assert none == {} and all(len(p) <= sum(min(1000, s) for s in (32 * 40, 16 * 20, 8 * 10, 4 * 5, 2 * 3)) for p in props)


def test_graspnet_id_map_reaches_the_trainer(osr):
    """make_trainer() must hand class_map to the trainer. With a sparse known-class list the PLN and softmax
    losses of trainer.step equal those of losses_forward (which always had the map); dataset ids that are not known classes
    never count as known."""
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr, "graspnet.yaml")
    K, C = cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES, cfg.MODEL.ROI_HEADS.NUM_CLASSES
    class_id = (torch.arange(K) * 3 + 2).to(torch.int64)  # 28 increasing ids, none equal to its own index
    assert int(class_id.max()) < C
    torch.manual_seed(0)
    model = M.build_model(cfg, class_id)
    known, unknown_only = [int(c) for c in class_id[:6]], [0, 1, 3, 4]  # ids 0/1/3/4 are valid dataset classes but NOT known ones
    seen = {}
    for classes, expect_known in ((known, True), (unknown_only, False)):
        data = _data(classes)
        g = torch.Generator().manual_seed(21)
        want = model.losses_forward(data, g)
        tr = model.make_trainer(lr=1e-5)
        assert tr.eng.id_map is not None and torch.equal(tr.eng.class_map.cpu(), class_id)
        g = torch.Generator().manual_seed(21)
        got = model.train_step(tr, data, g)
        for k in ("loss_dml", "loss_cls", "loss_box_reg", "loss_iou"):
            assert float(got[k]) == pytest.approx(float(want[k]), rel=1e-5, abs=1e-7), (k, classes)
        seen[expect_known] = float(got["loss_dml"])
    assert seen[True] != seen[False]  # known ground truth adds the intra / inter prototype hinges, unknown-only ground truth does not


def test_overflow_guard_skips_the_update_and_halves_the_scale(osr):
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr)
    torch.manual_seed(0)
    model = M.build_model(cfg)
    tr = model.make_trainer(lr=1e-4, loss_scale=1024.0)
    data = _data(list(range(20)))
    tensors = model._train_tensors(data, torch.Generator().manual_seed(1))
    tr.step(*tensors, update=False)  # gradients in the flat buffer, parameters untouched
    before = {k: v.clone() for k, v in tr.master.items()}
    mom = {k: v.clone() for k, v in tr.mom.items()}
    lowp = tr.eng.fc2_w.clone()
    tr.grad_flat[tr.grad_flat.numel() // 2] = float("inf")
    tr._update(1)
    torch.cuda.synchronize()
    assert all(torch.equal(v, before[k]) for k, v in tr.master.items()) and all(torch.equal(v, mom[k]) for k, v in tr.mom.items())
    assert torch.equal(tr.eng.fc2_w, lowp)
    assert tr.poll_overflow(wait=True) is True and tr.loss_scale == 512.0 and tr.overflow_steps == 1
    assert tr.poll_overflow() is False  # reported once
    tr.step(*tensors)  # a clean iteration applies
    torch.cuda.synchronize()
    assert tr.poll_overflow(wait=True) is False and tr.overflow_steps == 1
    assert not torch.equal(tr.master["fc2.w"], before["fc2.w"])
    tr.grad_flat[7] = float("nan")
    snap = tr.master["fc2.w"].clone()
    tr._update(1)
    assert tr.poll_overflow(wait=True) is True and tr.loss_scale == 256.0 and torch.equal(tr.master["fc2.w"], snap)


def test_back_to_back_steps_keep_every_overflow_verdict(osr):
    """with step() called in a loop the host runs ahead of the GPU; every update's verdict must still be applied.
    A loss scale of 2^40 overflows the fp16 activation gradients of every iteration: two steps issued back to back without any
    host sync in between, polled afterwards -> two skipped updates, scale quartered, parameters untouched."""
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr)
    torch.manual_seed(0)
    model = M.build_model(cfg)
    tr = model.make_trainer(lr=1e-4, loss_scale=float(2 ** 40))
    data = _data(list(range(20)))
    tensors = model._train_tensors(data, torch.Generator().manual_seed(1))
    torch.cuda.synchronize()
    before = {k: v.clone() for k, v in tr.master.items()}
    tr.step(*tensors)
    tr.step(*tensors)  # (its own poll at the top may or may not see the first verdict yet: either way nothing may be lost)
    assert tr.poll_overflow(wait=True) in (True, False)
    assert tr.overflow_steps == 2 and tr.loss_scale == float(2 ** 38)
    assert all(torch.equal(v, before[k]) for k, v in tr.master.items())


def test_non_finite_proposals_raise_floating_point_error_in_training(osr):
    """find_top_proposals.py:96-101 raises FloatingPointError when predicted boxes / scores are not finite in training and drops
    the rows silently at test time. Stand-alone ClsFreeRPN.forward raises at once (it reads the proposal counts on the host
    anyway); the model-level training path raises when the iteration's verdict is drained (no extra host sync)."""
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr)
    torch.manual_seed(0)
    model = M.build_model(cfg)
    data = _data(list(range(20)))
    model.train()
    images = model.preprocess_image(data)
    gt = [d["instances"] for d in data]
    features = model.backbone(images.tensor)
    props, _ = model.proposal_generator(images, features, gt)
    st = model.proposal_generator.storage  # classification_free_rpn.py:459-463,553-554
    assert set(st) == {"rpn/num_pos_anchors", "rpn/num_neg_anchors", "rpn/obj_num_pos_anchors", "rpn/obj_num_neg_anchors", "rpn/num_proposals"}
    assert 0 < st["rpn/num_pos_anchors"] <= 128 and st["rpn/num_pos_anchors"] + st["rpn/num_neg_anchors"] == 256
    assert st["rpn/num_proposals"] == sum(len(p) for p in props) / 2
    with torch.no_grad():
        model.proposal_generator.rpn_head.anchor_deltas.bias[1] = float("nan")
    model.refresh()
    with pytest.raises(FloatingPointError, match="Training has diverged"):
        model.proposal_generator(images, features, gt)
    model.eval()
    props, _ = model.proposal_generator(images, features)  # test time: silently filtered (every anchor of every level is dropped)
    assert all(len(p) == 0 for p in props)
    # model level: the trainer's forward queues the status word behind the update's overflow verdict
    model.train()
    tr = model.make_trainer(lr=1e-4, loss_scale=1024.0)
    tensors = model._train_tensors(data, torch.Generator().manual_seed(1))
    tr.step(*tensors)
    with pytest.raises(FloatingPointError, match="Predicted boxes or scores contain Inf/NaN"):
        tr.poll_overflow(wait=True)


def test_resumed_optimizer_state_carries_the_loss_scale(osr):
    """the dynamic loss scale (scale, clean-update count, skipped updates) is part of the optimizer state; a run
    that had backed off resumes at the scale it was written with."""
    from openset_rcnn_amd.host import modeling as M
    cfg = _cfg(osr)
    torch.manual_seed(0)
    model = M.build_model(cfg)
    tr = model.make_trainer(lr=1e-4, loss_scale=1024.0)
    data = _data(list(range(20)))
    tensors = model._train_tensors(data, torch.Generator().manual_seed(1))
    tr.step(*tensors, update=False)
    tr.grad_flat[3] = float("inf")
    tr._update(1)
    state = tr.export_optimizer_state()  # (drains the verdicts first)
    assert tr.loss_scale == 512.0 and float(state[tr.SCALE_KEY][0]) == 512.0 and int(state[tr.SCALE_KEY][2]) == 1
    tr2 = model.make_trainer(lr=1e-4, loss_scale=1024.0)
    assert tr2.loss_scale == 1024.0
    tr2.load_optimizer_state(state)
    assert tr2.loss_scale == 512.0 and tr2.overflow_steps == 1 and tr2.scaler.scale_max == 1024.0
    assert all(torch.equal(tr2.mom[k].cpu(), state[k]) for k in tr2.mom)
