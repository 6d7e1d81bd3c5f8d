"""CPU tests of the host pieces around the training loop: the solver mirror (build_optimizer / build_lr_scheduler), the
resume-exact training data stream and the configuration guards of the loss kernels."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(osr, extra=()):
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "voc_coco.yaml"))
    cfg.merge_from_list(list(extra))
    return cfg


def test_scheduler_mirror_follows_warmup_multistep(osr):
    from openset_rcnn_amd.host.solver import HipSGD, build_lr_scheduler
    from openset_rcnn_amd.host.train import warmup_multistep_lr
    cfg = _cfg(osr)
    opt = HipSGD(model=None, lr=cfg.SOLVER.BASE_LR, momentum=0.9, weight_decay=1e-4)
    sch = build_lr_scheduler(cfg, opt)
    kw = dict(base_lr=cfg.SOLVER.BASE_LR, steps=tuple(cfg.SOLVER.STEPS), gamma=cfg.SOLVER.GAMMA, warmup_iters=cfg.SOLVER.WARMUP_ITERS,
              warmup_factor=cfg.SOLVER.WARMUP_FACTOR)
    for it in range(0, 1000):
        assert opt.param_groups[0]["lr"] == warmup_multistep_lr(it, **kw)  # the lr iteration `it` trains with (train.py:147 logs it)
        sch.step()
    # resumed at iteration 84000: the first milestone has been passed
    opt2 = HipSGD(None, cfg.SOLVER.BASE_LR, 0.9, 1e-4)
    build_lr_scheduler(cfg, opt2, last_iter=83999)
    assert opt2.param_groups[0]["lr"] == pytest.approx(cfg.SOLVER.BASE_LR * 0.1)
    opt.zero_grad()  # accepted, a no-op


def test_unsupported_loss_types_are_rejected_not_silently_replaced(osr):
    from openset_rcnn_amd.host.engine import check_supported_losses, loss_types_of
    from openset_rcnn_amd.host.modeling import engine_cfg_from
    assert engine_cfg_from(_cfg(osr))["rpn_loc_weight"] == 0.5
    check_supported_losses(engine_cfg_from(_cfg(osr)))  # the yaml's choice passes
    # every loss the reference implements is accepted and reaches the kernels as (type, beta) ...
    for opt, key, want in ((["MODEL.RPN.BBOX_REG_LOSS_TYPE", "giou"], "rpn_box", ("giou", 0.0)),
                           (["MODEL.RPN.BBOX_REG_LOSS_TYPE", "smooth_l1", "MODEL.RPN.SMOOTH_L1_BETA", "0.11"], "rpn_box", ("smooth_l1", 0.11)),
                           (["MODEL.RPN.CTR_SMOOTH_L1_BETA", "0.1"], "rpn_ctr", ("smooth_l1", 0.1)),
                           (["MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE", "ciou"], "roi_box", ("ciou", 0.0)),
                           (["MODEL.ROI_BOX_HEAD.IOU_SMOOTH_L1_BETA", "1.0"], "roi_iou", ("smooth_l1", 1.0))):
        c = engine_cfg_from(_cfg(osr, opt))
        check_supported_losses(c)
        assert loss_types_of(c)[key] == want
    # ... and a name it does not implement is refused when a loss is first computed (engine.*_losses_forward), never at inference
    for opt in (["MODEL.RPN.BBOX_REG_LOSS_TYPE", "l2"], ["MODEL.RPN.CTR_REG_LOSS_TYPE", "giou"], ["MODEL.ROI_BOX_HEAD.IOU_REG_LOSS_TYPE", "l1"],
                ["MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE", "huber"]):
        with pytest.raises(NotImplementedError):
            check_supported_losses(engine_cfg_from(_cfg(osr, opt)))
    # RPN.LOSS_WEIGHT multiplies both CF-RPN losses (classification_free_rpn.py:273-276)
    assert engine_cfg_from(_cfg(osr, ["MODEL.RPN.LOSS_WEIGHT", "2.0"]))["rpn_ctr_weight"] == 1.0


@pytest.fixture()
def toy_dicts(tmp_path):
    from PIL import Image
    g = np.random.default_rng(0)
    out = []
    for i in range(7):
        f = tmp_path / f"im{i}.png"
        Image.fromarray(g.integers(0, 256, (40 + i, 60, 3), dtype=np.uint8)).save(f)
        out.append(dict(file_name=str(f), image_id=i, annotations=[dict(bbox=[5, 5, 30, 30], category_id=i % 3)]))
    return out


def test_train_loader_resume_continues_the_exact_stream(osr, toy_dicts, monkeypatch):
    """A loader started at iteration k yields what an uninterrupted loader yields from its k-th batch on (same images, same
    flips and sizes), without mapping any of the skipped samples; ranks see disjoint interleaved samples."""
    from openset_rcnn_amd.host import data as D
    cfg = _cfg(osr, ["INPUT.MIN_SIZE_TRAIN", "(32, 40, 48)", "INPUT.MAX_SIZE_TRAIN", "80"])
    calls = []
    real = D.read_image
    monkeypatch.setattr(D, "read_image", lambda f, fmt="BGR": (calls.append(f), real(f, fmt))[1])

    def take(loader, k):
        return [[(d["image_id"], tuple(d["image"].shape), d["image"].sum().item(), d["instances"].gt_boxes.tensor.tolist()) for d in next(loader)] for _ in range(k)]

    full = take(D.build_detection_train_loader(toy_dicts, D.DatasetMapper(cfg, True, seed=5), 4, seed=5, rank=0, world=1), 9)  # 36 samples: > 5 epochs of 7
    calls.clear()
    tail = take(D.build_detection_train_loader(toy_dicts, D.DatasetMapper(cfg, True, seed=5), 4, seed=5, rank=0, world=1, start_iter=6), 3)
    assert tail == full[6:9]
    assert len(calls) == 12  # only the 3 yielded batches were decoded, none of the 24 skipped samples
    assert len({b[0][1] for b in full}) > 1  # the augmentation does vary
    # two ranks: the union of their per-rank batches is the global batch of the single-rank stream, in interleaved order
    r0 = take(D.build_detection_train_loader(toy_dicts, D.DatasetMapper(cfg, True, seed=5), 4, seed=5, rank=0, world=2), 4)
    r1 = take(D.build_detection_train_loader(toy_dicts, D.DatasetMapper(cfg, True, seed=5), 4, seed=5, rank=1, world=2), 4)
    for it in range(4):
        assert [r0[it][0], r1[it][0], r0[it][1], r1[it][1]] == full[it]


def test_host_side_option_tables_and_pyramid_shapes(osr):
    """Host logic that needs no GPU: the (type, beta) -> osr_loss_options translation and the PLN distance table refuse unknown
    names; pyramid_shapes reproduces the stride arithmetic of the stem (7x7/2 pad 3), the 3x3/2 max pool, the stride-2 1x1 convs
    and LastLevelMaxPool for the two benchmark resolutions (800x1344 and GraspNet's 768x1344) and an odd size."""
    from openset_rcnn_amd.host import ops
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    o = ops._loss_options(("giou", 0.0), 0.25)
    assert (o.box_loss_type, o.box_smooth_l1_beta, o.aux_smooth_l1_beta) == (2, 0.0, 0.25)
    assert [ops._loss_options((k, 0.5), 0.0).box_loss_type for k in ("iou", "smooth_l1", "giou", "diou", "ciou")] == [0, 1, 2, 3, 4]
    with pytest.raises(ops.OsrError):
        ops._loss_options(("l2", 0.0), 0.0)
    assert [ops._pln_distance(k) for k in ("COS", "L1", "L2")] == [0, 1, 2]
    with pytest.raises(ops.OsrError):
        ops._pln_distance("cosine")
    assert OpensetRCNNEngine.pyramid_shapes(800, 1344) == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    assert OpensetRCNNEngine.pyramid_shapes(768, 1344) == [(192, 336), (96, 168), (48, 84), (24, 42), (12, 21)]
    assert OpensetRCNNEngine.pyramid_shapes(256, 352) == [(64, 88), (32, 44), (16, 22), (8, 11), (4, 6)]
