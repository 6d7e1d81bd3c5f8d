"""BASELINE config 1 ("Base-RCNN-FPN.yaml R50 on 2 synthetic COCO-format images"): the stock detectron2 modules the reference's base
yaml names -- RPN + StandardRPNHead (A = 3, objectness logits, Box2BoxTransform, per-level NMS 0.7, post-NMS top-k) and
StandardROIHeads + FastRCNNOutputLayers (softmax over 80 + 1, per-class NMS 0.5, 100 detections) -- on the HIP path, against the
oracle's restatement of detectron2's published algorithms (oracle/osr_oracle.py, "BASELINE config 1" section).

CPU: registry names, state-dict keys, cell anchors, the oracle's own known answers. GPU: two synthetic COCO-format images go
through dataset registration -> DatasetMapper -> model(batch); every index-producing stage is compared with the oracle fed the
engine's own inputs to that stage (bit-exact), dense stages at their rounding tolerance, and the parity (fp32) mode end to end."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_binding as CO
from oracle import osr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _cfg(osr, device="cpu"):
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "base_rcnn_fpn.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", device])
    return cfg


def test_stock_names_are_registered_and_keys_match_detectron2(osr):
    from openset_rcnn_amd.host import modeling as M
    for reg, name in ((M.PROPOSAL_GENERATOR_REGISTRY, "RPN"), (M.RPN_HEAD_REGISTRY, "StandardRPNHead"), (M.ROI_HEADS_REGISTRY, "StandardROIHeads")):
        assert name in reg
    cfg = _cfg(osr)
    assert cfg.MODEL.PROPOSAL_GENERATOR.NAME == "RPN" and cfg.MODEL.RPN.HEAD_NAME == "StandardRPNHead" and cfg.MODEL.ROI_HEADS.NAME == "StandardROIHeads"
    model = M.build_model(cfg)
    sd = model.state_dict()
    assert sd["proposal_generator.rpn_head.objectness_logits.weight"].shape == (3, 256, 1, 1)
    assert sd["proposal_generator.rpn_head.anchor_deltas.weight"].shape == (12, 256, 1, 1)
    assert sd["roi_heads.box_predictor.cls_score.weight"].shape == (81, 1024)
    assert sd["roi_heads.box_predictor.bbox_pred.weight"].shape == (4, 1024)  # CLS_AGNOSTIC_BBOX_REG: true in the reference's base yaml
    assert not any("centerness" in k or "dml" in k or "iou_pred" in k for k in sd)
    ref = "/root/reference/configs/Base-RCNN-FPN.yaml"
    if os.path.exists(ref):  # the reference's own file, unchanged, resolves to the same configuration tree
        from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
        c2 = get_cfg()
        add_openset_rcnn_config(c2)
        c2.merge_from_file(ref)
        a, b = _cfg(osr), c2
        a.MODEL.DEVICE = b.MODEL.DEVICE
        assert json.dumps(a, default=str, sort_keys=True) == json.dumps(b, default=str, sort_keys=True)


def test_cell_anchors_and_oracle_known_answers(osr):
    from openset_rcnn_amd.host.engine_std import cell_anchor_table
    t = cell_anchor_table((32, 64), (0.5, 1.0, 2.0))
    assert t.shape == (2, 3, 4)
    # ratio 1: the square; ratio 0.5: w = 32*sqrt(2), h = 32/sqrt(2) (area kept); ratio 2 is its transpose
    assert t[0, 1].tolist() == [-16.0, -16.0, 16.0, 16.0]
    assert t[0, 0, 2].item() == pytest.approx(16 * 2 ** 0.5, rel=1e-6) and t[0, 0, 3].item() == pytest.approx(16 / 2 ** 0.5, rel=1e-6)
    assert torch.allclose(t[0, 2, [1, 0, 3, 2]], t[0, 0]) and torch.allclose(t[1], 2 * t[0])
    assert torch.equal(O.anchor_grid([(2, 3)], strides=(4,), sizes=(32,), ratios=(0.5, 1.0, 2.0))[0][:3], t[0])  # cell (0,0): A innermost
    # Box2BoxTransform: zero deltas = identity; dw clamped at log(1000/16)
    b = torch.tensor([[10.0, 20.0, 50.0, 80.0]])
    assert torch.allclose(O.b2b_apply_deltas(torch.zeros(1, 4), b, (1, 1, 1, 1)), b)
    big = O.b2b_apply_deltas(torch.tensor([[0.0, 0.0, 100.0, 0.0]]), b, (1, 1, 1, 1))
    assert float(big[0, 2] - big[0, 0]) == pytest.approx(40 * 1000 / 16, rel=1e-5)
    # per-level NMS: identical boxes on two levels both survive, on one level only the better one does
    props = [torch.tensor([[[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 10.0]]]), torch.tensor([[[0.0, 0.0, 10.0, 10.0]]])]
    logits = [torch.tensor([[2.0, 1.0]]), torch.tensor([[0.5]])]
    (bx, sc, lv), = O.standard_find_top_rpn_proposals(props, logits, [(100, 100)], 0.7, 1000, 1000)
    assert sc.tolist() == [2.0, 0.5] and lv.tolist() == [0, 1]
    # class-agnostic deltas, two rows: softmax + threshold + per-class NMS; the background column never becomes a detection
    p = {"roi_heads.box_predictor.cls_score.weight": torch.zeros(3, 4), "roi_heads.box_predictor.cls_score.bias": torch.tensor([3.0, 0.0, 4.0]),
         "roi_heads.box_predictor.bbox_pred.weight": torch.zeros(4, 4), "roi_heads.box_predictor.bbox_pred.bias": torch.zeros(4)}
    cfg = dict(O.BASE_RCNN_CFG, num_classes=2)
    bb, ss, cc, rc = O.fast_rcnn_output_inference(torch.zeros(2, 4), torch.tensor([[0.0, 0.0, 10.0, 10.0], [1.0, 0.0, 10.0, 10.0]]), (50, 50), p, cfg)
    assert cc.tolist() == [0] and rc.tolist() == [[0, 0]] and float(ss[0]) == pytest.approx(float(F.softmax(torch.tensor([3.0, 0.0, 4.0]), 0)[0]))  # class 1 (p = 0.013) is below SCORE_THRESH_TEST


@pytest.fixture()
def coco_toy(tmp_path):
    """Two 480x640 JPEGs with three boxes each and a COCO-format instances json, generated on the fly."""
    from PIL import Image
    g = np.random.default_rng(5)
    root = tmp_path / "coco_toy"
    (root / "images").mkdir(parents=True)
    images, anns = [], []
    for i in range(2):
        Image.fromarray(g.integers(0, 256, (480, 640, 3), dtype=np.uint8)).save(root / "images" / f"{i}.jpg", quality=95)
        images.append(dict(id=i + 1, file_name=f"{i}.jpg", height=480, width=640))
        for j in range(3):
            x, y, w, h = 40 + 150 * j + 10 * i, 60 + 90 * j, 120.0, 100.0 + 20 * j
            anns.append(dict(id=len(anns) + 1, image_id=i + 1, category_id=(j % 2) + 1, bbox=[x, y, w, h], area=w * h, iscrowd=0))
    blob = dict(images=images, annotations=anns, categories=[dict(id=1, name="thing_a"), dict(id=2, name="thing_b")])
    (root / "instances.json").write_text(json.dumps(blob))
    return str(root)


def _nchw(t):
    return t.detach().cpu().float().permute(0, 3, 1, 2).contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["f16", "f32-parity"])
def test_base_rcnn_fpn_on_two_synthetic_coco_images(osr, coco_toy, dtype):
    from openset_rcnn_amd.host import datasets as D
    from openset_rcnn_amd.host import modeling as M
    from openset_rcnn_amd.host.data import DatasetMapper, build_detection_test_loader
    from openset_rcnn_amd.host.weights import random_standard_params
    cfg = _cfg(osr, DEV)
    cfg.merge_from_list(["INPUT.MIN_SIZE_TEST", "480", "INPUT.MAX_SIZE_TEST", "640"])
    dicts = D.load_coco_json(os.path.join(coco_toy, "instances.json"), os.path.join(coco_toy, "images"))
    assert len(dicts) == 2 and len(dicts[0]["annotations"]) == 3
    batch = next(iter(build_detection_test_loader(dicts, DatasetMapper(cfg, is_train=False), batch_size=2, rank=0, world=1)))
    params = random_standard_params(0)
    model = M.build_model(cfg)
    sd = model.state_dict()
    for k, v in params.items():
        if k in sd:
            sd[k] = v
        elif k.endswith(".bias") and k[:-5] + ".norm.bias" in sd:
            sd[k[:-5] + ".norm.bias"] = v
    model.load_state_dict(sd)
    model.kernel_dtype = dtype
    model.eval()
    out = model(batch)
    assert len(out) == 2 and all(0 < len(o["instances"]) <= 100 for o in out)
    # ---- stage by stage against the oracle, on the engine's own inputs ----
    eng = model.engine()
    keep = {}
    imgs = torch.stack([b["image"] for b in batch]).to(DEV)
    res = eng.forward(imgs, [(480, 640)] * 2, keep=keep)
    torch.cuda.synchronize()
    q = (lambda t: t) if dtype == torch.float32 else (lambda t: t.half().float())
    tol = 1e-4 if dtype == torch.float32 else 5e-3
    p = {k: (q(v) if v.dim() == 4 and k.startswith(("backbone.", "proposal_generator.rpn_head.conv")) else v) for k, v in params.items()}
    feats = {k: _nchw(v) for k, v in keep["feats"].items()}
    n = 2
    # StandardRPNHead on the engine's pyramid
    ds, ls = [], []
    for k in ("p2", "p3", "p4", "p5", "p6"):
        d, l = O.standard_rpn_head(feats[k], p)
        ds.append(d)
        ls.append(l)
    ds, ls = O.flatten_head_outputs(ds, ls)
    l_ref, d_ref = torch.cat([l.reshape(-1) for l in ls]), torch.cat([d.reshape(-1, 4) for d in ds])
    shapes = keep["rpn_shapes"]
    # engine layout: level-major, inside a level image-major (n, h*w*A): same as the concatenation of the flattened (N, HWA) blocks
    assert float((keep["rpn_logits"].cpu() - l_ref).abs().max()) < tol * max(1.0, float(l_ref.abs().max()))
    assert float((keep["rpn_deltas"].cpu() - d_ref).abs().max()) < tol * max(1.0, float(d_ref.abs().max()))
    # selection + per-level NMS + post-NMS top-k: oracle fed the engine's logits / deltas -> same proposals, same order
    anchors = O.anchor_grid(shapes, sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1.0, 2.0))
    lg, dl, off = [], [], 0
    for (h, w) in shapes:
        cnt = n * h * w * 3
        lg.append(keep["rpn_logits"][off:off + cnt].cpu().view(n, -1))
        dl.append(keep["rpn_deltas"][off:off + cnt].cpu().view(n, -1, 4))
        off += cnt
    props = [O.b2b_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4), (1.0, 1.0, 1.0, 1.0)).view(n, -1, 4) for d, a in zip(dl, anchors)]
    ref_props = O.standard_find_top_rpn_proposals(props, lg, [(480, 640)] * 2, 0.7, 1000, 1000)
    sel = keep["sel"]
    for i, (rb, rs, rl) in enumerate(ref_props):
        c = int(sel["counts"][i])
        assert c == len(rb), (c, len(rb))
        assert torch.equal(sel["scores"][i, :c].cpu(), rs)
        assert torch.allclose(sel["boxes"][i, :c].cpu(), rb, rtol=1e-5, atol=1e-4)  # expf on the device vs libm: last-ulp differences
    # box head + FastRCNNOutputLayers.inference: oracle fed the engine's proposals and box features
    counts = [int(c) for c in sel["counts"].cpu()]
    bf = keep["box_feats"].view(n, sel["cap"], -1)
    for i in range(n):
        eb = sel["boxes"][i, :counts[i]].cpu()
        rb, rs, rc, _ = O.fast_rcnn_output_inference(bf[i, :counts[i]].cpu(), eb, (480, 640), params)
        m = int(res[3][i])
        assert m == len(rb)
        # the same detections with the same scores; two detections whose scores agree to the comparison's tolerance may come in either order (the oracle's
        # torch-CPU matrix products and the device's exact-fp32 MFMA sum in different orders, so a near-tie can sort either way)
        gb, gs, gc = res[0][i, :m].cpu(), res[1][i, :m].cpu(), res[2][i, :m].cpu()
        assert torch.allclose(gs, rs, atol=1e-6)
        used = torch.zeros(m, dtype=torch.bool)
        for j in range(m):
            ok = (gc == rc[j]) & ((gs - rs[j]).abs() <= 1e-6 + 1e-5 * rs[j].abs()) & ((gb - rb[j]).abs().amax(dim=1) <= 1e-4 + 1e-5 * rb[j].abs().max()) & ~used
            assert bool(ok.any()), (i, j, int(rc[j]), float(rs[j]))
            k = int(torch.nonzero(ok)[0])
            assert abs(k - j) <= 2, (j, k)  # (only neighbours in the sorted list can swap)
            used[k] = True
    if dtype == torch.float32:  # parity mode: the whole network against the fp32 oracle
        ref, _ = O.standard_detector_inference([b["image"] for b in batch], params, roi_align_fn=CO.roi_align)
        matched = total = 0
        for i in range(n):
            m = int(res[3][i])
            gb, gs, gc = res[0][i, :m].cpu(), res[1][i, :m].cpu(), res[2][i, :m].cpu()
            rb, rs, rc, _ = ref[i]
            iou = O.pairwise_iou(rb, gb)
            used = torch.zeros(m, dtype=torch.bool)
            for j in range(len(rb)):
                ok = (iou[j] >= 0.99) & ((gs - rs[j]).abs() <= 1e-2) & (gc == rc[j]) & ~used
                if bool(ok.any()):
                    used[int(torch.nonzero(ok)[0])] = True
                    matched += 1
            total += max(m, len(rb))
        print(f"\n[config 1, parity mode] {matched}/{total} detections agree with the fp32 oracle")
        assert matched >= 0.95 * total
