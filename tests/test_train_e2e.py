"""Forward half of the training step on the GPU (OpensetRCNNEngine.forward_losses), checked stage by stage against
the CPU oracle. As in test_engine_e2e.py each oracle stage is fed the ENGINE's own inputs to that stage, so index
outputs (labels, sampled candidates) must match bit-exactly and loss scalars to 1e-5 relative."""
import pytest
import torch
import torch.nn.functional as F

from oracle import c_binding as CO
from oracle import osr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nchw(t):
    return t.detach().cpu().float().permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


@pytest.fixture(scope="module")
def run(osr):
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    params = random_params(0)
    eng = OpensetRCNNEngine(params, dtype=torch.float16, device=DEV)
    g = torch.Generator().manual_seed(17)
    n, h, w, gmax = 2, 250, 330, 6
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    sizes = [(250, 330), (240, 300)]
    gt = torch.zeros(n, gmax, 4)
    gcls = torch.zeros(n, gmax, dtype=torch.int64)
    gcnt = [4, 2]
    for i, c in enumerate(gcnt):
        ctr = torch.rand(c, 2, generator=g) * torch.tensor([sizes[i][1] * 0.8, sizes[i][0] * 0.8]) + 20
        size = torch.rand(c, 2, generator=g) * 100 + 24
        b = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
        b[:, 0::2].clamp_(0, sizes[i][1])
        b[:, 1::2].clamp_(0, sizes[i][0])
        gt[i, :c] = b
        gcls[i, :c] = torch.randint(0, 20, (c,), generator=g)
    hp, wp = 256, 352
    shapes = O.level_shapes(hp, wp)
    r = sum(a * b for a, b in shapes)
    # train-time selection capacity: min(2000, level size) summed over the levels
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = dict(rpn_reg=torch.rand(n, r, generator=g), rpn_obj=torch.rand(n, r, generator=g), roi=torch.rand(n, cap + gmax, generator=g))
    keep = {}
    image_hw = torch.tensor(sizes, dtype=torch.int32).to(DEV)
    out = eng.forward_losses(images.to(DEV), image_hw, hp, wp, gt.to(DEV), gcls.to(DEV), torch.tensor(gcnt, dtype=torch.int32).to(DEV),
                             {k: v.to(DEV) for k, v in keys.items()}, keep=keep)
    torch.cuda.synchronize()
    return dict(eng=eng, params=params, sizes=sizes, gt=gt, gcls=gcls, gcnt=gcnt, keys=keys, keep=keep, out=out, shapes=shapes, n=n,
                cap=cap, gmax=gmax)


def test_rpn_targets_and_losses(run):
    keep, n, shapes = run["keep"], run["n"], run["shapes"]
    anchors = torch.cat(O.anchor_grid(shapes))
    refs = [O.rpn_label_and_sample(anchors, run["gt"][i, :run["gcnt"][i]], run["keys"]["rpn_reg"][i], run["keys"]["rpn_obj"][i]) for i in range(n)]
    for i, ref in enumerate(refs):
        assert torch.equal(keep["labels_pre"][i].cpu(), ref["labels_pre"]) and torch.equal(keep["obj_labels_pre"][i].cpu(), ref["obj_labels_pre"])
        assert torch.equal(keep["labels"][i].cpu(), ref["labels"]) and torch.equal(keep["obj_labels"][i].cpu(), ref["obj_labels"])
        assert torch.equal(keep["matched_boxes"][i].cpu(), ref["matched_boxes"])
        assert torch.allclose(keep["ctr_target"][i].cpu(), ref["ctr_target"], rtol=2.4e-7, atol=0.0)
        assert int((ref["labels"] == 1).sum()) > 0
    # losses on the engine's own head outputs (level-major -> image-major for the oracle)
    dl, ct, off = [], [], 0
    for h, w in shapes:
        dl.append(keep["rpn_deltas"][off:off + n * h * w].cpu().view(n, h * w, 4))
        ct.append(keep["rpn_ctr"][off:off + n * h * w].cpu().view(n, h * w))
        off += n * h * w
    ref = O.rpn_losses(anchors, torch.cat(dl, 1), torch.cat(ct, 1), keep["labels"].cpu(), keep["obj_labels"].cpu(), keep["matched_boxes"].cpu(),
                       keep["ctr_target"].cpu())
    out = run["out"]
    assert float(out["loss_rpn_loc"]) == pytest.approx(float(ref["loss_rpn_loc"]), rel=1e-5)
    assert float(out["loss_rpn_ctr"]) == pytest.approx(float(ref["loss_rpn_ctr"]), rel=1e-5)
    assert [int(v) for v in out["rpn_anchor_counts"].cpu()] == [ref["num_pos"], ref["num_neg"], ref["obj_num_pos"], ref["obj_num_neg"]]


def test_roi_sampling_and_losses(run):
    keep, n, p, eng = run["keep"], run["n"], run["params"], run["eng"]
    sel, smp, cap, gmax = keep["sel"], keep["sampled"], run["cap"], run["gmax"]
    assert sel["cap"] == cap
    counts = [int(c) for c in sel["counts"].cpu()]
    rows = []
    for i in range(n):
        c, gc = counts[i], run["gcnt"][i]
        ki = torch.cat((run["keys"]["roi"][i, :c], run["keys"]["roi"][i, cap:cap + gc]))
        ref = O.roi_label_and_sample(sel["boxes"][i, :c].cpu(), sel["scores"][i, :c].cpu(), run["gt"][i, :gc], run["gcls"][i, :gc], ki)
        m = len(ref["sampled_idx"])
        assert smp["counts"][i].cpu().tolist() == [m, ref["num_fg"], ref["num_bg"]]
        assert torch.equal(smp["src"][i, :m].cpu().long(), ref["sampled_idx"])
        assert torch.equal(smp["gt_classes"][i, :m].cpu(), ref["gt_classes"]) and torch.equal(smp["ious"][i, :m].cpu(), ref["ious"])
        assert torch.equal(smp["boxes"][i, :m].cpu(), ref["boxes"]) and torch.equal(smp["gt_boxes"][i, :m].cpu(), ref["gt_boxes"])
        assert ref["num_fg"] >= gc  # every GT box is its own foreground candidate
        rows.append(m)
    bs = smp["boxes"].shape[1]
    valid = torch.cat([torch.arange(bs) < m for m in rows])
    # RoIAlign + box head on the sampled rows (fp16 operands, fp32 accumulate)
    feats = [nchw(keep["feats"][k]) for k in ("p2", "p3", "p4", "p5")]
    pooled_ref = O.roi_pooler_ref(feats, [smp["boxes"][i, :rows[i]].cpu() for i in range(n)], roi_align_fn=CO.roi_align)
    pe = keep["pooled"].cpu().float()[valid].permute(0, 3, 1, 2)
    assert float((pe - pooled_ref).abs().max()) < 2e-3 * max(1.0, float(pooled_ref.abs().max()))
    q16 = lambda t: t.half().float()  # noqa: E731
    h1 = q16(F.relu(F.linear(torch.flatten(pe, 1), q16(p["roi_heads.box_head.fc1.weight"]), p["roi_heads.box_head.fc1.bias"])))
    bf_ref = F.relu(F.linear(h1, q16(p["roi_heads.box_head.fc2.weight"]), p["roi_heads.box_head.fc2.bias"]))
    bf = keep["box_feats"].cpu()
    assert rel_err(bf[valid], bf_ref) < 5e-3
    # predictor, PLN encoder / decoder, classifier on the engine's box features (fp32 GEMMs)
    d_ref, iou_ref = O.box_predictor(bf[valid], p)
    pred = keep["pred"].cpu()
    assert rel_err(pred[valid, :4], d_ref) < 1e-4 and float((torch.sigmoid(pred[valid, 4]) - iou_ref.view(-1)).abs().max()) < 1e-5
    cls, ious = smp["gt_classes"].view(-1).cpu()[valid], smp["ious"].view(-1).cpu()[valid]
    c = eng.cfg
    lb, li = O.roi_box_losses(pred[valid, :4], torch.sigmoid(pred[valid, 4]), smp["boxes"].view(-1, 4).cpu()[valid],
                              smp["gt_boxes"].view(-1, 4).cpu()[valid], cls, ious, c["num_classes"], c["box_reg_weight"], c["iou_reg_weight"])
    out = run["out"]
    assert float(out["loss_box_reg"]) == pytest.approx(float(lb), rel=1e-5) and float(out["loss_iou"]) == pytest.approx(float(li), rel=1e-5)
    emb_ref, rec_ref, dml = O.pln_loss(bf[valid], cls, ious, p, c["pln_alpha"], c["pln_beta"], c["pln_loss_weight"], c["num_known"],
                                       c["pln_iou_threshold"])
    assert rel_err(keep["emb"].cpu()[valid], emb_ref) < 1e-4 and rel_err(keep["rec"].cpu()[valid], rec_ref) < 1e-4
    # loss kernels on the engine's own embeddings / logits
    _, _, dml_e = O.pln_loss(bf[valid], cls, ious, p, c["pln_alpha"], c["pln_beta"], c["pln_loss_weight"], c["num_known"], c["pln_iou_threshold"])
    assert float(out["loss_dml"]) == pytest.approx(float(dml_e), rel=1e-4)
    logits = keep["logits"].cpu()[valid]
    assert rel_err(logits, F.linear(keep["rec"].cpu()[valid], p["roi_heads.softmaxcls.cls_score.weight"], p["roi_heads.softmaxcls.cls_score.bias"])) < 1e-4
    ce = O.softmax_ce_loss(logits, cls, c["num_classes"], c["num_known"], c["cls_loss_weight"])
    assert float(out["loss_cls"]) == pytest.approx(float(ce), rel=1e-5)
    for k in ("loss_rpn_loc", "loss_rpn_ctr", "loss_box_reg", "loss_iou", "loss_dml", "loss_cls"):
        assert torch.isfinite(out[k]).all() and float(out[k]) > 0.0


def test_graspnet_id_map_losses(osr):
    """GraspNet configuration: known classes are a sorted subset of the dataset ids (class_map); PLN / classifier losses see
    id_map[gt_classes] (prototype_learning_network.py:146-147, softmax_classifier.py:224-229)."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    K, NC = 28, 88
    g = torch.Generator().manual_seed(5)
    class_map = torch.sort(torch.randperm(NC, generator=g)[:K])[0]
    params = random_params(0, num_known=K)
    eng = OpensetRCNNEngine(params, dict(num_classes=NC, num_known=K, unknown_id=1000, unk_thr=0.09), dtype=torch.float16, device=DEV, class_map=class_map)
    n, h, w, gmax = 1, 128, 160, 3
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    gt = torch.tensor([[[10.0, 20.0, 90.0, 100.0], [60.0, 30.0, 150.0, 120.0], [30.0, 60.0, 70.0, 110.0]]])
    gcls = class_map[torch.tensor([[0, 13, 27]])]
    shapes = O.level_shapes(h, w)
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = dict(rpn_reg=torch.rand(n, r, generator=g), rpn_obj=torch.rand(n, r, generator=g), roi=torch.rand(n, cap + gmax, generator=g))
    keep = {}
    out = eng.forward_losses(images.to(DEV), torch.tensor([(h, w)], dtype=torch.int32).to(DEV), h, w, gt.to(DEV), gcls.to(DEV),
                             torch.tensor([3], dtype=torch.int32).to(DEV), {k: v.to(DEV) for k, v in keys.items()}, keep=keep)
    smp = keep["sampled"]
    m = int(smp["counts"][0, 0])
    cls, ious = smp["gt_classes"].view(-1).cpu()[:m], smp["ious"].view(-1).cpu()[:m]
    assert int(((cls != NC)).sum()) > 0
    idm = torch.full((NC + 1,), -1, dtype=torch.int64)
    idm[class_map] = torch.arange(K)
    idm[NC] = K
    bf = keep["box_feats"].cpu()[:m]
    c = eng.cfg
    _, rec, dml = O.pln_loss(bf, idm[cls], ious, params, c["pln_alpha"], c["pln_beta"], c["pln_loss_weight"], K, c["pln_iou_threshold"])
    assert float(out["loss_dml"]) == pytest.approx(float(dml), rel=1e-4)
    logits = keep["logits"].cpu()[:m]
    ce = O.softmax_ce_loss(logits, idm[cls], K, K, c["cls_loss_weight"])  # after the remap the background id is K
    assert float(out["loss_cls"]) == pytest.approx(float(ce), rel=1e-5)
