"""Golden-vector tests on tests/golden/osr_golden_v1.npz (oracle-generated, see tests/golden/make_golden.py).

CPU half (`-m "not gpu"`): the oracle still reproduces the stored answers bit for bit (pins the oracle against drift).
GPU half (`-m gpu`): the HIP library reproduces the stored answers through the C ABI, without any oracle call."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_binding as CO
from oracle import osr_oracle as O

SHAPES, STRIDES, SIZES = [(12, 20), (6, 10), (3, 5)], (4, 8, 16), (32, 64, 128)
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "osr_golden_v1.npz")))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _split_levels(flat, n, last=()):
    out, off = [], 0
    for h, w in SHAPES:
        out.append(t(flat[off:off + n * h * w]).view(n, h * w, *last))
        off += n * h * w
    return out


# ------------------------------------------------------------------ CPU: oracle == golden
def test_oracle_matches_golden_selection(gold):
    n = 2
    anchors = O.anchor_grid(SHAPES, STRIDES, SIZES)
    ctr, deltas = _split_levels(gold["sel_ctr"], n), _split_levels(gold["sel_deltas"], n, (4,))
    props = [O.ltrb_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)).view(n, -1, 4) for d, a in zip(deltas, anchors)]
    ref = O.find_top_rpn_proposals(props, ctr, [tuple(s) for s in gold["sel_sizes"].tolist()], int(gold["sel_topk"]))
    for i, (b, s, idx) in enumerate(ref):
        assert np.array_equal(b.numpy(), gold[f"sel_boxes{i}"]) and np.array_equal(s.numpy(), gold[f"sel_scores{i}"])
        assert np.array_equal(idx.numpy(), gold[f"sel_src{i}"])


def test_oracle_matches_golden_roi_align_nms_pln(gold):
    boxes, bidx = t(gold["ra_boxes"]), t(gold["ra_bidx"])
    lv = O.assign_levels(boxes)
    assert np.array_equal(lv.numpy().astype(np.int32), gold["ra_levels"])
    for l, s in enumerate((0.25, 0.125, 0.0625, 0.03125)):
        ids = torch.nonzero(lv == l).squeeze(1)
        rois = torch.cat((bidx[ids].float().unsqueeze(1), boxes[ids]), dim=1)
        out = CO.roi_align(t(gold[f"ra_feat{l}"]), rois, s)
        assert np.allclose(out.numpy(), gold["ra_out"][ids.numpy()], rtol=1e-6, atol=1e-6)
        py = O.roi_align_ref(gold[f"ra_feat{l}"], rois.numpy(), s)  # the pure-numpy restatement agrees with the C one
        assert np.allclose(np.asarray(py), gold["ra_out"][ids.numpy()], rtol=1e-5, atol=1e-5)
    b, s, c = gold["nms_boxes"], gold["nms_scores"], gold["nms_cls"]
    assert np.array_equal(CO.nms(b, s, 0.5), gold["nms_keep_agnostic"]) and np.array_equal(O.nms_ref(b, s, 0.5), gold["nms_keep_agnostic"])
    assert np.array_equal(CO.batched_nms(b, s, c, 0.5), gold["nms_keep_per_class"])
    assert np.array_equal(O.batched_nms_ref(b, s, c, 0.5), gold["nms_keep_per_class"])
    assert np.array_equal(CO.nms(b, s, 1.0), gold["nms_sort_only"])
    rep = F.normalize(t(gold["pln_protos"]))
    dist = 1.0 - F.normalize(t(gold["pln_emb"])) @ rep.t()
    md, cls = dist.min(dim=1)
    cls = torch.where(md > 0.23, torch.full_like(cls, 80), cls)
    assert np.array_equal(cls.numpy(), gold["pln_class"]) and np.allclose(md.numpy(), gold["pln_min_dist"], atol=1e-6)


def test_oracle_matches_golden_training_targets(gold):
    anchors = torch.cat(O.anchor_grid(SHAPES, STRIDES, SIZES))
    gt, gcnt, gcls = t(gold["tr_gt"]), gold["tr_gt_count"], t(gold["tr_gt_classes"])
    for i in range(2):
        ref = O.rpn_label_and_sample(anchors, gt[i, :gcnt[i]], t(gold["tr_keys_reg"])[i], t(gold["tr_keys_obj"])[i], batch_size=32)
        for k in ("matched_idx", "matched_iou", "labels_pre", "obj_labels_pre", "labels", "obj_labels", "matched_boxes"):
            assert np.array_equal(ref[k].numpy(), gold[f"tr_{k}{i}"]), k
        assert np.allclose(ref["ctr_target"].numpy(), gold[f"tr_ctr_target{i}"], rtol=2.4e-7, atol=0)
        p, pcap = int(gold["tr_prop_count"][i]), gold["tr_prop_boxes"].shape[1]
        ki = torch.cat((t(gold["tr_roi_keys"])[i, :p], t(gold["tr_roi_keys"])[i, pcap:pcap + gcnt[i]]))
        r = O.roi_label_and_sample(t(gold["tr_prop_boxes"])[i, :p], t(gold["tr_prop_logits"])[i, :p], gt[i, :gcnt[i]], gcls[i, :gcnt[i]], ki, batch_size=16)
        assert np.array_equal(r["sampled_idx"].numpy(), gold[f"tr_roi_src{i}"]) and np.array_equal(r["gt_classes"].numpy(), gold[f"tr_roi_cls{i}"])
        assert np.array_equal(r["ious"].numpy(), gold[f"tr_roi_iou{i}"])


# ------------------------------------------------------------------ GPU: HIP == golden (no oracle in the loop)
@pytest.fixture(scope="module")
def ops(osr):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    osr._lib.load()
    return osr.ops


@pytest.mark.gpu
def test_hip_matches_golden_selection(ops, gold):
    n = 2
    lv = ops.make_rpn_levels(SHAPES, STRIDES, n, 1)
    cell = torch.tensor([[[-s / 2, -s / 2, s / 2, s / 2]] for s in SIZES], dtype=torch.float32).to(DEV)
    r = ops.rpn_select(lv, cell, t(gold["sel_ctr"]).to(DEV), t(gold["sel_deltas"]).to(DEV), n, t(gold["sel_sizes"]).to(DEV), int(gold["sel_topk"]))
    for i in range(n):
        c = int(r["counts"][i])
        assert c == len(gold[f"sel_src{i}"])
        assert np.array_equal(r["src_index"][i, :c].cpu().numpy().astype(np.int64), gold[f"sel_src{i}"])
        assert np.array_equal(r["boxes"][i, :c].cpu().numpy(), gold[f"sel_boxes{i}"]) and np.array_equal(r["scores"][i, :c].cpu().numpy(), gold[f"sel_scores{i}"])


@pytest.mark.gpu
def test_hip_matches_golden_roi_align_nms_pln(ops, gold):
    feats = [t(gold[f"ra_feat{l}"]).permute(0, 2, 3, 1).contiguous().to(DEV) for l in range(4)]
    out = ops.roi_align(feats, (0.25, 0.125, 0.0625, 0.03125), t(gold["ra_boxes"]).to(DEV), t(gold["ra_bidx"]).to(DEV), 7, torch.float32)
    assert np.allclose(out.cpu().permute(0, 3, 1, 2).numpy(), gold["ra_out"], rtol=1e-4, atol=1e-5)
    b, s, c = t(gold["nms_boxes"]).to(DEV), t(gold["nms_scores"]).to(DEV), t(gold["nms_cls"]).to(DEV)
    n = b.shape[0]
    ln = torch.tensor([n], dtype=torch.int32).to(DEV)
    for cls, thr, key in ((None, 0.5, "nms_keep_agnostic"), (c, 0.5, "nms_keep_per_class"), (None, 1.0, "nms_sort_only")):
        keep, cnt = ops.nms_topk(b, s, cls, None, 1, n, ln, thr, n)
        assert np.array_equal(keep[0, :int(cnt[0])].cpu().numpy(), gold[key]), key
    protos = ops.l2_normalize_rows(t(gold["pln_protos"]).to(DEV))
    pc, md = ops.pln_tail(t(gold["pln_emb"]).to(DEV), protos, 20, 1, 0.23, 80)
    assert np.array_equal(pc.cpu().numpy(), gold["pln_class"]) and np.allclose(md.cpu().numpy(), gold["pln_min_dist"], atol=1e-5)


@pytest.mark.gpu
def test_hip_matches_golden_training_targets(ops, gold):
    n = 2
    lv = ops.make_rpn_levels(SHAPES, STRIDES, n, 1)
    cell = torch.tensor([[[-s / 2, -s / 2, s / 2, s / 2]] for s in SIZES], dtype=torch.float32).to(DEV)
    gt, gcnt = t(gold["tr_gt"]).to(DEV), t(gold["tr_gt_count"]).to(DEV)
    midx, miou, lr, lo = ops.rpn_match_anchors(lv, cell, n, gt, gcnt)
    for i in range(n):
        assert np.array_equal(midx[i].cpu().numpy().astype(np.int64), gold[f"tr_matched_idx{i}"]) and np.array_equal(miou[i].cpu().numpy(), gold[f"tr_matched_iou{i}"])
        assert np.array_equal(lr[i].cpu().numpy(), gold[f"tr_labels_pre{i}"]) and np.array_equal(lo[i].cpu().numpy(), gold[f"tr_obj_labels_pre{i}"])
    ops.subsample_labels_(lr, t(gold["tr_keys_reg"]).to(DEV), 32, 0.5)
    ops.subsample_labels_(lo, t(gold["tr_keys_obj"]).to(DEV), 32, 1.0)
    mb, ct = ops.rpn_anchor_targets(lv, cell, n, gt, gcnt, midx, lo)
    for i in range(n):
        assert np.array_equal(lr[i].cpu().numpy(), gold[f"tr_labels{i}"]) and np.array_equal(lo[i].cpu().numpy(), gold[f"tr_obj_labels{i}"])
        assert np.array_equal(mb[i].cpu().numpy(), gold[f"tr_matched_boxes{i}"])
        assert np.allclose(ct[i].cpu().numpy(), gold[f"tr_ctr_target{i}"], rtol=2.4e-7, atol=0)
    o = ops.roi_match_and_sample(t(gold["tr_prop_boxes"]).to(DEV), t(gold["tr_prop_logits"]).to(DEV), t(gold["tr_prop_count"]).to(DEV), gt,
                                 t(gold["tr_gt_classes"]).to(DEV), gcnt, t(gold["tr_roi_keys"]).to(DEV), 81, 16, 0.25, 0.5)
    for i in range(n):
        m = len(gold[f"tr_roi_src{i}"])
        assert int(o["counts"][i, 0]) == m
        assert np.array_equal(o["src"][i, :m].cpu().numpy().astype(np.int64), gold[f"tr_roi_src{i}"])
        assert np.array_equal(o["gt_classes"][i, :m].cpu().numpy(), gold[f"tr_roi_cls{i}"]) and np.array_equal(o["ious"][i, :m].cpu().numpy(), gold[f"tr_roi_iou{i}"])
