"""Analytic known-answer tests that pin the CPU oracle (golden-vector list i-viii).

The reference ships no tests; these closed-form cases are what the oracle is anchored to.
"""
import math

import numpy as np
import pytest
import torch

from oracle import c_binding as C
from oracle import osr_oracle as O


# (i) anchors ---------------------------------------------------------------------------------
def test_anchor_grid_closed_form():
    shapes = O.level_shapes(800, 1344)
    assert shapes == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    assert sum(h * w for h, w in shapes) == 89523
    anchors = O.anchor_grid(shapes)
    for a, (h, w), s, z in zip(anchors, shapes, O.FPN_STRIDES, O.ANCHOR_SIZES):
        assert a.shape == (h * w, 4) and a.dtype == torch.float32
        # row-major: index i*w + j has centre (j*s, i*s)
        for (i, j) in [(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (h // 2, w // 3)]:
            exp = torch.tensor([j * s - z / 2, i * s - z / 2, j * s + z / 2, i * s + z / 2])
            assert torch.equal(a[i * w + j], exp)


def test_anchor_grid_three_ratios():
    # Base-RCNN-FPN.yaml:11 -> A=3, A innermost
    a = O.anchor_grid([(2, 2)], strides=(4,), sizes=(32,), ratios=(0.5, 1.0, 2.0))[0]
    assert a.shape == (12, 4)
    w05 = math.sqrt(32 * 32 / 0.5)
    np.testing.assert_allclose(a[0].numpy(), [-w05 / 2, -0.5 * w05 / 2, w05 / 2, 0.5 * w05 / 2], rtol=1e-6)
    np.testing.assert_allclose(a[1].numpy(), [-16, -16, 16, 16])
    np.testing.assert_allclose(a[4].numpy(), [4 - 16, -16, 4 + 16, 16])  # next cell, ratio 1


# (ii) ltrb transform -------------------------------------------------------------------------
def test_ltrb_round_trip_and_relu():
    g = torch.Generator().manual_seed(0)
    anchors = O.anchor_grid([(5, 7)], strides=(16,), sizes=(128,))[0]
    ctr = 0.5 * (anchors[:, :2] + anchors[:, 2:])
    ext = torch.rand(anchors.shape[0], 4, generator=g) * 200 + 1
    gt = torch.cat((ctr - ext[:, :2], ctr + ext[:, 2:]), dim=1)
    d = O.ltrb_get_deltas(anchors, gt)
    assert (d > 0).all()
    back = O.ltrb_apply_deltas(d, anchors)
    np.testing.assert_allclose(back.numpy(), gt.numpy(), rtol=1e-5, atol=1e-3)
    # negative deltas are clamped by the relu: box collapses onto the anchor centre
    z = O.ltrb_apply_deltas(-torch.ones(1, 4), anchors[:1])
    assert torch.equal(z[0], torch.cat((ctr[0], ctr[0])))
    # hand value: anchor [-16,-16,16,16] (size 32), delta (0.5,0.25,1,2) -> [-16,-8,32,64]
    a = torch.tensor([[-16.0, -16.0, 16.0, 16.0]])
    out = O.ltrb_apply_deltas(torch.tensor([[0.5, 0.25, 1.0, 2.0]]), a)
    assert torch.equal(out, torch.tensor([[-16.0, -8.0, 32.0, 64.0]]))


def test_b2b_round_trip_and_clamp():
    src = torch.tensor([[10.0, 20.0, 110.0, 220.0]])
    tgt = torch.tensor([[30.0, 10.0, 90.0, 250.0]])
    d = O.b2b_get_deltas(src, tgt)
    np.testing.assert_allclose(O.b2b_apply_deltas(d, src).numpy(), tgt.numpy(), rtol=1e-5)
    # dw clamp at log(1000/16): width multiplier is exactly 62.5
    big = torch.tensor([[0.0, 0.0, 1000.0, 0.0]])
    out = O.b2b_apply_deltas(big, src)
    np.testing.assert_allclose((out[0, 2] - out[0, 0]).item(), 62.5 * 100.0, rtol=1e-5)
    # zero deltas = identity
    np.testing.assert_allclose(O.b2b_apply_deltas(torch.zeros(1, 4), src).numpy(), src.numpy())


# (iii) centerness target ---------------------------------------------------------------------
def test_centerness_target():
    anchors = torch.tensor([[-16.0, -16.0, 16.0, 16.0], [84.0, 84.0, 116.0, 116.0], [0.0, 0.0, 32.0, 32.0]])
    gt = torch.tensor([[-50.0, -30.0, 50.0, 30.0]]).expand(3, 4)
    lab = torch.tensor([1, 1, 1])
    c = O.centerness_target(anchors, gt, lab)
    assert c[0].item() == pytest.approx(1.0)  # anchor centre == box centre
    assert c[1].item() == 0.0  # centre outside the box
    # centre (16,16): l=66 r=34 t=46 b=14 -> sqrt(34/66*14/46)
    assert c[2].item() == pytest.approx(math.sqrt(34 / 66 * 14 / 46), rel=1e-6)
    c0 = O.centerness_target(anchors, gt, torch.tensor([0, 1, 1]))
    assert c0[0].item() == 0.0  # forced to 0 where the objectness label is 0


# (iv) IoU identities -------------------------------------------------------------------------
def test_iou_identities():
    b = torch.tensor([[0.0, 0.0, 10.0, 10.0], [20.0, 20.0, 30.0, 40.0], [5.0, 0.0, 15.0, 10.0]])
    m = O.pairwise_iou(b, b)
    assert torch.equal(torch.diag(m), torch.ones(3))
    assert m[0, 1].item() == 0.0
    assert m[0, 2].item() == pytest.approx(50.0 / 150.0)
    assert torch.equal(O.elementwise_iou(b, b.roll(1, 0)), torch.diag(O.pairwise_iou(b, b.roll(1, 0))))


# (v) matcher ---------------------------------------------------------------------------------
def test_matcher_hand_matrix():
    q = torch.tensor([[0.80, 0.20, 0.25, 0.05, 0.50, 0.0],
                      [0.10, 0.25, 0.25, 0.02, 0.60, 0.0],
                      [0.00, 0.00, 0.10, 0.29, 0.10, 0.0]])
    m, lab = O.matcher(q, [0.3, 0.7], [0, -1, 1], low_quality=True)
    assert m.tolist()[:5] == [0, 1, 0, 2, 1]
    # col0 0.8 -> 1; col1 0.25 -> 0; col2 0.25 -> 0; col3 0.29 -> 0 but row-2 max (low quality) -> 1;
    # col4 0.6 -> -1 but it ties nothing: row1 max is 0.6 -> low quality -> 1; col5 0 -> 0
    assert lab.tolist() == [1, 0, 0, 1, 1, 0]
    m2, lab2 = O.matcher(q, [0.5], [0, 1], low_quality=False)
    assert lab2.tolist() == [1, 0, 0, 0, 1, 0]


# (vi) RoIAlign -------------------------------------------------------------------------------
def _ramp(n, c, h, w, ax, ay, a0):
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    f = torch.zeros(n, c, h, w)
    for i in range(n):
        for k in range(c):
            f[i, k] = a0 + (k + 1) * (ax * xs + ay * ys) + 10 * i
    return f


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_roi_align_constant_and_ramp(impl):
    fn = (lambda f, r, s: torch.from_numpy(O.roi_align_ref(f.numpy(), r.numpy(), s))) if impl == "numpy" else C.roi_align
    const = torch.full((2, 3, 20, 30), 2.5)
    rois = torch.tensor([[0, 8.0, 8.0, 72.0, 56.0], [1, 12.3, 9.1, 40.7, 33.3]])
    out = fn(const, rois, 0.25)
    np.testing.assert_allclose(out.numpy(), 2.5, rtol=1e-6)
    # bilinear interpolation of a linear ramp is exact, so the bin mean equals the ramp at the bin centre
    f = _ramp(2, 3, 20, 30, 0.5, -0.25, 1.0)
    out = fn(f, rois, 0.25)
    for r in range(2):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [(v * 0.25 - 0.5) for v in rois[r, 1:].tolist()]
        bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
        for k in range(3):
            for ph in range(7):
                for pw in range(7):
                    cx, cy = x1 + (pw + 0.5) * bw, y1 + (ph + 0.5) * bh
                    exp = 1.0 + (k + 1) * (0.5 * cx - 0.25 * cy) + 10 * b
                    assert out[r, k, ph, pw].item() == pytest.approx(exp, rel=2e-5, abs=2e-5)


def test_roi_align_border_rules_and_degenerate():
    f = _ramp(1, 2, 8, 8, 1.0, 1.0, 0.0)
    # RoI far outside the map: every sample has y>H or x>W -> zeros
    far = torch.tensor([[0, 400.0, 400.0, 500.0, 500.0]])
    assert torch.equal(C.roi_align(f, far, 0.25), torch.zeros(1, 2, 7, 7))
    # zero-size RoI (aligned=True: no min-size clamp): grid 0x0, count=max(0,1)=1 -> zeros
    deg = torch.tensor([[0, 10.0, 10.0, 10.0, 10.0]])
    assert torch.equal(C.roi_align(f, deg, 0.25), torch.zeros(1, 2, 7, 7))
    # RoI crossing the top-left border: samples in [-1,0] clamp to 0, samples < -1 contribute 0
    cross = torch.tensor([[0, -12.0, -12.0, 16.0, 16.0]])
    a = C.roi_align(f, cross, 0.25)
    b = torch.from_numpy(O.roi_align_ref(f.numpy(), cross.numpy(), 0.25))
    assert torch.equal(a, b)
    assert a[0, 0, 0, 0].item() == 0.0 and a[0, 0, 6, 6].item() > 0


def test_roi_align_c_matches_numpy_random():
    g = torch.Generator().manual_seed(3)
    f = torch.randn(2, 4, 25, 42, generator=g)
    xy = torch.rand(40, 2, generator=g) * torch.tensor([1300.0, 780.0])
    wh = torch.rand(40, 2, generator=g) * 600 + 1
    rois = torch.cat((torch.randint(0, 2, (40, 1), generator=g).float(), xy, xy + wh), dim=1)
    a = C.roi_align(f, rois, 1 / 32)
    b = torch.from_numpy(O.roi_align_ref(f.numpy(), rois.numpy(), 1 / 32))
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6)


def test_level_assignment_boundaries():
    def sq(s):
        return torch.tensor([[0.0, 0.0, float(s), float(s)]])
    assert O.assign_levels(sq(111.9)).item() == 0
    assert O.assign_levels(sq(112)).item() == 1  # sqrt(area)/224 = 0.5 -> level 3
    assert O.assign_levels(sq(223.9)).item() == 1
    assert O.assign_levels(sq(224)).item() == 2
    assert O.assign_levels(sq(447.9)).item() == 2
    assert O.assign_levels(sq(448)).item() == 3
    assert O.assign_levels(sq(5000)).item() == 3  # clamp at p5
    assert O.assign_levels(sq(1)).item() == 0  # clamp at p2
    assert O.assign_levels(sq(0)).item() == 0  # log2(1e-8) -> clamp


# (vii) NMS -----------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_nms_known_answers(impl):
    nms = O.nms_ref if impl == "numpy" else C.nms
    # identical boxes: only the first (highest score / lowest index) survives
    b = np.array([[0, 0, 10, 10]] * 3, dtype=np.float32)
    assert nms(b, np.array([0.5, 0.9, 0.9], dtype=np.float32), 0.5).tolist() == [1]
    # chain A-B-C: IoU(A,B)>thr, IoU(B,C)>thr, IoU(A,C)<thr -> keep A and C
    chain = np.array([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], dtype=np.float32)
    assert nms(chain, np.array([0.9, 0.8, 0.7], dtype=np.float32), 0.4).tolist() == [0, 2]
    # equal scores: stable order, lower index first
    far = np.array([[0, 0, 1, 1], [5, 5, 6, 6], [9, 9, 10, 10]], dtype=np.float32)
    assert nms(far, np.array([0.3, 0.3, 0.3], dtype=np.float32), 0.5).tolist() == [0, 1, 2]
    # thr = 1.0 never suppresses (iou > 1 impossible): pure descending sort
    assert nms(b, np.array([0.1, 0.7, 0.4], dtype=np.float32), 1.0).tolist() == [1, 2, 0]
    # IoU exactly at the threshold is NOT suppressed (strict >)
    half = np.array([[0, 0, 2, 1], [1, 0, 3, 1]], dtype=np.float32)  # inter 1, union 3
    assert nms(half, np.array([0.9, 0.8], dtype=np.float32), 1.0 / 3.0).tolist() == [0, 1]
    assert nms(half, np.array([0.9, 0.8], dtype=np.float32), 0.33).tolist() == [0]
    # zero-area boxes: 0/0 = nan, nan > thr is false -> both kept
    z = np.array([[1, 1, 1, 1], [1, 1, 1, 1]], dtype=np.float32)
    assert nms(z, np.array([0.9, 0.8], dtype=np.float32), 0.5).tolist() == [0, 1]


def test_batched_nms_per_class_and_c_vs_numpy():
    b = np.array([[0, 0, 10, 10], [1, 0, 11, 10], [0, 0, 10, 10], [50, 50, 60, 60]], dtype=np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.95], dtype=np.float32)
    c = np.array([0, 0, 1, 0], dtype=np.int64)
    assert O.batched_nms_ref(b, s, c, 0.5).tolist() == [3, 0, 2]
    assert C.batched_nms(b, s, c, 0.5).tolist() == [3, 0, 2]
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, 100, (300, 2)).astype(np.float32)
    wh = rng.uniform(5, 60, (300, 2)).astype(np.float32)
    bb = np.concatenate((xy, xy + wh), 1)
    ss = np.round(rng.uniform(0, 1, 300), 2).astype(np.float32)  # rounded -> many exact ties
    cc = rng.integers(0, 4, 300)
    assert O.batched_nms_ref(bb, ss, cc, 0.5).tolist() == C.batched_nms(bb, ss, cc, 0.5).tolist()
    assert O.nms_ref(bb, ss, 0.5).tolist() == C.nms(bb, ss, 0.5).tolist()


def test_stable_topk_tie_rule():
    s = torch.tensor([[0.5, 0.9, 0.5, 0.9, -0.0, 0.0, 0.1]])
    v, i = O.stable_topk(s, 6)
    assert i.tolist() == [[1, 3, 0, 2, 6, 4]]
    assert C.argsort_desc(s[0].numpy()).tolist() == [1, 3, 0, 2, 6, 4, 5]


# (viii) PLN ----------------------------------------------------------------------------------
def _pln_params(k=20, d=256, fdim=256):
    p = {
        "roi_heads.dml.encoder.weight": torch.eye(d, fdim),
        "roi_heads.dml.encoder.bias": torch.zeros(d),
        "roi_heads.dml.decoder.weight": 2.0 * torch.eye(fdim, d),
        "roi_heads.dml.decoder.bias": torch.ones(fdim),
        "roi_heads.dml.representatives": 3.0 * torch.eye(k, d),  # orthogonal, un-normalised
    }
    return p


def test_pln_inference_orthonormal_prototypes():
    p = _pln_params()
    f = torch.zeros(4, 256)
    f[0, 7] = 5.0  # exactly prototype 7 direction -> dist 0
    f[1, 3] = 1.0
    f[1, 4] = 1.0  # 45 degrees between 3 and 4 -> dist 1-1/sqrt2 = 0.2929 > 0.23 -> unknown
    f[2, 100] = 1.0  # orthogonal to all prototypes -> dist 1 -> unknown
    f[3, 5] = 1.0
    f[3, 6] = 0.6  # cos = 1/sqrt(1.36)=0.8575 -> dist .1425 -> known 5
    cls, rec, md, emb = O.pln_inference(f, p, unk_thr=0.23, unknown_id=80)
    assert cls.tolist() == [7, 80, 80, 5]
    np.testing.assert_allclose(md.numpy(), [0.0, 1 - 1 / math.sqrt(2), 1.0, 1 - 1 / math.sqrt(1.36)], atol=1e-6)
    # decoder acts on the UN-normalised embedding (prototype_learning_network.py:205)
    assert rec[0, 7].item() == pytest.approx(11.0)
    # ties (row 1 is equidistant from 3 and 4): argmin takes the lower class index
    cls2, _, _, _ = O.pln_inference(f, p, unk_thr=0.5, unknown_id=80)
    assert cls2[1].item() == 3
    # threshold is strict (>): dist == thr stays known
    thr = float(md[3])
    assert O.pln_inference(f, p, unk_thr=thr, unknown_id=80)[0][3].item() == 5


def test_pln_loss_terms():
    p = _pln_params()
    f = torch.zeros(3, 256)
    f[0, 2] = 1.0  # on its own prototype: intra 0, inter 1
    f[1, 2] = 1.0
    f[1, 9] = 1.0  # gt 2: intra = 1-1/sqrt2, inter (to 9) same
    f[2, 0] = 1.0  # background row, ignored
    gt = torch.tensor([2, 2, 80])
    iou = torch.tensor([0.9, 0.6, 0.0])
    _, _, loss = O.pln_loss(f, gt, iou, p, alpha=0.1, beta=0.9, loss_weight=0.5)
    d = 1 - 1 / math.sqrt(2)
    exp = (max(0 - 0.1, 0) + max(d - 0.1, 0)) + (max(0.9 - 1.0, 0) + max(0.9 - d, 0)) + 20 * max(1.0 - 1.0, 0)
    assert loss.item() == pytest.approx(exp * 0.5 / 3, rel=1e-5)
    # iou <= 0.5 rows are not foreground
    _, _, l2 = O.pln_loss(f, gt, torch.tensor([0.9, 0.5, 0.0]), p, alpha=0.1, beta=0.9, loss_weight=0.5)
    assert l2.item() == pytest.approx(0.0, abs=1e-7)


# end-to-end structural checks ------------------------------------------------------------------
def test_find_top_rpn_proposals_structure():
    g = torch.Generator().manual_seed(0)
    shapes = [(8, 10), (4, 5), (2, 3)]
    n = 2
    props, scores = [], []
    for h, w in shapes:
        xy = torch.rand(n, h * w, 2, generator=g) * 60
        props.append(torch.cat((xy, xy + torch.rand(n, h * w, 2, generator=g) * 30), dim=2))
        scores.append(torch.rand(n, h * w, generator=g))
    props[0][1, 5, 2] = float("nan")
    scores[0][1, 5] = 10.0  # would be selected first, then dropped as non-finite
    props[1][0, 3] = torch.tensor([70.0, 70.0, 90.0, 90.0])  # clipped to empty at (64, 64)
    scores[1][0, 3] = 5.0
    res = O.find_top_rpn_proposals(props, scores, [(64, 64), (64, 64)], pre_nms_topk=6)
    for i, (b, s, idx) in enumerate(res):
        assert len(b) == len(s) == len(idx)
        assert (b[:, 2] > b[:, 0]).all() and (b[:, 3] > b[:, 1]).all()
        assert b.min() >= 0 and b.max() <= 64
    assert len(res[0][0]) == 6 + 6 + 6 - 1 and len(res[1][0]) == 6 + 6 + 6 - 1
    # level-major, descending within level, never re-sorted across levels
    s0 = res[0][1]
    assert (s0[:6][:-1] >= s0[:6][1:]).all()
    with pytest.raises(FloatingPointError):
        O.find_top_rpn_proposals(props, scores, [(64, 64), (64, 64)], pre_nms_topk=6, training=True)


def test_fast_rcnn_inference_is_sort_topk():
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(50, 2, generator=g) * 50
    boxes = torch.cat((xy, xy + 5), dim=1)
    scores = torch.rand(50, 1, generator=g) * 0.2
    feats = torch.arange(50.0).view(50, 1)
    b, s, f, k = O.fast_rcnn_inference_single_image(boxes, scores, (100, 100), feats, 0.05, 1.0, 10)
    order = torch.sort(scores[:, 0], descending=True, stable=True)[1]
    order = order[scores[order, 0] > 0.05][:10]
    assert torch.equal(k, order) and torch.equal(f[:, 0], order.float())


# ------------------------------------------------------------------------------------------------------
# training targets / losses: hand-computed answers
# ------------------------------------------------------------------------------------------------------
def test_subsample_by_keys_kat():
    labels = torch.tensor([1, 1, 0, 0, -1, 1], dtype=torch.int8)
    keys = torch.tensor([0.5, 0.1, 0.9, 0.2, 0.0, 0.3])
    pos, neg = O.subsample_by_keys(labels, keys, 4, 0.5, 0)
    assert pos.tolist() == [1, 5] and neg.tolist() == [3, 2]
    pos, neg = O.subsample_by_keys(labels, keys, 6, 1.0, 0)  # the objectness sampler: positives first, negatives fill up
    assert pos.tolist() == [1, 5, 0] and neg.tolist() == [3, 2]
    pos, neg = O.subsample_by_keys(labels, torch.zeros(6), 2, 0.5, 0)  # all keys tie: lower index wins
    assert pos.tolist() == [0] and neg.tolist() == [2]


def test_rpn_label_and_sample_kat():
    anchors = torch.tensor([[0.0, 0, 10, 10], [5.0, 0, 15, 10], [20.0, 20, 30, 30]])
    gt = torch.tensor([[0.0, 0, 10, 10]])
    keys = torch.tensor([0.3, 0.2, 0.1])
    r = O.rpn_label_and_sample(anchors, gt, keys, keys)
    assert r["matched_iou"].tolist() == pytest.approx([1.0, 50.0 / 150.0, 0.0])
    assert r["labels_pre"].tolist() == [1, -1, 0]       # thresholds (0.3, 0.7)
    assert r["obj_labels_pre"].tolist() == [1, 1, 0]    # thresholds (0.1, 0.3)
    assert r["labels"].tolist() == [1, -1, 0] and r["obj_labels"].tolist() == [1, 1, 0]
    # anchor 0: centre (5,5) inside the GT, l=r=t=b -> 1; anchor 1: centre on the GT's right edge -> 0; anchor 2: label 0 -> 0
    assert r["ctr_target"].tolist() == pytest.approx([1.0, 0.0, 0.0])
    assert torch.equal(r["matched_boxes"], gt.expand(3, 4))
    # no GT: everything background, zero targets
    r0 = O.rpn_label_and_sample(anchors, torch.zeros(0, 4), keys, keys)
    assert r0["labels"].tolist() == [0, 0, 0] and float(r0["ctr_target"].abs().sum()) == 0.0


def test_rpn_losses_kat():
    anchors = torch.tensor([[0.0, 0, 10, 10], [20.0, 20, 30, 30]])
    # deltas (l,t,r,b)/size: anchor 0 -> box [0,0,10,10] (IoU 1 with its GT), anchor 1 -> [20,20,30,30] vs GT [20,20,30,40] (IoU .5)
    deltas = torch.tensor([[[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5]]])
    mboxes = torch.tensor([[[0.0, 0, 10, 10], [20.0, 20, 30, 40]]])
    out = O.rpn_losses(anchors, deltas, torch.tensor([[0.25, 0.5]]), torch.tensor([[1, 1]], dtype=torch.int8),
                       torch.tensor([[1, 0]], dtype=torch.int8), mboxes, torch.tensor([[1.0, 0.0]]), batch_size=2, w_loc=0.5, w_ctr=0.5)
    assert float(out["loss_rpn_loc"]) == pytest.approx((0.0 + 0.5) / 2 * 0.5)
    assert float(out["loss_rpn_ctr"]) == pytest.approx((0.75 + 0.5) / 2 * 0.5)
    assert (out["num_pos"], out["num_neg"], out["obj_num_pos"], out["obj_num_neg"]) == (2, 0, 1, 1)


def test_roi_label_and_sample_kat():
    props = torch.tensor([[0.0, 0, 10, 10], [50.0, 50, 60, 60], [0.0, 0, 10, 20]])
    gt, cls = torch.tensor([[0.0, 0, 10, 10]]), torch.tensor([7])
    r = O.roi_label_and_sample(props, torch.tensor([0.1, 0.2, 0.3]), gt, cls, torch.tensor([0.9, 0.5, 0.4, 0.1]), num_classes=81,
                               batch_size=4, pos_frac=0.25)
    # candidates: 3 proposals + the GT box; IoU 1, 0, .5, 1 -> fg {0, 2, 3}; one fg slot: the smallest key is the GT (0.1)
    assert r["sampled_idx"].tolist() == [3, 1] and r["gt_classes"].tolist() == [7, 81]
    assert r["ious"].tolist() == [1.0, 0.0] and r["num_fg"] == 1 and r["num_bg"] == 1
    assert float(r["logits"][0]) == pytest.approx(23.02585, rel=1e-5)  # log((1-1e-10)/1e-10)


def test_box_pln_ce_loss_kats():
    # proposal == gt -> zero deltas; prediction 0.1 everywhere -> L1 = 0.4 on the single foreground row of 2
    cls = torch.tensor([3, 81])
    b = torch.tensor([[0.0, 0, 10, 10], [5.0, 5, 9, 9]])
    lb, li = O.roi_box_losses(torch.full((2, 4), 0.1), torch.tensor([0.5, 0.9]), b, b, cls, torch.tensor([1.0, 0.0]))
    assert float(lb) == pytest.approx(0.4 / 2 * 0.5) and float(li) == pytest.approx(0.5 / 2 * 0.5)
    # uniform logits: CE = log(K+1)
    ce = O.softmax_ce_loss(torch.zeros(3, 21), torch.tensor([0, 81, 19]), 81, 20, 0.9)
    assert float(ce) == pytest.approx(0.9 * math.log(21.0), rel=1e-6)
    # PLN with orthonormal prototypes and an embedding equal to its own prototype: intra 0, inter 1, centre distances 1
    p = {"roi_heads.dml.encoder.weight": torch.eye(4), "roi_heads.dml.encoder.bias": torch.zeros(4),
         "roi_heads.dml.decoder.weight": torch.eye(4), "roi_heads.dml.decoder.bias": torch.zeros(4),
         "roi_heads.dml.representatives": torch.eye(4)[:3] * 2.0}
    feats = torch.tensor([[3.0, 0, 0, 0], [0.0, 1, 0, 0]])
    _, _, loss = O.pln_loss(feats, torch.tensor([0, 81]), torch.tensor([0.9, 0.9]), p, alpha=0.1, beta=1.5, loss_weight=2.0, num_known=3)
    # row 0: relu(0-.1)=0 + relu(1.5-1)=.5 ; prototypes: 3 * relu(1.6-1)=1.8 ; * 2 / 2 rows
    assert float(loss) == pytest.approx((0.5 + 1.8) * 2.0 / 2, rel=1e-6)


# (ix) branches of the inference tail the reference text fixes ---------------------------------
def test_pln_inference_single_and_zero_unknown_edges():
    """prototype_learning_network.py:218-222: `unknown = (min_dist > thr).nonzero().squeeze()` is a 0-d index when exactly one
    detection is unknown and an empty one when none is; `min_index[unknown] = 80` must then change exactly one / no entry."""
    p = _pln_params()
    f = torch.zeros(3, 256)
    f[0, 2] = 1.0
    f[1, 9] = 4.0
    f[2, 200] = 1.0  # orthogonal to every prototype: the ONLY unknown
    cls, _, md, _ = O.pln_inference(f, p, unk_thr=0.23, unknown_id=80)
    assert cls.tolist() == [2, 9, 80] and md.tolist() == pytest.approx([0.0, 0.0, 1.0], abs=1e-6)
    cls0, _, _, _ = O.pln_inference(f[:2], p, unk_thr=0.23, unknown_id=80)  # no unknown at all
    assert cls0.tolist() == [2, 9]
    cls1, _, _, _ = O.pln_inference(f[2:], p, unk_thr=0.23, unknown_id=80)  # a single detection, unknown
    assert cls1.tolist() == [80]
    # GraspNet form (:221-222): class_id maps the prototype index first, THEN the unknown entries are overwritten with 1000
    class_id = torch.arange(100, 120)
    cg, _, _, _ = O.pln_inference(f, p, unk_thr=0.23, unknown_id=1000, class_id=class_id)
    assert cg.tolist() == [102, 109, 1000]


def _cls_params(num_known=20, fdim=256):
    w = torch.zeros(num_known + 1, fdim)
    for k in range(num_known):
        w[k, k] = 10.0  # feature k lights up class k
    return {"roi_heads.softmaxcls.cls_score.weight": w, "roi_heads.softmaxcls.cls_score.bias": torch.zeros(num_known + 1)}


def test_softmax_classifier_all_known_and_all_unknown_branches():
    """softmax_classifier.py:317-344: with every detection known (`known.all()`) the result is the known leg alone -- no unknown
    call, no concatenation; with unknowns present the order is [unknown..., known...]; with NO known detection the known leg
    runs on an empty set and contributes nothing."""
    p = _cls_params()
    cfg = dict(O.VOC_COCO_CFG)
    boxes = torch.tensor([[10.0, 10.0, 50.0, 50.0], [60.0, 60.0, 90.0, 90.0], [12.0, 12.0, 52.0, 52.0]])
    scores = torch.tensor([0.9, 0.8, 0.7])
    rec = torch.zeros(3, 256)
    rec[0, 4] = 1.0
    rec[1, 7] = 1.0
    rec[2, 4] = 1.0
    # (a) all known: box 2 overlaps box 0 (IoU 0.82) in the same class 4 -> suppressed by the per-class NMS; classes 4 and 7 survive
    kb, ks, kc = O.softmax_classifier_inference(boxes, scores, torch.tensor([4, 7, 4]), rec, (100, 100), p, cfg)
    pk = float(torch.softmax(torch.tensor([10.0] + [0.0] * 20), 0)[0])
    assert sorted(kc.tolist()) == [4, 7] and len(kb) == 2
    assert ks.tolist() == pytest.approx([pk, pk], rel=1e-6)
    assert not (kc == 80).any()
    # (b) one unknown in the middle: it comes FIRST in the output, with its objectness score, then the known ones
    ob, os_, oc = O.softmax_classifier_inference(boxes, scores, torch.tensor([4, 80, 4]), rec, (100, 100), p, cfg)
    assert oc.tolist() == [80, 4] and os_.tolist() == pytest.approx([0.8, pk], rel=1e-6)
    assert ob[0].tolist() == [60.0, 60.0, 90.0, 90.0]
    # (c) nothing known: only the class-agnostic leg; boxes 0 and 2 overlap -> the lower-scored one is suppressed
    ub, us, uc = O.softmax_classifier_inference(boxes, scores, torch.tensor([80, 80, 80]), rec, (100, 100), p, cfg)
    assert uc.tolist() == [80, 80] and us.tolist() == pytest.approx([0.9, 0.8])
