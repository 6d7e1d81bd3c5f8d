"""Generator of tests/golden/osr_golden_v1.npz: small seeded input/output vectors for the hot path.

Provenance: ORACLE-generated (oracle/osr_oracle.py + oracle/osr_oracle_c.c), not reference-generated -- the reference
cannot be imported in the build container (detectron2 / fvcore / torchvision are absent), and it holds
no fixtures of its own. The file freezes the oracle's answers at the commit that introduced it, so that
  * tests/test_golden.py (CPU) notices any later drift of the oracle, and
  * tests/test_golden.py (GPU) checks the HIP library against stored numbers without running the oracle.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_binding as CO  # noqa: E402
from oracle import osr_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "osr_golden_v1.npz")
SHAPES, STRIDES, SIZES = [(12, 20), (6, 10), (3, 5)], (4, 8, 16), (32, 64, 128)


def g(seed):
    return torch.Generator().manual_seed(seed)


def rpn_select_case():
    gg, n, topk = g(101), 2, 24
    anchors = O.anchor_grid(SHAPES, STRIDES, SIZES)
    ctr = [((torch.rand(n, h * w, generator=gg) * 16).round() / 16) for h, w in SHAPES]  # ties
    deltas = [torch.randn(n, h * w, 4, generator=gg) * 1.2 for h, w in SHAPES]
    sizes = [(48, 80), (44, 70)]
    props = [O.ltrb_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)).view(n, -1, 4) for d, a in zip(deltas, anchors)]
    ref = O.find_top_rpn_proposals(props, ctr, sizes, topk)
    out = dict(sel_ctr=torch.cat([c.reshape(-1) for c in ctr]).numpy(), sel_deltas=torch.cat([d.reshape(-1, 4) for d in deltas]).numpy(),
               sel_sizes=np.array(sizes, np.int32), sel_topk=np.int32(topk))
    for i, (b, s, idx) in enumerate(ref):
        out[f"sel_boxes{i}"], out[f"sel_scores{i}"], out[f"sel_src{i}"] = b.numpy(), s.numpy(), idx.numpy()
    return out


def roi_align_case():
    gg = g(102)
    feats = [torch.randn(2, 8, 32 // s * 4, 48 // s * 4, generator=gg) for s in (4, 8, 16, 32)]  # 128x192 image
    m = 24
    ctr = torch.rand(m, 2, generator=gg) * torch.tensor([192.0, 128.0])
    size = torch.exp(torch.rand(m, 2, generator=gg) * 5.5)
    boxes = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
    boxes[0] = torch.tensor([-20.0, -10.0, 40.0, 30.0])
    boxes[1] = torch.tensor([5.0, 5.0, 5.0, 5.0])
    bidx = torch.randint(0, 2, (m,), generator=gg, dtype=torch.int32)
    lv = O.assign_levels(boxes)
    ref = torch.zeros(m, 8, 7, 7)
    for l, s in enumerate((0.25, 0.125, 0.0625, 0.03125)):
        ids = torch.nonzero(lv == l).squeeze(1)
        rois = torch.cat((bidx[ids].float().unsqueeze(1), boxes[ids]), dim=1)
        ref[ids] = CO.roi_align(feats[l], rois, s)
    out = dict(ra_boxes=boxes.numpy(), ra_bidx=bidx.numpy(), ra_levels=lv.numpy().astype(np.int32), ra_out=ref.numpy())
    for l, f in enumerate(feats):
        out[f"ra_feat{l}"] = f.numpy()
    return out


def nms_case():
    gg, n = g(103), 60
    xy = torch.rand(n, 2, generator=gg) * 40
    wh = torch.rand(n, 2, generator=gg) * 30 + 4
    boxes = torch.cat((xy, xy + wh), dim=1).numpy()
    scores = ((torch.rand(n, generator=gg) * 20).round() / 20).numpy()  # ties
    cls = torch.randint(0, 3, (n,), generator=gg).numpy().astype(np.int32)
    return dict(nms_boxes=boxes, nms_scores=scores, nms_cls=cls, nms_keep_agnostic=CO.nms(boxes, scores, 0.5).astype(np.int32),
                nms_keep_per_class=CO.batched_nms(boxes, scores, cls, 0.5).astype(np.int32),
                nms_sort_only=CO.nms(boxes, scores, 1.0).astype(np.int32))


def pln_case():
    gg = g(104)
    p = O.make_head_params(seed=5, num_known=20)
    feats = torch.randn(40, 1024, generator=gg)
    rep = torch.nn.functional.normalize(p["roi_heads.dml.representatives"])
    # half of the rows are pushed towards a prototype so that both known and unknown outcomes occur
    emb_target = rep[torch.randint(0, 20, (20,), generator=gg)] * 4.0
    w, b = p["roi_heads.dml.encoder.weight"], p["roi_heads.dml.encoder.bias"]
    feats[:20] = torch.linalg.lstsq(w, (emb_target - b).t()).solution.t()
    cls, rec, md, emb = O.pln_inference(feats, p, 0.23, 80, 20)
    # the encoder GEMM is covered elsewhere; the fixture starts at the embedding (osr_pln_tail's input) to stay small
    return dict(pln_protos=p["roi_heads.dml.representatives"].numpy(), pln_emb=emb.numpy(), pln_class=cls.numpy(), pln_min_dist=md.numpy())


def train_case():
    gg, n = g(105), 2
    anchors = torch.cat(O.anchor_grid(SHAPES, STRIDES, SIZES))
    r = anchors.shape[0]
    gt = torch.tensor([[[4.0, 6.0, 40.0, 44.0], [30.0, 10.0, 78.0, 40.0], [50.0, 20.0, 60.0, 30.0]], [[10.0, 8.0, 70.0, 46.0], [0, 0, 0, 0], [0, 0, 0, 0]]])
    gcnt = [3, 1]
    gcls = torch.tensor([[3, 7, 11], [5, 0, 0]])
    kr, ko = torch.rand(n, r, generator=gg), torch.rand(n, r, generator=gg)
    out = dict(tr_gt=gt.numpy(), tr_gt_count=np.array(gcnt, np.int32), tr_gt_classes=gcls.numpy(), tr_keys_reg=kr.numpy(), tr_keys_obj=ko.numpy())
    for i in range(n):
        ref = O.rpn_label_and_sample(anchors, gt[i, :gcnt[i]], kr[i], ko[i], batch_size=32)
        for k in ("matched_idx", "matched_iou", "labels_pre", "obj_labels_pre", "labels", "obj_labels", "matched_boxes", "ctr_target"):
            out[f"tr_{k}{i}"] = ref[k].numpy()
    pcap = 40
    pb = torch.zeros(n, pcap, 4)
    pl = torch.randn(n, pcap, generator=gg)
    keys = torch.rand(n, pcap + 3, generator=gg)
    pcnt = [40, 25]
    for i in range(n):
        src = gt[i, torch.randint(0, gcnt[i], (pcap,), generator=gg)]
        wh = (src[:, 2:] - src[:, :2]).repeat(1, 2)
        pb[i] = src + (torch.rand(pcap, 4, generator=gg) - 0.5) * 0.9 * wh
        ki = torch.cat((keys[i, :pcnt[i]], keys[i, pcap:pcap + gcnt[i]]))
        ref = O.roi_label_and_sample(pb[i, :pcnt[i]], pl[i, :pcnt[i]], gt[i, :gcnt[i]], gcls[i, :gcnt[i]], ki, batch_size=16)
        out[f"tr_roi_src{i}"], out[f"tr_roi_cls{i}"], out[f"tr_roi_iou{i}"] = ref["sampled_idx"].numpy(), ref["gt_classes"].numpy(), ref["ious"].numpy()
    out.update(tr_prop_boxes=pb.numpy(), tr_prop_logits=pl.numpy(), tr_prop_count=np.array(pcnt, np.int32), tr_roi_keys=keys.numpy())
    return out


def main():
    data = {}
    for fn in (rpn_select_case, roi_align_case, nms_case, pln_case, train_case):
        data.update(fn())
    np.savez_compressed(OUT, **data)
    print(OUT, os.path.getsize(OUT), "bytes,", len(data), "arrays")


if __name__ == "__main__":
    main()
