"""Detection-level agreement between two runs of the detector on the same images: the measure tests/test_e2e_parity.py and
bench.py's `parity` object report (how many of the reference run's final detections the other run reproduces: same class,
IoU >= iou_thr, |score difference| <= score_tol, matched one to one in the reference's order)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


def pairwise_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(len(a), len(b)) IoU of xyxy boxes ([d2] pairwise_iou: empty intersections give 0)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    return torch.where(inter > 0, inter / union, torch.zeros_like(inter))


def match_detections(got: Tuple[torch.Tensor, torch.Tensor, torch.Tensor], ref: Tuple[torch.Tensor, torch.Tensor, torch.Tensor],
                     iou_thr: float = 0.99, score_tol: float = 1e-2):
    """Greedy one-to-one matching in the reference's order. got / ref: (boxes (k,4), scores (k), classes (k)) on the CPU.
    Returns a list of (ref index, got index) pairs; a same-class candidate is preferred."""
    gb, gs, gc = got
    rb, rs, rc = ref
    pairs = []
    if len(rb) == 0 or len(gb) == 0:
        return pairs
    iou = pairwise_iou(rb.float(), gb.float())
    used = torch.zeros(len(gb), dtype=torch.bool)
    for i in range(len(rb)):
        ok = (iou[i] >= iou_thr) & ((gs - rs[i]).abs() <= score_tol) & ~used
        if bool(ok.any()):
            cand = torch.nonzero(ok).squeeze(1)
            pref = cand[gc[cand] == rc[i]]
            j = int(pref[0]) if len(pref) else int(cand[0])
            used[j] = True
            pairs.append((i, j))
    return pairs


def detection_agreement(got_list: Sequence, ref_list: Sequence, iou_thr: float = 0.99, score_tol: float = 1e-2) -> dict:
    """Over a batch: matched / max(#ref, #got), the class-consistent matches, and the largest box / score deviation over the
    matched pairs (box deviation in pixels and relative to the larger image-side extent of the reference box set)."""
    matched = same_cls = n_ref = n_got = 0
    max_box = max_score = 0.0
    for got, ref in zip(got_list, ref_list):
        pairs = match_detections(got, ref, iou_thr, score_tol)
        matched += len(pairs)
        n_ref += len(ref[0])
        n_got += len(got[0])
        for i, j in pairs:
            same_cls += int(got[2][j] == ref[2][i])
            max_box = max(max_box, float((got[0][j].float() - ref[0][i].float()).abs().max()))
            max_score = max(max_score, float((got[1][j].float() - ref[1][i].float()).abs()))
    denom = max(n_ref, n_got, 1)
    return dict(matched=matched, same_class=same_cls, reference_detections=n_ref, returned_detections=n_got,
                fraction=matched / denom, max_box_abs_diff_px=max_box, max_score_abs_diff=max_score)
