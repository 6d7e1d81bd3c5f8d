"""Whole-network parity on the GPU: the HIP engine against the fp32 oracle (O.detector_inference, no quantisation anywhere) on
seeded images, final detections compared one by one.

  * PARITY MODE (engine dtype float32: fp32 storage and fp32 products in every layer, osr_conv_f32.hip): the arithmetic the
    reference runs in. Asserted: >= 95 % of the final <= 100 detections per image agree with the oracle (same class, IoU >= 0.99,
    |score difference| <= 1e-2), and the dense stages agree at 1e-4 on the engine's own inputs (BASELINE.json north_star:
    "fp32 box/score/embedding within 1e-4").
  * FAST MODE (fp16 storage, the benchmark path): the same comparison is REPORTED and asserted at the level fp16 storage of ~50
    layers allows; it is the available proxy for "mAP_k within 0.1" without a dataset or checkpoint.

Weights: weights.random_params (seed 0, score-spreading head scales) + weights.with_known_unknown_mix, so that both kinds of
detections (known classes through the softmax / per-class NMS leg, unknown through the class-agnostic leg) are present."""
import pytest
import torch
import torch.nn.functional as F

from oracle import c_binding as CO
from oracle import osr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, H, W = 4, 256, 384


def _iou(a, b):
    return O.pairwise_iou(a, b)


def agreement(got, ref, iou_thr=0.99, score_tol=1e-2):
    """Greedy one-to-one matching in the oracle's order. Returns (matched, class-consistent matches, len(got), len(ref))."""
    gb, gs, gc = got
    rb, rs, rc = ref
    if len(rb) == 0 or len(gb) == 0:
        return 0, 0, len(gb), len(rb)
    iou = _iou(rb, gb)
    used = torch.zeros(len(gb), dtype=torch.bool)
    matched = same_cls = 0
    for i in range(len(rb)):
        ok = (iou[i] >= iou_thr) & ((gs - rs[i]).abs() <= score_tol) & ~used
        if bool(ok.any()):
            cand = torch.nonzero(ok).squeeze(1)
            pref = cand[gc[cand] == rc[i]]
            j = int(pref[0]) if len(pref) else int(cand[0])
            used[j] = True
            matched += 1
            same_cls += int(gc[j] == rc[i])
    return matched, same_cls, len(gb), len(rb)


@pytest.fixture(scope="module")
def world(osr):
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params, with_known_unknown_mix
    g = torch.Generator().manual_seed(2024)
    images = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    sizes = [(H, W), (H, W), (H - 16, W - 40), (H - 6, W)]
    base = random_params(0)
    keep = {}
    eng = OpensetRCNNEngine(base, dtype=torch.float32, device=DEV)
    eng.forward(images.to(DEV), sizes, keep=keep)
    cnt = keep["cnt1"].cpu()
    emb = torch.cat([keep["emb"].view(N, 1000, -1)[i, :int(cnt[i])] for i in range(N)])
    params = with_known_unknown_mix(base, emb)
    with torch.no_grad():  # O.detector_inference with the true image sizes the engine gets (it takes them from the tensors otherwise)
        batch, _ = O.preprocess_images([im for im in images])
        feats = O.resnet_fpn_forward(batch, params)
        props, _ = O.rpn_inference(feats, sizes, params, 1000)
        ref, _ = O.roi_heads_inference(feats, [(b, s) for b, s, _ in props], sizes, params, roi_align_fn=CO.roi_align)
    return dict(images=images, sizes=sizes, params=params, ref=ref, Engine=OpensetRCNNEngine)


def _run(world, dtype, keep=None):
    eng = world["Engine"](world["params"], dtype=dtype, device=DEV)
    out = eng.forward(world["images"].to(DEV), world["sizes"], keep=keep)
    torch.cuda.synchronize()
    insts = eng.to_instances(out, N)
    return eng, [(d["pred_boxes"], d["scores"], d["pred_classes"]) for d in insts]


def _report(tag, dets, ref):
    tot = [0, 0, 0, 0]
    lines = []
    for i, (g, r) in enumerate(zip(dets, ref)):
        m, c, ng, nr = agreement(g, r)
        lines.append(f"  image {i}: {m}/{nr} oracle detections matched ({c} with the same class), engine returned {ng}; "
                     f"known {int((r[2] != 80).sum())} / unknown {int((r[2] == 80).sum())} in the oracle")
        for k, v in enumerate((m, c, ng, nr)):
            tot[k] += v
    frac = tot[0] / max(tot[3], tot[2], 1)
    print(f"\n[{tag}] detection agreement {tot[0]}/{max(tot[3], tot[2])} = {frac:.3f} (IoU >= 0.99, |dscore| <= 1e-2)\n" + "\n".join(lines))
    return frac, tot


def test_parity_mode_detections_match_the_fp32_oracle(world):
    ref = world["ref"]
    assert sum(int((r[2] != 80).sum()) for r in ref) > 20 and sum(int((r[2] == 80).sum()) for r in ref) > 20, "weights must give both kinds"
    keep = {}
    eng, dets = _run(world, torch.float32, keep)
    frac, tot = _report("parity mode, fp32", dets, ref)
    assert frac >= 0.95, frac
    assert tot[1] == tot[0]  # class ids equal on every match
    world["keep32"] = keep
    # north_star: "fp32 box/score/embedding within 1e-4" -- END TO END, on the matched detections: box corners within 1e-4 of the
    # image extent (relative) and scores within 1e-4 absolute, not just IoU >= 0.99 / 1e-2
    from openset_rcnn_amd.host.agreement import detection_agreement
    ag = detection_agreement(dets, ref)
    extent = float(max(H, W))
    print(f"[parity mode] matched {ag['matched']}: max |box diff| {ag['max_box_abs_diff_px']:.3e} px "
          f"({ag['max_box_abs_diff_px'] / extent:.2e} of the image extent), max |score diff| {ag['max_score_abs_diff']:.3e}")
    assert ag["matched"] == tot[0]
    assert ag["max_box_abs_diff_px"] / extent <= 1e-4, ag
    assert ag["max_score_abs_diff"] <= 1e-4, ag


def test_parity_mode_dense_stages_within_1e_4(world):
    """Every dense stage of the parity mode against the oracle ON THE ENGINE'S OWN INPUTS to that stage, at 1e-4."""
    keep = world.get("keep32")
    if keep is None:
        keep = {}
        _run(world, torch.float32, keep)
    p, sizes = world["params"], world["sizes"]
    nchw = lambda t: t.detach().cpu().float().permute(0, 3, 1, 2).contiguous()  # noqa: E731
    rel = lambda a, b: float((a.detach().cpu().float() - b).abs().max() / b.abs().max().clamp(min=1e-6))  # noqa: E731
    # backbone + FPN: ~50 fp32 layers, oracle = torch CPU convolutions (another summation order)
    batch, _ = O.preprocess_images([im for im in world["images"]])
    feats = O.resnet_fpn_forward(batch, p)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        assert rel(nchw(keep["feats"][k]), feats[k]) < 1e-4, k
    # CF-RPN head on the engine's pyramid
    ef = {k: nchw(v) for k, v in keep["feats"].items()}
    ds, cs = [], []
    for k in ("p2", "p3", "p4", "p5", "p6"):
        d, c = O.cfrpn_head(ef[k], p)
        ds.append(d)
        cs.append(c)
    ds, cs = O.flatten_head_outputs(ds, cs)
    assert rel(keep["rpn_deltas"], torch.cat([d.reshape(-1, 4) for d in ds])) < 1e-4
    assert float((keep["rpn_ctr"].cpu() - torch.cat([c.reshape(-1) for c in cs])).abs().max()) < 1e-4
    # RoIAlign (fp32 out) -> FC1 -> FC2 on the engine's pyramid and proposals
    sel = keep["sel"]
    cap = sel["cap"]
    counts = [int(c) for c in sel["counts"].cpu()]
    boxes = [sel["boxes"][i, :counts[i]].cpu() for i in range(N)]
    pooled_ref = O.roi_pooler_ref([ef[k] for k in ("p2", "p3", "p4", "p5")], boxes, roi_align_fn=CO.roi_align)
    pooled = keep["pooled"].view(N, cap, 7, 7, 256)
    pe = torch.cat([pooled[i, :counts[i]] for i in range(N)]).cpu().float().permute(0, 3, 1, 2)
    assert float((pe - pooled_ref).abs().max()) < 1e-4 * max(1.0, float(pooled_ref.abs().max()))
    x = torch.flatten(pe, 1)
    h1 = F.relu(F.linear(x, p["roi_heads.box_head.fc1.weight"], p["roi_heads.box_head.fc1.bias"]))
    h1e = torch.cat([keep["h1"].view(N, cap, -1)[i, :counts[i]] for i in range(N)]).cpu()
    assert rel(h1e, h1) < 1e-4
    bf = F.relu(F.linear(h1e, p["roi_heads.box_head.fc2.weight"], p["roi_heads.box_head.fc2.bias"]))
    bfe = torch.cat([keep["box_feats"].view(N, cap, -1)[i, :counts[i]] for i in range(N)]).cpu()
    assert rel(bfe, bf) < 1e-4
    # predictor, PLN embeddings / reconstruction
    d_ref, _ = O.box_predictor(bfe, p)
    pd = keep["pred"]["pred_deltas"].view(N, cap, 4)
    assert rel(torch.cat([pd[i, :counts[i]] for i in range(N)]), d_ref) < 1e-4
    cnt1 = keep["cnt1"].cpu()
    for i in range(N):
        c = int(cnt1[i])
        _, rec, _, emb = O.pln_inference(keep["det_feats"][i, :c].cpu(), p, 0.23, 80, 20)
        assert rel(keep["emb"].view(N, 1000, -1)[i, :c], emb) < 1e-4 and rel(keep["rec"].view(N, 1000, -1)[i, :c], rec) < 1e-4


def test_fast_mode_detection_agreement_is_reported(world):
    """fp16 storage between ~50 layers moves scores by ~1e-3..1e-2 relative, which re-orders near-ties in the top-k / NMS
    cascade: the agreement below is what the fast path delivers against the fp32 reference on random-init weights."""
    ref = world["ref"]
    _, dets = _run(world, torch.float16)
    frac, tot = _report("fast mode, fp16 storage", dets, ref)
    loose = [agreement(g, r, iou_thr=0.9, score_tol=5e-2) for g, r in zip(dets, ref)]
    lfrac = sum(m for m, *_ in loose) / max(sum(max(ng, nr) for _, _, ng, nr in loose), 1)
    print(f"[fast mode] at IoU >= 0.9, |dscore| <= 5e-2: {lfrac:.3f}")
    # bf16 storage (8 significant bits) is a TRAINING storage type here (bench.py does not offer it for inference): reported only
    _, dets_bf = _run(world, torch.bfloat16)
    _report("fast mode, bf16 storage (training-only storage type)", dets_bf, ref)
    # measured 0.8975-0.91 on MI355X in rounds 2-5: a regression of the fp16 path's agreement below 0.88 fails
    assert frac >= 0.88 and lfrac >= frac, (frac, lfrac)


def test_which_fp16_storage_point_costs_the_agreement(world):
    """the fast path re-run with each fp16 storage point kept in fp32 in turn (engine fp32_points, a
    diagnostic: the layers behind the point run on the fp32 kernels). Reported; asserted only that no such point makes the
    agreement WORSE by more than noise, and that the all-fp32-heads-behind-an-fp16-backbone configuration is reported next to the
    fp32-backbone one -- the numbers say where the 9 % goes."""
    ref = world["ref"]
    res = {}
    for pts in ((), ("rpn_hidden",), ("pooled",), ("h1",), ("pooled", "h1"), ("rpn_hidden", "pooled", "h1"), ("backbone",)):
        eng = world["Engine"](world["params"], dtype=torch.float16, device=DEV, fp32_points=pts)
        out = eng.forward(world["images"].to(DEV), world["sizes"])
        torch.cuda.synchronize()
        dets = [(d["pred_boxes"], d["scores"], d["pred_classes"]) for d in eng.to_instances(out, N)]
        frac, _ = _report("fast mode, fp32 at " + ("+".join(pts) if pts else "(nothing: the benchmark path)"), dets, ref)
        res[pts] = frac
        del eng
    print("\n[storage points] " + "; ".join(f"{'+'.join(k) or 'none'}: {v:.3f}" for k, v in res.items()))
    base = res[()]
    assert base >= 0.85
    for k, v in res.items():
        assert v >= base - 0.05, (k, v, base)


def test_parity_mode_at_benchmark_resolution(osr):
    """One 3x800x1333 image (the benchmark's input size: 89 523 anchors, 4273 proposals) through the parity mode against the fp32
    oracle: the same agreement bar as at 256x384, so the result does not hinge on small feature maps."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params, with_known_unknown_mix
    g = torch.Generator().manual_seed(77)
    image = torch.randint(0, 256, (1, 3, 800, 1333), generator=g, dtype=torch.uint8)
    base = random_params(0)
    keep = {}
    eng = OpensetRCNNEngine(base, dtype=torch.float32, device=DEV)
    eng.forward(image.to(DEV), [(800, 1333)], keep=keep)
    params = with_known_unknown_mix(base, keep["emb"][: int(keep["cnt1"][0])])
    del eng, keep
    with torch.no_grad():
        ref = O.detector_inference([image[0]], params, params, roi_align_fn=CO.roi_align)
    eng = OpensetRCNNEngine(params, dtype=torch.float32, device=DEV)
    out = eng.forward(image.to(DEV), [(800, 1333)])
    torch.cuda.synchronize()
    d = eng.to_instances(out, 1)[0]
    m, c, ng, nr = agreement((d["pred_boxes"], d["scores"], d["pred_classes"]), ref[0])
    print(f"\n[parity mode, 800x1333] {m}/{nr} oracle detections matched ({c} with the same class), engine returned {ng}; "
          f"known {int((ref[0][2] != 80).sum())} / unknown {int((ref[0][2] == 80).sum())}")
    assert nr > 50 and m >= 0.95 * max(nr, ng) and c == m
