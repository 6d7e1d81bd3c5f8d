"""End-to-end composition test of the inference engine on the GPU, checked stage by stage against the CPU oracle.

fp16 storage between layers makes a whole-network comparison loose by construction, so each stage of
the oracle is fed the ENGINE's own inputs to that stage: dense stages are then compared at the tolerance of one
rounding of the stored result, and every index-producing stage (top-k selection, first-stage sort, PLN class,
NMS keep lists, final detections) must match bit-exactly given identical inputs.
"""
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


# storage dtype of activations / MFMA operands -> tolerance multiplier of the dense-stage checks (bf16 keeps 8 significant
# bits against fp16's 11); every index-producing stage is bit-exact for both, given the engine's own inputs
@pytest.fixture(scope="module", params=[torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def run(osr, request):
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    params = random_params(0)
    dt = request.param
    tol = 1.0 if dt == torch.float16 else 8.0
    q = (lambda t: t.half().float()) if dt == torch.float16 else (lambda t: t.bfloat16().float())
    eng = OpensetRCNNEngine(params, dtype=dt, device=DEV)
    g = torch.Generator().manual_seed(7)
    images = torch.randint(0, 256, (2, 3, 250, 330), generator=g, dtype=torch.uint8)
    sizes = [(250, 330), (240, 300)]  # second image: smaller valid area inside the same tensor
    keep = {}
    out = eng.forward(images.to(DEV), sizes, keep=keep)
    torch.cuda.synchronize()
    return dict(eng=eng, params=params, images=images, sizes=sizes, keep=keep, out=out, q=q, tol=tol)


def test_backbone_wiring(run):
    q16, TOL = run["q"], run["tol"]
    p = {k: (q16(v) if v.dim() == 4 else v) for k, v in run["params"].items()}
    batch, _ = O.preprocess_images([im for im in run["images"]])
    feats = O.resnet_fpn_forward(q16(batch), p, quant=q16)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        e = nchw(run["keep"]["feats"][k])
        assert e.shape == feats[k].shape
        assert rel_err(e, feats[k]) < 3e-2 * TOL, f"{k}: rel err {rel_err(e, feats[k])}"  # ~50 rounded layers deep
    for k in ("res2", "res3", "res4", "res5"):
        assert nchw(run["keep"][k]).shape[1] == {"res2": 256, "res3": 512, "res4": 1024, "res5": 2048}[k]


def test_rpn_head_and_selection(run):
    keep, p = run["keep"], run["params"]
    q16, TOL = run["q"], run["tol"]
    feats = {k: nchw(v) for k, v in keep["feats"].items()}
    n = 2
    # dense part: oracle on the engine's pyramid (fp16 conv weights, fp32 everywhere else)
    pq = dict(p)
    pq["proposal_generator.rpn_head.conv.weight"] = q16(p["proposal_generator.rpn_head.conv.weight"])
    ds, cs = [], []
    for k in ("p2", "p3", "p4", "p5", "p6"):
        d, c = O.cfrpn_head(feats[k], pq)
        ds.append(d)
        cs.append(c)
    ds, cs = O.flatten_head_outputs(ds, cs)
    d_ref = torch.cat([d.reshape(-1, 4) for d in ds])
    c_ref = torch.cat([c.reshape(-1) for c in cs])
    assert rel_err(keep["rpn_deltas"], d_ref) < 5e-3 * TOL  # hidden state stored in the low-precision dtype before the normalise
    assert float((keep["rpn_ctr"].cpu() - c_ref).abs().max()) < 2e-3 * TOL
    # index part: oracle selection on the engine's own deltas / centerness -> bit-exact
    shapes = keep["rpn_shapes"]
    anchors = O.anchor_grid(shapes)
    ctr, dl, off = [], [], 0
    for h, w in shapes:
        ctr.append(keep["rpn_ctr"][off:off + n * h * w].cpu().view(n, h * w))
        dl.append(keep["rpn_deltas"][off:off + n * h * w].cpu().view(n, h * w, 4))
        off += n * h * w
    props = [O.ltrb_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)).view(n, -1, 4) for d, a in zip(dl, anchors)]
    ref = O.find_top_rpn_proposals(props, ctr, run["sizes"], 1000)
    sel = keep["sel"]
    for i, (rb, rs, ri) in enumerate(ref):
        c = int(sel["counts"][i])
        assert c == len(rb)
        assert torch.equal(sel["src_index"][i, :c].cpu().long(), ri)
        assert torch.equal(sel["boxes"][i, :c].cpu(), rb)
        assert torch.equal(sel["scores"][i, :c].cpu(), rs)


def test_roi_heads_stagewise(run):
    keep, p, eng = run["keep"], run["params"], run["eng"]
    q16, TOL = run["q"], run["tol"]
    sel = keep["sel"]
    n, cap = 2, sel["cap"]
    counts = [int(c) for c in sel["counts"].cpu()]
    feats = [nchw(keep["feats"][k]) for k in ("p2", "p3", "p4", "p5")]
    boxes = [sel["boxes"][i, :counts[i]].cpu() for i in range(n)]
    # RoIAlign (fp16 out) vs the C oracle on the engine's pyramid and proposals
    pooled_ref = O.roi_pooler_ref(feats, boxes, roi_align_fn=CO.roi_align)
    pooled = keep["pooled"].view(n, cap, 7, 7, 256)
    pe = torch.cat([pooled[i, :counts[i]] for i in range(n)]).cpu().float().permute(0, 3, 1, 2)
    assert float((pe - pooled_ref).abs().max()) < 2e-3 * TOL * max(1.0, float(pooled_ref.abs().max()))
    assert float(pooled[0, counts[0]:].abs().max()) == 0.0  # padded rows are zero
    # box head on the engine's pooled features (fp16 operands, fp32 accumulate)
    pq = dict(p)
    pq["roi_heads.box_head.fc1.weight"] = q16(p["roi_heads.box_head.fc1.weight"])
    pq["roi_heads.box_head.fc2.weight"] = q16(p["roi_heads.box_head.fc2.weight"])
    x = torch.flatten(pe, 1)
    h1 = q16(F.relu(F.linear(x, pq["roi_heads.box_head.fc1.weight"], p["roi_heads.box_head.fc1.bias"])))
    bf_ref = F.relu(F.linear(h1, pq["roi_heads.box_head.fc2.weight"], p["roi_heads.box_head.fc2.bias"]))
    bf = keep["box_feats"].view(n, cap, -1)
    bfe = torch.cat([bf[i, :counts[i]] for i in range(n)]).cpu()
    assert rel_err(bfe, bf_ref) < 5e-3 * TOL
    # predictor + first-stage filtering on the engine's box features
    d_ref, iou_ref = O.box_predictor(bfe, p)
    pd = keep["pred"]["pred_deltas"].view(n, cap, 4)
    assert rel_err(torch.cat([pd[i, :counts[i]] for i in range(n)]), d_ref) < 1e-4
    score = keep["pred"]["score"].view(n, cap).cpu()
    cand = keep["pred"]["cand"].view(n, cap).cpu()
    keep1, cnt1 = keep["keep1"].cpu(), keep["cnt1"].cpu()
    for i in range(n):
        ids = torch.nonzero(cand[i, :counts[i]]).squeeze(1).numpy()
        order = ids[CO.argsort_desc(score[i, ids].numpy())][:1000]
        assert int(cnt1[i]) == len(order)
        assert keep1[i, :len(order)].tolist() == order.tolist()  # first-stage "NMS" (thr 1.0) == stable sort + top-1000
    # PLN on the engine's gathered features
    det_feats = keep["det_feats"].cpu()
    for i in range(n):
        c = int(cnt1[i])
        cls, rec, md, emb = O.pln_inference(det_feats[i, :c], p, eng.cfg["unk_thr"], 80, 20)
        assert rel_err(keep["emb"].view(n, 1000, -1)[i, :c], emb) < 1e-4
        assert rel_err(keep["rec"].view(n, 1000, -1)[i, :c], rec) < 1e-4
        rep = F.normalize(p["roi_heads.dml.representatives"])
        dist = 1.0 - F.normalize(emb) @ rep.t()
        top2 = dist.topk(2, dim=1, largest=False)[0]
        safe = ((top2[:, 1] - top2[:, 0]) > 1e-5) & ((md - eng.cfg["unk_thr"]).abs() > 1e-5)
        assert torch.equal(keep["pln_class"].view(n, 1000)[i, :c].cpu()[safe], cls[safe])


def test_final_detections_given_engine_logits(run):
    """Softmax classifier + both NMS passes + assembly, oracle fed the engine's det boxes/scores/classes/logits."""
    keep, eng = run["keep"], run["eng"]
    n = 2
    cnt1 = keep["cnt1"].cpu()
    ob, osc, ocl, on = [t.cpu() for t in run["out"]]
    cfg = dict(O.VOC_COCO_CFG)
    for i in range(n):
        c = int(cnt1[i])
        b = keep["det_boxes"][i, :c].cpu()
        s = keep["det_scores"][i, :c, 0].cpu()
        cls = keep["pln_class"].view(n, 1000)[i, :c].cpu()
        lg = keep["logits"].view(n, 1000, 21)[i, :c].cpu()
        known = cls != 80
        probs = F.softmax(lg[known], dim=-1)
        kb, ks, kc = O.softmax_known_inference(b[known], probs, run["sizes"][i], cfg["known_score_thresh"], cfg["known_nms_thresh"], cfg["known_topk"])
        if not bool(known.all()):
            ub, us, uc = O.softmax_unknown_inference(b[~known], s[~known], run["sizes"][i], cfg["unknown_score_thresh"],
                                                     cfg["unknown_nms_thresh"], cfg["unknown_topk"], 80)
            rb, rs, rc = torch.cat((ub, kb)), torch.cat((us, ks)), torch.cat((uc, kc))
        else:
            rb, rs, rc = kb, ks, kc
        m = int(on[i])
        assert m == len(rb), f"image {i}: {m} detections vs oracle {len(rb)}"
        assert torch.equal(ocl[i, :m], rc)
        assert torch.equal(ob[i, :m], rb)
        assert float((osc[i, :m] - rs).abs().max()) < 1e-6
    insts = eng.to_instances(run["out"], n)
    assert len(insts) == n and set(insts[0]) == {"pred_boxes", "scores", "pred_classes"}


def test_engine_is_deterministic(run):
    eng = run["eng"]
    a = eng.forward(run["images"].to(DEV), run["sizes"])
    b = eng.forward(run["images"].to(DEV), run["sizes"])
    for x, y, z in zip(a, b, run["out"]):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_micro_batch_streams_give_the_same_detections(run):
    """Splitting the batch over HIP streams is pure intra-GPU data parallelism: per-image results do not change."""
    eng = run["eng"]
    imgs = torch.cat([run["images"], run["images"].flip(0)]).to(DEV)  # 4 images
    hw = torch.tensor(run["sizes"] + run["sizes"][::-1], dtype=torch.int32, device=DEV)
    a = eng.forward_device(imgs, hw, 256, 352)
    b = eng.forward_device_streams(imgs, hw, 256, 352, nstreams=2)
    c = eng.forward_device_streams(imgs, hw, 256, 352, nstreams=4)
    torch.cuda.synchronize()
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_hipgraph_replay_matches_eager(run):
    eng = run["eng"]
    imgs = torch.cat([run["images"], run["images"].flip(0)]).to(DEV)
    hw = torch.tensor(run["sizes"] + run["sizes"][::-1], dtype=torch.int32, device=DEV)
    eager = [t.clone() for t in eng.forward_device(imgs, hw, 256, 352)]
    graph, out = eng.capture(imgs, hw, 256, 352, nstreams=2)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for x, y in zip(eager, out):
        assert torch.equal(x, y)
    imgs.copy_(imgs.flip(0))  # new content in the captured input buffer -> replay follows it
    graph.replay()
    torch.cuda.synchronize()
    ref = eng.forward_device(imgs, hw, 256, 352)
    for x, y in zip(ref, out):
        assert torch.equal(x, y)


def test_graspnet_configuration_tail(osr):
    """GraspNet yaml (28 known of 88 classes, UNK_THR 0.09, unknown id 1000): PLN classes, softmax over 28+1 logits, both NMS
    passes and the final remap of the known ids through the sorted `class_id` table (prototype_learning_network.py:222,
    softmax_classifier.py:300-334), oracle fed the engine's own tensors -> classes / boxes / counts bit-exact."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    K, UNK = 28, 1000
    params = random_params(0, num_known=K)
    class_map = torch.arange(0, 88, 3)[:K].to(torch.int64) + 1  # 28 increasing dataset ids, none of them its own index
    g = torch.Generator().manual_seed(11)
    images = torch.randint(0, 256, (2, 3, 250, 330), generator=g, dtype=torch.uint8)
    sizes = [(250, 330), (236, 310)]
    # the yaml's threshold first (random prototypes: everything is "unknown"), then the median distance of that run, which
    # splits the detections into known and unknown ones
    thr, seen = 0.09, {}
    for attempt in range(2):
        eng = OpensetRCNNEngine(params, dict(num_known=K, num_classes=88, unknown_id=UNK, unk_thr=thr), torch.float16, DEV, class_map)
        keep = {}
        out = [t.cpu() for t in eng.forward(images.to(DEV), sizes, keep=keep)]
        seen[attempt] = _check_graspnet_tail(eng, params, keep, out, sizes, thr, K, UNK, class_map)
        md = keep["min_dist"].view(2, 1000).cpu()
        thr = float(torch.cat([md[i, :int(keep["cnt1"][i])] for i in range(2)]).median())
    assert seen[1][0] > 0 and seen[1][1] > 0, "the median threshold should produce both kinds of detections"


def _check_graspnet_tail(eng, params, keep, out, sizes, thr, K, UNK, class_map):
    ob, osc, ocl, on = out
    n, cnt1 = 2, keep["cnt1"].cpu()
    cfg = dict(O.VOC_COCO_CFG)
    seen_known = seen_unknown = 0
    for i in range(n):
        c = int(cnt1[i])
        det_feats = keep["det_feats"][i, :c].cpu()
        cls_ref, _, md, emb = O.pln_inference(det_feats, params, thr, UNK, K)
        rep = F.normalize(params["roi_heads.dml.representatives"])
        top2 = (1.0 - F.normalize(emb) @ rep.t()).topk(2, dim=1, largest=False)[0]
        safe = ((top2[:, 1] - top2[:, 0]) > 1e-5) & ((md - thr).abs() > 1e-5)
        cls = keep["pln_class"].view(n, 1000)[i, :c].cpu()
        exp = cls_ref.clone()  # the PLN reports the known classes through the class_id table (:222)
        exp[cls_ref != UNK] = class_map[cls_ref[cls_ref != UNK]]
        assert torch.equal(cls[safe], exp[safe])
        b = keep["det_boxes"][i, :c].cpu()
        s = keep["det_scores"][i, :c, 0].cpu()
        lg = keep["logits"].view(n, 1000, K + 1)[i, :c].cpu()
        known = cls != UNK
        parts = []
        if bool((~known).any()):
            parts.append(O.softmax_unknown_inference(b[~known], s[~known], sizes[i], cfg["unknown_score_thresh"], cfg["unknown_nms_thresh"],
                                                     cfg["unknown_topk"], UNK))
        if bool(known.any()):
            kb, ks, kc = O.softmax_known_inference(b[known], F.softmax(lg[known], dim=-1), sizes[i], cfg["known_score_thresh"],
                                                   cfg["known_nms_thresh"], cfg["known_topk"])
            parts.append((kb, ks, class_map[kc]))
        rb, rs, rc = (torch.cat([p_[j] for p_ in parts]) for j in range(3))
        m = int(on[i])
        assert m == len(rb)
        assert torch.equal(ocl[i, :m], rc) and torch.equal(ob[i, :m], rb)
        assert float((osc[i, :m] - rs).abs().max()) < 1e-6
        seen_known += int((rc != UNK).sum())
        seen_unknown += int((rc == UNK).sum())
    return seen_known, seen_unknown


def test_graspnet_resolution_batch_of_eight(osr):
    """BASELINE config 3's per-GPU workload: eight 1280x720 frames after ResizeShortestEdge(800, 1333) = 750x1333, padded to 768x1344
    (SURVEY 8d), GraspNet head sizes (28 known of 88 classes, unknown id 1000), inference and one training step at full size.
    Size-independent properties: every count within its capacity, boxes inside the image, scores descending per group, the same
    batch split 5 + 3 gives the same detections (images are independent), the hipGraph replay reproduces the eager pass, and the
    training step's six losses are finite with every gradient bucket written."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params, with_known_unknown_mix
    K, UNK, n, h, w = 28, 1000, 8, 750, 1333
    params = random_params(0, num_known=K)
    class_map = torch.arange(0, 88, 3)[:K].to(torch.int64) + 1
    cfg = dict(num_known=K, num_classes=88, unknown_id=UNK, unk_thr=0.09)
    g = torch.Generator().manual_seed(21)
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).to(DEV)
    hw = torch.tensor([(h, w)] * n, dtype=torch.int32, device=DEV)
    cal = OpensetRCNNEngine(params, cfg, torch.float16, DEV, class_map)
    keep = {}
    cal.forward_device(images[:2], hw[:2], 768, 1344, keep)
    cnt = keep["cnt1"].cpu()
    emb = torch.cat([keep["emb"].view(2, -1, keep["emb"].shape[-1])[i, :int(cnt[i])] for i in range(2)]).cpu()
    params = with_known_unknown_mix(params, emb, unk_thr=0.09)
    eng = OpensetRCNNEngine(params, cfg, torch.float16, DEV, class_map)
    ob, osc, ocl, on = [t.cpu() for t in eng.forward_device(images, hw, 768, 1344)]
    assert ob.shape == (n, 100, 4) and int(on.min()) > 0 and int(on.max()) <= 100
    known_ids = set(class_map.tolist())
    seen_known = seen_unknown = 0
    for i in range(n):
        c = int(on[i])
        b, s, k = ob[i, :c], osc[i, :c], ocl[i, :c]
        assert bool((b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 2] <= w).all() and (b[:, 3] <= h).all())
        assert bool(torch.isfinite(s).all()) and set(k.tolist()) <= known_ids | {UNK}
        unk = k == UNK
        seen_known += int((~unk).sum()); seen_unknown += int(unk.sum())
        for grp in (s[unk], s[~unk]):  # [unknown..., known...], each sorted by score (softmax_classifier.py:334)
            assert bool((grp[1:] <= grp[:-1]).all())
    assert seen_known > 0 and seen_unknown > 0
    a = [t.cpu() for t in eng.forward_device(images[:5], hw[:5], 768, 1344)]
    b2 = [t.cpu() for t in eng.forward_device(images[5:], hw[5:], 768, 1344)]
    for full, pa, pb in zip((ob, osc, ocl, on), a, b2):
        assert torch.equal(full, torch.cat((pa, pb)))
    graph, gout = eng.capture(images, hw, 768, 1344, 2)
    graph.replay()
    torch.cuda.synchronize()
    for full, got in zip((ob, osc, ocl, on), gout):
        assert torch.equal(full, got.cpu())
    # one training step of the same configuration (id_map through class_map), 6 ground-truth boxes per frame
    tr = OpensetRCNNTrainer(params, cfg=cfg, dtype=torch.float16, device=DEV, lr=1e-5, loss_scale=512.0, class_map=class_map)
    ngt = 6
    ctr = torch.rand(n, ngt, 2, generator=g) * torch.tensor([w * 0.8, h * 0.8]) + 40
    size = torch.rand(n, ngt, 2, generator=g) * 300 + 32
    gt = torch.cat((ctr - size / 2, ctr + size / 2), dim=2)
    gt[..., 0::2].clamp_(0, w); gt[..., 1::2].clamp_(0, h)
    gcls = class_map[torch.randint(0, K, (n, ngt), generator=g)]
    gcnt = torch.full((n,), ngt, dtype=torch.int32)
    shapes = eng.pyramid_shapes(768, 1344)
    assert shapes == [(192, 336), (96, 168), (48, 84), (24, 42), (12, 21)]
    r = sum(a_ * b_ for a_, b_ in shapes)
    cap = sum(min(2000, a_ * b_) for a_, b_ in shapes)
    keys = {k_: torch.rand(s_, generator=g).to(DEV) for k_, s_ in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + ngt)))}
    tr.grad_flat.fill_(float("nan"))
    losses = tr.step(images, hw, 768, 1344, gt.to(DEV), gcls.to(DEV), gcnt.to(DEV), keys, update=False)
    vals = {k_: float(v) for k_, v in losses.items()}
    assert set(vals) == {"loss_rpn_loc", "loss_rpn_ctr", "loss_box_reg", "loss_iou", "loss_dml", "loss_cls"}
    assert all(v == v and abs(v) < 1e3 for v in vals.values()), vals
    for name, gview in tr.grad.items():
        assert bool(torch.isfinite(gview).all()), name  # every parameter's gradient was written (no NaN left from the fill)
