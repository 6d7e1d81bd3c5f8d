"""GPU parity of the loss / per-row backward kernels against torch autograd applied to the oracle's forward functions."""
import pytest
import torch
import torch.nn.functional as F

from oracle import osr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops(osr):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    osr._lib.load()
    return osr.ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def test_rpn_losses_and_tail_backward(ops):
    import tests.test_train_fwd as TF
    shapes, strides, sizes = [(24, 40), (12, 20), (6, 10), (3, 5)], (4, 8, 16, 32), (32, 64, 128, 256)
    c = TF._rpn_case(ops, 61, shapes, strides, sizes, 2, (96, 160), [5, 3])
    lr, lo, mb, ct = TF._check_rpn_targets(ops, c)
    n, gg = c["n"], g(62)
    # hidden state t per level (level-major rows), tail weights
    rows = [n * h * w for h, w in shapes]
    t = F.relu(torch.randn(sum(rows), 256, generator=gg)).half()
    w_tail = torch.randn(5, 256, generator=gg) * 0.5
    b_tail = torch.tensor([0.4, 0.4, 0.4, 0.4, 0.0])
    tt = t.float().requires_grad_(True)
    wt = w_tail.clone().requires_grad_(True)
    bt = b_tail.clone().requires_grad_(True)
    u = tt / tt.norm(dim=1, keepdim=True).clamp(min=1e-12)
    o = u @ wt.t() + bt
    deltas_lm, ctr_lm = o[:, :4], torch.sigmoid(o[:, 4])
    # level-major -> image-major (n, R, .)
    dl, cl, off = [], [], 0
    for (h, w), r in zip(shapes, rows):
        dl.append(deltas_lm[off:off + r].view(n, h * w, 4))
        cl.append(ctr_lm[off:off + r].view(n, h * w))
        off += r
    ref = O.rpn_losses(c["anchors"], torch.cat(dl, 1), torch.cat(cl, 1), lr.cpu(), lo.cpu(), mb.cpu(), ct.cpu())
    loss = ref["loss_rpn_loc"] + ref["loss_rpn_ctr"]
    loss.backward()
    scale = 64.0
    d5 = ops.rpn_losses_bwd(c["lv"], c["cell"], n, deltas_lm.detach().contiguous().to(DEV), ctr_lm.detach().contiguous().to(DEV), lr, lo, mb, ct,
                            loss_scale=scale)
    # d o: from autograd through the sigmoid
    o2 = o.detach().clone().requires_grad_(True)
    dl2, cl2, off = [], [], 0
    for (h, w), r in zip(shapes, rows):
        dl2.append(o2[off:off + r, :4].reshape(n, h * w, 4))
        cl2.append(torch.sigmoid(o2[off:off + r, 4]).view(n, h * w))
        off += r
    ref2 = O.rpn_losses(c["anchors"], torch.cat(dl2, 1), torch.cat(cl2, 1), lr.cpu(), lo.cpu(), mb.cpu(), ct.cpu())
    (ref2["loss_rpn_loc"] + ref2["loss_rpn_ctr"]).backward()
    assert rel(d5 / scale, o2.grad) < 1e-4
    assert int((o2.grad.abs().sum(1) > 0).sum()) > 20
    dt, dw, db = ops.cfrpn_tail_bwd(t.to(DEV), w_tail.to(DEV), d5)
    assert rel(dw / scale, wt.grad) < 1e-4 and rel(db / scale, bt.grad) < 1e-4
    assert rel(dt.float() / scale, tt.grad * (t.float() > 0)) < 2e-3  # stored in fp16


def test_roi_head_losses_backward(ops):
    gg = g(71)
    m, K, NC, d = 512, 20, 81, 256
    cls = torch.randint(0, K, (m,), generator=gg)
    cls[torch.rand(m, generator=gg) < 0.6] = NC
    cls[m - 20:] = -1  # padding rows
    prop = torch.rand(m, 4, generator=gg) * 300
    prop[:, 2:] = prop[:, :2] + 8 + torch.rand(m, 2, generator=gg) * 200
    gtb = prop + torch.randn(m, 4, generator=gg) * 6
    gtb[:, 2:] = torch.max(gtb[:, 2:], gtb[:, :2] + 2)
    gi = torch.rand(m, generator=gg)
    pred = torch.randn(m, 5, generator=gg)
    ok = cls >= 0
    # box / IoU losses
    p = pred.clone().requires_grad_(True)
    lb, li = O.roi_box_losses(p[ok][:, :4], torch.sigmoid(p[ok][:, 4]), prop[ok], gtb[ok], cls[ok], gi[ok], NC)
    (lb + li).backward()
    dp = ops.roi_box_losses_bwd(pred.to(DEV), prop.to(DEV), gtb.to(DEV), cls.to(DEV), gi.to(DEV), NC, loss_scale=8.0)
    assert rel(dp / 8.0, p.grad) < 1e-4
    # cross entropy
    logits = torch.randn(m, K + 1, generator=gg) * 2
    lg = logits.clone().requires_grad_(True)
    O.softmax_ce_loss(lg[ok], cls[ok], NC, K, 0.9).backward()
    dl = ops.softmax_ce_loss_bwd(logits.to(DEV), cls.to(DEV), NC, 0.9, loss_scale=4.0)
    assert rel(dl / 4.0, lg.grad) < 1e-4
    # PLN: gradient w.r.t. the embedding and the raw prototypes (through both normalisations)
    protos = (torch.randn(K, d, generator=gg) * 1.5).requires_grad_(True)
    emb = (torch.randn(m, d, generator=gg) + 0.8 * protos.detach()[cls.clamp(0, K - 1)]).requires_grad_(True)
    new = F.normalize(emb[ok])
    rep = F.normalize(protos)
    c_ok, i_ok = cls[ok], gi[ok]
    fg = torch.nonzero((c_ok >= 0) & (c_ok < K) & (i_ok > 0.5)).squeeze(1)
    dist = 1.0 - new[fg] @ rep.t()
    ar = torch.arange(dist.shape[0])
    intra = dist[ar, c_ok[fg]]
    d2 = dist.clone()
    d2[ar, c_ok[fg]] = 1000
    inter = d2.min(dim=1)[0]
    cd = (1.0 - rep @ rep.t()).clone()
    cd[torch.arange(K), torch.arange(K)] = 1000
    cdist = cd.min(dim=1)[0]
    alpha, beta = 0.3, 1.1
    loss = (torch.clamp(intra - alpha, min=0).sum() + torch.clamp(beta - inter, min=0).sum() + torch.clamp(beta + alpha - cdist, min=0).sum()) * 0.5 / int(ok.sum())
    loss.backward()
    assert float(torch.clamp(beta + alpha - cdist, min=0).sum()) > 0 and float(torch.clamp(intra - alpha, min=0).sum()) > 0
    de, dpr = ops.pln_loss_bwd(emb.detach().to(DEV), protos.detach().to(DEV), cls.to(DEV), gi.to(DEV), 0.5, alpha, beta, 0.5, loss_scale=16.0)
    assert rel(de / 16.0, emb.grad) < 1e-4
    assert rel(dpr / 16.0, protos.grad) < 1e-4
    # the forward kernel agrees with this formulation
    fw = ops.pln_loss_fwd(emb.detach().to(DEV), F.normalize(protos.detach()).to(DEV), cls.to(DEV), gi.to(DEV), 0.5, alpha, beta, 0.5)
    assert float(fw[0]) == pytest.approx(float(loss), rel=1e-5)


def test_roi_align_backward(ops):
    gg = g(81)
    n, c = 2, 32
    shapes = [(32, 48), (16, 24), (8, 12), (4, 6)]
    m = 60
    ctr = torch.rand(m, 2, generator=gg) * torch.tensor([192.0, 128.0])
    size = torch.exp(torch.rand(m, 2, generator=gg) * 5.0 + 0.5)
    boxes = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
    boxes[0] = torch.tensor([-20.0, -10.0, 40.0, 30.0])
    boxes[1] = torch.tensor([0.0, 0.0, 192.0, 128.0])  # large RoI: table overflow -> per-sample path
    bidx = torch.randint(0, n, (m,), generator=gg, dtype=torch.int32)
    bidx[5] = -1
    dout = torch.randn(m, 7, 7, c, generator=gg)
    scales = (0.25, 0.125, 0.0625, 0.03125)
    got = ops.roi_align_bwd(dout.to(DEV), shapes, n, scales, boxes.to(DEV), bidx.to(DEV))
    # reference: the forward is linear in the features, so d feat = J^T dout with J probed by autograd on the torch restatement
    lv = O.assign_levels(boxes)
    for l, (h, w) in enumerate(shapes):
        feat = torch.zeros(n, c, h, w, requires_grad=True)
        ids = torch.nonzero((lv == l) & (bidx >= 0)).squeeze(1)
        if len(ids) == 0:
            assert float(got[l].abs().max()) == 0.0
            continue
        rois = torch.cat((bidx[ids].float().unsqueeze(1), boxes[ids]), dim=1)
        out = O.roi_align_torch(feat, rois, scales[l])
        out.backward(dout[ids].permute(0, 3, 1, 2))
        assert rel(got[l].permute(0, 3, 1, 2), feat.grad) < 1e-4, f"level {l}"


def test_roi_align_backward_extreme_aspect_footprints(ops):
    """The streaming backward (one atomic per footprint pixel) and its per-bin fallback on wide-thin, tall-thin, sub-pixel and
    border-crossing boxes of one level, against autograd through the torch restatement of the forward."""
    gg = g(82)
    n, c, h, w = 2, 8, 120, 340
    boxes = torch.tensor([
        [4.0, 100.0, 1300.0, 112.0], [300.0, 2.0, 330.0, 470.0], [10.0, 10.0, 700.0, 400.0], [50.3, 60.2, 51.1, 61.0],
        [-40.0, -30.0, 90.0, 50.0], [1200.0, 400.0, 1400.0, 520.0], [0.0, 0.0, 1360.0, 480.0], [600.0, 200.0, 640.0, 203.0],
    ])
    bidx = torch.tensor([0, 1, 0, 1, 0, 1, 1, 0], dtype=torch.int32)
    dout = torch.randn(len(boxes), 7, 7, c, generator=gg)
    got = ops.roi_align_bwd(dout.to(DEV), [(h, w)], n, (0.25,), boxes.to(DEV), bidx.to(DEV), min_level=2)[0].cpu()
    feat = torch.zeros(n, c, h, w, requires_grad=True)
    out = O.roi_align_torch(feat, torch.cat((bidx.float().unsqueeze(1), boxes), dim=1), 0.25)
    out.backward(dout.permute(0, 3, 1, 2))
    assert rel(got.permute(0, 3, 1, 2), feat.grad) < 1e-4


def _image_major(boxes, bidx, dout, n):
    """Reorder a RoI list into the (n, S) image-major layout the pixel-centric backward wants (padding rows: batch_idx -1)."""
    per = [torch.nonzero(bidx == b).squeeze(1) for b in range(n)]
    S = max(len(p) for p in per) + 2  # at least two padding rows per image
    bx = torch.zeros(n * S, 4)
    bi = torch.full((n * S,), -1, dtype=torch.int32)
    do = torch.zeros((n * S,) + tuple(dout.shape[1:]))
    for b, ids in enumerate(per):
        bx[b * S:b * S + len(ids)] = boxes[ids]
        bi[b * S:b * S + len(ids)] = b
        do[b * S:b * S + len(ids)] = dout[ids]
    do[bi < 0] = 7.0  # (garbage behind padding rows must not matter)
    return bx, bi, do, S


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_roi_align_backward_dense_matches_autograd_and_the_scatter_kernel(ops, dt):
    """osr_roi_align_bwd_dense (pixel-centric gather, no atomics) on the lists of the two tests above, reordered image-major:
    against autograd through the torch restatement of the forward, against the scatter kernel, and bit-reproducible."""
    gg = g(83)
    n, c = 2, 32
    shapes = [(32, 48), (16, 24), (8, 12), (4, 6)]
    m = 60
    ctr = torch.rand(m, 2, generator=gg) * torch.tensor([192.0, 128.0])
    size = torch.exp(torch.rand(m, 2, generator=gg) * 5.0 + 0.5)
    boxes = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
    boxes[0] = torch.tensor([-20.0, -10.0, 40.0, 30.0])
    boxes[1] = torch.tensor([0.0, 0.0, 192.0, 128.0])
    boxes[2] = torch.tensor([50.3, 60.2, 51.1, 61.0])    # sub-pixel
    boxes[3] = torch.tensor([4.0, 100.0, 190.0, 104.0])  # wide and thin
    bidx = torch.randint(0, n, (m,), generator=gg, dtype=torch.int32)
    dout = torch.randn(m, 7, 7, c, generator=gg).to(dt).float()
    scales = (0.25, 0.125, 0.0625, 0.03125)
    bx, bi, do, S = _image_major(boxes, bidx, dout, n)
    got = ops.roi_align_bwd(do.to(dt).to(DEV), shapes, n, scales, bx.to(DEV), bi.to(DEV), rois_per_image=S)
    again = ops.roi_align_bwd(do.to(dt).to(DEV), shapes, n, scales, bx.to(DEV), bi.to(DEV), rois_per_image=S)
    scat = ops.roi_align_bwd(do.to(dt).to(DEV), shapes, n, scales, bx.to(DEV), bi.to(DEV))
    if dt != torch.float32:  # the storage dtype straight from the kernel = the fp32 sums rounded once
        low = ops.roi_align_bwd(do.to(dt).to(DEV), shapes, n, scales, bx.to(DEV), bi.to(DEV), rois_per_image=S, out_dtype=dt)
        for l in range(len(shapes)):
            assert low[l].dtype == dt and torch.equal(low[l], got[l].to(dt))
    lv = O.assign_levels(boxes)
    for l, (h, w) in enumerate(shapes):
        assert torch.equal(got[l], again[l]), "the gather has a fixed summation order"
        assert rel(got[l], scat[l]) < 1e-5, f"level {l}: gather vs scatter"
        feat = torch.zeros(n, c, h, w, requires_grad=True)
        ids = torch.nonzero(lv == l).squeeze(1)
        if len(ids) == 0:
            assert float(got[l].abs().max()) == 0.0
            continue
        rois = torch.cat((bidx[ids].float().unsqueeze(1), boxes[ids]), dim=1)
        O.roi_align_torch(feat, rois, scales[l]).backward(dout[ids].permute(0, 3, 1, 2))
        assert rel(got[l].permute(0, 3, 1, 2), feat.grad) < 1e-4, f"level {l}"


def test_roi_align_backward_dense_extreme_footprints_and_ragged_tiles(ops):
    gg = g(84)
    n, c, h, w = 2, 8, 117, 339  # not multiples of the 8 x 8 tile
    boxes = torch.tensor([
        [4.0, 100.0, 1300.0, 112.0], [300.0, 2.0, 330.0, 465.0], [10.0, 10.0, 700.0, 400.0], [50.3, 60.2, 51.1, 61.0],
        [-40.0, -30.0, 90.0, 50.0], [1200.0, 400.0, 1400.0, 520.0], [0.0, 0.0, 1356.0, 468.0], [600.0, 200.0, 640.0, 203.0],
    ])
    bidx = torch.tensor([0, 1, 0, 1, 0, 1, 1, 0], dtype=torch.int32)
    dout = torch.randn(len(boxes), 7, 7, c, generator=gg)
    bx, bi, do, S = _image_major(boxes, bidx, dout, n)
    got = ops.roi_align_bwd(do.to(DEV), [(h, w)], n, (0.25,), bx.to(DEV), bi.to(DEV), min_level=2, rois_per_image=S)[0].cpu()
    feat = torch.zeros(n, c, h, w, requires_grad=True)
    O.roi_align_torch(feat, torch.cat((bidx.float().unsqueeze(1), boxes), dim=1), 0.25).backward(dout.permute(0, 3, 1, 2))
    assert rel(got.permute(0, 3, 1, 2), feat.grad) < 1e-4


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_pool_bwd_two_byte_vector_path(ops, dt):
    """osr_pool_bwd on f16 / bf16 with c % 8 == 0 takes the 16-byte kernel: same sums in the same order as the scalar kernel (base, then
    the fine pixels row by row, fp32, one rounding) -- odd fine sizes (the last coarse row / column has fewer than four sources), with
    and without base, both modes."""
    gg = g(92)
    n, hf, wf, c = 2, 9, 7, 24
    fine = torch.randn(n, hf, wf, c, generator=gg).to(dt)
    base = torch.randn(n, 5, 4, c, generator=gg).to(dt)
    for b in (base, None):
        got = ops.pool_bwd(fine.to(DEV), (5, 4), None if b is None else b.to(DEV), 0).cpu()
        ref = torch.zeros(n, 5, 4, c) if b is None else b.float().clone()
        for dy in range(2):
            for dx in range(2):
                part = fine.float()[:, dy::2, dx::2]
                ref[:, :part.shape[1], :part.shape[2]] += part
        assert got.dtype == dt and torch.equal(got, ref.to(dt))
    src = torch.randn(n, 5, 4, c, generator=gg).to(dt)
    got = ops.pool_bwd(src.to(DEV), (hf, wf), fine.to(DEV), 1).cpu()
    ref = fine.float().clone()
    ref[:, ::2, ::2] += src.float()
    assert torch.equal(got, ref.to(dt))
    # a channel count the vector kernel does not take: the scalar kernel, same answer
    f2, b2 = fine[..., :20].contiguous(), base[..., :20].contiguous()
    got = ops.pool_bwd(f2.to(DEV), (5, 4), b2.to(DEV), 0).cpu()
    ref = b2.float().clone()
    for dy in range(2):
        for dx in range(2):
            part = f2.float()[:, dy::2, dx::2]
            ref[:, :part.shape[1], :part.shape[2]] += part
    assert torch.equal(got, ref.to(dt))


def test_elementwise_and_sgd(ops):
    gg = g(91)
    a = torch.randn(2, 9, 7, 16, generator=gg)
    act = torch.randn(2, 9, 7, 16, generator=gg)
    gt = a.clone().half().to(DEV)
    ops.relu_mask_(gt, act.half().to(DEV))
    assert torch.equal(gt.cpu(), (a.half() * (act.half() > 0)))
    out = ops.add_cast(a.to(DEV), act.half().to(DEV), torch.float16)
    assert torch.equal(out.cpu(), (a + act.half().float()).half())
    # FPN top-down backward: fine (9x7) -> coarse (5x4)
    base = torch.randn(2, 5, 4, 16, generator=gg)
    got = ops.pool_bwd(a.to(DEV), (5, 4), base.to(DEV), 0).cpu()
    up = torch.zeros(2, 5, 4, 16, requires_grad=True)
    fine = up.permute(0, 3, 1, 2).repeat_interleave(2, 2).repeat_interleave(2, 3)[:, :, :9, :7]
    fine.backward(a.permute(0, 3, 1, 2))
    assert rel(got, base + up.grad) < 1e-6
    # p6 = p5[::2, ::2] backward: src (5x4) -> out (9x7)
    src = torch.randn(2, 5, 4, 16, generator=gg)
    got = ops.pool_bwd(src.to(DEV), (9, 7), a.to(DEV), 1).cpu()
    ref = a.clone()
    ref[:, ::2, ::2] += src
    assert rel(got, ref) < 1e-6
    # SGD with momentum, weight decay, loss scale and a folded row scale, two steps, against torch.optim.SGD
    p0 = torch.randn(6, 10, generator=gg)
    rs = torch.rand(6, generator=gg) + 0.5
    tp = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([tp], lr=0.02, momentum=0.9, weight_decay=1e-2)
    p, buf = p0.clone().to(DEV), torch.zeros(6, 10).to(DEV)
    lowp = torch.empty(6, 10, dtype=torch.float16, device=DEV)
    for step in range(2):
        gfold = torch.randn(6, 10, generator=gg)  # gradient w.r.t. the folded weight w * rs
        tp.grad = gfold * rs[:, None]
        opt.step()
        ops.sgd_step_(p, (gfold * 32.0).to(DEV), buf, 0.02, 0.9, 1e-2, grad_scale=1.0 / 32.0, row_scale=rs.to(DEV), lowp=lowp)
    assert rel(p, tp.detach()) < 1e-6
    assert torch.equal(lowp.cpu(), (p.cpu() * rs[:, None]).half())


def test_pack_dgrad_weight_kernel_and_overflow_flag(ops):
    """osr_pack_dgrad_weight against the host packing (weights.pack_dgrad_weight: flip + permute), ragged channel counts, both
    element sizes, in-place refill of a preallocated buffer; osr_check_finite clears its flag on inf / NaN only."""
    from openset_rcnn_amd.host.weights import pack_conv_weight, pack_dgrad_weight
    gg = g(77)
    for cout, cin, k, dt in ((40, 96, 3, torch.float16), (256, 64, 1, torch.bfloat16), (33, 17, 5, torch.float32), (128, 128, 3, torch.float16)):
        w = torch.randn(cout, cin, k, k, generator=gg)
        fwd = pack_conv_weight(w, dt).to(DEV)           # (cout,kh,kw,cin): what the forward kernels read
        want = pack_dgrad_weight(w, dt)                 # (cin,kh,kw,cout), flipped
        got = ops.pack_dgrad_weight(fwd)
        assert got.shape == want.shape and torch.equal(got.cpu(), want)
        buf = torch.zeros_like(got)
        assert ops.pack_dgrad_weight(fwd, buf) is buf and torch.equal(buf.cpu(), want)
    m = torch.randn(21, 1024, generator=gg).to(DEV)
    assert torch.equal(ops.pack_dgrad_weight(m).cpu(), m.t().cpu())
    flag = torch.ones(1, dtype=torch.int32, device=DEV)
    x = torch.randn(100003, generator=gg).to(DEV)
    x = x[: 100000].contiguous()
    assert int(ops.check_finite_(x, flag)) == 1
    x[99999] = float("nan")
    assert int(ops.check_finite_(x, flag)) == 0
    flag.fill_(1)
    x[99999] = 0.0
    x[12345] = float("-inf")
    assert int(ops.check_finite_(x, flag)) == 0


def test_gemm_f32_tn_is_dy_transposed_times_x(ops):
    """osr_gemm_f32_tn (dW = dy^T x of the fp32 heads, autograd of F.linear at prototype_learning_network.py:204-205) against a
    float64 product: the head shapes of the training step (5-, 21-, 256-, 1024-wide dy over 8192 samples: split over samples and
    reduced), ragged shapes that take the scalar-load path, a single split, writing into a preallocated view; tolerance 1e-5 of
    the result scale (fp32 accumulation over up to 8192 samples)."""
    gg = g(91)
    for k, m, n in ((8192, 21, 1024), (8192, 5, 1024), (8192, 256, 1024), (8192, 1024, 256), (1000, 67, 130), (48, 64, 64), (7, 3, 5), (130, 200, 64)):
        a = torch.randn(k, m, generator=gg)
        b = torch.randn(k, n, generator=gg)
        want = a.double().t() @ b.double()
        got = ops.gemm_f32_tn(a.to(DEV), b.to(DEV)).cpu().double()
        assert got.shape == want.shape
        assert (got - want).abs().max() <= 1e-5 * want.abs().max(), (k, m, n)
    flat = torch.zeros(21 * 1024 + 8, device=DEV)
    a, b = torch.randn(8192, 21, generator=gg), torch.randn(8192, 1024, generator=gg)
    view = flat[8:].view(21, 1024)
    assert ops.gemm_f32_tn(a.to(DEV), b.to(DEV), out=view) is view
    assert torch.equal(view, ops.gemm_f32_tn(a.to(DEV), b.to(DEV)))           # fixed reduction order: bit-identical reruns
    assert float(flat[:8].abs().sum()) == 0.0


@pytest.mark.parametrize("box_loss,aux_beta", [(("iou", 0.0), 0.0), (("smooth_l1", 0.0), 0.3), (("smooth_l1", 0.11), 0.0), (("giou", 0.0), 0.05),
                                               (("diou", 0.0), 0.0), (("ciou", 0.0), 0.2)])
def test_every_box_regression_loss_of_the_reference(ops, box_loss, aux_beta):
    """BBOX_REG_LOSS_TYPE "smooth_l1" | "iou" | "giou" | "diou" | "ciou" with SMOOTH_L1_BETA, and CTR_ / IOU_SMOOTH_L1_BETA
    (box_regression_w_iou.py:13-85, classification_free_rpn.py:466-481, osrcnn_fast_rcnn.py:330-368) for the CF-RPN and the RoI box
    head: loss values against the oracle (1e-5) and gradients against autograd over the oracle (1e-4)."""
    import tests.test_train_fwd as TF
    # ---- CF-RPN ----
    shapes, strides, sizes = [(24, 40), (12, 20), (6, 10), (3, 5)], (4, 8, 16, 32), (32, 64, 128, 256)
    c = TF._rpn_case(ops, 161, shapes, strides, sizes, 2, (96, 160), [5, 3])
    lr, lo, mb, ct = TF._check_rpn_targets(ops, c)
    n, gg = c["n"], g(162)
    rows = [n * h * w for h, w in shapes]
    o = torch.randn(sum(rows), 5, generator=gg) * 0.8 + torch.tensor([0.5, 0.5, 0.5, 0.5, 0.0])
    if box_loss[0] == "ciou":  # (a box that the ReLU of apply_deltas collapses to zero width AND height has no aspect ratio: atan(0/0) is NaN in
        o[:, :4] = o[:, :4].abs() + 0.05  # the reference's ciou_loss as well; keep the predicted boxes non-degenerate for this loss)

    def image_major(x):
        dl, cl, off = [], [], 0
        for (h, w), r in zip(shapes, rows):
            dl.append(x[off:off + r, :4].reshape(n, h * w, 4))
            cl.append(torch.sigmoid(x[off:off + r, 4]).view(n, h * w))
            off += r
        return torch.cat(dl, 1), torch.cat(cl, 1)
    o2 = o.clone().requires_grad_(True)
    d_im, c_im = image_major(o2)
    ref = O.rpn_losses(c["anchors"], d_im, c_im, lr.cpu(), lo.cpu(), mb.cpu(), ct.cpu(), box_loss=box_loss, ctr_beta=aux_beta)
    (ref["loss_rpn_loc"] + ref["loss_rpn_ctr"]).backward()
    deltas_lm, ctr_lm = o[:, :4].contiguous().to(DEV), torch.sigmoid(o[:, 4]).contiguous().to(DEV)
    got = ops.rpn_losses_fwd(c["lv"], c["cell"], n, deltas_lm, ctr_lm, lr, lo, mb, ct, box_loss=box_loss, ctr_beta=aux_beta).cpu()
    assert float(got[0]) == pytest.approx(float(ref["loss_rpn_loc"]), rel=1e-5, abs=1e-7)
    assert float(got[1]) == pytest.approx(float(ref["loss_rpn_ctr"]), rel=1e-5, abs=1e-7)
    d5 = ops.rpn_losses_bwd(c["lv"], c["cell"], n, deltas_lm, ctr_lm, lr, lo, mb, ct, loss_scale=16.0, box_loss=box_loss, ctr_beta=aux_beta)
    assert rel(d5 / 16.0, o2.grad) < 1e-4
    assert int((o2.grad[:, :4].abs().sum(1) > 0).sum()) > 10 and int((o2.grad[:, 4].abs() > 0).sum()) > 20
    # ---- RoI box head ----
    gg = g(171)
    m, K, NC = 512, 20, 81
    cls = torch.randint(0, K, (m,), generator=gg)
    cls[torch.rand(m, generator=gg) < 0.5] = NC
    cls[m - 20:] = -1
    prop = torch.rand(m, 4, generator=gg) * 300
    prop[:, 2:] = prop[:, :2] + 8 + torch.rand(m, 2, generator=gg) * 200
    gtb = prop + torch.randn(m, 4, generator=gg) * 6
    gtb[:, 2:] = torch.max(gtb[:, 2:], gtb[:, :2] + 2)
    gi = torch.rand(m, generator=gg)
    pred = torch.randn(m, 5, generator=gg)
    pred[::7, 2] = 30.0  # beyond the scale clamp of Box2BoxTransform.apply_deltas: zero gradient there for the decoded-box losses
    ok = cls >= 0
    p = pred.clone().requires_grad_(True)
    lb, li = O.roi_box_losses(p[ok][:, :4], torch.sigmoid(p[ok][:, 4]), prop[ok], gtb[ok], cls[ok], gi[ok], NC, box_loss=box_loss, iou_beta=aux_beta)
    (lb + li).backward()
    pd = pred.to(DEV)
    out3 = ops.roi_box_losses_fwd(pd[:, :4], pd[:, 4], prop.to(DEV), gtb.to(DEV), cls.to(DEV), gi.to(DEV), NC, iou_is_logit=True, box_loss=box_loss,
                                  iou_beta=aux_beta).cpu()
    assert float(out3[0]) == pytest.approx(float(lb), rel=2e-5) and float(out3[1]) == pytest.approx(float(li), rel=2e-5)
    dp = ops.roi_box_losses_bwd(pd, prop.to(DEV), gtb.to(DEV), cls.to(DEV), gi.to(DEV), NC, loss_scale=8.0, box_loss=box_loss, iou_beta=aux_beta)
    assert rel(dp / 8.0, p.grad) < 1e-4


def test_loss_type_is_validated(ops):
    from openset_rcnn_amd.host.engine import check_supported_losses
    from openset_rcnn_amd.host.ops import OsrError
    check_supported_losses(dict(loss_types=dict(rpn_box=("ciou", 0.0), roi_box=("giou", 0.0), rpn_ctr=("smooth_l1", 0.2), roi_iou=("smooth_l1", 0.1))))
    with pytest.raises(NotImplementedError):
        check_supported_losses(dict(loss_types=dict(rpn_box=("l2", 0.0))))
    with pytest.raises(NotImplementedError):
        check_supported_losses(dict(loss_types=dict(roi_iou=("giou", 0.0))))
    with pytest.raises(OsrError):
        ops._loss_options(("huber", 0.0), 0.0)


@pytest.mark.parametrize("distance,reps,d", [("COS", 1, 256), ("COS", 3, 256), ("L1", 1, 256), ("L2", 1, 256), ("L1", 2, 128), ("L2", 5, 256)])
def test_pln_loss_distances_and_prototypes_per_class(ops, distance, reps, d):
    """MODEL.PLN.DISTANCE_TYPE 'COS' | 'L1' | 'L2' and REPS_PER_CLASS >= 1 (prototype_learning_network.py:149-187: a class's distance
    is the minimum over its prototypes; the prototype term excludes the own-class block): loss against the oracle (1e-5), gradients
    w.r.t. the embeddings and the raw prototypes against autograd (1e-4), inference classes / distances (pln_tail) against the oracle."""
    gg = g(211 + reps)
    m, K, NC = 384, 20, 81
    cls = torch.randint(0, K, (m,), generator=gg)
    cls[torch.rand(m, generator=gg) < 0.5] = NC
    cls[m - 12:] = -1
    gi = torch.rand(m, generator=gg)
    protos = (torch.randn(K * reps, d, generator=gg) * 1.5).requires_grad_(True)
    anchor = protos.detach()[(cls.clamp(0, K - 1) * reps + torch.randint(0, reps, (m,), generator=gg))]
    emb = (torch.randn(m, d, generator=gg) + 0.9 * anchor).requires_grad_(True)
    # hinge constants that leave all three terms active for every distance (the distances' scales differ)
    with torch.no_grad():
        dd = O.pln_distance(F.normalize(emb), F.normalize(protos), distance)
        alpha, beta = float(dd.min(dim=1)[0].median()) * 0.9, float(dd.median()) * 1.05
    ok = cls >= 0
    loss = O.pln_loss_terms(F.normalize(emb[ok]), F.normalize(protos), cls[ok], gi[ok], alpha, beta, K, 0.5, reps, distance) * 0.5 / float(ok.sum())
    loss.backward()
    got = ops.pln_loss_fwd(emb.detach().to(DEV), F.normalize(protos.detach()).to(DEV), cls.to(DEV), gi.to(DEV), 0.5, alpha, beta, 0.5, reps=reps,
                           distance=distance)
    assert float(got) == pytest.approx(float(loss), rel=2e-5)
    de, dp = ops.pln_loss_bwd(emb.detach().to(DEV), protos.detach().to(DEV), cls.to(DEV), gi.to(DEV), 0.5, alpha, beta, 0.5, loss_scale=32.0, reps=reps,
                              distance=distance)
    assert float(emb.grad.abs().sum()) > 0 and float(protos.grad.abs().sum()) > 0
    assert rel(de / 32.0, emb.grad) < 1e-4
    assert rel(dp / 32.0, protos.grad) < 1e-4
    # inference: nearest class over min-over-reps distances, unknown beyond the threshold
    md, mi = dd.reshape(m, K, reps).min(dim=2)[0].min(dim=1)
    thr = float(md.median())
    pc, mind = ops.pln_tail(emb.detach().to(DEV), F.normalize(protos.detach()).to(DEV), K, reps, thr, 80, distance=distance)
    want = mi.clone()
    want[md > thr] = 80
    near = (md - thr).abs() < 1e-5 * max(thr, 1.0)
    assert torch.equal(pc.cpu()[~near], want[~near])
    assert rel(mind, md) < 1e-5
