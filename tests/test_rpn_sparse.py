"""GPU tests of the sparse backward of the CF-RPN head's 3x3 convolution (csrc/osr_rpn_sparse.hip): the loss of ClsFreeRPN
(classification_free_rpn.py:446-490) reaches the sampled anchors only, so the head's backward runs on the rows that carry a
gradient. Checked: the row list (bit-exact index work), the gathered im2col rows (bit-exact copies), and the three results of the
chain -- weight gradient, bias gradient, data gradient joined with an existing feature gradient -- against torch autograd of the
dense 3x3 convolution on the CPU and against the trainer's dense launches."""
import pytest
import torch
import torch.nn.functional as F

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


SHAPES = ((24, 40), (12, 20), (6, 10), (3, 5), (2, 3))
STRIDES = (4, 8, 16, 32, 64)


def sparse_d5(gen, rows, k):
    d5 = torch.zeros(rows, 5)
    idx = torch.randperm(rows, generator=gen)[:k]
    d5[idx] = torch.randn(k, 5, generator=gen) * 0.05
    d5[idx[0]] = torch.tensor([0.0, 0.0, 0.0, 0.0, 1e-3])   # only the centerness term
    d5[idx[1]] = torch.tensor([0.0, -0.0, 0.0, 0.0, 0.0])   # a sampled row whose gradient is exactly zero: not listed
    return d5, idx


def test_sparse_rows_list_and_map(ops):
    gen = g(1)
    rows = 2 * sum(h * w for h, w in SHAPES) + 3
    d5, idx = sparse_d5(gen, rows, 300)
    want = torch.nonzero((d5 != 0).any(dim=1)).squeeze(1).int()
    assert len(want) == 299
    ids, rmap, cnt = ops.rpn_sparse_rows(d5.to(DEV), 512)
    ids, rmap, cnt = ids.cpu(), rmap.cpu(), cnt.cpu()
    assert cnt.tolist() == [299, 299]
    assert torch.equal(ids[:299], want) and bool((ids[299:] == -1).all())
    inv = torch.full((rows,), -1, dtype=torch.int32)
    inv[want.long()] = torch.arange(299, dtype=torch.int32)
    assert torch.equal(rmap, inv)
    # a list shorter than what is found: the first cap rows, the rest dropped and reported
    ids, rmap, cnt = ops.rpn_sparse_rows(d5.to(DEV), 100)
    assert cnt.cpu().tolist() == [100, 299] and torch.equal(ids.cpu(), want[:100])
    assert int((rmap.cpu() >= 0).sum()) == 100
    # nothing to list
    ids, rmap, cnt = ops.rpn_sparse_rows(torch.zeros(1000, 5, device=DEV), 64)
    assert cnt.cpu().tolist() == [0, 0] and bool((ids.cpu() == -1).all()) and bool((rmap.cpu() == -1).all())
    # a NaN gradient is listed (it has to reach the overflow check)
    d5n = torch.zeros(500, 5)
    d5n[17, 2] = float("nan")
    ids, _, cnt = ops.rpn_sparse_rows(d5n.to(DEV), 8)
    assert cnt.cpu().tolist() == [1, 1] and int(ids[0]) == 17


def pyramid(gen, n, c=256):
    return [(torch.randn(n, h, w, c, generator=gen) * 0.5).half() for h, w in SHAPES]


def row_position(n, rid):
    off = 0
    for l, (h, w) in enumerate(SHAPES):
        if rid < off + n * h * w:
            local = rid - off
            return l, local // (h * w), (local % (h * w)) // w, local % w
        off += n * h * w
    raise AssertionError


def test_gather_cols_is_the_im2col_row(ops):
    gen = g(2)
    n = 2
    feats = pyramid(gen, n)
    rows = n * sum(h * w for h, w in SHAPES)
    d5, _ = sparse_d5(gen, rows, 200)
    d5[0, 0] = 0.3            # the first pixel of p2: taps above / left of the map
    d5[rows - 1, 4] = -0.2    # the last pixel of p6
    lv = ops.make_rpn_levels(SHAPES, STRIDES, n, 1)
    ids, rmap, cnt = ops.rpn_sparse_rows(d5.to(DEV), 256)
    cols, d5r = ops.rpn_gather_cols(lv, [f.to(DEV) for f in feats], n, ids, d5.to(DEV))
    ids, cols, d5r = ids.cpu(), cols.cpu().view(256, 9, 256), d5r.cpu()
    k = int(cnt[0])
    assert k == 201 and int(ids[0]) == 0 and int(ids[k - 1]) == rows - 1
    assert float(cols[k:].abs().max()) == 0.0 and float(d5r[k:].abs().max()) == 0.0
    assert torch.equal(d5r[:k], d5[ids[:k].long()])
    for j in list(range(0, k, 7)) + [k - 1]:
        l, img, y, x = row_position(n, int(ids[j]))
        h, w = SHAPES[l]
        for tap in range(9):
            yy, xx = y + tap // 3 - 1, x + tap % 3 - 1
            want = feats[l][img, yy, xx] if 0 <= yy < h and 0 <= xx < w else torch.zeros(256, dtype=torch.float16)
            assert torch.equal(cols[j, tap], want), (j, tap)


def dense_reference(feats, w, b, w_tail, d5, base, n):
    """Autograd of t = relu(conv3x3(p_l)), u = t / ||t||, o = u . w_tail^T over every level with d o = d5 (fp32, CPU): d W, d b, and
    base_l + d p_l."""
    ws = w.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)  # (cout, cin, kh, kw)
    bs = b.clone().requires_grad_(True)
    fl = [f.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True) for f in feats]
    outs = []
    for f in fl:
        t = F.relu(F.conv2d(f, ws, bs, padding=1)).half().float()  # (rounded as stored; the cast passes the gradient through)
        u = F.normalize(t.permute(0, 2, 3, 1).reshape(-1, 256), dim=1)
        outs.append(u @ w_tail.t())
    (torch.cat(outs) * d5).sum().backward()
    dps = [bl.float() + f.grad.permute(0, 2, 3, 1) for bl, f in zip(base, fl)]
    return ws.grad.permute(0, 2, 3, 1).contiguous(), bs.grad, dps


def sparse_chain(ops, feats, w, b, w_tail, d5, base, n, cap):
    lv = ops.make_rpn_levels(SHAPES, STRIDES, n, 1)
    fd = [f.to(DEV) for f in feats]
    d5d = d5.to(DEV)
    ids, rmap, cnt = ops.rpn_sparse_rows(d5d, cap)
    cols, d5r = ops.rpn_gather_cols(lv, fd, n, ids, d5d)
    w3 = w.to(DEV).view(256, 2304)
    t_rows = ops.linear(cols, w3, b.to(DEV), relu=True)
    dt_rows, dw_tail, db_tail = ops.cfrpn_tail_bwd(t_rows, w_tail.to(DEV), d5r)
    dw = ops.conv2d_wgrad(cols.view(1, cap, 1, 2304), dt_rows.view(1, cap, 1, 256), 1, 1).view(256, 3, 3, 256)
    db = ops.bias_grad(dt_rows)
    y = ops.linear(dt_rows, w3.t().contiguous(), torch.zeros(2304, device=DEV), out_dtype=torch.float32)
    grads = [bl.clone().to(DEV) for bl in base]
    ops.rpn_scatter_cols_add_(lv, n, rmap, y, grads)
    return dw.cpu(), db.cpu(), [x.cpu() for x in grads], (t_rows.cpu(), ids.cpu(), int(cnt[0]), dw_tail.cpu(), db_tail.cpu())


def test_sparse_chain_matches_autograd_of_the_dense_convolution(ops):
    gen = g(3)
    n, cap = 2, 512
    feats = pyramid(gen, n)
    w = (torch.randn(256, 3, 3, 256, generator=gen) * 0.02).half()
    b = torch.randn(256, generator=gen) * 0.1
    w_tail = torch.randn(5, 256, generator=gen) * 0.3
    rows = n * sum(h * w_ for h, w_ in SHAPES)
    d5, _ = sparse_d5(gen, rows, 330)
    # neighbours in a row / column / diagonal, image borders and corners: the col2im sums several taps into one pixel
    for r in (0, 1, 2, 40, 41, 81, 39, 24 * 40 - 1, 24 * 40, rows - 1, rows - 2, rows - 4):
        d5[r] = torch.randn(5, generator=gen) * 0.05
    base = [(torch.randn(n, h, w_, 256, generator=gen) * 0.01).half() for h, w_ in SHAPES]
    base[4].zero_()
    dw_ref, db_ref, dp_ref = dense_reference(feats, w, b, w_tail, d5, base, n)
    dw, db, dps, (t_rows, ids, k, _, _) = sparse_chain(ops, feats, w, b, w_tail, d5, base, n, cap)
    assert k == int((d5 != 0).any(dim=1).sum())
    # the recomputed hidden state of the listed rows is the dense convolution's
    for j in (0, 5, k - 1):
        l, img, y, x = row_position(n, int(ids[j]))
        t_ref = F.relu(F.conv2d(feats[l][img:img + 1].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1))[0, :, y, x]
        assert float((t_rows[j].float() - t_ref).abs().max()) <= 2e-3 * max(1.0, float(t_ref.abs().max()))
    assert float((dw - dw_ref).abs().max()) <= 2e-3 * float(dw_ref.abs().max())
    assert float((db - db_ref).abs().max()) <= 2e-3 * float(db_ref.abs().max())
    touched = 0
    for l in range(5):
        got, want = dps[l].float(), dp_ref[l]
        assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 1e-6, l
        same = (got == base[l].float()).all(dim=3)
        touched += int((~same).sum())
        # pixels no listed anchor reaches keep their bits
        reach = torch.zeros(n, *SHAPES[l], dtype=torch.bool)
        for j in range(k):
            ll, img, y, x = row_position(n, int(ids[j]))
            if ll == l:
                reach[img, max(0, y - 1):y + 2, max(0, x - 1):x + 2] = True
        assert bool(same[~reach].all()), l
    assert touched > k


def test_sparse_chain_is_reproducible_and_handles_an_empty_list(ops):
    gen = g(4)
    n, cap = 2, 512
    feats = pyramid(gen, n)
    w = (torch.randn(256, 3, 3, 256, generator=gen) * 0.02).half()
    b = torch.randn(256, generator=gen) * 0.1
    w_tail = torch.randn(5, 256, generator=gen) * 0.3
    rows = n * sum(h * w_ for h, w_ in SHAPES)
    d5, _ = sparse_d5(gen, rows, 400)
    base = [(torch.randn(n, h, w_, 256, generator=gen) * 0.01).half() for h, w_ in SHAPES]
    a = sparse_chain(ops, feats, w, b, w_tail, d5, base, n, cap)
    c = sparse_chain(ops, feats, w, b, w_tail, d5, base, n, cap)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and all(torch.equal(x, y) for x, y in zip(a[2], c[2]))
    z = sparse_chain(ops, feats, w, b, w_tail, torch.zeros(rows, 5), base, n, cap)
    assert float(z[0].abs().max()) == 0.0 and float(z[1].abs().max()) == 0.0 and all(torch.equal(x, y) for x, y in zip(z[2], base))
    assert float(z[3][3].abs().max()) == 0.0


def test_trainer_sparse_and_dense_rpn_backward_agree(osr):
    """The whole training step with the head's backward on the listed rows against the dense launches of rounds 1-3: same batch,
    same weights, every gradient tensor agrees to the fp16 storage of the data gradients (the hidden state of the listed rows is
    recomputed with another K order than the fused head kernel's, so single fp16 roundings may differ)."""
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params
    from oracle import osr_oracle as O
    params = random_params(0)
    gen = g(23)
    n, h, w, gmax = 2, 128, 160, 4
    images = torch.randint(0, 256, (n, 3, h, w), generator=gen, dtype=torch.uint8)
    gt = torch.zeros(n, gmax, 4)
    gcls = torch.zeros(n, gmax, dtype=torch.int64)
    for i, c in enumerate((3, 2)):
        ctr = torch.rand(c, 2, generator=gen) * torch.tensor([w * 0.7, h * 0.7]) + 16
        size = torch.rand(c, 2, generator=gen) * 60 + 24
        bx = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
        bx[:, 0::2].clamp_(0, w)
        bx[:, 1::2].clamp_(0, h)
        gt[i, :c] = bx
        gcls[i, :c] = torch.randint(0, 20, (c,), generator=gen)
    shapes = O.level_shapes(h, w)
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = dict(rpn_reg=torch.rand(n, r, generator=gen), rpn_obj=torch.rand(n, r, generator=gen), roi=torch.rand(n, cap + gmax, generator=gen))
    args = (images.to(DEV), torch.tensor([(h, w)] * n, dtype=torch.int32).to(DEV), h, w, gt.to(DEV), gcls.to(DEV),
            torch.tensor([3, 2], dtype=torch.int32).to(DEV), {k: v.to(DEV) for k, v in keys.items()})
    grads, losses = {}, {}
    for sparse in (True, False):
        tr = OpensetRCNNTrainer(params, dtype=torch.float16, device=DEV, lr=5e-5, loss_scale=512.0)
        tr.sparse_rpn_bwd = sparse
        for _ in range(2):
            tr.grad_flat.fill_(float("nan"))
            out = tr.step(*args, update=False)
        torch.cuda.synchronize()
        grads[sparse] = {k: v.clone().cpu() for k, v in tr.grad.items()}
        losses[sparse] = {k: float(v) for k, v in out.items() if k.startswith("loss")}
    assert losses[True] == losses[False]  # the forward is the same launches (only the hidden state is not written)
    for k, a in grads[True].items():
        b = grads[False][k]
        assert bool(torch.isfinite(a).all()), k
        cos = float(F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
        assert cos >= 0.9999 and abs(float(a.norm() / b.norm().clamp(min=1e-30)) - 1.0) <= 2e-3, (k, cos)


def test_sparse_rows_fuzz(ops):
    """Random sizes (around the 256-row walk and the 1024-workgroup split), densities from empty to full, caps below and above the
    count: list == torch.nonzero, map == its inverse, counts as specified -- index work, bit-exact."""
    gen = g(99)
    sizes = [1, 5, 255, 256, 257, 1023, 1024, 1025, 4099, 65536, 262144 + 7, 262144 * 4 + 3]
    for rows in sizes:
        for dens in (0.0, 0.001, 0.03, 0.5, 1.0):
            hit = torch.rand(rows, generator=gen) < dens
            d5 = torch.zeros(rows, 5)
            col = torch.randint(0, 5, (rows,), generator=gen)
            d5[torch.arange(rows)[hit], col[hit]] = 1.0 + torch.rand(int(hit.sum()), generator=gen)
            want = torch.nonzero(hit).squeeze(1).int()
            for cap in {1, max(1, len(want) // 2), len(want) + 3}:
                ids, rmap, cnt = ops.rpn_sparse_rows(d5.to(DEV), cap)
                ids, rmap, cnt = ids.cpu(), rmap.cpu(), cnt.cpu().tolist()
                k = min(cap, len(want))
                assert cnt == [k, len(want)], (rows, dens, cap, cnt)
                assert torch.equal(ids[:k], want[:k]) and bool((ids[k:] == -1).all()), (rows, dens, cap)
                inv = torch.full((rows,), -1, dtype=torch.int32)
                inv[want[:k].long()] = torch.arange(k, dtype=torch.int32)
                assert torch.equal(rmap, inv), (rows, dens, cap)
