"""Multi-level launches (include/osr.h: osr_conv2d_fwd_levels, osr_cfrpn_head_fwd_levels): the FPN's output convolutions ([d2] FPN.forward,
Base-RCNN-FPN.yaml:3-8) and ClsFreeRPNHead.forward's loop over the pyramid (classification_free_rpn.py:157-161) as ONE grid of 256 x 256
tiles with a level table. Every tile runs the single-level kernel's code on its level's fields and every kernel of the family walks the K
slices in the same order per output element, so each level's result must equal the per-level launch BIT FOR BIT (osr_conv2d_fwd without
its split-K tail, osr_cfrpn_head_fwd on either of its tiles) -- at pyramid shapes with ragged last tiles per level, levels smaller than
one tile, one and two column tiles, both storage dtypes."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops(osr):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    osr._lib.load()
    return osr.ops


def _pyramid(g, n, shapes, cin, dtype):
    return [(torch.randn(n, h, w, cin, generator=g) * 0.5).to(dtype).to(DEV) for h, w in shapes]


# (n, level shapes, cin, cout, k): the bench's pyramid at batch 2; a 1 x 1 layer with two K slices and two column tiles; a single level
CASES = [(2, [(200, 336), (100, 168), (50, 84), (25, 42)], 256, 256, 3),
         (3, [(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)], 256, 256, 3),
         (2, [(64, 65), (7, 9), (1, 1)], 128, 512, 1),
         (1, [(40, 52)], 192, 256, 3)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_conv_levels_equal_the_per_level_launches(ops, case, relu, dtype):
    n, shapes, cin, cout, k = case
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + k)
    xs = _pyramid(g, n, shapes, cin, dtype)
    wt = (torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin)).to(dtype).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    outs = ops.conv2d_levels(xs, wt, b, relu=relu)
    assert outs is not None and len(outs) == len(xs)
    prev = ops.SPLIT_K_TAIL
    ops.SPLIT_K_TAIL = False  # (the split-K tail of a single-level launch adds its K ranges separately: not the same sum order)
    try:
        for x, o in zip(xs, outs):
            ref = ops.conv2d(x, wt, b, 1, k // 2, relu=relu)
            assert o.shape == ref.shape and torch.equal(o, ref)
    finally:
        ops.SPLIT_K_TAIL = prev
    # against an fp32 convolution of the same rounded operands
    import torch.nn.functional as F
    x0 = xs[-1].float().permute(0, 3, 1, 2)
    ref = F.conv2d(x0, wt.float().permute(0, 3, 1, 2), b, padding=k // 2)
    if relu:
        ref = ref.relu()
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    assert (outs[-1].float().permute(0, 3, 1, 2) - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_levels_with_one_weight_per_level(ops, dtype):
    """The FPN has one output conv per level: every level brings its own weights and bias."""
    g = torch.Generator().manual_seed(77)
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    xs = _pyramid(g, 2, shapes, 256, dtype)
    ws = [(torch.randn(256, 3, 3, 256, generator=g) / 48.0).to(dtype).to(DEV) for _ in shapes]
    bs = [torch.randn(256, generator=g).to(DEV) for _ in shapes]
    outs = ops.conv2d_levels(xs, ws, bs)
    assert outs is not None
    prev = ops.SPLIT_K_TAIL
    ops.SPLIT_K_TAIL = False
    try:
        for x, w_, b_, o in zip(xs, ws, bs, outs):
            assert torch.equal(o, ops.conv2d(x, w_, b_, 1, 1))
    finally:
        ops.SPLIT_K_TAIL = prev


def test_conv_levels_writes_into_given_buffers_and_refuses_what_it_cannot_take(ops):
    g = torch.Generator().manual_seed(5)
    xs = _pyramid(g, 2, [(20, 30), (10, 15)], 256, torch.float16)
    wt = (torch.randn(256, 3, 3, 256, generator=g) / 48.0).half().to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    outs = [torch.full((2, h, w, 256), 7.0, dtype=torch.float16, device=DEV) for h, w in [(20, 30), (10, 15)]]
    got = ops.conv2d_levels(xs, wt, b, outs=outs)
    assert got is not None and all(a.data_ptr() == o.data_ptr() for a, o in zip(got, outs))
    assert torch.equal(outs[1], ops.conv2d(xs[1], wt, b, 1, 1))
    # cout not a multiple of 256 / a single K slice: outside the envelope, nothing launched, the caller takes the per-level path
    w128 = (torch.randn(128, 3, 3, 256, generator=g) / 48.0).half().to(DEV)
    assert ops.conv2d_levels(xs, w128, torch.zeros(128, device=DEV)) is None
    x64 = _pyramid(g, 1, [(8, 8)], 64, torch.float16)
    assert ops.conv2d_levels(x64, (torch.randn(256, 1, 1, 64, generator=g) / 8.0).half().to(DEV), b) is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shapes", [[(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)], [(31, 45), (16, 23), (8, 12), (4, 6), (2, 3)]])
def test_cfrpn_head_levels_equal_the_per_level_fused_head(ops, shapes, dtype):
    n = 2
    g = torch.Generator().manual_seed(len(shapes) * 100 + shapes[0][0])
    xs = _pyramid(g, n, shapes, 256, dtype)
    wt = (torch.randn(256, 3, 3, 256, generator=g) / 48.0).to(dtype).to(DEV)
    b = (torch.randn(256, generator=g) * 0.1).to(DEV)
    wtail = (torch.randn(5, 256, generator=g) * 0.05).to(DEV)
    btail = (torch.randn(5, generator=g) * 0.1).to(DEV)
    rows = [n * h * w for h, w in shapes]
    total = sum(rows)
    deltas = torch.full((total, 4), float("nan"), device=DEV)
    ctr = torch.full((total,), float("nan"), device=DEV)
    hid = torch.zeros((total, 256), dtype=dtype, device=DEV)
    offs = [sum(rows[:i]) for i in range(len(rows))]
    ok = ops.cfrpn_head_fused_levels(xs, wt, b, wtail, btail, [deltas[o:o + r] for o, r in zip(offs, rows)], [ctr[o:o + r] for o, r in zip(offs, rows)],
                                     [hid[o:o + r] for o, r in zip(offs, rows)])
    assert ok
    assert torch.isfinite(deltas).all() and torch.isfinite(ctr).all()  # every row of every level was written
    for x, o, r in zip(xs, offs, rows):
        h_ref = torch.empty((r, 256), dtype=dtype, device=DEV)
        d_ref, c_ref = ops.cfrpn_head_fused(x, wt, b, wtail, btail, hidden_out=h_ref)
        assert torch.equal(deltas[o:o + r], d_ref) and torch.equal(ctr[o:o + r], c_ref) and torch.equal(hid[o:o + r], h_ref)
    # without hidden outputs: the same deltas / centerness
    d2, c2 = torch.empty_like(deltas), torch.empty_like(ctr)
    assert ops.cfrpn_head_fused_levels(xs, wt, b, wtail, btail, [d2[o:o + r] for o, r in zip(offs, rows)], [c2[o:o + r] for o, r in zip(offs, rows)])
    assert torch.equal(d2, deltas) and torch.equal(c2, ctr)


def test_levels_repeat_screen_under_a_second_stream(ops):
    """Same inputs, many launches, a second stream hammering memory: every result equals the first (the level table changes which rows a
    tile's LDS-DMA units fetch, not the ring's counted waits -- a unit landing late would show as a changed tile)."""
    g = torch.Generator().manual_seed(11)
    xs = _pyramid(g, 2, [(100, 168), (50, 84), (25, 42), (13, 21)], 256, torch.float16)
    wt = (torch.randn(256, 3, 3, 256, generator=g) / 48.0).half().to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    first = [o.clone() for o in ops.conv2d_levels(xs, wt, b)]
    hammer = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    for it in range(12):
        with torch.cuda.stream(side):
            hammer.add_(1)
        outs = ops.conv2d_levels(xs, wt, b)
        for a, f in zip(outs, first):
            assert torch.equal(a, f), f"launch {it} differs"
    torch.cuda.synchronize()


# ---- two convolutions of one input in one launch (osr_conv2d_fwd_pair: shortcut + conv1 of a stage's first bottleneck) ------------------
PAIR_CASES = [  # (n, h, w, cin, cout_a, cout_b, k, stride): res3.0 / res4.0 / res5.0 geometry, a stride-1 pair, a 3 x 3 pair, ragged M tiles
    (2, 40, 56, 256, 512, 128, 1, 2), (2, 21, 29, 512, 1024, 256, 1, 2), (1, 13, 19, 1024, 2048, 512, 1, 2),
    (2, 17, 23, 256, 256, 128, 1, 1), (1, 15, 22, 128, 128, 256, 3, 1)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", PAIR_CASES)
def test_conv_pair_equals_the_two_launches(ops, case, dtype):
    n, h, w, cin, ca, cb, k, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).to(dtype).to(DEV)
    wa = (torch.randn(ca, k, k, cin, generator=g) / math.sqrt(k * k * cin)).to(dtype).to(DEV)
    wb = (torch.randn(cb, k, k, cin, generator=g) / math.sqrt(k * k * cin)).to(dtype).to(DEV)
    ba, bb = torch.randn(ca, generator=g).to(DEV), torch.randn(cb, generator=g).to(DEV)
    got = ops.conv2d_pair(x, wa, ba, False, wb, bb, True, stride, k // 2)
    assert got is not None
    ya, yb = got
    ra = ops.conv2d(x, wa, ba, stride, k // 2, relu=False)
    rb = ops.conv2d(x, wb, bb, stride, k // 2, relu=True)
    assert ya.shape == ra.shape and yb.shape == rb.shape
    assert torch.equal(ya, ra) and torch.equal(yb, rb)
    assert float(yb.min()) >= 0.0 and float(ya.min()) < 0.0  # the ReLU flag is per convolution


def test_conv_pair_refuses_what_it_cannot_take(ops):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1, 8, 8, 256, generator=g)).half().to(DEV)
    w64 = (torch.randn(64, 1, 1, 256, generator=g) / 16).half().to(DEV)
    w128 = (torch.randn(128, 1, 1, 256, generator=g) / 16).half().to(DEV)
    assert ops.conv2d_pair(x, w128, torch.zeros(128, device=DEV), False, w64, torch.zeros(64, device=DEV), True) is None  # 64 output channels
