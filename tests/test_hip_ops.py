"""GPU parity tests: every HIP entry point (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact for index / integer outputs (top-k order, kept lists, counts, classes) and for the fp32 box
arithmetic that feeds them (decode, clip); 1e-4 relative for fp32 values produced by reductions (the kernels
sum in a different order than torch-CPU); fp16/bf16 MFMA contractions are compared on identically rounded
inputs with fp32 accumulation on both sides.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_binding as CO
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


def nhwc(x):  # (N,C,H,W) -> contiguous NHWC
    return x.permute(0, 2, 3, 1).contiguous()


def assert_close(a, b, rtol=1e-4, atol=None, name=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if atol is None:
        atol = rtol * max(float(b.abs().max()), 1e-6)
    bad = (a - b).abs() > atol + rtol * b.abs()
    assert not bool(bad.any()), f"{name}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err {float((a - b).abs().max()):.3e}"


# ------------------------------------------------------------------------------------------------------
def test_preprocess(ops):
    img = torch.randint(0, 256, (2, 3, 37, 50), generator=g(0), dtype=torch.uint8)
    mean, std = (103.53, 116.28, 123.675), (1.0, 2.0, 0.5)
    for src in (img, img.float() + 0.25):
        out = ops.preprocess(src.to(DEV), 64, 64, mean, std, torch.float16).cpu()
        assert out.shape == (2, 70, 72, 4)
        ref = torch.zeros(2, 70, 72, 4)
        nrm = (src.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
        ref[:, 3:40, 3:53, :3] = nrm.permute(0, 2, 3, 1)
        assert torch.equal(out.float(), ref.half().float())


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, pad, relu, res_mode, dtype, out_dtype
    (2, 20, 28, 64, 128, 3, 1, 1, True, 0, torch.float16, torch.float32),
    (2, 20, 28, 64, 64, 3, 1, 1, True, 0, torch.float16, torch.float16),      # BN=64 tile
    (1, 23, 17, 96, 200, 1, 1, 0, False, 0, torch.float16, torch.float32),    # M and N tails
    (2, 21, 30, 128, 256, 1, 2, 0, False, 0, torch.float16, torch.float32),   # strided 1x1 (MSRA)
    (2, 14, 18, 64, 256, 1, 1, 0, True, 1, torch.float16, torch.float16),     # bottleneck shortcut add
    (1, 10, 12, 256, 256, 1, 1, 0, False, 2, torch.float16, torch.float16),   # FPN top-down upsample add
    (1, 9, 11, 32, 8, 3, 1, 1, False, 0, torch.float16, torch.float32),       # smallest legal channels
    (2, 20, 28, 64, 128, 3, 1, 1, True, 1, torch.bfloat16, torch.bfloat16),
    (1, 16, 16, 512, 512, 3, 2, 1, True, 0, torch.bfloat16, torch.float32),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(ops, case):
    n, h, w, cin, cout, k, stride, pad, relu, res_mode, dt, odt = case
    gg = g(hash(case[:8]) % 1000)
    x = (torch.randn(n, cin, h, w, generator=gg)).to(dt)
    wt = (torch.randn(cout, cin, k, k, generator=gg) / math.sqrt(cin * k * k)).to(dt)
    b = torch.randn(cout, generator=gg)
    ref = F.conv2d(x.float(), wt.float(), b, stride=stride, padding=pad)
    res = None
    if res_mode == 1:
        res = torch.randn(ref.shape, generator=gg).to(dt)
        ref = ref + res.float()
    elif res_mode == 2:
        res = torch.randn(n, cout, (ref.shape[2] + 1) // 2, (ref.shape[3] + 1) // 2, generator=gg).to(dt)
        ref = ref + F.interpolate(res.float(), scale_factor=2.0, mode="nearest")[:, :, : ref.shape[2], : ref.shape[3]]
    if relu:
        ref = F.relu(ref)
    out = ops.conv2d(nhwc(x).to(DEV), wt.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), stride=stride, pad=pad, relu=relu,
                     residual=None if res is None else nhwc(res).to(DEV), res_mode=res_mode, out_dtype=odt)
    out = out.cpu().float().permute(0, 3, 1, 2)
    if odt == torch.float32:
        assert_close(out, ref, rtol=1e-4, name="conv fp32 out")
    else:
        eps = 2.0 ** -10 if odt == torch.float16 else 2.0 ** -7  # one rounding of the stored result
        assert_close(out, ref, rtol=eps, atol=eps * float(ref.abs().max()) * 0.01 + 1e-6, name="conv half out")


@pytest.mark.parametrize("cin,odt", [(64, torch.float16), (128, torch.float32), (32, torch.float16)])
def test_conv2d_post_mask(ops, cin, odt):
    """osr_conv2d_fwd_masked: (conv + bias + residual) zeroed where the mask tensor is <= 0, one launch on the BK=64 kernel
    (cin % 64 == 0); cin = 32 takes the documented two-launch route (OSR_ERR_UNSUPPORTED -> conv + relu_mask)."""
    gg = g(300 + cin)
    n, h, w, cout, k = 2, 19, 23, 64, 3
    x = torch.randn(n, cin, h, w, generator=gg).half()
    wt = (torch.randn(cout, cin, k, k, generator=gg) / math.sqrt(cin * k * k)).half()
    b = torch.randn(cout, generator=gg)
    res = torch.randn(n, cout, h, w, generator=gg).half()
    mask = torch.randn(n, cout, h, w, generator=gg).half()
    mask[0, :, 3, :] = 0.0  # exactly zero: masked (the ReLU output was not positive)
    ref = (F.conv2d(x.float(), wt.float(), b, padding=1) + res.float()) * (mask.float() > 0)
    out = ops.conv2d(nhwc(x).to(DEV), wt.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), pad=1, residual=nhwc(res).to(DEV), res_mode=1,
                     out_dtype=odt, post_mask=nhwc(mask).to(DEV)).cpu().float().permute(0, 3, 1, 2)
    assert bool(((out == 0) | (mask.float() > 0)).all())  # exact zeros where masked
    if odt == torch.float32:
        assert_close(out, ref, rtol=1e-4, name="masked conv fp32 out")
    else:
        assert_close(out, ref, rtol=2.0 ** -10, atol=2.0 ** -10 * float(ref.abs().max()) * 0.01 + 1e-6, name="masked conv half out")
    # no residual
    out0 = ops.conv2d(nhwc(x).to(DEV), wt.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), pad=1, out_dtype=torch.float32,
                      post_mask=nhwc(mask).to(DEV)).cpu().permute(0, 3, 1, 2)
    assert_close(out0, F.conv2d(x.float(), wt.float(), b, padding=1) * (mask.float() > 0), rtol=1e-4, name="masked conv, no residual")


def test_stem_conv_matches_7x7(ops):
    gg = g(5)
    img = torch.randint(0, 256, (2, 3, 61, 90), generator=gg, dtype=torch.uint8)
    hp, wp = 64, 96
    mean, std = (103.53, 116.28, 123.675), (1.0, 1.0, 1.0)
    w7 = torch.randn(64, 3, 7, 7, generator=gg) * 0.02
    b = torch.randn(64, generator=gg)
    xpad = ops.preprocess(img.to(DEV), hp, wp, mean, std, torch.float16)
    # stem view weight: [co][kh][0][kw*4 + c], 8th tap and 4th channel zero
    wv = torch.zeros(64, 7, 1, 32)
    wv.view(64, 7, 8, 4)[:, :, :7, :3] = w7.half().float().permute(0, 2, 3, 1)
    out = ops.stem_conv(xpad, wv.half().to(DEV), b.to(DEV), hp, wp, relu=True).cpu().float().permute(0, 3, 1, 2)
    batch, _ = O.preprocess_images([im for im in img], mean, std)
    batch = F.pad(batch, (0, wp - batch.shape[3], 0, hp - batch.shape[2]))
    ref = F.relu(F.conv2d(batch.half().float(), w7.half().float(), b, stride=2, padding=3))
    assert out.shape == ref.shape
    eps = 2.0 ** -10
    assert_close(out, ref, rtol=eps, atol=eps * float(ref.abs().max()) * 0.01 + 1e-5, name="stem")
    # the product packing: 8-row view (K = 256) -> BK=64 direct-to-LDS kernel, two taps per K slice
    from openset_rcnn_amd.host.weights import pack_stem_weight
    out8 = ops.stem_conv(xpad, pack_stem_weight(w7, torch.float16).to(DEV), b.to(DEV), hp, wp, relu=True).cpu().float().permute(0, 3, 1, 2)
    assert_close(out8, ref, rtol=eps, atol=eps * float(ref.abs().max()) * 0.01 + 1e-5, name="stem bk64")


def test_linear_large_k(ops):
    gg = g(6)
    x = torch.randn(300, 12544, generator=gg).half()
    w = (torch.randn(1024, 12544, generator=gg) / 112).half()
    b = torch.randn(1024, generator=gg)
    out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), relu=True, out_dtype=torch.float32).cpu()
    ref = F.relu(x.double() @ w.double().t() + b.double()).float()
    assert_close(out, ref, rtol=1e-4, name="fc1")


def test_maxpool_and_subsample(ops):
    x = torch.randn(2, 64, 21, 30, generator=g(7)).half()
    out = ops.maxpool3x3s2(nhwc(x).to(DEV)).cpu().permute(0, 3, 1, 2)
    assert torch.equal(out, F.max_pool2d(x.float(), 3, 2, 1).half())
    out = ops.subsample2(nhwc(x).to(DEV)).cpu().permute(0, 3, 1, 2)
    assert torch.equal(out, F.max_pool2d(x.float(), 1, 2, 0).half())


@pytest.mark.parametrize("m,n,k,relu", [(1000, 256, 1024, False), (333, 1024, 256, True), (130, 21, 1024, False), (1, 7, 16, False)])
def test_gemm_f32(ops, m, n, k, relu):
    gg = g(m + n)
    a = torch.randn(m, k, generator=gg)
    w = torch.randn(n, k, generator=gg) / math.sqrt(k)
    b = torch.randn(n, generator=gg)
    out = ops.gemm_f32(a.to(DEV), w.to(DEV), b.to(DEV), relu=relu).cpu()
    ref = a.double() @ w.double().t() + b.double()
    if relu:
        ref = F.relu(ref)
    assert_close(out, ref.float(), rtol=1e-5, name="gemm_f32")


# ------------------------------------------------------------------------------------------------------
def test_cfrpn_head_tail(ops):
    p = O.make_head_params(0)
    for dt in (torch.float32, torch.float16):
        t = F.relu(torch.randn(5000, 256, generator=g(8))).to(dt)
        t[17] = 0  # zero row: eps clamp path
        d, c = ops.cfrpn_head_tail(t.to(DEV), p["proposal_generator.rpn_head.anchor_deltas.weight"].view(4, 256).to(DEV),
                                   p["proposal_generator.rpn_head.anchor_deltas.bias"].to(DEV),
                                   p["proposal_generator.rpn_head.centerness.weight"].view(1, 256).to(DEV),
                                   p["proposal_generator.rpn_head.centerness.bias"].to(DEV))
        dr, cr = O.cfrpn_head_tail(t.float(), p)
        assert_close(d, dr, rtol=1e-4, atol=1e-5, name="deltas")
        assert_close(c, cr, rtol=1e-4, atol=1e-6, name="ctr")


def _run_select(ops, shapes, strides, sizes, n, topk, image_sizes, seed, ties=False, poison=False):
    gg = g(seed)
    anchors = O.anchor_grid(shapes, strides, sizes)
    ctr = [torch.rand(n, h * w, generator=gg) for h, w in shapes]
    if ties:
        ctr = [(c * 20).round() / 20 for c in ctr]  # many exact ties, incl. at the k-th value
    deltas = [torch.randn(n, h * w, 4, generator=gg) * 1.5 for h, w in shapes]
    if poison:
        deltas[0][0, 3, 1] = float("nan")
        ctr[0][0, 3] = 2.0
        deltas[0][n - 1, 5, 2] = float("inf")
        ctr[0][n - 1, 5] = 3.0
        deltas[1][0, 0] = torch.tensor([-1.0, -1.0, -1.0, -1.0])  # collapses to a point -> empty
        ctr[1][0, 0] = 4.0
        deltas[0][0, 7, 0] = float("-inf")  # relu(-inf) = 0: stays valid
        ctr[0][0, 7] = 5.0
    props = [O.ltrb_apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)).view(n, -1, 4)
             for d, a in zip(deltas, anchors)]
    ref = O.find_top_rpn_proposals(props, ctr, image_sizes, topk)
    lv = ops.make_rpn_levels(shapes, strides, n, 1)
    cell = torch.tensor([[[-s / 2, -s / 2, s / 2, s / 2]] for s in sizes], dtype=torch.float32)
    ctr_cat = torch.cat([c.reshape(-1) for c in ctr]).to(DEV)
    del_cat = torch.cat([d.reshape(-1, 4) for d in deltas]).to(DEV)
    hw = torch.tensor(image_sizes, dtype=torch.int32).to(DEV)
    r = ops.rpn_select(lv, cell.to(DEV), ctr_cat, del_cat, n, hw, topk)
    counts = r["counts"].cpu()
    for i, (rb, rs, ri) in enumerate(ref):
        c = int(counts[i])
        assert c == len(rb), f"image {i}: count {c} vs {len(rb)}"
        assert torch.equal(r["src_index"][i, :c].cpu().long(), ri), f"image {i}: selected anchor indices differ"
        assert torch.equal(r["scores"][i, :c].cpu(), rs)
        assert torch.equal(r["boxes"][i, :c].cpu(), rb), f"image {i}: decoded boxes not bit-exact"
        assert bool((r["batch_idx"].view(n, -1)[i, :c].cpu() == i).all()) and bool((r["batch_idx"].view(n, -1)[i, c:].cpu() == -1).all())
    return r, ref


def test_rpn_select_small_with_ties_and_poison(ops):
    shapes, strides, sizes = [(24, 40), (12, 20), (6, 10), (3, 5)], (4, 8, 16, 32), (32, 64, 128, 256)
    r, _ = _run_select(ops, shapes, strides, sizes, 3, 100, [(96, 160), (90, 150), (96, 100)], 11, ties=True, poison=True)
    assert int(r["status_flags"].cpu()[0]) != 0
    r, _ = _run_select(ops, shapes, strides, sizes, 2, 2000, [(96, 160)] * 2, 12)  # k >= level size: take all
    assert int(r["status_flags"].cpu()[0]) == 0


def test_rpn_select_many_seeds(ops):
    """Top-k order, decoded and clipped boxes and counts are bit-exact: twelve more random score / delta fields (every
    other one with ties) on a mid-size pyramid, k below and above the level sizes."""
    shapes, strides, sizes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)], (4, 8, 16, 32, 64), (32, 64, 128, 256, 512)
    for seed in range(12):
        k = (50, 300, 2000)[seed % 3]
        _run_select(ops, shapes, strides, sizes, 2, k, [(160, 256), (150, 250)], 500 + seed, ties=seed % 2 == 0)


def test_rpn_select_full_size(ops):
    shapes = O.level_shapes(800, 1344)
    r, ref = _run_select(ops, shapes, O.FPN_STRIDES, O.ANCHOR_SIZES, 2, 1000, [(800, 1333), (750, 1333)], 13, ties=True)
    assert r["cap"] == 4273
    _run_select(ops, shapes, O.FPN_STRIDES, O.ANCHOR_SIZES, 1, 2000, [(800, 1333)], 14)  # train-time k: cap 7323


# ------------------------------------------------------------------------------------------------------
def _roi_case(seed, n=2, c=256, hw=(64, 96)):
    gg = g(seed)
    h, w = hw
    feats = [torch.randn(n, c, h // s, w // s, generator=gg) for s in (4, 8, 16, 32)]
    m = 300
    ctr = torch.rand(m, 2, generator=gg) * torch.tensor([w * 1.0, h * 1.0])
    size = torch.exp(torch.rand(m, 2, generator=gg) * 6.0)  # 1 .. 400 px: all four levels, tiny and huge
    boxes = torch.cat((ctr - size / 2, ctr + size / 2), dim=1)
    boxes[0] = torch.tensor([10.0, 10.0, 10.0, 10.0])      # zero area
    boxes[1] = torch.tensor([-50.0, -40.0, 30.0, 20.0])    # crosses the top-left border
    boxes[2] = torch.tensor([0.0, 0.0, float(w), float(h)])  # whole image
    boxes[3] = torch.tensor([w - 5.0, h - 5.0, w + 300.0, h + 300.0])  # mostly outside
    bidx = torch.randint(0, n, (m,), generator=gg, dtype=torch.int32)
    bidx[7] = -1
    return feats, boxes, bidx


@pytest.mark.parametrize("dt,odt", [(torch.float32, torch.float32), (torch.float16, torch.float32), (torch.float16, torch.float16),
                                    (torch.bfloat16, torch.float32)])
def test_roi_align_vs_oracle(ops, dt, odt):
    feats, boxes, bidx = _roi_case(21)
    fq = [f.to(dt) for f in feats]
    out = ops.roi_align([nhwc(f).to(DEV) for f in fq], (0.25, 0.125, 0.0625, 0.03125), boxes.to(DEV), bidx.to(DEV), 7, odt)
    out = out.cpu().float().permute(0, 3, 1, 2)  # (m,7,7,c) -> (m,c,7,7)
    lv = O.assign_levels(boxes)
    ref = torch.zeros_like(out)
    for l, s in enumerate((0.25, 0.125, 0.0625, 0.03125)):
        ids = torch.nonzero((lv == l) & (bidx >= 0)).squeeze(1)
        rois = torch.cat((bidx[ids].float().unsqueeze(1), boxes[ids]), dim=1)
        ref[ids] = CO.roi_align(fq[l].float(), rois, s)
    assert float(out[7].abs().max()) == 0.0 and float(out[0].abs().max()) == 0.0
    if odt == torch.float32:
        assert_close(out, ref, rtol=1e-4, atol=1e-5, name="roi_align")
    else:
        assert_close(out, ref, rtol=2.0 ** -10, atol=2e-3, name="roi_align f16 out")


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_roi_align_every_stream_length_and_inner_depth(ops, dt):
    """Round 6 (ra_stream: q * D steps in a rotation of D register groups + r < D steps in groups of their own, D = 2..8 by the
    pixels per step): RoIs whose footprint is 1..40 pixels along the streamed axis and 1..6 pixels per bin along the other, in both
    orientations, so that every (q, r) split of every pipeline depth and the no-rotation case (n < D) run -- against the C oracle."""
    gg = g(29)
    h, w, c = 56, 72, 16
    f = torch.randn(1, c, h, w, generator=gg).to(dt).float()
    boxes = []
    for n_out in range(1, 41):            # footprint length along the streamed (shorter or equal) side, in level pixels
        for per_bin in (0.4, 1.0, 2.3, 3.1, 4.2, 5.5):  # bin size along the other side: 1..6 pixels per bin
            long_side = min(7 * per_bin, 52.0)
            short_side = max(n_out - 1.3, 0.2)
            x0, y0 = 3.3 + (n_out % 5) * 0.37, 2.6 + (n_out % 3) * 0.41
            boxes.append([x0, y0, x0 + long_side, y0 + short_side])    # streamed along y
            boxes.append([y0, x0, y0 + short_side, x0 + long_side])    # streamed along x
    boxes = torch.tensor(boxes, dtype=torch.float32) * 4.0  # one level at stride 4 (min_level = 2): image coordinates
    boxes[:, 2].clamp_(max=w * 4.0 - 1)
    boxes[:, 3].clamp_(max=h * 4.0 - 1)
    bidx = torch.zeros(len(boxes), dtype=torch.int32)
    lib_out = ops.roi_align([nhwc(f).to(dt).to(DEV)], (0.25,), boxes.to(DEV), bidx.to(DEV), 7, torch.float32, min_level=2)
    out = lib_out.cpu().float().permute(0, 3, 1, 2)
    ref = CO.roi_align(f, torch.cat((bidx.float().unsqueeze(1), boxes), dim=1), 0.25)
    assert_close(out, ref, rtol=1e-4, atol=1e-5, name="roi_align stream sweep")


def test_roi_align_wide_bins_take_the_sample_loop(ops):
    # one coarse level only: a 600 px RoI at stride 4 has 21 px bins (> the 13 px LDS table) -> 4-tap fallback
    gg = g(22)
    f = torch.randn(1, 8, 160, 200, generator=gg)
    boxes = torch.tensor([[20.0, 30.0, 620.0, 500.0], [100.0, 100.0, 140.0, 130.0]])
    bidx = torch.zeros(2, dtype=torch.int32)
    out = ops.roi_align([nhwc(f).to(DEV)], (0.25,), boxes.to(DEV), bidx.to(DEV), 7, torch.float32, min_level=2).cpu().permute(0, 3, 1, 2)
    ref = CO.roi_align(f, torch.cat((bidx.float().unsqueeze(1), boxes), 1), 0.25)
    assert_close(out, ref, rtol=1e-4, atol=1e-5, name="roi_align fallback")


def test_roi_align_extreme_aspect_footprints(ops):
    """Every path of the RoIAlign kernel on one level: wide-thin boxes (columns streamed in chunks of six rows: footprint taller
    than six pixel rows per bin when transposed), tall-thin boxes (rows streamed), a footprint wider than 64 columns, boxes
    much smaller than a pixel (a pixel in all seven bins: per-bin loop), boxes crossing the border, a full-image box whose
    bins overflow nothing on the 64-column tables, and one on a map wide enough to overflow them (per-sample loop)."""
    gg = g(23)
    f = torch.randn(2, 12, 120, 340, generator=gg)
    boxes = torch.tensor([
        [4.0, 100.0, 1300.0, 112.0],     # 324 x 3 px footprint: rows streamed, 47-column bins
        [300.0, 2.0, 330.0, 470.0],      # 7 x 117: columns streamed, 17-row bins (chunks of 6)
        [10.0, 10.0, 700.0, 400.0],      # big box, 25 x 14 px bins
        [50.3, 60.2, 51.1, 61.0],        # a fifth of a pixel: every bin samples the same pixels
        [-40.0, -30.0, 90.0, 50.0],      # crosses the top-left border
        [1200.0, 400.0, 1400.0, 520.0],  # crosses the bottom-right border
        [0.0, 0.0, 1360.0, 480.0],       # the whole map
        [600.0, 200.0, 640.0, 203.0],    # 10 x 0.75
    ])
    bidx = torch.tensor([0, 1, 0, 1, 0, 1, 1, 0], dtype=torch.int32)
    out = ops.roi_align([nhwc(f).to(DEV)], (0.25,), boxes.to(DEV), bidx.to(DEV), 7, torch.float32, min_level=2).cpu().permute(0, 3, 1, 2)
    ref = CO.roi_align(f, torch.cat((bidx.float().unsqueeze(1), boxes), 1), 0.25)
    assert_close(out, ref, rtol=1e-4, atol=1e-5, name="roi_align extreme aspect")
    # a map wider than 64 * 7 columns: the whole-map box has 70-column bins -> table overflow -> per-sample loop
    f2 = torch.randn(1, 4, 24, 500, generator=gg)
    b2 = torch.tensor([[0.0, 0.0, 2000.0, 96.0], [8.0, 8.0, 1900.0, 20.0]])
    z = torch.zeros(2, dtype=torch.int32)
    out2 = ops.roi_align([nhwc(f2).to(DEV)], (0.25,), b2.to(DEV), z.to(DEV), 7, torch.float32, min_level=2).cpu().permute(0, 3, 1, 2)
    assert_close(out2, CO.roi_align(f2, torch.cat((z.float().unsqueeze(1), b2), 1), 0.25), rtol=1e-4, atol=1e-5, name="roi_align overflow")


@pytest.mark.parametrize("dt", [torch.float32, torch.float16], ids=["f32", "f16"])
def test_roi_align_256_channels_every_path(ops, dt):
    """The model's own channel count (one 512-byte pixel per wave-load) through every path of the kernel: footprints streamed along
    either axis, wider than the 192-step window tables (per-bin loop), bins wider than the 32-column weight tables (per-sample
    loop), bins narrower than a pixel, footprints clipped by every border, degenerate and out-of-image boxes, padding rows."""
    gg = g(29)
    f = torch.randn(2, 256, 60, 500, generator=gg).to(dt)
    boxes = torch.tensor([
        [4.0, 100.0, 1300.0, 112.0],     # 324 x 3 px footprint: 21 chunks per row, rows in all seven bin rows -> fixed windows
        [300.0, 2.0, 330.0, 230.0],      # 7 x 57 px: one short chunk per row, tall bins, sliding window
        [10.0, 10.0, 700.0, 200.0],      # big box: 25-column bins, several chunks per row
        [50.3, 60.2, 51.1, 61.0],        # a fifth of a pixel: every bin samples the same pixels on both axes
        [-40.0, -30.0, 90.0, 50.0],      # crosses the top-left border (clamped samples, samples below -1 dropped)
        [1900.0, 200.0, 2100.0, 260.0],  # crosses the bottom-right border
        [0.0, 0.0, 2000.0, 240.0],       # the whole map: 72-column bins -> table overflow -> per-sample loop
        [600.0, 100.0, 640.0, 103.0],    # 10 x 0.75 px
        [700.0, 50.0, 764.0, 114.0],     # 16 x 16 px: exactly one chunk per row
        [701.0, 51.0, 769.5, 113.0],     # 17 columns: a second chunk holding one pixel
        [100.0, 100.0, 100.0, 100.0],    # zero area
        [5000.0, 5000.0, 5100.0, 5100.0],  # entirely outside: every sample invalid -> zeros
        [820.0, 20.0, 1020.0, 26.0],     # 50 x 1.5 px: three passes over two rows
    ])
    bidx = torch.tensor([0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1], dtype=torch.int32)
    gb = torch.rand(40, 4, generator=gg)
    ctr = gb[:, :2] * torch.tensor([2000.0, 240.0])
    size = torch.exp(gb[:, 2:] * 5.5)
    boxes = torch.cat((boxes, torch.cat((ctr - size / 2, ctr + size / 2), dim=1)))
    bidx = torch.cat((bidx, torch.randint(0, 2, (40,), generator=gg, dtype=torch.int32)))
    bidx[20] = -1
    for odt in ((torch.float32,) if dt == torch.float32 else (torch.float32, torch.float16)):
        out = ops.roi_align([nhwc(f).to(DEV)], (0.25,), boxes.to(DEV), bidx.to(DEV), 7, odt, min_level=2).cpu().float().permute(0, 3, 1, 2)
        valid = bidx >= 0
        ref = torch.zeros_like(out)
        ref[valid] = CO.roi_align(f.float(), torch.cat((bidx[valid].float().unsqueeze(1), boxes[valid]), 1), 0.25)
        assert float(out[20].abs().max()) == 0.0 and float(out[11].abs().max()) == 0.0
        if odt == torch.float32:
            assert_close(out, ref, rtol=1e-4, atol=1e-5, name="roi_align rows")
        else:
            assert_close(out, ref, rtol=2.0 ** -10, atol=2e-3, name="roi_align rows f16 out")


def test_roi_align_linear_ramp_is_exact(ops):
    ys, xs = torch.meshgrid(torch.arange(50.0), torch.arange(84.0), indexing="ij")
    f = (0.5 * xs - 0.25 * ys + 3.0).view(1, 1, 50, 84).expand(1, 4, 50, 84).contiguous()
    boxes = torch.tensor([[100.0, 80.0, 400.0, 390.0]])
    out = ops.roi_align([nhwc(f).to(DEV)], (1 / 16,), boxes.to(DEV), torch.zeros(1, dtype=torch.int32, device=DEV), 7, torch.float32,
                        min_level=4).cpu()[0, :, :, 0]
    x1, y1, x2, y2 = [(v / 16 - 0.5) for v in boxes[0].tolist()]
    for ph in range(7):
        for pw in range(7):
            cx, cy = x1 + (pw + 0.5) * (x2 - x1) / 7, y1 + (ph + 0.5) * (y2 - y1) / 7
            assert out[ph, pw].item() == pytest.approx(0.5 * cx - 0.25 * cy + 3.0, rel=1e-5)


def test_linear_split_k_tail_round(ops, osr):
    """Deep-K FC layers whose tile grid leaves a mostly empty last dispatch round are cut along K for that round (three launches:
    full rounds, ksplit partial-sum launches of the tail tiles, fixed-order reduction + epilogue). Same result as the single launch
    up to the fp32 summation order of the tail rows, reproducible bit for bit, and equal to the fp64 reference at fp32 accuracy."""
    gg = g(41)
    m, k, n = 17000, 3072, 1024  # 67 x 4 tiles of 256 x 256 (or 133 x 8 of 128 x 128) on 256 x occ slots, 48 K slices
    x = (torch.randn(m, k, generator=gg) * 0.5).half().to(DEV)
    w = (torch.randn(n, k, generator=gg) / math.sqrt(k)).half().to(DEV)
    b = torch.randn(n, generator=gg).to(DEV)
    L = osr._lib
    p = L.ConvParams()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = 1, m, 1, k, m, 1, n
    p.kh = p.kw = p.stride_h = p.stride_w = 1
    p.in_stride_n, p.in_stride_h, p.in_stride_w = m * k, k, k
    p.out_stride_n, p.out_stride_h, p.out_stride_w = m * n, n, n
    p.in_dtype = p.out_dtype = L.OSR_F16
    import ctypes
    assert L.load().osr_conv2d_fwd_workspace_bytes(ctypes.byref(p)) > 0, "this shape must qualify for the split (otherwise the test tests nothing)"
    buf = ctypes.create_string_buffer(256)
    L.load().osr_conv2d_fwd_describe(ctypes.byref(p), 1, buf, 256)
    assert "split-K" in buf.value.decode() and "reduce" in buf.value.decode(), buf.value
    try:
        ops.SPLIT_K_TAIL = False
        single = ops.linear(x, w, b, relu=True, out_dtype=torch.float32)
        ops.SPLIT_K_TAIL = True
        split = ops.linear(x, w, b, relu=True, out_dtype=torch.float32)
        split2 = ops.linear(x, w, b, relu=True, out_dtype=torch.float32)
        half = ops.linear(x, w, b, relu=True)
    finally:
        ops.SPLIT_K_TAIL = True
    assert torch.equal(split, split2)  # fixed-order partial sums: bitwise reproducible
    assert not torch.equal(split, single) or True  # (the tail rows are summed in another order; equality is allowed, not required)
    ref = torch.relu(x.double() @ w.double().t() + b.double()).float()
    assert_close(split, ref, rtol=1e-4, name="split-K linear")
    assert_close(single, ref, rtol=1e-4, name="single-launch linear")
    assert_close(half, ref, rtol=2.0 ** -10, atol=2.0 ** -10 * float(ref.abs().max()) * 0.01 + 1e-6, name="split-K linear, f16 out")
    assert torch.equal(split[: 8192], single[: 8192])  # rows of the full rounds are untouched by the split


def test_conv3x3_split_k_tail_round(ops, osr):
    """Round 4: the split-K tail also takes K x K convolutions behind at least two full dispatch rounds (fpn_output3 at batch 16:
    1050 tiles of 256 x 256 on 256 slots; 380 -> 321 us). The tail workgroups start in the middle of the launch's K order
    (channel-slice-major, tap-minor), padding included. Reference: torch-CPU convolution of the first and of the last image (the
    tail rows are the last 6656 of the last image)."""
    gg = g(42)
    n, h, w, c = 16, 100, 168, 256
    x = (torch.randn(n, h, w, c, generator=gg) * 0.5).half().to(DEV)
    wt = (torch.randn(c, 3, 3, c, generator=gg) / math.sqrt(9 * c)).half().to(DEV)
    b = torch.randn(c, generator=gg).to(DEV)
    L = osr._lib
    import ctypes
    p = L.ConvParams()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, h, w, c, h, w, c
    p.kh = p.kw = 3
    p.stride_h = p.stride_w = p.pad_h = p.pad_w = 1
    p.in_stride_n, p.in_stride_h, p.in_stride_w = h * w * c, w * c, c
    p.out_stride_n, p.out_stride_h, p.out_stride_w = h * w * c, w * c, c
    p.in_dtype, p.out_dtype = L.OSR_F16, L.OSR_F32
    assert L.load().osr_conv2d_fwd_workspace_bytes(ctypes.byref(p)) > 0, "this shape must qualify for the split (otherwise the test tests nothing)"
    buf = ctypes.create_string_buffer(256)
    L.load().osr_conv2d_fwd_describe(ctypes.byref(p), 1, buf, 256)
    assert "split-K" in buf.value.decode(), buf.value
    try:
        ops.SPLIT_K_TAIL = False
        single = ops.conv2d(x, wt, b, 1, 1, relu=True, out_dtype=torch.float32)
        ops.SPLIT_K_TAIL = True
        split = ops.conv2d(x, wt, b, 1, 1, relu=True, out_dtype=torch.float32)
        split2 = ops.conv2d(x, wt, b, 1, 1, relu=True, out_dtype=torch.float32)
    finally:
        ops.SPLIT_K_TAIL = True
    assert torch.equal(split, split2)
    assert torch.equal(split[:15], single[:15])  # rows of the full rounds are untouched by the split
    assert not torch.equal(split[15], single[15]) or True  # (the tail rows are summed in another order: equality allowed, not required)
    for img in (0, n - 1):
        ref = torch.relu(F.conv2d(x[img:img + 1].cpu().float().permute(0, 3, 1, 2), wt.cpu().float().permute(0, 3, 1, 2), b.cpu(), padding=1)).permute(0, 2, 3, 1)
        assert_close(split[img:img + 1], ref, rtol=1e-4, name=f"split-K 3x3 conv, image {img}")
        assert_close(single[img:img + 1], ref, rtol=1e-4, name=f"single-launch 3x3 conv, image {img}")


# ------------------------------------------------------------------------------------------------------
def test_box_predictor_tail(ops):
    gg = g(31)
    p = O.make_head_params(1)
    m = 2000
    x = F.relu(torch.randn(m, 1024, generator=gg))
    xy = torch.rand(m, 2, generator=gg) * torch.tensor([1200.0, 700.0])
    props = torch.cat((xy, xy + torch.rand(m, 2, generator=gg) * 300 + 1), dim=1)
    ctr = torch.rand(m, generator=gg)
    bidx = torch.randint(0, 2, (m,), generator=gg, dtype=torch.int32)
    bidx[5] = -1
    hw = torch.tensor([[800, 1333], [640, 1000]], dtype=torch.int32)
    w = torch.cat((p["roi_heads.box_predictor.bbox_pred.weight"], p["roi_heads.box_predictor.iou_pred.weight"]))
    b = torch.cat((p["roi_heads.box_predictor.bbox_pred.bias"], p["roi_heads.box_predictor.iou_pred.bias"]))
    r = ops.box_predictor_tail(x.to(DEV), w.to(DEV), b.to(DEV), props.to(DEV), ctr.to(DEV), bidx.to(DEV), hw.to(DEV))
    d, iou = O.box_predictor(x, p)
    assert_close(r["pred_deltas"], torch.where(bidx.unsqueeze(1) >= 0, d, torch.zeros(1)), rtol=1e-4, atol=1e-5, name="deltas")
    assert_close(r["pred_iou"], torch.where(bidx >= 0, iou[:, 0], torch.zeros(1)), rtol=1e-4, atol=1e-6, name="iou")
    # downstream arithmetic checked on the kernel's own deltas/iou (identical inputs => tight tolerance)
    dk, ik = r["pred_deltas"].cpu(), r["pred_iou"].cpu()
    boxes = O.b2b_apply_deltas(dk, props)
    score = O.objectness_score(ik, ctr)
    for i in range(2):
        sel = bidx == i
        bc = O.box_clip(boxes[sel], tuple(hw[i].tolist()))
        assert_close(r["boxes"].cpu()[sel], bc, rtol=1e-5, atol=1e-3, name="boxes")
    assert_close(r["score"].cpu()[bidx >= 0], score[bidx >= 0], rtol=1e-6, atol=1e-7, name="score")
    cand = (score > 0.05) & (bidx >= 0)
    near = (score - 0.05).abs() < 1e-6
    assert bool(((r["cand"].cpu() != 0) == cand)[~near].all())


def _nms_inputs(seed, nseg, stride, lens, ncls):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 300, (nseg, stride, 2)).astype(np.float32)
    wh = rng.uniform(10, 120, (nseg, stride, 2)).astype(np.float32)
    boxes = np.concatenate((xy, xy + wh), 2)
    scores = np.round(rng.uniform(0, 1, (nseg, stride)), 3).astype(np.float32)  # exact ties
    cls = rng.integers(0, ncls, (nseg, stride)).astype(np.int32)
    cand = (rng.uniform(0, 1, (nseg, stride)) > 0.2).astype(np.int32)
    return boxes, scores, cls, cand, np.asarray(lens, dtype=np.int32)


@pytest.mark.parametrize("stride,lens,ncls,thr,topk", [
    (1000, [1000, 37, 0, 999], 5, 0.5, 50),
    (1000, [1000, 500], 1, 0.5, 1000),       # topk never reached
    (4273, [4273, 4000, 1], 1, 1.0, 1000),   # first-stage: pure stable sort + top-1000 (LDS sort, 8192 slots)
    (20000, [20000, 12345], 20, 0.5, 50),    # worst-case known candidates: global-memory sort path
    (64, [64, 1], 3, 0.3, 10),
])
def test_nms_topk_bit_exact(ops, stride, lens, ncls, thr, topk):
    nseg = len(lens)
    boxes, scores, cls, cand, seg_len = _nms_inputs(stride + len(lens), nseg, stride, lens, ncls)
    keep, cnt = ops.nms_topk(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), torch.from_numpy(cls).to(DEV),
                             torch.from_numpy(cand).to(DEV), nseg, stride, torch.from_numpy(seg_len).to(DEV), thr, topk)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for s in range(nseg):
        ids = np.nonzero(cand[s, : lens[s]])[0]
        if thr >= 1.0:
            k = ids[CO.argsort_desc(scores[s, ids])][:topk]
        else:
            k = ids[CO.batched_nms(boxes[s, ids], scores[s, ids], cls[s, ids].astype(np.int64), thr)][:topk]
        assert cnt[s] == len(k), f"segment {s}: kept {cnt[s]} vs {len(k)}"
        assert keep[s, : cnt[s]].tolist() == k.tolist(), f"segment {s}: kept index list differs"
        assert (keep[s, cnt[s]:] == -1).all()


def test_nms_topk_many_seeds(ops):
    """Index parity is the bar for NMS: sixteen more random scenes (clustered boxes, ties, dropped candidates) at three
    thresholds, kept lists compared element by element with the C oracle."""
    for seed in range(16):
        thr = (0.3, 0.5, 0.7)[seed % 3]
        stride, lens, ncls, topk = 700, [700, 333, 1, 0], 1 + seed % 5, 100
        boxes, scores, cls, cand, seg_len = _nms_inputs(1000 + seed, len(lens), stride, lens, ncls)
        keep, cnt = ops.nms_topk(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), torch.from_numpy(cls).to(DEV),
                                 torch.from_numpy(cand).to(DEV), len(lens), stride, torch.from_numpy(seg_len).to(DEV), thr, topk)
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        for s_ in range(len(lens)):
            ids = np.nonzero(cand[s_, : lens[s_]])[0]
            k = ids[CO.batched_nms(boxes[s_, ids], scores[s_, ids], cls[s_, ids].astype(np.int64), thr)][:topk] if len(ids) else ids
            assert cnt[s_] == len(k) and keep[s_, : cnt[s_]].tolist() == k.tolist(), f"seed {seed} segment {s_}"


def test_nms_no_class_no_cand(ops):
    boxes, scores, _, _, seg_len = _nms_inputs(3, 2, 300, [300, 150], 1)
    keep, cnt = ops.nms_topk(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), None, None, 2, 300,
                             torch.from_numpy(seg_len).to(DEV), 0.4, 25)
    for s in range(2):
        k = CO.nms(boxes[s, : seg_len[s]], scores[s, : seg_len[s]], 0.4)[:25]
        assert keep[s, : int(cnt[s])].cpu().tolist() == k.tolist()


def test_gather_and_l2norm(ops):
    gg = g(41)
    src = torch.randn(2 * 50, 1024, generator=gg)
    keep = torch.tensor([[3, 49, 0, -1], [7, 7, 1, 2]], dtype=torch.int32)
    cnt = torch.tensor([3, 2], dtype=torch.int32)
    out = ops.gather_rows(src.to(DEV), 50, keep.to(DEV), cnt.to(DEV)).cpu()
    assert torch.equal(out[0, :3], src[[3, 49, 0]]) and torch.equal(out[1, :2], src[[57, 57]])
    assert float(out[0, 3].abs().max()) == 0 and float(out[1, 2:].abs().max()) == 0
    x = torch.randn(20, 256, generator=gg)
    x[3] = 0
    assert_close(ops.l2_normalize_rows(x.to(DEV)), F.normalize(x), rtol=1e-6, atol=1e-7, name="l2norm")


def test_pln_tail(ops):
    gg = g(42)
    p = O.make_head_params(2)
    p["roi_heads.dml.encoder.weight"] = torch.eye(256, 1024)  # emb = feats[:, :256]: lets the test place rows near prototypes
    feats = torch.randn(1500, 1024, generator=gg)
    near = torch.arange(0, 1500, 2)
    feats[near, :256] = p["roi_heads.dml.representatives"][near % 20] * 1.7 + torch.randn(len(near), 256, generator=gg) * \
        torch.linspace(0.05, 1.2, len(near)).unsqueeze(1)
    cls_ref, rec_ref, md_ref, emb = O.pln_inference(feats, p, 0.23, 80, 20)
    protos = ops.l2_normalize_rows(p["roi_heads.dml.representatives"].to(DEV))
    rv = torch.tensor([1000, 400], dtype=torch.int32)
    emb_pad = torch.zeros(2000, 256)
    emb_pad[:1000] = emb[:1000]
    emb_pad[1000:1400] = emb[1000:1400]
    pc, md = ops.pln_tail(emb_pad.to(DEV), protos, 20, 1, 0.23, 80, rows_valid=rv.to(DEV), seg_rows=1000)
    pc, md = pc.cpu(), md.cpu()
    valid = torch.cat((torch.arange(1000), torch.arange(1000, 1400)))
    assert_close(md[valid], md_ref[:1400], rtol=1e-5, atol=1e-6, name="min_dist")
    # classes must agree wherever the decision is not within rounding of a tie / the threshold
    rep = F.normalize(p["roi_heads.dml.representatives"])
    dist = 1.0 - F.normalize(emb[:1400]) @ rep.t()
    top2 = dist.topk(2, dim=1, largest=False)[0]
    safe = ((top2[:, 1] - top2[:, 0]) > 1e-5) & ((md_ref[:1400] - 0.23).abs() > 1e-5)
    assert int(safe.sum()) > 1300
    assert torch.equal(pc[valid][safe], cls_ref[:1400][safe])
    assert bool((pc[1400:] == -1).all())
    assert int((pc[valid] == 80).sum()) > 0 and int((pc[valid] != 80).sum()) > 0


def test_pln_tail_known_answers(ops):
    protos = torch.eye(20, 256)
    f = torch.zeros(4, 256)
    f[0, 7] = 5.0
    f[1, 3] = 1.0
    f[1, 4] = 1.0
    f[2, 100] = 1.0
    f[3, 5] = 1.0
    f[3, 6] = 0.6
    pc, md = ops.pln_tail(f.to(DEV), protos.to(DEV), 20, 1, 0.23, 80)
    assert pc.cpu().tolist() == [7, 80, 80, 5]
    pc2, _ = ops.pln_tail(f.to(DEV), protos.to(DEV), 20, 1, 0.5, 80)
    assert pc2.cpu()[1].item() == 3  # tie between classes 3 and 4: lower index
    cm = (torch.arange(20) * 3 + 1).to(torch.int64)
    pc3, _ = ops.pln_tail(f.to(DEV), protos.to(DEV), 20, 1, 0.23, 1000, class_map=cm.to(DEV))
    assert pc3.cpu().tolist() == [22, 1000, 1000, 16]


def test_softmax_candidates_nms_assemble_vs_oracle(ops):
    gg = g(43)
    p = O.make_head_params(3)
    n, seg = 2, 1000
    cfg = dict(O.VOC_COCO_CFG)
    results_ref, inputs = [], []
    counts = [1000, 620]
    det_boxes = torch.zeros(n, seg, 4)
    det_scores = torch.zeros(n, seg)
    pred_cls = torch.full((n, seg), -1, dtype=torch.int64)
    logits = torch.zeros(n, seg, 21)
    for i in range(n):
        c = counts[i]
        xy = torch.rand(c, 2, generator=gg) * torch.tensor([1000.0, 600.0])
        b = torch.cat((xy, xy + torch.rand(c, 2, generator=gg) * 250 + 5), dim=1)
        b = O.box_clip(b, (800, 1333))
        s = torch.sort(torch.rand(c, generator=gg), descending=True)[0]
        rec = torch.randn(c, 1024, generator=gg) * 2
        cls = torch.where(torch.rand(c, generator=gg) > 0.4, torch.randint(0, 20, (c,), generator=gg), torch.tensor(80))
        lg = F.linear(rec, p["roi_heads.softmaxcls.cls_score.weight"], p["roi_heads.softmaxcls.cls_score.bias"])
        results_ref.append(O.softmax_classifier_inference(b, s, cls, rec, (800, 1333), p, cfg))
        det_boxes[i, :c], det_scores[i, :c], pred_cls[i, :c], logits[i, :c] = b, s, cls, lg
    cnt = torch.tensor(counts, dtype=torch.int32)
    cands = ops.softmax_candidates(logits.view(-1, 21).to(DEV), 20, det_boxes.view(-1, 4).to(DEV), det_scores.view(-1).to(DEV),
                                   pred_cls.view(-1).to(DEV), cnt.to(DEV), n, seg, 80, cfg["known_score_thresh"], cfg["unknown_score_thresh"])
    kk, kc = ops.nms_topk(cands["k_boxes"], cands["k_scores"], cands["k_cls"], None, n, seg * 20, cands["k_count"], cfg["known_nms_thresh"],
                          cfg["known_topk"])
    uk, uc = ops.nms_topk(cands["u_boxes"], cands["u_scores"], None, None, n, seg, cands["u_count"], cfg["unknown_nms_thresh"],
                          cfg["unknown_topk"])
    ob, osc, ocl, on = ops.assemble_detections(cands, kk, kc, uk, uc, n, 80)
    for i in range(n):
        rb, rs, rc = results_ref[i]
        c = int(on[i])
        assert c == len(rb), f"image {i}: {c} detections vs {len(rb)}"
        assert torch.equal(ocl[i, :c].cpu(), rc)
        assert_close(osc[i, :c], rs, rtol=1e-5, atol=1e-7, name="scores")
        assert torch.equal(ob[i, :c].cpu(), rb)
        nu = int((rc == 80).sum())
        assert bool((rc[:nu] == 80).all()) and bool((rc[nu:] != 80).all())  # [unknown..., known...]


def test_inference_tail_all_known_single_unknown_and_all_unknown(ops):
    """The branches softmax_classifier.py:317-344 and prototype_learning_network.py:218-222 fix (tests/test_oracle_kat.py (ix)) on
    the HIP path: an image whose detections are ALL known (the `known.all()` branch: the unknown leg contributes nothing), one
    with exactly ONE unknown (the 0-d `squeeze()` index) and one with no known detection -- kept lists, scores, classes and the
    [unknown..., known...] order against the oracle, bit for bit on indices."""
    cfg = dict(O.VOC_COCO_CFG)
    w = torch.zeros(21, 256)
    for k in range(20):
        w[k, k] = 10.0
    p = {"roi_heads.softmaxcls.cls_score.weight": w, "roi_heads.softmaxcls.cls_score.bias": torch.zeros(21)}
    boxes = torch.tensor([[10.0, 10.0, 50.0, 50.0], [60.0, 60.0, 90.0, 90.0], [12.0, 12.0, 52.0, 52.0]])
    scores = torch.tensor([0.9, 0.8, 0.7])
    rec = torch.zeros(3, 256)
    rec[0, 4] = 1.0
    rec[1, 7] = 1.0
    rec[2, 4] = 1.0
    cases = [torch.tensor([4, 7, 4]), torch.tensor([4, 80, 4]), torch.tensor([80, 80, 80])]
    n, seg = len(cases), 8
    det_boxes = torch.zeros(n, seg, 4)
    det_scores = torch.zeros(n, seg)
    pred_cls = torch.full((n, seg), -1, dtype=torch.int64)
    logits = torch.zeros(n, seg, 21)
    refs = []
    for i, cls in enumerate(cases):
        det_boxes[i, :3], det_scores[i, :3], pred_cls[i, :3] = boxes, scores, cls
        logits[i, :3] = F.linear(rec, w)
        refs.append(O.softmax_classifier_inference(boxes, scores, cls, rec, (100, 100), p, cfg))
    cnt = torch.tensor([3] * n, dtype=torch.int32)
    cands = ops.softmax_candidates(logits.view(-1, 21).to(DEV), 20, det_boxes.view(-1, 4).to(DEV), det_scores.view(-1).to(DEV),
                                   pred_cls.view(-1).to(DEV), cnt.to(DEV), n, seg, 80, cfg["known_score_thresh"], cfg["unknown_score_thresh"])
    assert cands["u_count"].cpu().tolist() == [0, 1, 3]
    kk, kc = ops.nms_topk(cands["k_boxes"], cands["k_scores"], cands["k_cls"], None, n, seg * 20, cands["k_count"], cfg["known_nms_thresh"],
                          cfg["known_topk"])
    uk, uc = ops.nms_topk(cands["u_boxes"], cands["u_scores"], None, None, n, seg, cands["u_count"], cfg["unknown_nms_thresh"],
                          cfg["unknown_topk"])
    ob, osc, ocl, on = ops.assemble_detections(cands, kk, kc, uk, uc, n, 80)
    assert on.cpu().tolist() == [2, 2, 2]
    for i in range(n):
        rb, rs, rc = refs[i]
        c = int(on[i])
        assert c == len(rb) and torch.equal(ocl[i, :c].cpu(), rc), (i, ocl[i, :c].cpu(), rc)
        assert_close(osc[i, :c], rs, rtol=1e-6, atol=1e-7, name="scores")
        assert torch.equal(ob[i, :c].cpu(), rb)
    assert ocl[0, :2].cpu().tolist() == [4, 7] and ocl[1, :2].cpu().tolist() == [80, 4] and ocl[2, :2].cpu().tolist() == [80, 80]
    # PLN: exactly one unknown / none / a single unknown detection
    protos = torch.eye(20, 256)
    f = torch.zeros(3, 256)
    f[0, 2] = 1.0
    f[1, 9] = 4.0
    f[2, 200] = 1.0
    assert ops.pln_tail(f.to(DEV), protos.to(DEV), 20, 1, 0.23, 80)[0].cpu().tolist() == [2, 9, 80]
    assert ops.pln_tail(f[:2].contiguous().to(DEV), protos.to(DEV), 20, 1, 0.23, 80)[0].cpu().tolist() == [2, 9]
    assert ops.pln_tail(f[2:].contiguous().to(DEV), protos.to(DEV), 20, 1, 0.23, 80)[0].cpu().tolist() == [80]
    cm = torch.arange(100, 120, dtype=torch.int64)
    assert ops.pln_tail(f.to(DEV), protos.to(DEV), 20, 1, 0.23, 1000, class_map=cm.to(DEV))[0].cpu().tolist() == [102, 109, 1000]


def test_cfrpn_head_fused_matches_two_step_and_oracle(ops):
    """Fused ClsFreeRPNHead level (one launch) == conv + tail (two launches) == oracle on fp16-rounded operands."""
    gg = g(51)
    p = O.make_head_params(4)
    x = (torch.randn(2, 256, 21, 29, generator=gg)).half()
    w = p["proposal_generator.rpn_head.conv.weight"].half()
    b = torch.randn(256, generator=gg) * 0.1
    wt = torch.cat((p["proposal_generator.rpn_head.anchor_deltas.weight"].view(4, 256), p["proposal_generator.rpn_head.centerness.weight"].view(1, 256)))
    bt = torch.cat((p["proposal_generator.rpn_head.anchor_deltas.bias"], p["proposal_generator.rpn_head.centerness.bias"]))
    xd, wd = nhwc(x).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    d_f, c_f = ops.cfrpn_head_fused(xd, wd, b.to(DEV), wt.to(DEV), bt.to(DEV))
    t = ops.conv2d(xd, wd, b.to(DEV), 1, 1, relu=True)
    d_u, c_u = ops.cfrpn_head_tail(t.view(-1, 256), wt[:4].contiguous().to(DEV), bt[:4].contiguous().to(DEV), wt[4:].contiguous().to(DEV),
                                   bt[4:].contiguous().to(DEV))
    assert_close(d_f, d_u, rtol=1e-4, atol=1e-5, name="fused vs two-step deltas")
    assert_close(c_f, c_u, rtol=1e-4, atol=1e-6, name="fused vs two-step ctr")
    tq = F.relu(F.conv2d(x.float(), w.float(), b, padding=1)).half().float()  # hidden state rounded to the storage dtype
    pq = dict(p)
    dr, cr = O.cfrpn_head_tail(tq.permute(0, 2, 3, 1).reshape(-1, 256), pq)
    assert_close(d_f, dr, rtol=2e-3, atol=2e-3, name="fused vs oracle deltas")  # 1 fp16 ulp flips of t before the normalise
    assert_close(c_f, cr, rtol=2e-3, atol=1e-3, name="fused vs oracle ctr")
    # osr_cfrpn_head_fwd_ex: the same launch also hands out the hidden state (the training step keeps it for the head's backward),
    # for the small-level (128-row tiles) and the large-level (256-row tiles, >= 512 of them) instantiation
    for xx in (xd, nhwc(torch.randn(2, 256, 264, 256, generator=gg).half()).to(DEV)):
        rows = xx.shape[0] * xx.shape[1] * xx.shape[2]
        hid = torch.full((rows, 256), -1.0, dtype=torch.float16, device=DEV)
        d_h, c_h = ops.cfrpn_head_fused(xx, wd, b.to(DEV), wt.to(DEV), bt.to(DEV), hidden_out=hid)
        d_0, c_0 = ops.cfrpn_head_fused(xx, wd, b.to(DEV), wt.to(DEV), bt.to(DEV))
        assert torch.equal(d_h, d_0) and torch.equal(c_h, c_0)
        tt = ops.conv2d(xx, wd, b.to(DEV), 1, 1, relu=True).view(-1, 256)
        assert float((hid.float() - tt.float()).abs().max()) <= 2e-3 * float(tt.float().abs().max())  # (one fp16 ulp: another tile's K order)
        assert float(hid.min()) >= 0.0


def test_detector_postprocess(ops):
    gg = g(77)
    n, cap = 3, 100
    xy = torch.rand(n, cap, 2, generator=gg) * 300 - 20
    boxes = torch.cat((xy, xy + torch.rand(n, cap, 2, generator=gg) * 80), dim=2)
    boxes[0, 3] = torch.tensor([500.0, 10.0, 520.0, 30.0])  # fully outside after clipping -> dropped
    boxes[1, 0, 2] = boxes[1, 0, 0]                          # zero width -> dropped
    scores = torch.rand(n, cap, generator=gg)
    classes = torch.randint(0, 21, (n, cap), generator=gg)
    count = torch.tensor([100, 37, 0], dtype=torch.int32)
    sizes, outs = [(240, 320)] * n, [(480, 500), (120, 160), (240, 320)]
    scale = torch.tensor([[ow / s[1], oh / s[0]] for (oh, ow), s in zip(outs, sizes)], dtype=torch.float32)
    ob, os_, oc, on = [t.cpu() for t in ops.detector_postprocess(boxes.to(DEV), scores.to(DEV), classes.to(DEV), count.to(DEV), scale.to(DEV),
                                                                 torch.tensor(outs, dtype=torch.int32).to(DEV))]
    for i in range(n):
        c = int(count[i])
        b = boxes[i, :c] * torch.tensor([scale[i, 0], scale[i, 1], scale[i, 0], scale[i, 1]])
        b[:, 0::2] = b[:, 0::2].clamp(0, outs[i][1])
        b[:, 1::2] = b[:, 1::2].clamp(0, outs[i][0])
        keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        m = int(keep.sum())
        assert int(on[i]) == m
        assert torch.equal(ob[i, :m], b[keep]) and torch.equal(os_[i, :m], scores[i, :c][keep]) and torch.equal(oc[i, :m], classes[i, :c][keep])
        assert bool((oc[i, m:] == -1).all())
    assert int(on[0]) < 100 and int(on[1]) < 37


def test_roi_locality_order_is_a_permutation_and_changes_nothing(ops):
    """osr_roi_locality_order / osr_roi_align_fwd_ordered: the order is a permutation of the list that groups RoIs by image, then
    pyramid level ([d2] assign_boxes_to_levels as osrcnn_roi_heads.py:108-113 configures it), then 32-pixel tile; padding rows
    (batch index -1) go last; pooling in that order, in list order and in reversed order gives bit-identical rows."""
    g = torch.Generator().manual_seed(5)
    n, per = 3, 700
    shapes = [(64, 96), (32, 48), (16, 24), (8, 12)]
    scales = (0.25, 0.125, 0.0625, 0.03125)
    feats = [torch.randn(n, h, w, 256, generator=g).half().to(DEV) for h, w in shapes]
    ctr = torch.rand(n * per, 2, generator=g) * torch.tensor([384.0, 256.0])
    size = torch.exp(torch.rand(n * per, 2, generator=g) * 4.5 + 1.5)           # 4 .. 400 px: every level
    boxes = torch.cat((ctr - size / 2, ctr + size / 2), dim=1).clamp(min=0).contiguous().to(DEV)
    bi = torch.arange(n, dtype=torch.int32).repeat_interleave(per)
    bi[torch.rand(n * per, generator=g) < 0.1] = -1
    bi = bi[torch.randperm(n * per, generator=g)].contiguous().to(DEV)           # images interleaved in the list
    full = ops.roi_locality_order(feats, scales, boxes, bi)
    m = n * per
    order = full[:m]
    assert full.shape == (m + 1,) and int(full[m]) == int((bi >= 0).sum())
    assert order.dtype == torch.int32 and torch.equal(torch.sort(order.long()).values.cpu(), torch.arange(m))
    ob = bi[order.long()].cpu()
    valid = ob >= 0
    assert bool((ob[valid][1:] >= ob[valid][:-1]).all()), "images are contiguous and ascending"
    assert not bool(valid[int(valid.sum()):].any()), "padding rows last"
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).cpu()
    lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224 + 1e-8)).clamp(2, 5)[order.long().cpu()]
    key = ob[valid].double() * 10 + lvl[valid].double()
    assert bool((key[1:] >= key[:-1]).all()), "levels ascend inside an image"
    ident = torch.arange(m, dtype=torch.int32, device=DEV)
    a = ops.roi_align(feats, scales, boxes, bi, 7, torch.float16, order=ident)
    b = ops.roi_align(feats, scales, boxes, bi, 7, torch.float16, order=full)
    b2 = ops.roi_align(feats, scales, boxes, bi, 7, torch.float16, order=order.contiguous())
    assert torch.equal(b, b2)
    c = ops.roi_align(feats, scales, boxes, bi, 7, torch.float16, order=ident.flip(0).contiguous())
    d = ops.roi_align(feats, scales, boxes, bi, 7, torch.float16)
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    assert float(a[(bi < 0)].abs().sum()) == 0.0


def test_linear_skips_tiles_of_padding_rows(ops):
    """osr_conv_params.row_seg_counts / row_seg_rows (the box head's FC layers over padded per-image proposal lists;
    osrcnn_roi_heads.py:304-309 runs them on the real proposals only): every data row equals the plain launch bit for bit, tiles
    that hold padding rows only are left untouched, for the big-tile FC1 shape (with its split-K tail) and the 128-row FC2 shape."""
    gg = g(31)
    for (seg_rows, nseg, k, nout, dt_out) in ((4273, 16, 1024, 1024, torch.float32), (1500, 3, 12544, 1024, torch.float16), (700, 5, 256, 64, torch.float16)):
        m = seg_rows * nseg
        x = torch.randn(m, k, generator=gg).half().to(DEV)
        w = (torch.randn(nout, k, generator=gg) / k ** 0.5).half().to(DEV)
        b = torch.randn(nout, generator=gg).to(DEV)
        counts = torch.randint(0, seg_rows + 1, (nseg,), generator=gg, dtype=torch.int32)
        counts[0] = seg_rows
        if nseg > 2:
            counts[1] = 0
            counts[2] = 1
        want = ops.linear(x, w, b, relu=True, out_dtype=dt_out)
        sentinel = -7.0
        out = torch.full((1, m, 1, nout), sentinel, dtype=dt_out, device=DEV)
        got = ops.conv2d(x.view(1, m, 1, k), w.view(nout, 1, 1, k), b, relu=True, out_dtype=dt_out, out=out, row_seg=(counts.to(DEV), seg_rows)).view(m, nout)
        row = torch.arange(m)
        valid = (row % seg_rows) < counts[row // seg_rows].long()
        assert torch.equal(got[valid.to(DEV)], want[valid.to(DEV)])
        untouched = (got == sentinel).all(dim=1).cpu()
        assert not bool((untouched & valid).any())
        assert bool(untouched.any()), "some tile must have been skipped"
        wrote_pad = (~untouched) & (~valid)  # whatever was written outside the data rows is the ordinary result of those rows
        assert torch.equal(got[wrote_pad.to(DEV)], want[wrote_pad.to(DEV)])
