"""GPU parity tests of the tile-centric RoIAlign (osr_roi_align_fwd_tiled + osr_roi_align_fwd_masked, csrc/osr_roi_tiled.hip)
against the C oracle's restatement of torchvision roi_align / [d2] ROIPooler (osrcnn_roi_heads.py:108-113,306) and against
the wave-per-RoI kernel, and of the planar second output of the convolution that feeds it.

Tolerance (fp32 out): 1e-4 of the value + 1e-5 -- the kernel sums in another order and its x weights are fp16 hi + lo pairs (2^-22)."""
import pytest
import torch

from oracle import c_binding as CO
from oracle import osr_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SCALES = (0.25, 0.125, 0.0625, 0.03125)


@pytest.fixture(scope="module")
def ops(osr):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    osr._lib.load()
    return osr.ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def assert_close(a, b, rtol=1e-4, atol=1e-5, name=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    bad = (a - b).abs() > atol + rtol * b.abs()
    assert not bool(bad.any()), f"{name}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err {float((a - b).abs().max()):.3e}"


def oracle_pool(feats_nchw, boxes, bidx):
    """[d2] ROIPooler: level per box, torchvision roi_align (aligned, sampling_ratio 0) on that level; padding rows stay zero."""
    lv = O.assign_levels(boxes)
    ref = torch.zeros((boxes.shape[0], feats_nchw[0].shape[1], 7, 7))
    for l, s in enumerate(SCALES[:len(feats_nchw)]):
        ids = torch.nonzero((lv == l) & (bidx >= 0)).squeeze(1)
        if len(ids):
            rois = torch.cat((bidx[ids].float().unsqueeze(1), boxes[ids]), dim=1)
            ref[ids] = CO.roi_align(feats_nchw[l].float(), rois, s)
    return ref


def tiled(ops, feats_nchw, boxes, bidx, odt=torch.float32):
    fl = [nhwc(f).half().to(DEV) for f in feats_nchw]
    pl = [ops.to_planes(f) for f in fl]
    out, rid = ops.roi_align_tiled(fl, pl, SCALES[:len(fl)], boxes.to(DEV), bidx.to(DEV), 7, odt, return_rid=True)
    c = fl[0].shape[3]
    return ops.slice_major_to_bin_major(out, c).cpu().float().permute(0, 3, 1, 2), rid.cpu()


def proposal_like_boxes(gen, n_img, m, w_img, h_img):
    """Boxes with the statistics of the proposal lists (wider than tall, 8 .. 600 px, all four levels), a few degenerate ones and
    padding rows."""
    cx = torch.rand(m, generator=gen) * w_img
    cy = torch.rand(m, generator=gen) * h_img
    w = torch.exp(torch.rand(m, generator=gen) * 4.3 + 2.0)          # 7 .. 540 px
    h = w * (0.25 + torch.rand(m, generator=gen) * 1.0)
    boxes = torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), dim=1)
    boxes[:, 0::2].clamp_(0, w_img)
    boxes[:, 1::2].clamp_(0, h_img)
    bidx = torch.randint(0, n_img, (m,), generator=gen, dtype=torch.int32)
    bidx[::17] = -1                                                    # padding rows
    boxes[5] = torch.tensor([40.0, 40.0, 40.0, 90.0])                  # zero width
    boxes[6] = torch.tensor([50.3, 60.2, 51.1, 61.0])                  # a fifth of a pixel
    boxes[7] = torch.tensor([0.0, 0.0, float(w_img), float(h_img)])    # the whole image
    boxes[8] = torch.tensor([-60.0, -30.0, 90.0, 50.0])                # crosses the top-left border (unclipped input)
    boxes[9] = torch.tensor([w_img - 50.0, h_img - 40.0, w_img + 120.0, h_img + 80.0])  # crosses the bottom-right border
    boxes[10] = torch.tensor([4.0, 100.0, w_img - 4.0, 112.0])         # wider than 64 columns on its level
    return boxes, bidx


@pytest.mark.parametrize("odt", [torch.float32, torch.float16])
def test_tiled_roi_align_vs_oracle(ops, odt):
    """Four levels of a 480 x 800 image (p2 120 x 200: several half-overlapping regions; p4 / p5 smaller than one region), 256
    channels, 900 boxes over two images."""
    gen = g(31)
    n, hh, ww = 2, 480, 800
    feats = [torch.randn(n, 256, hh // s, ww // s, generator=gen).half().float() for s in (4, 8, 16, 32)]
    boxes, bidx = proposal_like_boxes(gen, n, 900, ww, hh)
    out, rid = tiled(ops, feats, boxes, bidx, odt)
    ref = oracle_pool(feats, boxes, bidx)
    assert int((rid >= 0).sum()) > 0.8 * int((bidx >= 0).sum()), "most of the list must go through the tiled kernel"
    assert torch.equal(rid == -2, bidx < 0) and int((rid[5:11] >= 0).sum()) >= 1 and int(rid[10]) == -1
    assert float(out[bidx < 0].abs().max()) == 0.0
    if odt == torch.float32:
        assert_close(out, ref, name="tiled roi_align")
    else:
        assert_close(out, ref, rtol=2.0 ** -10, atol=2e-3, name="tiled roi_align f16 out")


def test_tiled_equals_the_wave_per_roi_kernel_on_the_rows_it_leaves(ops):
    """The rows the tiled kernel does not take (rid -1) are written by the wave-per-RoI kernel in the slice-major layout: bit for bit
    what that kernel writes in its own layout; every other row agrees with it to the summation-order tolerance."""
    gen = g(32)
    n, hh, ww = 2, 416, 640
    feats = [torch.randn(n, 256, hh // s, ww // s, generator=gen).half().float() for s in (4, 8, 16, 32)]
    boxes, bidx = proposal_like_boxes(gen, n, 400, ww, hh)
    out, rid = tiled(ops, feats, boxes, bidx)
    fl = [nhwc(f).half().to(DEV) for f in feats]
    old = ops.roi_align(fl, SCALES, boxes.to(DEV), bidx.to(DEV), 7, torch.float32).cpu().permute(0, 3, 1, 2)
    left = rid < 0
    assert int((rid == -1).sum()) >= 3
    assert torch.equal(out[left], old[left])
    assert_close(out, old, name="tiled vs wave-per-RoI")


def test_tiled_linear_ramp_is_exact(ops):
    """Known answer (SURVEY 8c vi): bilinear interpolation of a linear ramp is exact, so every bin of a box inside the map is the
    ramp at the bin centre. f(y, x) = 2 x + 3 y + channel on one level of 80 x 112."""
    h, w, c = 80, 112, 32
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    f = (2 * xs + 3 * ys)[None, None] + torch.arange(c, dtype=torch.float32)[None, :, None, None]
    boxes = torch.tensor([[40.0, 40.0, 120.0, 96.0], [100.5, 60.25, 340.0, 180.0], [8.0, 8.0, 36.0, 22.0], [200.0, 110.0, 420.0, 290.0]])
    bidx = torch.zeros(4, dtype=torch.int32)
    fl = [nhwc(f).half().to(DEV)]  # (integers up to 500: exact in fp16)
    out, rid = ops.roi_align_tiled(fl, [ops.to_planes(fl[0])], (0.25,), boxes.to(DEV), bidx.to(DEV), 7, torch.float32, return_rid=True)
    out = ops.slice_major_to_bin_major(out, c).cpu()
    assert bool((rid.cpu() >= 0).all())
    for i, b in enumerate(boxes):
        x1, y1, x2, y2 = [float(v) * 0.25 - 0.5 for v in b]
        bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
        for ph in range(7):
            for pw in range(7):
                want = 2 * (x1 + (pw + 0.5) * bw) + 3 * (y1 + (ph + 0.5) * bh)
                got = out[i, ph, pw]
                assert float((got - (want + torch.arange(c))).abs().max()) < 2e-3, (i, ph, pw)


def test_tiled_result_does_not_depend_on_the_list(ops):
    """A RoI's row is a function of its box alone: pooled inside a long list or alone, first or last, the bits are the same (the path
    a RoI takes depends on its own geometry; regions, list order and the other RoIs do not enter its arithmetic)."""
    gen = g(33)
    n, hh, ww = 2, 416, 640
    feats = [torch.randn(n, 256, hh // s, ww // s, generator=gen).half().float() for s in (4, 8, 16, 32)]
    boxes, bidx = proposal_like_boxes(gen, n, 300, ww, hh)
    full, _ = tiled(ops, feats, boxes, bidx)
    perm = torch.randperm(300, generator=gen)
    shuf, _ = tiled(ops, feats, boxes[perm], bidx[perm])
    assert torch.equal(shuf, full[perm])
    few = torch.tensor([12, 40, 41, 250])
    part, _ = tiled(ops, feats, boxes[few], bidx[few])
    assert torch.equal(part, full[few])
    again, _ = tiled(ops, feats, boxes, bidx)
    assert torch.equal(again, full)


def test_conv_writes_the_planar_copy(ops):
    """osr_conv_params.out2_planar16: the 3 x 3 output convolution's second output is the same values in (n, c/16, h, w, 16)."""
    gen = g(34)
    x = torch.randn(2, 37, 53, 256, generator=gen).half().to(DEV)
    w = (torch.randn(256, 3, 3, 256, generator=gen) * 0.02).half().to(DEV)
    b = torch.randn(256, generator=gen).to(DEV)
    planes = torch.full((2, 16, 37, 53, 16), float("nan"), dtype=torch.float16, device=DEV)
    y = ops.conv2d(x, w, b, 1, 1, planes_out=planes)
    y0 = ops.conv2d(x, w, b, 1, 1)
    assert torch.equal(y, y0)
    assert torch.equal(planes, ops.to_planes(y))


def test_engine_with_the_tiled_path_agrees_with_the_default_engine(osr, ops):
    """OpensetRCNNEngine(tiled_roi=True): the FPN output convolutions write the planar copy, RoIAlign runs tile-centric and FC1 reads
    slice-major rows through the re-packed weight -- pooled rows within one fp16 rounding of the default engine's, box features
    within the fp16 box head's tolerance."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    params = random_params(0)
    gen = g(35)
    images = torch.randint(0, 256, (2, 3, 256, 384), generator=gen, dtype=torch.uint8).to(DEV)
    ka, kb = {}, {}
    OpensetRCNNEngine(params, device=DEV).forward(images, keep=ka)
    eng = OpensetRCNNEngine(params, device=DEV, tiled_roi=True)
    eng.forward(images, keep=kb)
    assert "p2_planes" in kb["feats"] and torch.equal(kb["feats"]["p3_planes"], ops.to_planes(kb["feats"]["p3"]))
    assert torch.equal(ka["sel"]["boxes"], kb["sel"]["boxes"])
    pa, pb = ka["pooled"].float().cpu(), kb["pooled"].float().cpu()
    assert float((pa - pb).abs().max()) <= 2e-3 * max(1.0, float(pa.abs().max()))
    cnt = [int(c) for c in ka["sel"]["counts"].cpu()]
    cap = ka["sel"]["cap"]
    fa, fb = ka["box_feats"].view(2, cap, -1).cpu(), kb["box_feats"].view(2, cap, -1).cpu()
    for i in range(2):
        d = (fa[i, :cnt[i]] - fb[i, :cnt[i]]).abs().max()
        assert float(d) <= 5e-3 * max(1.0, float(fa[i, :cnt[i]].abs().max()))
