"""The fused res2 bottleneck (osr_bottleneck_fwd, one launch) against (a) the three / four separate osr_conv2d_fwd launches it
replaces and (b) the oracle's torch-CPU convolutions on identically rounded operands. Shapes cover whole tiles, ragged edges
(height / width not multiples of the 8 x 16 tile: partial tiles, halo rows outside the image on every side), one-pixel-wide
inputs, several images, both block kinds (identity shortcut with cin 256; projection shortcut with cin 64) and both dtypes."""
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


def _block(seed, cin, proj, dtype):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, k=1.0: torch.randn(*s, generator=g) * k  # noqa: E731
    w = dict(w1=rnd(64, cin, 1, 1, k=(2.0 / cin) ** 0.5), b1=rnd(64, k=0.3), w2=rnd(64, 64, 3, 3, k=(2.0 / 576) ** 0.5), b2=rnd(64, k=0.3),
             w3=rnd(256, 64, 1, 1, k=(1.0 / 64) ** 0.5), b3=rnd(256, k=0.3))
    if proj:
        w.update(wsc=rnd(256, cin, 1, 1, k=(1.0 / cin) ** 0.5), bsc=rnd(256, k=0.3))
    return {k: (v.to(dtype) if k.startswith("w") else v.float()) for k, v in w.items()}


def _packed(w, dev):
    return {k: (v.permute(0, 2, 3, 1).contiguous() if k.startswith("w") else v.contiguous()).to(dev) for k, v in w.items()}


def _separate(ops, x, p, proj):
    sc = ops.conv2d(x, p["wsc"], p["bsc"]) if proj else x
    o = ops.conv2d(x, p["w1"], p["b1"], relu=True)
    o = ops.conv2d(o, p["w2"], p["b2"], 1, 1, relu=True)
    return ops.conv2d(o, p["w3"], p["b3"], relu=True, residual=sc, res_mode=1)


def _torch_ref(x_nhwc, w, proj, dtype):
    """fp32 math on the rounded operands, intermediates rounded to the storage dtype where the kernels round them."""
    x = x_nhwc.float().permute(0, 3, 1, 2)
    r = lambda t: t.to(dtype).float()  # noqa: E731
    o = r(F.relu(F.conv2d(x, w["w1"].float(), w["b1"])))
    o = r(F.relu(F.conv2d(o, w["w2"].float(), w["b2"], padding=1)))
    y = F.conv2d(o, w["w3"].float(), w["b3"])
    sc = F.conv2d(x, w["wsc"].float(), w["bsc"]) if proj else x
    return r(F.relu(y + sc)).permute(0, 2, 3, 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("proj", [False, True])
@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 8, 16), (3, 21, 37), (1, 1, 1), (2, 5, 50), (1, 40, 7), (1, 9, 17)])
def test_fused_block_matches_separate_launches_and_torch(ops, shape, proj, dtype):
    n, h, w_ = shape
    cin = 64 if proj else 256
    wts = _block(hash((shape, proj)) % 1000, cin, proj, dtype)
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(n, h, w_, cin, generator=g).clamp_(min=-0.5)).to(dtype)  # (mostly non-negative, like a post-ReLU activation)
    p = _packed(wts, DEV)
    y = ops.bottleneck(x.to(DEV), p["w1"], p["b1"], p["w2"], p["b2"], p["w3"], p["b3"], p.get("wsc"), p.get("bsc"))
    assert y is not None and tuple(y.shape) == (n, h, w_, 256) and y.dtype == dtype
    y_sep = _separate(ops, x.to(DEV), p, proj)
    torch.cuda.synchronize()
    ref = _torch_ref(x, wts, proj, dtype)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7  # one unit in the last place of the storage dtype, relative
    scale = float(ref.abs().max())
    # against torch: fp32 summation order differs and an intermediate may round the other way -> a few ulps of the output
    err = float((y.float().cpu() - ref).abs().max())
    assert err <= 4 * eps * scale, (err, scale)
    err_sep = float((y.float() - y_sep.float()).abs().max())
    if proj:
        assert err_sep <= 4 * eps * scale, err_sep  # the shortcut is not rounded to the storage dtype on its own here
    else:
        # same K order, same rounding points, identical MFMA products: the fused block reproduces the separate launches bit for bit
        assert torch.equal(y, y_sep), f"max abs diff {err_sep:.3e} (scale {scale:.3f})"


def test_fused_block_edges_do_not_see_relu_of_the_bias(ops):
    """conv2's zero padding applies to conv1's OUTPUT: a halo pixel outside the image must contribute 0, not relu(b1). A large
    positive b1 makes the difference O(1) on every border pixel."""
    dtype = torch.float16
    wts = _block(3, 256, False, dtype)
    wts["b1"] = torch.full((64,), 5.0)
    x = torch.zeros(1, 8, 16, 256, dtype=dtype)
    p = _packed(wts, DEV)
    y = ops.bottleneck(x.to(DEV), p["w1"], p["b1"], p["w2"], p["b2"], p["w3"], p["b3"])
    ref = _torch_ref(x, wts, False, dtype)
    assert float((y.float().cpu() - ref).abs().max()) <= 4 * 2.0 ** -10 * float(ref.abs().max())
    inner, border = ref[0, 3, 8], ref[0, 0, 0]
    assert float((inner - border).abs().max()) > 0.1  # (the test has teeth: border and interior pixels differ)


def test_unsupported_shapes_fall_back(ops):
    dtype = torch.float16
    x = torch.zeros(1, 8, 16, 128, dtype=dtype, device=DEV)
    w1 = torch.zeros(64, 1, 1, 128, dtype=dtype, device=DEV)
    w2 = torch.zeros(64, 3, 3, 64, dtype=dtype, device=DEV)
    w3 = torch.zeros(256, 1, 1, 64, dtype=dtype, device=DEV)
    b = lambda k: torch.zeros(k, device=DEV)  # noqa: E731
    assert ops.bottleneck(x, w1, b(64), w2, b(64), w3, b(256)) is None  # cin 128: not a res2 shape -> the caller runs the separate launches


def test_engine_uses_the_fused_block_and_agrees_with_the_unfused_engine(osr):
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    g = torch.Generator().manual_seed(5)
    images = torch.randint(0, 256, (2, 3, 160, 224), generator=g, dtype=torch.uint8).to(DEV)
    params = random_params(0)
    eng = OpensetRCNNEngine(params, dtype=torch.float16, device=DEV)
    assert eng.fuse_res2
    keep_f, keep_u = {}, {}
    eng.forward(images, keep=keep_f)
    eng.fuse_res2 = False
    eng.forward(images, keep=keep_u)
    torch.cuda.synchronize()
    a, b = keep_f["res2"].float(), keep_u["res2"].float()
    assert float((a - b).abs().max()) <= 4 * 2.0 ** -10 * float(b.abs().max())  # (block 0's shortcut rounds once less)
    for k in ("p2", "p3", "p4", "p5"):
        fa, fb = keep_f["feats"][k].float(), keep_u["feats"][k].float()
        assert float((fa - fb).abs().max()) <= 2e-2 * float(fb.abs().max()), k


def test_fused_blocks_in_a_chain_between_other_launches_are_exact(ops, osr):
    """Regression test of a write-after-read race in the conv2 tap ring (round 3): a wave's last fragment reads of a weight slot
    could still be in flight when another wave's LDS-DMA for the tap three ahead landed in that slot. It showed as one wrong tile
    in a few thousand, only when the fused launches ran between launches of OTHER kernels and batch sizes (steady-state
    back-to-back launches of the same kernel never showed it). The scenario that reproduced it six times in thirty chains: unfused
    chains of 8, 8, 5 and 3 images, then the fused res2 chain; the identity blocks must reproduce the separate launches bit for
    bit on their own inputs, every time."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    n, h, w = 8, 750, 1333
    g = torch.Generator().manual_seed(21)
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).to(DEV)
    eng = OpensetRCNNEngine(random_params(0), None, torch.float16, DEV)
    c = eng.cfg

    def front(imgs, fused):
        eng.fuse_res2 = fused
        xpad = ops.preprocess(imgs, 768, 1344, c["pixel_mean"], c["pixel_std"], eng.dtype)
        x = ops.stem_conv(xpad, eng.w["backbone.bottom_up.stem.conv1.w"], eng.w["backbone.bottom_up.stem.conv1.b"], 768, 1344, relu=True)
        x = ops.maxpool3x3s2(x)
        outs = [x]
        for b in range(3):
            x = eng._bottleneck(x, f"backbone.bottom_up.res2.{b}", b == 0, 1)
            outs.append(x)
        torch.cuda.synchronize()
        return outs

    for rep in range(4):
        for nn in (8, 8, 5, 3):
            front(images[:nn], False)
        runs = [front(images, True), front(images, True), front(images[:5], True), front(images[5:], True), front(images, True)]
        eng.fuse_res2 = False
        for o in runs:
            for b in (1, 2):
                want = eng._bottleneck(o[b], f"backbone.bottom_up.res2.{b}", False, 1)
                assert torch.equal(o[b + 1], want), (rep, b, int((o[b + 1] != want).sum()))
        assert all(torch.equal(u, v) for u, v in zip(runs[0], runs[1])) and all(torch.equal(u, v) for u, v in zip(runs[0], runs[4]))
        del runs
        torch.cuda.empty_cache()
