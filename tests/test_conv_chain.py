"""osr_conv2d_chain_fwd (a bottleneck's conv2 -> conv3 + residual in one launch, the res3 blocks) against the two osr_conv2d_fwd
launches it replaces -- bit for bit: same K order, same rounding point of the intermediate -- and against torch-CPU convolutions on
identically rounded operands. Shapes cover whole tiles, a ragged last tile (rows not a multiple of 128), one-pixel images (every tap
but the centre in the padding), a 1x1 first convolution, stride 2, and both storage dtypes."""
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


def _weights(seed, cin, k, dtype):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc  # noqa: E731
    return dict(w2=rnd(128, cin, k, k, sc=(2.0 / (cin * k * k)) ** 0.5).to(dtype), b2=rnd(128, sc=0.3), w3=rnd(512, 128, 1, 1, sc=(1.0 / 128) ** 0.5).to(dtype),
                b3=rnd(512, sc=0.3))


def _nhwc(w):
    return w.permute(0, 2, 3, 1).contiguous().to(DEV)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 16, 32, 128, 3, 1, 1), (1, 8, 16, 128, 3, 1, 1), (3, 21, 37, 128, 3, 1, 1), (1, 1, 1, 128, 3, 1, 1),
                                  (2, 13, 50, 64, 1, 1, 0), (1, 40, 28, 256, 3, 2, 1), (1, 100, 168, 128, 3, 1, 1)])
def test_chain_matches_two_launches_and_torch(ops, case, dtype):
    n, h, w_, cin, k, stride, pad = case
    wt = _weights(sum(case), cin, k, dtype)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, h, w_, cin, generator=g).to(dtype)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w_ + 2 * pad - k) // stride + 1
    res = torch.randn(n, ho, wo, 512, generator=g).to(dtype)
    xd, rd = x.to(DEV), res.to(DEV)
    w2, w3, b2, b3 = _nhwc(wt["w2"]), _nhwc(wt["w3"]), wt["b2"].to(DEV), wt["b3"].to(DEV)
    y = ops.conv2d_chain(xd, w2, b2, w3, b3, rd, stride, pad)
    assert y is not None and y.shape == (n, ho, wo, 512)
    o2 = ops.conv2d(xd, w2, b2, stride, pad, relu=True)
    sep = ops.conv2d(o2, w3, b3, relu=True, residual=rd, res_mode=1)
    torch.cuda.synchronize()
    assert torch.equal(y, sep), f"fused != separate launches: {(y.float() - sep.float()).abs().max().item()}"
    # osr_conv2d_chain_fwd_ex: the first convolution's output stored as well (the training step keeps it): the separate launch's bits
    y2, mid = ops.conv2d_chain(xd, w2, b2, w3, b3, rd, stride, pad, keep_mid=True)
    torch.cuda.synchronize()
    assert torch.equal(y2, sep) and mid.shape == o2.shape and torch.equal(mid, o2)
    if n * ho * wo <= 4096:  # fp32 math on the rounded operands, the intermediate rounded where the kernels round it
        r = lambda t: t.to(dtype).float()  # noqa: E731
        o = r(F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), wt["w2"].float(), wt["b2"], stride=stride, padding=pad)))
        ref = r(F.relu(F.conv2d(o, wt["w3"].float(), wt["b3"]) + res.float().permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
        tol = 2e-2 if dtype == torch.float16 else 1.5e-1
        assert (y.float().cpu() - ref).abs().max().item() <= tol


def test_chain_between_other_launches_is_exact(ops):
    """The weight-stage ring is restaged behind counted waits + a barrier: a write-after-read race there would show as a wrong
    tile only between other kernels (the res2 block's history, tests/test_bottleneck.py). Chains of fused launches interleaved with
    unrelated convolutions, several batch sizes, every result compared with the separate launches."""
    dtype = torch.float16
    wt = _weights(5, 128, 3, dtype)
    w2, w3, b2, b3 = _nhwc(wt["w2"]), _nhwc(wt["w3"]), wt["b2"].to(DEV), wt["b3"].to(DEV)
    g = torch.Generator().manual_seed(3)
    for n in (1, 3, 5, 8):
        x = torch.randn(n, 50, 84, 128, generator=g).to(dtype).to(DEV)
        res = torch.randn(n, 50, 84, 512, generator=g).to(dtype).to(DEV)
        sep = ops.conv2d(ops.conv2d(x, w2, b2, 1, 1, relu=True), w3, b3, relu=True, residual=res, res_mode=1)
        for _ in range(6):
            junk = ops.conv2d(res, w3.permute(3, 1, 2, 0).contiguous(), b2)  # an unrelated launch in between (512 -> 128)
            y = ops.conv2d_chain(x, w2, b2, w3, b3, res, 1, 1)
            assert torch.equal(y, sep)
            del junk


def test_chain_declines_other_shapes(ops):
    x = torch.zeros(1, 8, 8, 64, dtype=torch.float16, device=DEV)
    w2 = torch.zeros(64, 3, 3, 64, dtype=torch.float16, device=DEV)
    w3 = torch.zeros(256, 1, 1, 64, dtype=torch.float16, device=DEV)
    res = torch.zeros(1, 8, 8, 256, dtype=torch.float16, device=DEV)
    assert ops.conv2d_chain(x, w2, torch.zeros(64, device=DEV), w3, torch.zeros(256, device=DEV), res, 1, 1) is None


def test_engine_runs_res3_through_the_chain_and_is_bit_identical_without_it(osr):
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    g = torch.Generator().manual_seed(5)
    images = torch.randint(0, 256, (2, 3, 160, 224), generator=g, dtype=torch.uint8).to(DEV)
    eng = OpensetRCNNEngine(random_params(0), dtype=torch.float16, device=DEV)
    assert eng.chain_res3
    eng.profile = []
    keep_c, keep_u = {}, {}
    eng.forward(images, keep=keep_c)
    assert sum("chained" in t[0] for t in eng.profile) == 4  # the four res3 blocks
    eng.profile = None
    eng.chain_res3 = False
    eng.forward(images, keep=keep_u)
    torch.cuda.synchronize()
    assert torch.equal(keep_c["res3"], keep_u["res3"])
    for k in ("p2", "p3", "p4", "p5"):
        assert torch.equal(keep_c["feats"][k], keep_u["feats"][k]), k


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_chain_under_load_from_a_second_stream(osr, dtype):
    """The scenario that exposed a store-data hazard in the first form of the kernel: whole passes on two streams at once, so that the
    chained launches of one micro-batch run beside the other micro-batch's RPN / RoI-head kernels. A 16-byte buffer store reads its
    data registers after issue; overwritten at once, lanes 12-15 of every 16 stored the next value (one run in three, never in a
    single stream). Twenty passes, every detection tensor compared with the single-stream pass."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    g = torch.Generator().manual_seed(7)
    images = torch.randint(0, 256, (2, 3, 250, 330), generator=g, dtype=torch.uint8)
    imgs = torch.cat([images, images.flip(0)]).to(DEV)
    hw = torch.tensor([(250, 330), (240, 300), (240, 300), (250, 330)], dtype=torch.int32, device=DEV)
    eng = OpensetRCNNEngine(random_params(0), dtype=dtype, device=DEV)
    assert eng.chain_res3
    ref = eng.forward_device(imgs, hw, 256, 352)
    torch.cuda.synchronize()
    for rep in range(20):
        out = eng.forward_device_streams(imgs, hw, 256, 352, nstreams=2 + 2 * (rep % 2))
        torch.cuda.synchronize()
        for a, b in zip(ref, out):
            assert torch.equal(a, b), f"pass {rep}"
