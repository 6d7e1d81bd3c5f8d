"""GPU parity test of the fused stem (osr_stem_maxpool_fwd, csrc/osr_stem_pool.hip): [d2] BasicStem.forward = 7x7/s2/p3 conv (FrozenBN
folded) -> ReLU -> max_pool2d(3, 2, 1), /root/reference/configs/Base-RCNN-FPN.yaml:3-8 -- against the two separate launches it replaces
(bit for bit: same K order, same rounding points) and against torch-CPU fp32 on the same fp16-rounded operands."""
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


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,h,w", [(2, 96, 160), (1, 64, 64), (3, 224, 352)])
def test_fused_stem_equals_conv_then_pool(osr, ops, dt, n, h, w):
    from openset_rcnn_amd.host.weights import pack_stem_weight
    g = torch.Generator().manual_seed(7 + h)
    img = torch.randint(0, 256, (n, 3, h - 5, w - 9), generator=g, dtype=torch.uint8).to(DEV)  # ragged: the /32 padding is part of the case
    wt = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    b = (torch.randn(64, generator=g) * 0.5).to(DEV)
    wv = pack_stem_weight(wt, dt).to(DEV)
    xpad = ops.preprocess(img, h, w, (103.53, 116.28, 123.675), (57.0, 57.0, 58.0), dt)
    ref = ops.maxpool3x3s2(ops.stem_conv(xpad, wv, b, h, w, relu=True))
    got = ops.stem_maxpool(xpad, wv, b, h, w)
    assert got.shape == ref.shape == (n, h // 4, w // 4, 64)
    assert torch.equal(got, ref), f"max abs diff {float((got.float() - ref.float()).abs().max())}"
    # and both against torch-CPU fp32 on the same rounded operands (the stem's output is rounded to the storage dtype before the pool)
    x = xpad[:, 3:3 + h, 3:3 + w, :3].float().cpu().permute(0, 3, 1, 2)
    wq = wt.to(dt).float()
    y = F.relu(F.conv2d(x, wq, b.cpu(), stride=2, padding=3)).to(dt).float()
    want = F.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1)
    tol = 2.0 ** (-9 if dt == torch.float16 else -6)
    assert float((got.float().cpu() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))


def test_engine_uses_the_fused_stem_and_keeps_its_results(osr, ops):
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    eng = OpensetRCNNEngine(random_params(0), device=DEV)
    images = torch.randint(0, 256, (2, 3, 256, 384), generator=torch.Generator().manual_seed(3), dtype=torch.uint8).to(DEV)
    assert eng.fuse_stem
    a = eng.forward(images)
    eng.fuse_stem = False
    b = eng.forward(images)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("src_dt", [torch.uint8, torch.float32])
def test_raw_image_variant_equals_preprocess_then_fused_stem(osr, ops, src_dt):
    """osr_stem_maxpool_fwd_raw: [d2] preprocess_image + ImageList.from_tensors folded into the staging -- the same bits as
    osr_preprocess -> osr_stem_maxpool_fwd on uint8 and float32 batches, with an image smaller than its /32-padded size."""
    from openset_rcnn_amd.host.weights import pack_stem_weight
    g = torch.Generator().manual_seed(11)
    n, h, w, hp, wp = 3, 123, 181, 128, 192
    img = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    img = (img.float() + torch.rand(n, 3, h, w, generator=g)).to(DEV) if src_dt == torch.float32 else img.to(DEV)
    wv = pack_stem_weight(torch.randn(64, 3, 7, 7, generator=g) * 0.05, torch.float16).to(DEV)
    b = (torch.randn(64, generator=g) * 0.5).to(DEV)
    mean, std = (103.53, 116.28, 123.675), (57.375, 57.12, 58.395)
    want = ops.stem_maxpool(ops.preprocess(img, hp, wp, mean, std, torch.float16), wv, b, hp, wp)
    got = ops.stem_maxpool_raw(img, hp, wp, mean, std, wv, b)
    assert got.shape == (n, hp // 4, wp // 4, 64) and torch.equal(got, want)
