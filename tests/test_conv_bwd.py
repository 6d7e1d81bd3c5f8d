"""GPU parity of the dense-layer backward kernels against torch autograd (fp32 on the same fp16-rounded operands).
Tolerance: fp32 accumulation on both sides, different summation order -> 2e-3 of the tensor's max (weight gradients sum up to
~10^4 products of fp16-rounded factors); data gradients are stored in fp16 -> one rounding."""
import ctypes

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


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


CASES = [  # n, cin, cout, h, w, k, stride, pad
    (2, 256, 256, 13, 21, 3, 1, 1),
    (2, 512, 128, 25, 42, 1, 1, 0),
    (2, 256, 512, 26, 42, 1, 2, 0),
    (1, 64, 192, 17, 9, 3, 1, 1),      # partial channel tiles
    (1, 1024, 256, 300, 1, 1, 1, 0),   # fully connected: rows = h
    (3, 128, 128, 7, 5, 3, 1, 1),      # M = 105: one short step
]


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad", CASES)
def test_wgrad_and_dgrad_vs_autograd(ops, osr, n, cin, cout, h, w, k, stride, pad):
    from openset_rcnn_amd.host.weights import pack_dgrad_weight
    gg = g(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=gg).half().float().requires_grad_(True)
    wt = (torch.randn(cout, cin, k, k, generator=gg) * 0.05).half().float().requires_grad_(True)
    y = F.conv2d(x, wt, None, stride, pad)
    dy = torch.randn(y.shape, generator=gg).half().float()
    y.backward(dy)
    xd, dyd = nhwc(x.detach()).half().to(DEV), nhwc(dy).half().to(DEV)
    dw = ops.conv2d_wgrad(xd, dyd, k, k, stride, pad)
    assert tuple(dw.shape) == (cout, k, k, cin)
    assert rel(dw.permute(0, 3, 1, 2), wt.grad) < 2e-3, f"wgrad rel err {rel(dw.permute(0, 3, 1, 2), wt.grad)}"
    dw2 = ops.conv2d_wgrad(xd, dyd, k, k, stride, pad, dw=dw.clone(), accumulate=True)
    assert rel(dw2, 2 * dw) < 1e-6
    assert torch.equal(ops.conv2d_wgrad(xd, dyd, k, k, stride, pad), dw), "fixed-order split-K reduction must be bitwise reproducible"
    db = ops.bias_grad(dyd)
    assert rel(db, dy.sum(dim=(0, 2, 3))) < 1e-4
    # data gradient: plain, with the ReLU mask of the layer below, and with a second gradient added
    wd = pack_dgrad_weight(wt.detach(), torch.float16).to(DEV)
    dx = ops.conv2d_dgrad(dyd, wd, (h, w), stride, pad, out_dtype=torch.float32)
    assert rel(dx.permute(0, 3, 1, 2), x.grad) < 2e-3
    act = torch.randn(n, h, w, cin, generator=gg).half()
    dxm = ops.conv2d_dgrad(dyd, wd, (h, w), stride, pad, mask=act.to(DEV), out_dtype=torch.float32)
    assert rel(dxm, nhwc(x.grad) * (act.float() > 0)) < 2e-3
    other = torch.randn(n, h, w, cin, generator=gg).half()
    dxa = ops.conv2d_dgrad(dyd, wd, (h, w), stride, pad, add=other.to(DEV), out_dtype=torch.float32)
    ref = nhwc(x.grad) + other.float()
    if stride == 2:  # the addend is only read at the pixels the strided layer writes; the rest of dx stays zero
        keep = torch.zeros_like(ref)
        keep[:, ::2, ::2] = 1
        ref = ref * keep
    assert rel(dxa, ref) < 2e-3


def test_wgrad_single_split_writes_dw_directly(ops):
    """A layer with enough weight tiles to fill the GPU on its own (FC1: 4 x 49 tiles of 256 x 256) runs as ONE split, and without an
    accumulate the kernel then writes dw itself -- no workspace copy, no reduction launch. Checked against the fp32 product on the CPU,
    into a view of a larger buffer (neighbouring elements untouched), and against the accumulating form (which keeps the reduction)."""
    gg = g(77)
    m, k, nout = 300, 12544, 1024
    x = (torch.randn(m, k, generator=gg) * 0.5).half()
    dy = (torch.randn(m, nout, generator=gg) * 0.1).half()
    ref = dy.float().t() @ x.float()
    flat = torch.full((nout * k + 64,), 7.0, dtype=torch.float32, device=DEV)
    dw = flat[32:32 + nout * k].view(nout, 1, 1, k)
    ops.conv2d_wgrad(x.to(DEV).view(1, m, 1, k), dy.to(DEV).view(1, m, 1, nout), 1, 1, dw=dw)
    assert rel(dw.view(nout, k), ref) < 2e-3
    assert float(flat[:32].min()) == 7.0 and float(flat[32 + nout * k:].max()) == 7.0
    first = dw.clone()
    ops.conv2d_wgrad(x.to(DEV).view(1, m, 1, k), dy.to(DEV).view(1, m, 1, nout), 1, 1, dw=dw, accumulate=True)
    assert rel(dw, 2 * first) < 1e-6
    assert torch.equal(ops.linear_wgrad(x.to(DEV), dy.to(DEV)), first.view(nout, k))


# (n, cin, cout, h, w, k, stride): splits of >= 48 sixty-four-row steps -> the 8-phase loop of the 256 x 256 weight-gradient tile (round 5); the
# cases above are all short splits (the one-barrier loop). 3x3 with zero padding, a strided 1x1 (the gather's row / image wraps), the FC
# layers' (m, 1) view (one pixel per row: a wrap at every row), bf16.
P8_CASES = [(8, 256, 256, 100, 168, 3, 1, torch.float16), (16, 1024, 1024, 100, 168, 1, 2, torch.float16), (1, 12544, 1024, 8192, 1, 1, 1, torch.float16),
            (8, 256, 256, 100, 168, 3, 1, torch.bfloat16), (3, 1024, 1024, 211, 97, 1, 1, torch.float16), (8, 1024, 1024, 157, 211, 1, 2, torch.float16)]


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,dtype", P8_CASES)
def test_wgrad_8phase_loop_vs_fp32_reference(ops, osr, n, cin, cout, h, w, k, stride, dtype):
    L = osr._lib
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    # the split the library will use: rows per split / 64 >= 48 selects the 8-phase loop (osr_conv_bwd.hip, WG_P8_MIN_STEPS)
    p = L.ConvParams()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, h, w, cin, ho, wo, cout
    p.kh = p.kw = k; p.stride_h = p.stride_w = stride; p.pad_h = p.pad_w = pad
    p.in_dtype = p.out_dtype = L.OSR_F16 if dtype == torch.float16 else L.OSR_BF16
    splits = int(L.load().osr_conv2d_wgrad_workspace_bytes(ctypes.byref(p))) // (cout * k * k * cin * 4)
    assert (n * ho * wo) // max(splits, 1) // 64 >= 48, "this case no longer reaches the 8-phase loop"
    gg = torch.Generator(device=DEV).manual_seed(cin + h)
    x = (torch.randn(n, h, w, cin, generator=gg, device=DEV) * 0.5).to(dtype)
    dy = (torch.randn(n, ho, wo, cout, generator=gg, device=DEV) * 0.1).to(dtype)
    dw = ops.conv2d_wgrad(x, dy, k, k, stride, pad)
    assert torch.equal(ops.conv2d_wgrad(x, dy, k, k, stride, pad), dw), "not reproducible"
    if k == 1:  # dw[co][ci] = sum over the (strided) pixels
        xs = x[:, ::stride, ::stride].reshape(-1, cin).float()
        ref = (dy.reshape(-1, cout).float().t() @ xs).view(cout, 1, 1, cin)
    else:
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (cout, cin, k, k), dy.float().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    err = float((dw - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, err  # fp32 accumulation in another order over up to 134 400 rows
