"""The 8-phase K loop of the 256 x 256 conv / FC tile (csrc/osr_conv_gemm64.hip, TWO == 2: FC1, fpn_output2/3, the fused CF-RPN head):
  * against fp32 / fp64 convolutions of the identically rounded operands (torch), and
  * BIT FOR BIT against the 128 x 128 one-barrier-per-slice kernel, which walks the same K slices in the same order per output element
    (an independent kernel with a different staging scheme: a unit landing late or restaged early in the 8-phase ring shows here),
at shapes that cover an even / odd / minimal number of K tiles, a ragged last M tile, per-image padded row lists, both storage
dtypes, fp32 output, the split-K tail launch (whose partial-sum workgroups run the same loop from a K offset), and the fused head.
A second stream hammers memory during a repeat screen (the ring's counted waits must hold whatever the load latency is)."""
import ctypes as C
import math

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


def _describe(osr, n, h, w, cin, cout, k, conc, ws=0):
    L = osr._lib
    p = L.ConvParams()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, h, w, cin, h, w, cout
    p.kh = p.kw = k; p.stride_h = p.stride_w = 1; p.pad_h = p.pad_w = k // 2
    p.in_stride_n, p.in_stride_h, p.in_stride_w = h * w * cin, w * cin, cin
    p.out_stride_n, p.out_stride_h, p.out_stride_w = h * w * cout, w * cout, cout
    p.in_dtype = p.out_dtype = L.OSR_F16
    p.concurrency = conc
    buf = C.create_string_buffer(256)
    L.load().osr_conv2d_fwd_describe(C.byref(p), ws, buf, 256)
    return buf.value.decode()


# (n, h, w, cin, cout, k): K tiles = k*k*cin/64 -- 36 (even), 15 (odd), 4 and 5 (the shortest loops the cost model sends to this tile: little
# more than the ring's prologue and drain), 16 with a ragged last tile
CASES = [(4, 100, 168, 256, 256, 3), (1, 70000, 1, 960, 1024, 1), (1, 131000, 1, 256, 256, 1), (1, 131000, 1, 320, 512, 1), (1, 70001, 1, 1024, 1024, 1)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_8phase_matches_torch_and_the_128_tile_kernel(osr, ops, case, dtype):
    n, h, w, cin, cout, k = case
    assert _describe(osr, n, h, w, cin, cout, k, 2).startswith("256x256"), "the cost model no longer sends this shape to the 256 x 256 tile"
    assert _describe(osr, n, h, w, cin, cout, k, 1).startswith("128x"), "expected a 128-row tile without the concurrency hint"
    g = torch.Generator().manual_seed(sum(case))
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).to(dtype).to(DEV)
    wt = (torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin)).to(dtype).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    prev = ops.SPLIT_K_TAIL
    ops.SPLIT_K_TAIL = False
    try:
        with ops.concurrent_streams(2):  # a tile-selection hint only (osr_conv_params.concurrency)
            y = ops.conv2d(x, wt, b, 1, k // 2, relu=True)
            y32 = ops.conv2d(x, wt, b, 1, k // 2, relu=False, out_dtype=torch.float32)
        small = ops.conv2d(x, wt, b, 1, k // 2, relu=True)
        small32 = ops.conv2d(x, wt, b, 1, k // 2, relu=False, out_dtype=torch.float32)
    finally:
        ops.SPLIT_K_TAIL = prev
    torch.cuda.synchronize()
    assert torch.equal(y, small) and torch.equal(y32, small32), f"256-tile != 128-tile kernel: {(y32 - small32).abs().max().item()}"
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), b, padding=k // 2).permute(0, 2, 3, 1)
    err = (y32 - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err  # fp32 accumulation in another order
    rows = torch.randint(0, n * h * w, (64,), generator=g)
    if k == 1:  # fp64 on a sample of rows
        r64 = x.view(-1, cin)[rows.to(DEV)].double() @ wt.view(cout, cin).double().t() + b.double()
        assert (y32.view(-1, cout)[rows.to(DEV)].double() - r64).abs().max().item() <= 1e-4 * max(1.0, r64.abs().max().item())


def test_8phase_residual_epilogues_on_deep_k_layers(osr, ops):
    """From 18 K slices on a residual layer may take the 256 x 256 tile (the data gradients of the 3 x 3 FPN output convolutions: ReLU mask
    of the layer below = res_mode 3, gradient sum = res_mode 1, osr_conv2d_fwd_masked's post-mask): the epilogue reads the residual
    unprefetched; results equal the 128-row kernel's, which prefetches it under the K loop."""
    n, h, w, c, k = 4, 100, 168, 256, 3
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(n, h, w, c, generator=g) * 0.5).half().to(DEV)
    wt = (torch.randn(c, k, k, c, generator=g) / 48).half().to(DEV)
    b = torch.zeros(c, device=DEV)
    res = torch.randn(n, h, w, c, generator=g).half().to(DEV)
    mask = torch.randn(n, h, w, c, generator=g).half().to(DEV)
    for mode in (1, 3):
        pm = mask if mode == 1 else None  # (osr_conv2d_fwd_masked takes res_mode 0 / 1)
        with ops.concurrent_streams(2):
            big = ops.conv2d(x, wt, b, 1, 1, relu=False, residual=res, res_mode=mode, post_mask=pm)
        small = ops.conv2d(x, wt, b, 1, 1, relu=False, residual=res, res_mode=mode, post_mask=pm)
        torch.cuda.synchronize()
        assert torch.equal(big, small), mode
        y = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        ref = torch.where(mask.float() > 0, y + res.float(), torch.zeros_like(y)) if mode == 1 else torch.where(res.float() > 0, y, torch.zeros_like(y))
        assert (big.float() - ref).abs().max().item() <= 2e-2


def test_8phase_ragged_row_lists_and_split_k_tail(osr, ops):
    """FC1 as the engine runs it: 16 padded lists of 4273 slots, tiles without a data row skipped; then the same layer with the
    partial last dispatch round cut along K (osr_conv2d_fwd + workspace): full rounds + split-K tail workgroups + reduce."""
    g = torch.Generator().manual_seed(21)
    m, kdim, nout = 16 * 4273, 12544, 1024
    x = (torch.randn(m, kdim, generator=g) * 0.5).half().to(DEV)
    wt = (torch.randn(nout, kdim, generator=g) / 112).half().to(DEV)
    b = torch.randn(nout, generator=g).to(DEV)
    counts = [4273, 3000, 4000, 100, 0, 4273, 2500, 3999, 4273, 1, 255, 257, 4100, 3800, 2900, 4273]
    keep = torch.zeros(m, dtype=torch.bool, device=DEV)
    for i, c in enumerate(counts):
        keep[i * 4273:i * 4273 + c] = True
    assert "256x256" in _describe(osr, 1, m, 1, kdim, nout, 1, 1, ws=1) and "split-K" in _describe(osr, 1, m, 1, kdim, nout, 1, 1, ws=1)
    full = ops.linear(x, wt, b, relu=True)  # with the split-K tail
    seg = ops.linear(x, wt, b, relu=True, row_seg=(torch.tensor(counts, dtype=torch.int32, device=DEV), 4273))
    prev = ops.SPLIT_K_TAIL
    ops.SPLIT_K_TAIL = False
    try:
        single = ops.linear(x, wt, b, relu=True)
    finally:
        ops.SPLIT_K_TAIL = prev
    torch.cuda.synchronize()
    assert torch.equal(seg[keep], full[keep])  # (same launch plan: full rounds + split-K tail; the lists only skip tiles)
    head = 65536  # rows of the full dispatch rounds: identical with and without the tail split
    assert torch.equal(full[:head], single[:head])
    idx = torch.randint(0, m, (96,), generator=g).to(DEV)
    r64 = F.relu(x[idx].double() @ wt.double().t() + b.double())
    for out in (full, single):
        assert (out[idx].double() - r64).abs().max().item() <= 2e-3 * max(1.0, r64.abs().max().item())  # fp16 output rounding


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_8phase_fused_head_matches_the_unfused_launches(ops, dtype):
    """The fused CF-RPN head on 256-row tiles (>= 512 tiles: the kernel with the 8-phase loop) against osr_conv2d_fwd +
    osr_cfrpn_head_tail; the hidden state it parks in LDS is the separate launch's output bit for bit. The tail runs on the matrix
    cores (fp32 tail weights as three storage-dtype terms: three 8-bit significands for bf16): within 1e-5 of the un-fused fp32 FMA tail."""
    g = torch.Generator().manual_seed(9)
    n, h, w = 2, 200, 336
    x = (torch.randn(n, h, w, 256, generator=g) * 0.5).to(dtype).to(DEV)
    wt = (torch.randn(256, 3, 3, 256, generator=g) / 48).to(dtype).to(DEV)
    b = (torch.randn(256, generator=g) * 0.1).to(DEV)
    wtail = (torch.randn(5, 256, generator=g) * 0.05).to(DEV)
    btail = (torch.randn(5, generator=g) * 0.1).to(DEV)
    hidden = torch.empty(n * h * w, 256, dtype=dtype, device=DEV)
    d, c = ops.cfrpn_head_fused(x, wt, b, wtail, btail, hidden_out=hidden)
    t = ops.conv2d(x, wt, b, 1, 1, relu=True)
    d2, c2 = ops.cfrpn_head_tail(t.view(-1, 256), wtail[:4].contiguous(), btail[:4].contiguous(), wtail[4:].contiguous(), btail[4:].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(hidden.view_as(t), t)
    assert (d - d2.view_as(d)).abs().max().item() <= 1e-5 and (c - c2.view_as(c)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("scale", [1e-4, 1e-2, 30.0])
def test_fused_head_tail_weights_of_any_magnitude_keep_fp32_accuracy(ops, dtype, scale):
    """the tail's fp32 weights enter the matrix cores as three storage-dtype terms. fp16 terms run out of exponent range for
    small weights (mid subnormal, lo flushed) unless each tail row is first scaled by a power of two: with weights around 1e-4 (what
    weight decay drives the rows towards), at the init scale and at 30, the fused head stays within fp32 rounding of the fp64 value
    computed from the parked hidden state -- relative to the row's own magnitude, so a truncated low term would show."""
    g = torch.Generator().manual_seed(21)
    n, h, w = 1, 64, 96
    x = (torch.randn(n, h, w, 256, generator=g) * 0.5).to(dtype).to(DEV)
    wt = (torch.randn(256, 3, 3, 256, generator=g) / 48).to(dtype).to(DEV)
    b = (torch.randn(256, generator=g) * 0.1).to(DEV)
    wtail = torch.randn(5, 256, generator=g) * scale
    wtail[1] *= 37.0  # rows of different magnitude: the scale is per row
    wtail[2, ::3] *= 2.0 ** -9  # weights far below their row's largest
    wtail = wtail.to(DEV)
    btail = torch.zeros(5, device=DEV)
    hidden = torch.empty(n * h * w, 256, dtype=dtype, device=DEV)
    d, c = ops.cfrpn_head_fused(x, wt, b, wtail, btail, hidden_out=hidden)
    torch.cuda.synchronize()
    t = hidden.double()
    tn = t / t.norm(dim=1, keepdim=True).clamp_min(1e-12)
    ref = tn @ wtail.double().t()  # (rows, 5)
    got = torch.cat((d.view(-1, 4).double(), torch.logit(c.view(-1, 1).double().clamp(1e-7, 1 - 1e-7))), dim=1)
    for q in range(4):  # the four delta rows: relative to the row's typical output (|t_n| = 1, so ~ |w_q| in the mean)
        mag = float(wtail[q].abs().max()) * 4
        err = float((got[:, q] - ref[:, q]).abs().max())
        assert err <= 3e-6 * mag, (q, err, mag)  # fp32 accumulation of 256 terms; the fp16 truncation was ~3e-6 x 16 x |w| and up
    # centerness goes through a sigmoid: checked where it is not saturated
    keep = ref[:, 4].abs() < 6
    assert float((got[keep, 4] - ref[keep, 4]).abs().max()) <= 1e-5 * max(1.0, float(wtail[4].abs().max()) * 4) + 2e-5


def test_8phase_repeat_screen_beside_a_second_stream(ops):
    """Forty launches each of a 3 x 3 and a deep 1 x 1 layer while another stream streams 256 MB through HBM: every output equal to the
    first (a unit read before it landed, or restaged before its last read retired, would differ in some tile)."""
    g = torch.Generator().manual_seed(4)
    side = torch.cuda.Stream()
    hammer = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    with ops.concurrent_streams(2):
        for (n, h, w, cin, cout, k) in [(4, 100, 168, 256, 256, 3), (1, 70000, 1, 2048, 512, 1)]:
            x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().to(DEV)
            wt = (torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin)).half().to(DEV)
            b = torch.randn(cout, generator=g).to(DEV)
            ref = ops.conv2d(x, wt, b, 1, k // 2, relu=True).clone()
            for i in range(40):
                if i % 2:
                    with torch.cuda.stream(side):
                        hammer.fill_(i & 255); hammer.add_(1)
                assert torch.equal(ops.conv2d(x, wt, b, 1, k // 2, relu=True), ref), f"launch {i}"
    torch.cuda.synchronize()
