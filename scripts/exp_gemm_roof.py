"""Calibration (not part of the product): what the vendor's tuned GEMM (hipBLASLt behind torch.matmul, fp16 in / fp32 accumulate) reaches on THIS box
on the GEMM shapes of the product's MFMA-bound layers, beside the product's own kernels on the same shapes. A plain GEMM has no im2col
gather, no halo, no fused epilogue: it is an upper reference for the K loop alone at the clock the box holds under a dense MFMA load
(MI355X_MICROARCH.md: DVFS gives clock back as the matrix pipe fills), not a replacement (library GEMMs need a materialised im2col matrix:
9x the activation bytes for a 3 x 3 layer)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)


def timed(fn, reps=10, rounds=5):
    best, med = 1e30, []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e3
        best = min(best, t); med.append(t)
    return sorted(med)[len(med) // 2], best


def gemm(m, k, n):
    a = (torch.randn(m, k, generator=g) * 0.5).half().cuda()
    b = (torch.randn(n, k, generator=g) / math.sqrt(k)).half().cuda()
    bt = b.t().contiguous()
    out = torch.empty(m, n, dtype=torch.float16, device="cuda")
    return (lambda: torch.matmul(a, b.t(), out=out)), (lambda: torch.matmul(a, bt, out=out)), a, b


rows = [("fc1 68368 x 12544 -> 1024", 68368, 12544, 1024, None),
        ("fpn_output2 as GEMM (16x200x336 rows, K 2304, N 256)", 16 * 200 * 336, 2304, 256, (16, 200, 336, 256, 3)),
        ("fpn_output3 as GEMM (16x100x168 rows)", 16 * 100 * 168, 2304, 256, (16, 100, 168, 256, 3)),
        ("res4 conv2 as GEMM (16x50x84 rows, K 2304, N 256)", 16 * 50 * 84, 2304, 256, (16, 50, 84, 256, 3)),
        ("res5 conv2 as GEMM (16x25x42 rows, K 4608, N 512)", 16 * 25 * 42, 4608, 512, (16, 25, 42, 512, 3)),
        ("res4 conv1 1x1 (67200 x 1024 -> 256)", 67200, 1024, 256, (16, 50, 84, 1024, 1)),
        ("res4 conv3 1x1 (67200 x 256 -> 1024)", 67200, 256, 1024, (16, 50, 84, 256, 1)),
        ("square 8192^3", 8192, 8192, 8192, None)]
print(f"{'shape':56s} {'hipBLASLt NT us (TF/s)':>26s} {'NN us (TF/s)':>22s} {'product kernel us (TF/s)':>28s}", flush=True)
for name, m, k, n, conv in rows:
    nt, nn, a, b = gemm(m, k, n)
    fl = 2.0 * m * k * n
    t_nt, t_nn = timed(nt), timed(nn)
    if conv is None and m != 8192:
        bias = torch.zeros(n, device="cuda")
        mine = timed(lambda: ops.linear(a, b, bias, relu=True))
    elif conv is not None:
        nb, h, w, cin, kk = conv
        x = (torch.randn(nb, h, w, cin, generator=g) * 0.5).half().cuda()
        wt = (torch.randn(n, kk, kk, cin, generator=g) / math.sqrt(kk * kk * cin)).half().cuda()
        bias = torch.zeros(n, device="cuda")
        mine = timed(lambda: ops.conv2d(x, wt, bias, 1, kk // 2, relu=True))
    else:
        bias = torch.zeros(n, device="cuda")
        mine = timed(lambda: ops.linear(a, b, bias, relu=False))
    f = lambda t: f"{t[0]:8.1f} / {t[1]:8.1f} ({fl / t[1] / 1e6:6.0f})"
    print(f"{name:56s} {f(t_nt):>26s} {f(t_nn):>22s} {f(mine):>28s}", flush=True)
    del a, b, nt, nn
    torch.cuda.empty_cache()
