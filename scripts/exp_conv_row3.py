"""Recorded experiment (not in the product): row-shared pixel staging of the 3x3 convolutions. Apply scripts/exp_conv_row3.patch to
csrc/osr_conv_gemm64.hip, build with OSR_EXTRA_HIPCC_FLAGS=-DOSR_EXPERIMENT, run this once with OSR_CONV_ROW3 unset, =1 (128 x 128 tiles)
and =2 (256 x 256 too): per-layer time and a checksum of every output, so that the runs can be compared bit for bit. Result (DESIGN.md
section 3): bit-identical, a third fewer staged bytes, 10-30 % SLOWER on res3-res5 / FPN p4 and 2-4 % slower on the 256 x 256 layers."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg._lib.load()
ops = pkg.ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
dt = torch.float16
cases = [("res3.conv2", 16, 100, 168, 128, 128), ("res4.conv2", 16, 50, 84, 256, 256), ("res5.conv2", 16, 25, 42, 512, 512),
         ("fpn_out3", 16, 100, 168, 256, 256), ("fpn_out4", 16, 50, 84, 256, 256), ("fpn_out5", 16, 25, 42, 256, 256),
         ("fpn_out2", 16, 200, 336, 256, 256), ("ragged", 3, 21, 37, 128, 128), ("tiny", 1, 1, 1, 64, 128), ("thin", 2, 5, 1, 64, 64), ("wide", 1, 3, 200, 64, 128)]
for name, n, h, w, cin, cout in cases:
    x = torch.randn(n, h, w, cin, generator=g).to(dt).to(dev)
    wt = (torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dt).to(dev)
    b = (torch.randn(cout, generator=g) * 0.3).to(dev)
    y = ops.conv2d(x, wt, b, 1, 1, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.conv2d(x, wt, b, 1, 1, relu=True)
    e0.record()
    for _ in range(20):
        ops.conv2d(x, wt, b, 1, 1, relu=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * n * h * w * cout * 9 * cin
    print(f"{name:12s} {us:9.1f} us {fl / us / 1e6:8.1f} TFLOP/s  sha {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
if len(sys.argv) > 1:  # the chained kernel too
    x = torch.randn(16, 100, 168, 128, generator=g).to(dt).to(dev)
    res = torch.randn(16, 100, 168, 512, generator=g).to(dt).to(dev)
    w2 = (torch.randn(128, 3, 3, 128, generator=g) * (2.0 / 1152) ** 0.5).to(dt).to(dev)
    w3 = (torch.randn(512, 1, 1, 128, generator=g) * (1.0 / 128) ** 0.5).to(dt).to(dev)
    b2, b3 = torch.zeros(128, device=dev), torch.zeros(512, device=dev)
    y = ops.conv2d_chain(x, w2, b2, w3, b3, res, 1, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_chain(x, w2, b2, w3, b3, res, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    print(f"chain        {e0.elapsed_time(e1) / 20 * 1e3:9.1f} us  sha {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16]}")
