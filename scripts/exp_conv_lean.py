"""Recorded experiment: the 128 x 128 single-buffer kernel at four workgroups per CU (one fragment set, no residual prefetch:
OSR_CONV_LEAN=1 without residual layers, =2 with them; -DOSR_EXPERIMENT build). Per-layer time + checksum."""
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
cases = [("res3.conv1", 16, 100, 168, 512, 128, 1, False), ("res4.conv1", 16, 50, 84, 1024, 256, 1, False), ("res4.conv2", 16, 50, 84, 256, 256, 3, False),
         ("res4.conv3", 16, 50, 84, 256, 1024, 1, True), ("res5.conv1", 16, 25, 42, 2048, 512, 1, False), ("res5.conv2", 16, 25, 42, 512, 512, 3, False),
         ("res5.conv3", 16, 25, 42, 512, 2048, 1, True), ("fpn_out4", 16, 50, 84, 256, 256, 3, False), ("lateral4", 16, 50, 84, 1024, 256, 1, False)]
tot = 0.0
for name, n, h, w, cin, cout, k, res in cases:
    x = torch.randn(n, h, w, cin, generator=g).to(dt).to(dev)
    wt = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(dt).to(dev)
    b = (torch.randn(cout, generator=g) * 0.3).to(dev)
    r = torch.randn(n, h, w, cout, generator=g).to(dt).to(dev) if res else None
    f = lambda: ops.conv2d(x, wt, b, 1, k // 2, relu=True, residual=r, res_mode=1 if res else 0)  # noqa: E731
    y = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        f()
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print(f"{name:12s} {us:8.1f} us {2.0 * n * h * w * cout * k * k * cin / us / 1e6:8.1f} TFLOP/s sha {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
print(f"sum {tot:.1f} us")
