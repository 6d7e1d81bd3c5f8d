"""Diagnostic (not part of the product): phase times of the chained conv2 -> conv3 kernel (osr_conv2d_chain_fwd) from s_memrealtime
stamps. Needs a library built with OSR_EXTRA_HIPCC_FLAGS=-DC64_STAMPS (python openset-rcnn_amd/build.py --force)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg._lib.load()
ops = pkg.ops
lib.osr_debug_set_conv_stamps.argtypes = [C.c_void_p]
lib.osr_debug_set_conv_stamps.restype = None
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
dt = torch.float16
n = 16
x = torch.randn(n, 100, 168, 128, generator=g).to(dt).to(dev)
res = torch.randn(n, 100, 168, 512, generator=g).to(dt).to(dev)
w2 = (torch.randn(128, 3, 3, 128, generator=g) * (2.0 / 1152) ** 0.5).to(dt).to(dev)
w3 = (torch.randn(512, 1, 1, 128, generator=g) * (1.0 / 128) ** 0.5).to(dt).to(dev)
b2 = (torch.randn(128, generator=g) * 0.3).to(dev)
b3 = (torch.randn(512, generator=g) * 0.3).to(dev)
stamps = torch.zeros((1 << 16) * 4, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.conv2d_chain(x, w2, b2, w3, b3, res, 1, 1)
torch.cuda.synchronize()
lib.osr_debug_set_conv_stamps(C.c_void_p(stamps.data_ptr()))
ops.conv2d_chain(x, w2, b2, w3, b3, res, 1, 1)
torch.cuda.synchronize()
lib.osr_debug_set_conv_stamps(None)
s = stamps.view(-1, 4).cpu()
s = s[s[:, 3] > 0].double()
t = s * 0.01
span = float(t[:, 3].max() - t[:, 0].min())
ph = [(t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2]), (t[:, 3] - t[:, 0])]
start = t[:, 0] - t[:, 0].min()
print(f"{len(s)} workgroups, stamped span {span:.1f} us")
for nm, p in zip(("entry->first slice landed", "K loop (conv2, 18 slices)", "park + 8 stages of conv3", "whole workgroup"), ph):
    print(f"    {nm:28s} mean {float(p.mean()):6.2f} us   p10 {float(p.quantile(0.1)):6.2f}   p50 {float(p.quantile(0.5)):6.2f}   p90 {float(p.quantile(0.9)):6.2f}")
print(f"    workgroup start times: p10 {float(start.quantile(0.1)):.1f}  p50 {float(start.quantile(0.5)):.1f}  p90 {float(start.quantile(0.9)):.1f} us;"
      f" concurrency = sum(wg time)/span = {float(ph[3].sum()) / span:.1f} workgroups")
