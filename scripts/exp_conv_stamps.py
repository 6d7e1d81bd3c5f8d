"""Diagnostic (not part of the product): per-workgroup phase times of conv_igemm64_kernel from s_memrealtime stamps.
Needs a library built with OSR_EXTRA_HIPCC_FLAGS=-DC64_STAMPS (python openset-rcnn_amd/build.py --force)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
if os.environ.get("OSR_VARIANT_LIB"):
    pkg._lib.LIB_PATH = os.environ["OSR_VARIANT_LIB"]
lib = pkg._lib.load()
from openset_rcnn_amd.host import ops
from openset_rcnn_amd.host.weights import pack_conv_weight

lib.osr_debug_set_conv_stamps.argtypes = [C.c_void_p]
lib.osr_debug_set_conv_stamps.restype = None
dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def case(name, n, h, w, cin, cout, k, stride=1, pad=0, residual=False, relu=True):
    x = torch.randn(n, h, w, cin, generator=g).half().to(dev)
    wt = pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * 0.05, torch.float16).to(dev)
    b = torch.zeros(cout).to(dev)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, ho, wo, cout, generator=g).half().to(dev) if residual else None
    stamps = torch.zeros((1 << 16) * 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.conv2d(x, wt, b, stride, pad, relu, res, 1 if residual else 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d(x, wt, b, stride, pad, relu, res, 1 if residual else 0)
    e1.record()
    torch.cuda.synchronize()
    lib.osr_debug_set_conv_stamps(C.c_void_p(stamps.data_ptr()))
    ops.conv2d(x, wt, b, stride, pad, relu, res, 1 if residual else 0)
    torch.cuda.synchronize()
    lib.osr_debug_set_conv_stamps(None)
    s = stamps.view(-1, 4).cpu()
    s = s[s[:, 3] > 0].double()
    t = s * 0.01  # 100 MHz -> us
    span = float(t[:, 3].max() - t[:, 0].min())
    ph = [(t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2]), (t[:, 3] - t[:, 0])]
    start = t[:, 0] - t[:, 0].min()
    print(f"{name}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch (un-stamped), {len(s)} workgroups, stamped span {span:.1f} us")
    for nm, p in zip(("entry->first slice landed", "K loop", "epilogue", "whole workgroup"), ph):
        print(f"    {nm:28s} mean {float(p.mean()):6.2f} us   p10 {float(p.quantile(0.1)):6.2f}   p50 {float(p.quantile(0.5)):6.2f}   p90 {float(p.quantile(0.9)):6.2f}")
    print(f"    workgroup start times: p10 {float(start.quantile(0.1)):.1f}  p50 {float(start.quantile(0.5)):.1f}  p90 {float(start.quantile(0.9)):.1f} us;"
          f" concurrency = sum(wg time)/span = {float(ph[3].sum()) / span:.1f} workgroups")


case("res4.conv3  1x1 256->1024 +res  M=67200 ", 16, 50, 84, 256, 1024, 1, residual=True)
case("res4.conv1  1x1 1024->256       M=67200 ", 16, 50, 84, 1024, 256, 1)
case("res4.conv2  3x3 256->256        M=67200 ", 16, 50, 84, 256, 256, 3, pad=1)
case("res3.conv3  1x1 128->512 +res   M=268800", 16, 100, 168, 128, 512, 1, residual=True)
case("res2.conv3  1x1 64->256 +res    M=1075200", 16, 200, 336, 64, 256, 1, residual=True)
case("res2.conv1  1x1 256->64         M=1075200", 16, 200, 336, 256, 64, 1)
case("fpn_out3    3x3 256->256        M=268800", 16, 100, 168, 256, 256, 3, pad=1, relu=False)
