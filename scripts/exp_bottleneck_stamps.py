"""Diagnostic (not part of the product): per-workgroup phase times of bottleneck64_kernel from s_memrealtime stamps.
Needs a library built with OSR_EXTRA_HIPCC_FLAGS=-DBN_STAMPS (python openset-rcnn_amd/build.py --force)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine
from openset_rcnn_amd.host.weights import random_params
lib.osr_debug_set_bn_stamps.argtypes = [C.c_void_p]; lib.osr_debug_set_bn_stamps.restype = None
DEV = "cuda:0"
eng = OpensetRCNNEngine(random_params(0), None, torch.float16, DEV)
g = torch.Generator().manual_seed(3)
for name, cin, pre, first in (("block 1 (cin 256, identity)", 256, "backbone.bottom_up.res2.1", False), ("block 0 (cin 64, projection)", 64, "backbone.bottom_up.res2.0", True)):
    x = torch.randn(16, 200, 336, cin, generator=g).clamp_(min=0).half().to(DEV)
    for _ in range(3):
        eng._bottleneck(x, pre, first, 1)
    stamps = torch.zeros(16 * 25 * 21 * 8, dtype=torch.int64, device=DEV)
    lib.osr_debug_set_bn_stamps(C.c_void_p(stamps.data_ptr()))
    eng._bottleneck(x, pre, first, 1)
    torch.cuda.synchronize()
    lib.osr_debug_set_bn_stamps(None)
    t = stamps.view(-1, 8).cpu().double() * 0.01
    span = float(t[:, 5].max() - t[:, 0].min())
    print(f"{name}: {len(t)} workgroups, stamped span {span:.1f} us")
    names = ("entry -> conv1 may start (w1 and the pixels landed)", "conv1 -> mid1", "conv2 (9 taps) -> mid2", "conv3 + epilogue issue", "stores drained", "whole workgroup")
    ph = [t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 5] - t[:, 0]]
    for nm, p in zip(names, ph):
        print(f"    {nm:58s} mean {float(p.mean()):6.2f} us   p10 {float(p.quantile(0.1)):6.2f}   p50 {float(p.quantile(0.5)):6.2f}   p90 {float(p.quantile(0.9)):6.2f}")
    print(f"    concurrency = sum(workgroup time) / span = {float(ph[5].sum()) / span:.1f} workgroups (512 slots)")
