"""Experiment driver (not part of the product): the split-K tail on the 3 x 3 layers of res4 / FPN p4 (1050 tiles of 128 x 128 on 768
slots), on against off, HIP-event time per launch."""
import os, sys, math, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, (n, h, w, cin, cout, k) in {"res4.conv2": (16, 50, 84, 256, 256, 3), "res4.conv1": (16, 50, 84, 1024, 256, 1), "res5.conv2": (16, 25, 42, 512, 512, 3),
                                      "fpn_out3": (16, 100, 168, 256, 256, 3), "res3.conv1": (16, 100, 168, 512, 128, 1)}.items():
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().cuda()
    wt = (torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin)).half().cuda()
    b = torch.randn(cout, generator=g).cuda()
    res = {}
    for flag in (False, True):
        ops.SPLIT_K_TAIL = flag
        res[flag] = t(lambda: ops.conv2d(x, wt, b, 1, k // 2, relu=True))
    ops.SPLIT_K_TAIL = True
    L = pkg._lib
    p = L.ConvParams()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, h, w, cin, h, w, cout
    p.kh = p.kw = k; p.stride_h = p.stride_w = 1; p.pad_h = p.pad_w = k // 2
    p.in_stride_n, p.in_stride_h, p.in_stride_w = h * w * cin, w * cin, cin
    p.out_stride_n, p.out_stride_h, p.out_stride_w = h * w * cout, w * cout, cout
    p.in_dtype = p.out_dtype = L.OSR_F16
    buf = ctypes.create_string_buffer(256)
    L.load().osr_conv2d_fwd_describe(ctypes.byref(p), 1, buf, 256)
    print(f"{name:12s} single {res[False]:7.1f} us   with split tail {res[True]:7.1f} us   plan: {buf.value.decode()}", flush=True)
