"""Experiment driver (round 6): per-layer time of osr_conv2d_wgrad on the training step's layer shapes (batch 16 at 800 x 1344), for the
A/B of the XCD-aware work order (build flag -DWG_XCD_REMAP=0|1; scripts/ab_script.sh). Prints us, TFLOP/s and a checksum of dw (the two
orders must give the same bits: every partial sum lands in the same slot)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)
def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cases = {"res3.conv2 3x3 128->128 @16x100x168": (16, 100, 168, 128, 128, 3, 1), "res3.conv1 1x1 512->128": (16, 100, 168, 512, 128, 1, 1),
         "res3.conv3 1x1 128->512": (16, 100, 168, 128, 512, 1, 1), "res3.0.shortcut 1x1 s2 256->512 @16x200x336": (16, 200, 336, 256, 512, 1, 2),
         "res4.conv2 3x3 256->256 @16x50x84": (16, 50, 84, 256, 256, 3, 1), "res4.conv1 1x1 1024->256": (16, 50, 84, 1024, 256, 1, 1),
         "res4.conv3 1x1 256->1024": (16, 50, 84, 256, 1024, 1, 1), "res5.conv2 3x3 512->512 @16x25x42": (16, 25, 42, 512, 512, 3, 1),
         "res5.conv1 1x1 2048->512": (16, 25, 42, 2048, 512, 1, 1), "res5.conv3 1x1 512->2048": (16, 25, 42, 512, 2048, 1, 1),
         "fpn_output2 3x3 256->256 @16x200x336": (16, 200, 336, 256, 256, 3, 1), "fpn_output3 @16x100x168": (16, 100, 168, 256, 256, 3, 1),
         "fpn_lateral2 1x1 256->256 @16x200x336": (16, 200, 336, 256, 256, 1, 1), "fpn_lateral4 1x1 1024->256 @16x50x84": (16, 50, 84, 1024, 256, 1, 1),
         "fc1 8192x12544->1024": (1, 8192, 1, 12544, 1024, 1, 1), "fc2 8192x1024->1024": (1, 8192, 1, 1024, 1024, 1, 1)}
tot = 0.0
for name, (n, h, w, cin, cout, k, st) in cases.items():
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().cuda()
    ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    dy = (torch.randn(n, ho, wo, cout, generator=g) * 0.1).half().cuda()
    fn = lambda: ops.conv2d_wgrad(x, dy, k, k, st, k // 2)
    fl = 2.0 * n * ho * wo * cout * k * k * cin
    dw = fn(); torch.cuda.synchronize()
    ts = [timed(fn) for _ in range(5)]
    t = statistics.median(ts); tot += t
    print(f"{name:46s} {t:9.1f} us {fl / t / 1e6:7.0f} TFLOP/s   checksum {float(dw.double().sum()):+.9e} {float(dw.double().abs().max()):.6e}", flush=True)
    del x, dy, dw
print(f"sum {tot:.1f} us")
