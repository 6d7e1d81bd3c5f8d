"""Experiment driver (needs a -DOSR_EXPERIMENT build): the 8-phase loop of conv_wgrad_kernel<256x256> (OSR_WGRAD_PH8=1) against the
vmcnt(0)-per-step loop (=0), same process: bit-identity of dw, a repeat screen, interleaved timing rounds on random operands."""
import os, sys, math, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)
os.environ["OSR_WGRAD_PH8_MINSTEPS"] = "0"  # the 8-phase loop whatever the split length
ROUNDS = int(os.environ.get("ROUNDS", 7)); REPS = int(os.environ.get("REPS", 10)); SCREEN = int(os.environ.get("SCREEN", 20))
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cases = {"fpn_output2 3x3 256->256 @16x200x336": (16, 200, 336, 256, 256, 3, 1), "fpn_output3 @16x100x168": (16, 100, 168, 256, 256, 3, 1),
         "res4.conv2 3x3 256->256 @16x50x84": (16, 50, 84, 256, 256, 3, 1), "res5.conv2 3x3 512->512 @16x25x42": (16, 25, 42, 512, 512, 3, 1),
         "res4.conv3 1x1 256->1024 @16x50x84": (16, 50, 84, 256, 1024, 1, 1), "res5.conv1 1x1 2048->512 @16x25x42": (16, 25, 42, 2048, 512, 1, 1),
         "res4.0.conv1 1x1 s2 512->256 @16x100x168": (16, 100, 168, 512, 256, 1, 2), "fc1 8192x12544->1024": (1, 8192, 1, 12544, 1024, 1, 1),
         "fc2 8192x1024->1024": (1, 8192, 1, 1024, 1024, 1, 1), "short: 3x3 256->256 @1x13x21": (1, 13, 21, 256, 256, 3, 1)}
print(f"{'case':44s} {'identical':>9s} {'screen':>7s} {'old us med/min':>18s} {'8-phase us med/min':>20s} {'TF/s old -> new':>16s}", flush=True)
for name, (n, h, w, cin, cout, k, st) in cases.items():
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().cuda()
    ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    dy = (torch.randn(n, ho, wo, cout, generator=g) * 0.1).half().cuda()
    fn = lambda: ops.conv2d_wgrad(x, dy, k, k, st, k // 2)
    fl = 2.0 * n * ho * wo * cout * k * k * cin
    os.environ["OSR_WGRAD_PH8"] = "0"; ref = fn().clone()
    os.environ["OSR_WGRAD_PH8"] = "1"; new = fn().clone(); torch.cuda.synchronize()
    same = torch.equal(ref, new)
    bad = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(SCREEN))
    t0, t1 = [], []
    for _ in range(ROUNDS):
        os.environ["OSR_WGRAD_PH8"] = "0"; fn(); t0.append(timed(fn, REPS))
        os.environ["OSR_WGRAD_PH8"] = "1"; fn(); t1.append(timed(fn, REPS))
    print(f"{name:44s} {str(same):>9s} {bad:3d}/{SCREEN:<3d} {statistics.median(t0):9.1f}/{min(t0):8.1f} {statistics.median(t1):11.1f}/{min(t1):8.1f} "
          f"{fl / statistics.median(t0) / 1e6:7.0f} -> {fl / statistics.median(t1) / 1e6:5.0f}", flush=True)
