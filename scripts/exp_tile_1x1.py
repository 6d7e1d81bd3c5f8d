"""Experiment driver (needs a -DOSR_EXPERIMENT build; run once per OSR_CONV_FORCE_TILE value): the HBM-bound 1x1 layers of res3-res5 / FPN under every
conv tile shape, without a residual so that all shapes are admissible. Round 5 result (us; 0 = cost model, 1 = 128x128/1, 2 = 128x128/2, 3 = 256x256
8-phase, 4 = 128x256, 5 = 256x128, 7 = 128x64/2): res4.conv3 68 / 72 / 82 / 85 / 75 / 78 / 92, res4.conv1 53 / 54 / 65 / 67 / 67 / 67 / 61, lateral2 350 / 352 / 385 / 355 / 361 /
346 / 443: the shapes that move fewer bytes from L2 per FLOP are not faster -- these layers are held by the per-tile latency chain of a 4-slice K loop,
not by L2 -> CU bandwidth (which an A-stationary kernel would have relieved)."""
import os, sys, math, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
f = os.environ.get("OSR_CONV_FORCE_TILE", "0")
g = torch.Generator().manual_seed(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
for name, (n, h, w, cin, cout) in {"res4.conv3 256->1024": (16, 50, 84, 256, 1024), "res4.conv1 1024->256": (16, 50, 84, 1024, 256), "res5.conv3 512->2048": (16, 25, 42, 512, 2048),
                                   "lateral2 256->256": (16, 200, 336, 256, 256), "res3.conv1 512->128": (16, 100, 168, 512, 128)}.items():
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().cuda()
    wt = (torch.randn(cout, 1, 1, cin, generator=g) / math.sqrt(cin)).half().cuda()
    b = torch.randn(cout, generator=g).cuda()
    out.append(f"{name}: {t(lambda: ops.conv2d(x, wt, b, relu=True)):.1f}")
print("tile", f, " | ".join(out), flush=True)
