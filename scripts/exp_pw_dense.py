"""Experiment driver: the dense 1 x 1 layers of the pass (res3-res5 conv1 / conv3, FPN-free) under a VARIANT library (OSR_VARIANT_LIB), per-layer
time with HIP events; with CHECK=<file> the outputs are saved / compared bit for bit against the first library run."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("OSR_VARIANT_LIB"):
    pkg._lib.LIB_PATH = os.environ["OSR_VARIANT_LIB"]
pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)
def t(fn, reps=20):
    best = 1e9
    for _ in range(4):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
cases = [("res4.conv3 256->1024 +res", 16, 50, 84, 256, 1024, True, 1), ("res4.conv1 1024->256", 16, 50, 84, 1024, 256, False, 0),
         ("res3.conv3 128->512 +res", 16, 100, 168, 128, 512, True, 1), ("res3.conv1 512->128", 16, 100, 168, 512, 128, False, 0),
         ("res5.conv3 512->2048 +res", 16, 25, 42, 512, 2048, True, 1), ("res5.conv1 2048->512", 16, 25, 42, 2048, 512, False, 0),
         ("dgrad-like 256->1024 mask (res_mode 3)", 16, 50, 84, 256, 1024, True, 3), ("fc2 68368x1024->1024 f32", 1, 68368, 1, 1024, 1024, False, 0), ("fc1 68368x12544->1024", 1, 68368, 1, 12544, 1024, False, 0),
         ("lateral-like 256->256 @16x200x336", 16, 200, 336, 256, 256, False, 0)]
outs = {}
tot = 0.0
for name, n, h, w, cin, cout, res, mode in cases:
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().cuda()
    wt = (torch.randn(cout, 1, 1, cin, generator=g) / math.sqrt(cin)).half().cuda()
    b = torch.randn(cout, generator=g).cuda()
    r = (torch.randn(n, h, w, cout, generator=g)).half().cuda() if res else None
    od = torch.float32 if "f32" in name else None
    fn = lambda: ops.conv2d(x, wt, b, 1, 0, relu=(mode != 3), residual=r, res_mode=mode, out_dtype=od)
    us = t(fn); tot += us
    outs[name] = fn().cpu()
    print(f"{name:44s} {us:8.1f} us", flush=True)
print(f"{'sum':44s} {tot:8.1f} us   [{os.path.basename(os.environ.get('OSR_VARIANT_LIB', 'product'))}]", flush=True)
chk = os.environ.get("CHECK")
if chk:
    if os.path.exists(chk):
        ref = torch.load(chk)
        print("  identical to the first library:", all(torch.equal(outs[k], ref[k]) for k in outs), flush=True)
    else:
        torch.save(outs, chk)
