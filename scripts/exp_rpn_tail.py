"""Experiment driver: the merged CF-RPN head launch (p2..p6 at batch 16) and the merged FPN output launch under a VARIANT library
(OSR_VARIANT_LIB); prints times and, with CHECK=<file>, the largest difference of the head's outputs against the first library run."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("OSR_VARIANT_LIB"):
    pkg._lib.LIB_PATH = os.environ["OSR_VARIANT_LIB"]
pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
n = 16
xs = [(torch.randn(n, h, w, 256, generator=g) * 0.5).half().cuda() for h, w in shapes]
wt = (torch.randn(256, 3, 3, 256, generator=g) / 48.0).half().cuda()
b = (torch.randn(256, generator=g) * 0.1).cuda()
wtail = (torch.randn(5, 256, generator=g) * 0.05).cuda(); btail = (torch.randn(5, generator=g) * 0.1).cuda()
rows = [n * h * w for h, w in shapes]; offs = [sum(rows[:i]) for i in range(5)]
deltas = torch.empty(sum(rows), 4, device="cuda"); ctr = torch.empty(sum(rows), device="cuda")
dl = [deltas[o:o + r] for o, r in zip(offs, rows)]; cl = [ctr[o:o + r] for o, r in zip(offs, rows)]
head = lambda: ops.cfrpn_head_fused_levels(xs, wt, b, wtail, btail, dl, cl)
ws = [wt] * 4; bs = [b] * 4
fpn = lambda: ops.conv2d_levels(xs[:4], ws, bs)
def t(fn, reps=10):
    best = 1e9
    for _ in range(4):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
print(f"{os.path.basename(os.environ.get('OSR_VARIANT_LIB', 'product')):28s} head {t(head):8.1f} us   fpn outputs {t(fpn):8.1f} us", flush=True)
chk = os.environ.get("CHECK")
if chk:
    head(); torch.cuda.synchronize()
    cur = (deltas.cpu().clone(), ctr.cpu().clone())
    if os.path.exists(chk):
        ref = torch.load(chk)
        print("   max |d deltas| %.3g (max |deltas| %.3g)   max |d ctr| %.3g" % ((cur[0] - ref[0]).abs().max(), ref[0].abs().max(), (cur[1] - ref[1]).abs().max()), flush=True)
    else:
        torch.save(cur, chk)
