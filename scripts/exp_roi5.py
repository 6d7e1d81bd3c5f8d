"""Experiment driver (not part of the product): forward RoIAlign on the bench's proposals, list order against the locality order
(osr_roi_locality_order), timed kernel-only with HIP events; checks that the two outputs are bit-identical and prints the
histogram of RoIs per level. Build variants (-DRA_WPR=1|7 ...) are compared by scripts/ab_roi.sh on one box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine
from openset_rcnn_amd.host.weights import random_params
from openset_rcnn_amd.host import ops
eng = OpensetRCNNEngine(random_params(0), device="cuda:0")
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (16, 3, 800, 1333), generator=g, dtype=torch.uint8).cuda()
hw = torch.tensor([(800, 1333)] * 16, dtype=torch.int32, device="cuda")
keep = {}
eng.forward_device(images, hw, 800, 1344, keep)
feats, sel = keep["feats"], keep["sel"]
bb, ii = sel["boxes"].view(-1, 4).contiguous(), sel["batch_idx"].view(-1).contiguous()
fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
SC = (0.25, 0.125, 0.0625, 0.03125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, tag, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-56s %.3f ms" % (tag, e0.elapsed_time(e1) / reps), flush=True)


ident = torch.arange(bb.shape[0], dtype=torch.int32, device="cuda")
order = ops.roi_locality_order(fl, SC, bb, ii)
assert torch.equal(torch.sort(order[:-1].long()).values, ident.long()), "not a permutation"
a = ops.roi_align(fl, SC, bb, ii, 7, torch.float16, order=ident)
b = ops.roi_align(fl, SC, bb, ii, 7, torch.float16, order=order)
assert torch.equal(a, b), "order changed the result"
t(lambda: ops.roi_align(fl, SC, bb, ii, 7, torch.float16, order=ident), "list (score) order")
t(lambda: ops.roi_align(fl, SC, bb, ii, 7, torch.float16, order=order), "locality order (precomputed)")
t(lambda: ops.roi_locality_order(fl, SC, bb, ii), "osr_roi_locality_order alone")
t(lambda: ops.roi_align(fl, SC, bb, ii, 7, torch.float16), "order + pool (the product path)")
area = (bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1])
lvl = torch.floor(4 + torch.log2(torch.sqrt(area.clamp(min=1e-6)) / 224 + 1e-8)).clamp(2, 5)
print("rois", bb.shape[0], "valid", int((ii >= 0).sum()), "per level", [int(((lvl == l) & (ii >= 0)).sum()) for l in (2, 3, 4, 5)])
