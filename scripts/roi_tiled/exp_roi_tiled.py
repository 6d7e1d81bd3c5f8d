"""Experiment driver (not part of the product): RoIAlign on the bench's proposals, wave-per-RoI kernel (locality order) against the
tile-centric path (ops.roi_align_tiled), kernel time with HIP events; prints how many RoIs each path takes."""
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
pl = [feats[k + "_planes"] for k in ("p2", "p3", "p4", "p5")]
for f, p in zip(fl, pl):
    assert torch.equal(ops.to_planes(f), p), "planar copy differs"
SC = (0.25, 0.125, 0.0625, 0.03125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, tag, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-56s %.3f ms" % (tag, e0.elapsed_time(e1) / reps), flush=True)


a = ops.roi_align(fl, SC, bb, ii, 7, torch.float16)
b, rid = ops.roi_align_tiled(fl, pl, SC, bb, ii, 7, torch.float16, return_rid=True)
bm = ops.slice_major_to_bin_major(b, 256)
d = (a.float() - bm.float()).abs()
print("rows", bb.shape[0], "tiled", int((rid >= 0).sum()), "left", int((rid == -1).sum()), "padding", int((rid == -2).sum()),
      "max |diff| vs wave-per-RoI", float(d.max()), "of max", float(a.float().abs().max()))
t(lambda: ops.roi_align(fl, SC, bb, ii, 7, torch.float16), "wave per RoI (order + pool: the round-3 path)")
t(lambda: ops.roi_align_tiled(fl, pl, SC, bb, ii, 7, torch.float16), "tiled (plan + descriptors + main + leftovers)")
