"""Experiment driver (not part of the product): does the processing ORDER of the RoIs matter for osr_roi_align_fwd?
Same RoIs, permuted on the host: as selected (level-major, score order), sorted by (image, level, row band, x), random."""
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
b = sel["boxes"].view(-1, 4).contiguous()
bi = sel["batch_idx"].view(-1).contiguous()
fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
SC = (0.25, 0.125, 0.0625, 0.03125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(boxes, bidx, tag, reps=10):
    f = lambda: ops.roi_align(fl, SC, boxes, bidx, 7, torch.float16)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    print("%-70s %.3f ms" % (tag, e0.elapsed_time(e1) / reps))


area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
lvl = torch.floor(4 + torch.log2(torch.sqrt(area.clamp(min=1e-6)) / 224 + 1e-8)).clamp(2, 5)
sc = torch.tensor(SC, device=b.device)[(lvl - 2).long()]
cy, cx = (b[:, 1] + b[:, 3]) * 0.5 * sc, (b[:, 0] + b[:, 2]) * 0.5 * sc
pad = bi < 0
t(b, bi, "as selected (level-major, score order inside a level)")
for band in (4, 8, 16, 32):
    key = ((bi.double().clamp(min=0) * 4 + (lvl.double() - 2)) * 1e4 + torch.floor(cy.double() / band)) * 1e4 + cx.double()
    key = torch.where(pad, torch.full_like(key, 1e18), key)
    perm = torch.argsort(key)
    t(b[perm].contiguous(), bi[perm].contiguous(), "sorted by (image, level, %d-row band, x)" % band)
key = ((lvl.double() - 2) * 16 + bi.double().clamp(min=0)) * 1e8 + torch.floor(cy.double() / 8) * 1e4 + cx.double()
key = torch.where(pad, torch.full_like(key, 1e18), key)
perm = torch.argsort(key)
t(b[perm].contiguous(), bi[perm].contiguous(), "sorted by (level, image, 8-row band, x)")
perm = torch.randperm(b.shape[0], device=b.device)
t(b[perm].contiguous(), bi[perm].contiguous(), "random order")
bi0 = torch.where(bi >= 0, torch.zeros_like(bi), bi)
t(b, bi0, "all RoIs on image 0 (pyramid of one image: 46 MB)")
# region-local but not line-local: RoIs of one (image, level, row band) together, random order inside the band
for band in (8, 16, 32, 64):
    rnd = torch.rand(b.shape[0], device=b.device, dtype=torch.float64)
    key = ((bi.double().clamp(min=0) * 4 + (lvl.double() - 2)) * 1e4 + torch.floor(cy.double() / band)) + rnd * 0.5
    key = torch.where(pad, torch.full_like(key, 1e18), key)
    perm = torch.argsort(key)
    t(b[perm].contiguous(), bi[perm].contiguous(), "(image, level, %d-row band), random inside the band" % band)
rnd = torch.rand(b.shape[0], device=b.device, dtype=torch.float64)
key = torch.where(pad, torch.full_like(rnd, 1e18), bi.double().clamp(min=0) * 4 + (lvl.double() - 2) + rnd * 0.5)
perm = torch.argsort(key)
t(b[perm].contiguous(), bi[perm].contiguous(), "(image, level), random inside")
key = torch.where(pad, torch.full_like(rnd, 1e18), bi.double().clamp(min=0) + rnd * 0.5)
perm = torch.argsort(key)
t(b[perm].contiguous(), bi[perm].contiguous(), "(image), random inside")
