"""Experiment driver (not part of the product): time osr_roi_align_fwd alone on the bench's real proposals."""
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
b = sel["boxes"].view(-1, 4)
wh = (b[:, 2:] - b[:, :2])
print("boxes w/h mean", wh.mean(0).tolist(), "max", wh.max(0)[0].tolist(), "counts", sel["counts"].tolist()[:4])
fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
def run():
    return ops.roi_align(fl, (0.25, 0.125, 0.0625, 0.03125), b, sel["batch_idx"], 7, torch.float16)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("OSR_ROI_DEBUG=%s roi_align %.3f ms" % (os.environ.get("OSR_ROI_DEBUG", "0"), e0.elapsed_time(e1) / 10))
# locality experiment: all RoIs read image 0's pyramid / a tiny region
bi0 = torch.where(sel["batch_idx"] >= 0, torch.zeros_like(sel["batch_idx"]), sel["batch_idx"])
def run0():
    return ops.roi_align(fl, (0.25, 0.125, 0.0625, 0.03125), b, bi0, 7, torch.float16)
for _ in range(3): run0()
torch.cuda.synchronize(); e0.record()
for _ in range(10): run0()
e1.record(); torch.cuda.synchronize()
print("all RoIs on image 0: %.3f ms" % (e0.elapsed_time(e1) / 10))
bs = (b * 0.1).contiguous()  # all boxes squeezed into a 133x80 px corner: everything L2 resident
def run1():
    return ops.roi_align(fl, (0.25, 0.125, 0.0625, 0.03125), bs, bi0, 7, torch.float16)
for _ in range(3): run1()
torch.cuda.synchronize(); e0.record()
for _ in range(10): run1()
e1.record(); torch.cuda.synchronize()
print("tiny boxes (1/10 size) on image 0: %.3f ms" % (e0.elapsed_time(e1) / 10))
# spatial-locality experiment: same RoIs, processed in (image, level, y-tile, x) order
area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
lvl = torch.floor(4 + torch.log2(torch.sqrt(area.clamp(min=1e-6)) / 224 + 1e-8)).clamp(2, 5)
cy = (b[:, 1] + b[:, 3]) * 0.5
cx = (b[:, 0] + b[:, 2]) * 0.5
bi = sel["batch_idx"].clone()
key = (bi.double().clamp(min=0) * 4 + (lvl.double() - 2)) * 1e6 + torch.floor(cy.double() / 64) * 1e3 + torch.floor(cx.double() / 64)
key = torch.where(bi >= 0, key, torch.full_like(key, 1e12))
perm = torch.argsort(key)
bp, bip = b[perm].contiguous(), bi[perm].contiguous()
def run2():
    return ops.roi_align(fl, (0.25, 0.125, 0.0625, 0.03125), bp, bip, 7, torch.float16)
for _ in range(3): run2()
torch.cuda.synchronize(); e0.record()
for _ in range(10): run2()
e1.record(); torch.cuda.synchronize()
print("spatially sorted RoIs: %.3f ms" % (e0.elapsed_time(e1) / 10))
# scaling experiment: the first k images' RoIs only (tail / launch effects show up as non-proportional times)
cap = sel["boxes"].shape[1]
for k in (1, 2, 4, 8, 16):
    bk, bik = b[: k * cap].contiguous(), sel["batch_idx"].view(-1)[: k * cap].contiguous()
    def runk():
        return ops.roi_align(fl, (0.25, 0.125, 0.0625, 0.03125), bk, bik, 7, torch.float16)
    for _ in range(3): runk()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): runk()
    e1.record(); torch.cuda.synchronize()
    print("first %d image(s): %.3f ms" % (k, e0.elapsed_time(e1) / 10))
# per-level cost: RoIs of one level only (others marked as padding)
for L in (2, 3, 4, 5):
    bil = torch.where(lvl == L, sel["batch_idx"].view(-1), torch.full_like(sel["batch_idx"].view(-1), -1))
    def runl():
        return ops.roi_align(fl, (0.25, 0.125, 0.0625, 0.03125), b, bil, 7, torch.float16)
    for _ in range(3): runl()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): runl()
    e1.record(); torch.cuda.synchronize()
    print("level %d only (%d RoIs): %.3f ms" % (L, int((bil >= 0).sum()), e0.elapsed_time(e1) / 10))
# footprint statistics: column steps of the current streaming direction vs the transposed one
sc = torch.tensor([0.25, 0.125, 0.0625, 0.03125], device=b.device)[(lvl - 2).long()]
valid = sel["batch_idx"].view(-1) >= 0
fw = ((b[:, 2] - b[:, 0]) * sc + 2)[valid]
fh = ((b[:, 3] - b[:, 1]) * sc + 2)[valid]
print("footprint cols mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | rows mean %.1f p90 %.1f p99 %.1f max %.1f" % (
    fw.mean(), fw.median(), fw.quantile(0.9), fw.quantile(0.99), fw.max(), fh.mean(), fh.quantile(0.9), fh.quantile(0.99), fh.max()))
steps_now = 7 * fw
steps_best = 7 * torch.minimum(fw, fh)
loads = 7 * fw * (fh / 7 + 2)
print("sum column steps now %.3g, with the shorter axis streamed %.3g (%.2fx); sum loads %.3g" % (steps_now.sum(), steps_best.sum(), steps_now.sum() / steps_best.sum(), loads.sum()))
