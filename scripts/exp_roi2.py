"""Experiment driver (not part of the product): cost decomposition of osr_roi_align_fwd on the bench's real proposals --
stores only (all rows padding), setup + stores (boxes shrunk to ~1 feature pixel), everything; plus footprint statistics."""
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


def t(boxes, bidx, tag, out_dtype=torch.float16, reps=10):
    f = lambda: ops.roi_align(fl, SC, boxes, bidx, 7, out_dtype)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-58s %.3f ms" % (tag, ms))
    return ms


valid = bi >= 0
print("rows %d valid %d" % (bi.numel(), int(valid.sum())))
t(b, bi, "all (f16 out)")
t(b, bi, "all (f32 out)", torch.float32)
t(b, torch.full_like(bi, -1), "stores only: every row padding (zeros)")
ctr = (b[:, :2] + b[:, 2:]) * 0.5
tiny = torch.cat((ctr - 2.0, ctr + 2.0), dim=1).contiguous()  # 4x4 px boxes: level p2, ~1 feature pixel + bilinear neighbours
t(tiny, bi, "setup + stores: 4x4 px boxes at the same centres")
for s in (0.25, 0.5, 0.75):
    wh = (b[:, 2:] - b[:, :2]) * s * 0.5
    t(torch.cat((ctr - wh, ctr + wh), dim=1).contiguous(), bi, "boxes scaled by %.2f about their centres" % s)
# footprint statistics
area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
lvl = torch.floor(4 + torch.log2(torch.sqrt(area.clamp(min=1e-6)) / 224 + 1e-8)).clamp(2, 5)
sc = torch.tensor(SC, device=b.device)[(lvl - 2).long()]
fw = ((b[:, 2] - b[:, 0]) * sc + 2)[valid]
fh = ((b[:, 3] - b[:, 1]) * sc + 2)[valid]
print("footprint cols mean %.1f p50 %.1f p90 %.1f max %.1f | rows mean %.1f p50 %.1f p90 %.1f max %.1f | px mean %.0f sum %.3g (x512 B = %.2f GB)" % (
    fw.mean(), fw.median(), fw.quantile(0.9), fw.max(), fh.mean(), fh.median(), fh.quantile(0.9), fh.max(), (fw * fh).mean(), (fw * fh).sum(), float((fw * fh).sum()) * 512 / 1e9))
for L in (2, 3, 4, 5):
    m = (lvl == L) & valid
    print("level %d: %d RoIs, footprint %.1f x %.1f" % (L, int(m.sum()), float(((b[:, 2] - b[:, 0]) * sc + 2)[m].mean()), float(((b[:, 3] - b[:, 1]) * sc + 2)[m].mean())))
