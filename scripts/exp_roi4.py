"""Experiment driver (not part of the product): RoI processing order INSIDE each image (padding rows stay at every image's
tail, so the per-XCD share of real work is unchanged): as selected, sorted by (level, row band[, x]), random."""
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
n, cap = 16, sel["cap"]
b = sel["boxes"].view(n, cap, 4)
bi = sel["batch_idx"].view(n, cap)
fl = [feats[k] for k in ("p2", "p3", "p4", "p5")]
SC = (0.25, 0.125, 0.0625, 0.03125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(key, tag, reps=10):
    if key is None:
        bb, ii = b.reshape(-1, 4).contiguous(), bi.reshape(-1).contiguous()
    else:
        key = torch.where(bi < 0, torch.full_like(key, 1e18), key)  # padding rows stay last inside their image
        perm = torch.argsort(key, dim=1)
        bb = torch.gather(b, 1, perm[:, :, None].expand(-1, -1, 4)).reshape(-1, 4).contiguous()
        ii = torch.gather(bi, 1, perm).reshape(-1).contiguous()
    f = lambda: ops.roi_align(fl, SC, bb, ii, 7, torch.float16)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    print("%-64s %.3f ms" % (tag, e0.elapsed_time(e1) / reps))


area = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
lvl = torch.floor(4 + torch.log2(torch.sqrt(area.clamp(min=1e-6)) / 224 + 1e-8)).clamp(2, 5).double()
sc = torch.tensor(SC, device=b.device, dtype=torch.float64)[(lvl - 2).long()]
cy, cx = ((b[..., 1] + b[..., 3]) * 0.5).double() * sc, ((b[..., 0] + b[..., 2]) * 0.5).double() * sc
rnd = torch.rand(n, cap, device=b.device, dtype=torch.float64)
t(None, "as selected")
t(rnd, "random inside the image")
t(lvl + rnd * 0.5, "by level, random inside")
for band in (4, 8, 16, 32):
    t(lvl * 1e6 + torch.floor(cy / band) * 1e3 + rnd * 0.5, "by (level, %d-row band), random inside" % band)
    t(lvl * 1e6 + torch.floor(cy / band) * 1e3 + cx, "by (level, %d-row band, x)" % band)
t(lvl * 1e6 + cy, "by (level, y)")
