"""RoIAlign backward at the training step's shape (16 images x 512 sampled RoIs, R50-FPN pyramid of 800 x 1344): the scatter kernel
(fp32 atomics into a zeroed pyramid) against the pixel-centric gather (osr_roi_align_bwd_dense)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg._lib.load()
ops = pkg.ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
n, S, c = 16, 512, 256
shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
scales = (0.25, 0.125, 0.0625, 0.03125)
ctr = torch.rand(n * S, 2, generator=g) * torch.tensor([1333.0, 800.0])
size = torch.exp(torch.rand(n * S, 2, generator=g) * 3.2 + 2.8)  # 16 .. 400 px
boxes = torch.cat((ctr - size / 2, ctr + size / 2), dim=1).clamp_(min=0)
boxes[:, 2].clamp_(max=1333); boxes[:, 3].clamp_(max=800)
if len(sys.argv) > 1 and sys.argv[1] == "clustered":
    # what a training step's sampler produces: a quarter of each image's 512 RoIs are positives, jittered copies of its 8 ground-truth boxes
    # (IoU >= 0.5), so the tiles under a ground-truth box are reached by dozens of RoIs each
    for i in range(n):
        gtc = torch.rand(8, 2, generator=g) * torch.tensor([1100.0, 650.0]) + 100
        gts = torch.exp(torch.rand(8, 2, generator=g) * 2.0 + 3.6)  # 36 .. 270 px
        which = torch.randint(0, 8, (128,), generator=g)
        pc = gtc[which] + (torch.rand(128, 2, generator=g) - 0.5) * 0.25 * gts[which]
        s2 = gts[which] * (0.8 + 0.4 * torch.rand(128, 2, generator=g))
        boxes[i * S:i * S + 128] = torch.cat((pc - s2 / 2, pc + s2 / 2), dim=1).clamp_(min=0)
    boxes[:, 2].clamp_(max=1333); boxes[:, 3].clamp_(max=800)
bidx = torch.arange(n, dtype=torch.int32).repeat_interleave(S)
dout = torch.randn(n * S, 7, 7, c, generator=g).half().to(dev)
boxes, bidx = boxes.to(dev), bidx.to(dev)


def timeit(f, it=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


a = ops.roi_align_bwd(dout, shapes, n, scales, boxes, bidx)
b = ops.roi_align_bwd(dout, shapes, n, scales, boxes, bidx, rois_per_image=S)
for l in range(4):
    d = (a[l] - b[l]).abs().max().item() / max(a[l].abs().max().item(), 1e-9)
    print(f"level {l}: rel diff {d:.2e}")
print(f"scatter (zero fill + atomics): {timeit(lambda: ops.roi_align_bwd(dout, shapes, n, scales, boxes, bidx)):.3f} ms")
print(f"gather  (dense, no fill):      {timeit(lambda: ops.roi_align_bwd(dout, shapes, n, scales, boxes, bidx, rois_per_image=S)):.3f} ms")
