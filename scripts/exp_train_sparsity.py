"""How much of the feature-gradient pyramid of the training step is non-zero? (bench's synthetic workload: 16 images, 8 GT boxes each)
Prints, per level, the fraction of pixels whose RoI-head gradient (osr_roi_align_bwd_dense output) has any non-zero channel, and the
fraction of 128-pixel row-major tiles that contain such a pixel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
import bench
from openset_rcnn_amd.host import ops
from openset_rcnn_amd.host.train import OpensetRCNNTrainer
from openset_rcnn_amd.host.weights import random_params

dev = torch.device("cuda:0")
params = random_params(0)
tr = OpensetRCNNTrainer(params, dtype=torch.float16, device=dev, lr=1e-4, loss_scale=1024.0)
n = 16
g = torch.Generator().manual_seed(0)
images = torch.randint(0, 256, (n, 3, 800, 1333), generator=g, dtype=torch.uint8).to(dev)
image_hw = torch.tensor([(800, 1333)] * n, dtype=torch.int32, device=dev)
gt, gcls, gcnt = bench.synthetic_gt(n, 800, 1333)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
r = sum(a * b for a, b in shapes)
cap = sum(min(2000, a * b) for a, b in shapes)
keys = {k: torch.rand(s, generator=g).to(dev) for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + gt.shape[1])))}
args = (images, image_hw, 800, 1344, gt.to(dev), gcls.to(dev), gcnt.to(dev), keys)
seen = {}
orig = ops.roi_align_bwd
def spy(*a, **k):
    out = orig(*a, **k)
    seen["d_feat"] = [o.clone() for o in out]
    return out
ops.roi_align_bwd = spy
import openset_rcnn_amd.host.train as T
T.ops.roi_align_bwd = spy
for _ in range(2):
    tr.step(*args)
torch.cuda.synchronize()
for l, d in enumerate(seen["d_feat"]):
    nz = (d != 0).any(dim=3)
    flat = nz.reshape(-1)
    pad = (-flat.numel()) % 128
    tiles = torch.nn.functional.pad(flat, (0, pad)).view(-1, 128).any(dim=1)
    rows3 = nz.clone()
    rows3[:, 1:] |= nz[:, :-1]; rows3[:, :-1] |= nz[:, 1:]
    rows3[:, :, 1:] |= rows3[:, :, :-1].clone(); rows3[:, :, :-1] |= rows3[:, :, 1:].clone()
    print(f"p{l + 2}: non-zero pixels {float(nz.float().mean()):.3f}, 128-pixel tiles with one {float(tiles.float().mean()):.3f}, pixels within one pixel of one {float(rows3.float().mean()):.3f}")
