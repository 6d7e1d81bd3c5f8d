"""Experiment driver (not part of the product): dump the bench's proposal boxes (16 images, seed 1234, calibrated weights as bench.py builds them)
to gpurun_out/rois.pt so that RoIAlign work decompositions can be simulated on the CPU."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine
from openset_rcnn_amd.host.weights import random_params
eng = OpensetRCNNEngine(random_params(0), device="cuda:0")
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (16, 3, 800, 1333), generator=g, dtype=torch.uint8).cuda()
hw = torch.tensor([(800, 1333)] * 16, dtype=torch.int32, device="cuda")
keep = {}
eng.forward_device(images, hw, 800, 1344, keep)
sel = keep["sel"]
os.makedirs("gpurun_out", exist_ok=True)
torch.save(dict(boxes=sel["boxes"].cpu(), batch_idx=sel["batch_idx"].cpu(), counts=sel["counts"].cpu(), scores=sel["scores"].cpu()), "gpurun_out/rois.pt")
print("saved", sel["boxes"].shape, int(sel["counts"].sum()))
