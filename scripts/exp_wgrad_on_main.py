"""Experiment driver (round 6): the config-3 train step (batch 16 at 800 x 1333) with the weight gradients of the LAST blocks of the backward
riding on the main stream (OpensetRCNNTrainer.wgrad_on_main). profiles/r06_b_train_step_timeline.txt shows the data-gradient chain ending
1.7 ms before the weight-gradient stream has drained its backlog; rounds 4-5 measured this knob at +-0 when the chain was the longer one.
Interleaved rounds, one process."""
import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
import bench
from openset_rcnn_amd.host.train import OpensetRCNNTrainer
from openset_rcnn_amd.host.weights import random_params
dev = "cuda:0"
n = 16
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (n, 3, 800, 1333), generator=g, dtype=torch.uint8).to(dev)
hw = torch.tensor([(800, 1333)] * n, dtype=torch.int32, device=dev)
tr = OpensetRCNNTrainer(random_params(0), dtype=torch.float16, device=dev, lr=1e-4, loss_scale=1024.0)
gt, gcls, gcnt = bench.synthetic_gt(n, 800, 1333)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
r = sum(a * b for a, b in shapes); cap = sum(min(2000, a * b) for a, b in shapes)
g2 = torch.Generator().manual_seed(0)
keys = {k: torch.rand(s, generator=g2).to(dev) for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + gt.shape[1])))}
args = (images, hw, 800, 1344, gt.to(dev), gcls.to(dev), gcnt.to(dev), keys)
P = "backbone.bottom_up."
sets = {"none (product)": set(), "res3.0": {P + "res3.0"}, "res3.0-1": {P + "res3.0", P + "res3.1"}, "res3.0-2": {P + f"res3.{i}" for i in range(3)},
        "all of res3": {P + f"res3.{i}" for i in range(4)}, "res3 + res4.0": {P + f"res3.{i}" for i in range(4)} | {P + "res4.0"}}
def run(k):
    for _ in range(2): tr.step(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): tr.step(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
res = {k: [] for k in sets}
for rnd in range(4):
    for name, s in sets.items():
        tr.wgrad_on_main = s
        res[name].append(run(6))
for name, v in res.items():
    print(f"{name:18s} median {statistics.median(v):7.3f} ms   " + " ".join(f"{x:7.3f}" for x in v), flush=True)
