"""Per-layer listing of the MFMA convolution family on the bench workload (single stream, events around every launch): time,
algorithmic TFLOP/s and GB/s, and the layer's own floor max(flops / 2.5 PFLOP/s, bytes / 6.3 TB/s achievable HBM) -- where the family's
time goes relative to what each layer could at best take. Experiment record."""
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
for _ in range(3):
    eng.forward_device(images, hw, 800, 1344)
torch.cuda.synchronize()
acc = {}
REPS = 5
for rep in range(REPS):
    eng.profile = []
    eng.forward_device(images, hw, 800, 1344)
    torch.cuda.synchronize()
    for i, (name, fl, e0, e1, nb, _nominal) in enumerate(eng.resolve_profile()[0]):
        k = (i, name)
        a = acc.setdefault(k, [fl, nb, 0.0])
        a[2] += e0.elapsed_time(e1) / REPS
eng.profile = None
rows = []
for (i, name), (fl, nb, ms) in sorted(acc.items()):
    floor = max(fl / 2.5e15, nb / 6.3e12) * 1e3
    rows.append((name, ms, fl / ms / 1e9, nb / ms / 1e6, floor, "mfma" if fl / 2.5e15 > nb / 6.3e12 else "hbm"))
tot = sum(r[1] for r in rows); totf = sum(r[4] for r in rows)
print(f"{'layer':46s} {'ms':>7s} {'TF/s':>7s} {'GB/s':>7s} {'floor ms':>8s} {'bound':>5s} {'excess ms':>9s}")
for r in sorted(rows, key=lambda r: -(r[1] - r[4])):
    print(f"{r[0][-46:]:46s} {r[1]:7.3f} {r[2]:7.1f} {r[3]:7.0f} {r[4]:8.3f} {r[5]:>5s} {r[1] - r[4]:9.3f}")
print(f"total {tot:.3f} ms, sum of floors {totf:.3f} ms; hbm-bound layers: {sum(r[1] for r in rows if r[5] == 'hbm'):.3f} ms (floors {sum(r[4] for r in rows if r[5] == 'hbm'):.3f}), "
      f"mfma-bound: {sum(r[1] for r in rows if r[5] == 'mfma'):.3f} ms (floors {sum(r[4] for r in rows if r[5] == 'mfma'):.3f})")
