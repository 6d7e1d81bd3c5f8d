"""Experiment driver (not part of the product): the fused raw-image stem of a VARIANT library (OSR_VARIANT_LIB = a path from
scripts/build_variant.sh: a previous revision of osr_stem_pool.hip, or one with sections stubbed out -- the round-5 ablations, -DSP_ABL=1..4, are in the history of that file) at the bench's size; prints one time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("OSR_VARIANT_LIB"):
    pkg._lib.LIB_PATH = os.environ["OSR_VARIANT_LIB"]
pkg._lib.load()
from openset_rcnn_amd.host import ops
from openset_rcnn_amd.host.weights import pack_stem_weight
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (16, 3, 800, 1333), generator=g, dtype=torch.uint8).cuda()
wv = pack_stem_weight(torch.randn(64, 3, 7, 7, generator=g) * 0.05, torch.float16).cuda()
b = torch.randn(64, generator=g).cuda()
fn = lambda: ops.stem_maxpool_raw(img, 800, 1344, (103.53, 116.28, 123.675), (1.0, 1.0, 1.0), wv, b)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
print("%-60s %.1f us" % (os.path.basename(os.environ.get("OSR_VARIANT_LIB", "product")), best), flush=True)
if os.environ.get("CHECK"):
    ref = torch.load(os.environ["CHECK"]) if os.path.exists(os.environ["CHECK"]) else None
    out = fn().cpu()
    if ref is None: torch.save(out, os.environ["CHECK"])
    else: print("  identical to the product library:", torch.equal(out, ref), flush=True)
