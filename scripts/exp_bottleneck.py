"""Time the fused res2 bottleneck (osr_bottleneck_fwd) against the separate launches at the benchmark's shape (16 x 200 x 336)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine
from openset_rcnn_amd.host.weights import random_params
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = OpensetRCNNEngine(random_params(0), None, torch.float16, DEV)
g = torch.Generator().manual_seed(3)
x64 = torch.randn(16, 200, 336, 64, generator=g).clamp_(min=0).half().to(DEV)
x256 = torch.randn(16, 200, 336, 256, generator=g).clamp_(min=0).half().to(DEV)
def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for fused in (True, False):
    eng.fuse_res2 = fused
    t0 = timeit(lambda: eng._bottleneck(x64, "backbone.bottom_up.res2.0", True, 1))
    t1 = timeit(lambda: eng._bottleneck(x256, "backbone.bottom_up.res2.1", False, 1))
    gb0 = (x64.numel() + x256.numel()) * 2 / 1e9
    gb1 = 2 * x256.numel() * 2 / 1e9
    print(f"fused={fused}: block 0 (cin 64, projection) {t0:7.1f} us ({gb0 / t0 * 1e6:6.0f} GB/s algorithmic), block 1 (cin 256) {t1:7.1f} us ({gb1 / t1 * 1e6:6.0f} GB/s)")
