"""Experiment driver (not part of the product): the fused stem (osr_stem_maxpool_fwd) against stem conv + max pool at the bench's size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
from openset_rcnn_amd.host.weights import pack_stem_weight
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (16, 3, 800, 1333), generator=g, dtype=torch.uint8).cuda()
wv = pack_stem_weight(torch.randn(64, 3, 7, 7, generator=g) * 0.05, torch.float16).cuda()
b = torch.randn(64, generator=g).cuda()
xpad = ops.preprocess(img, 800, 1344, (103.53, 116.28, 123.675), (1.0, 1.0, 1.0), torch.float16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, tag, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-40s %.1f us" % (tag, e0.elapsed_time(e1) / reps * 1e3), flush=True)
t(lambda: ops.stem_conv(xpad, wv, b, 800, 1344), "stem conv (view, generic kernel)")
y = ops.stem_conv(xpad, wv, b, 800, 1344)
t(lambda: ops.maxpool3x3s2(y), "max pool")
t(lambda: ops.stem_maxpool(xpad, wv, b, 800, 1344), "fused stem")
assert torch.equal(ops.stem_maxpool(xpad, wv, b, 800, 1344), ops.maxpool3x3s2(y))
t(lambda: ops.preprocess(img, 800, 1344, (103.53, 116.28, 123.675), (1.0, 1.0, 1.0), torch.float16), "preprocess")
t(lambda: ops.stem_maxpool_raw(img, 800, 1344, (103.53, 116.28, 123.675), (1.0, 1.0, 1.0), wv, b), "fused stem from the raw batch")
assert torch.equal(ops.stem_maxpool_raw(img, 800, 1344, (103.53, 116.28, 123.675), (1.0, 1.0, 1.0), wv, b), ops.maxpool3x3s2(y))
