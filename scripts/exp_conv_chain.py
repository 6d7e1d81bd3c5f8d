"""conv2 -> conv3 chain (osr_conv2d_chain_fwd) against the two separate launches at the res3 shape of the bench (16 x 100 x 168,
128 -> 128 3x3 -> 512 1x1 + residual): equality and time per block. python scripts/exp_conv_chain.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg._lib.load()
ops = pkg.ops
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
dt = torch.float16
x = (torch.randn(n, 100, 168, 128, generator=g)).to(dt).to(dev)
res = torch.randn(n, 100, 168, 512, generator=g).to(dt).to(dev)
w2 = (torch.randn(128, 3, 3, 128, generator=g) * (2.0 / 1152) ** 0.5).to(dt).to(dev)
w3 = (torch.randn(512, 1, 1, 128, generator=g) * (1.0 / 128) ** 0.5).to(dt).to(dev)
b2 = (torch.randn(128, generator=g) * 0.3).to(dev)
b3 = (torch.randn(512, generator=g) * 0.3).to(dev)


def sep():
    return ops.conv2d(ops.conv2d(x, w2, b2, 1, 1, relu=True), w3, b3, relu=True, residual=res, res_mode=1)


def fused():
    return ops.conv2d_chain(x, w2, b2, w3, b3, res, 1, 1)


a, b = sep(), fused()
torch.cuda.synchronize()
print("equal:", torch.equal(a, b), "max diff", (a.float() - b.float()).abs().max().item())


def timeit(f, it=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for rep in range(3):
    print(f"separate {timeit(sep):8.1f} us   fused {timeit(fused):8.1f} us")
