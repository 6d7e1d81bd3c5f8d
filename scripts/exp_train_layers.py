"""Per-launch timing of the training step's MFMA launches (forward convs, dgrad, wgrad): one step with a device sync around each
call, so a line per layer with its shape, time and algorithmic TFLOP/s. Experiment record (not a benchmark): finds the layers whose
launch plan is off. Usage: python scripts/exp_train_layers.py [--kind wgrad|dgrad|conv]."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="wgrad,dgrad,conv")
    ap.add_argument("--min-us", type=float, default=0.0)
    args = ap.parse_args()
    pkg = ge.load_package()
    pkg._lib.load()
    from openset_rcnn_amd.host import ops
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params
    dev = "cuda:0"
    tr = OpensetRCNNTrainer(random_params(0), dtype=torch.float16, device=dev, lr=1e-5, loss_scale=1024.0)
    g = torch.Generator().manual_seed(99)
    n, h, w, ngt = 16, 800, 1333, 8
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).to(dev)
    hw = torch.tensor([(h, w)] * n, dtype=torch.int32, device=dev)
    ctr = torch.rand(n, ngt, 2, generator=g) * torch.tensor([w * 0.8, h * 0.8]) + 40
    size = torch.rand(n, ngt, 2, generator=g) * 480 + 32
    gt = torch.cat((ctr - size / 2, ctr + size / 2), dim=2)
    gt[..., 0::2].clamp_(0, w)
    gt[..., 1::2].clamp_(0, h)
    gcls = torch.randint(0, 20, (n, ngt), generator=g)
    gcnt = torch.full((n,), ngt, dtype=torch.int32)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = {k: torch.rand(s, generator=g).to(dev) for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + ngt)))}
    a = (images, hw, 800, 1344, gt.to(dev), gcls.to(dev), gcnt.to(dev), keys)
    for _ in range(2):
        tr.step(*a)
    torch.cuda.synchronize()
    rows = []

    def wrap(name, flops_of):
        fn = getattr(ops, name)

        def timed(*aa, **kw):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn(*aa, **kw)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t) * 1e6
            fl, desc = flops_of(*aa, **kw)
            rows.append((name, desc, us, fl / us / 1e6))
            return out
        setattr(ops, name, timed)

    def f_wgrad(x, dy, kh, kw_, stride=1, pad=0, **kw):
        nn, ho, wo, co = dy.shape
        return 2.0 * nn * ho * wo * co * kh * kw_ * x.shape[3], f"x{tuple(x.shape)} dy{tuple(dy.shape)} k{kh} s{stride}"

    def f_dgrad(dy, wd, out_hw, *aa, **kw):
        nn, ho, wo, co = dy.shape
        return 2.0 * nn * out_hw[0] * out_hw[1] * wd.numel(), f"dy{tuple(dy.shape)} w{tuple(wd.shape)} -> {out_hw}"

    def f_conv(x, wgt, *aa, **kw):
        return 0.0, f"x{tuple(x.shape)} w{tuple(wgt.shape)}"
    kinds = args.kind.split(",")
    if "wgrad" in kinds:
        wrap("conv2d_wgrad", f_wgrad)
    if "dgrad" in kinds:
        wrap("conv2d_dgrad", f_dgrad)
    if "conv" in kinds:
        wrap("conv2d", f_conv)
    tr.step(*a)
    torch.cuda.synchronize()
    tot = {}
    for name, desc, us, tf in rows:
        tot[name] = tot.get(name, 0.0) + us
        if us >= args.min_us:
            print(f"{name:14s} {us:9.1f} us {tf:7.1f} TF/s  {desc}")
    print({k: round(v / 1e3, 2) for k, v in tot.items()})


if __name__ == "__main__":
    main()
