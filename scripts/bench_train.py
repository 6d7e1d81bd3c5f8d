"""Training-step benchmark (BASELINE.json configs[2]: VOC-COCO yaml, train step with the PLN loss, batch 16 per GPU, 3x800x1333).
Not the driver's bench (bench.py measures the headline inference metric); same launch contract: one process per GPU,
`python -m torch.distributed.run --nproc-per-node N scripts/bench_train.py --gpus N`, gradients all-reduced over RCCL.
Prints one JSON line: images/s over all ranks and ms per iteration (forward + backward + all-reduce + SGD)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--gt", type=int, default=8, help="ground-truth boxes per image")
    ap.add_argument("--graph", action="store_true", help="capture one iteration into a hipGraph and replay it (single GPU)")
    ap.add_argument("--no-side-wgrad", action="store_true", help="weight gradients in the main stream (A/B of the second stream)")
    ap.add_argument("--wgrad-on-main", default="", help="comma-separated block prefixes (e.g. backbone.bottom_up.res3.0) whose weight gradients ride on the main stream (A/B)")
    ap.add_argument("--no-overlap-targets", action="store_true", help="anchor targets in the main stream (A/B of the side stream)")
    ap.add_argument("--phases", action="store_true", help="also time forward / backward / update separately (extra syncs)")
    args = ap.parse_args()
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("OSR_DIST_BACKEND", "nccl")  # gloo: rehearsal of the multi-rank path on a 1-GPU box
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    pkg = ge.load_package()
    pkg._lib.load()
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params
    dev = f"cuda:{local_rank}"
    tr = OpensetRCNNTrainer(random_params(0), dtype=torch.float16, device=dev, lr=1e-5, loss_scale=1024.0)
    tr.overlap_targets = not args.no_overlap_targets
    tr.side_wgrad = not args.no_side_wgrad
    tr.wgrad_on_main = set(x for x in args.wgrad_on_main.split(',') if x)
    g = torch.Generator().manual_seed(99 + rank)
    n, h, w = args.batch, 800, 1333
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).to(dev)
    hw = torch.tensor([(h, w)] * n, dtype=torch.int32, device=dev)
    ctr = torch.rand(n, args.gt, 2, generator=g) * torch.tensor([w * 0.8, h * 0.8]) + 40
    size = torch.rand(n, args.gt, 2, generator=g) * 480 + 32  # 32..512 px
    gt = torch.cat((ctr - size / 2, ctr + size / 2), dim=2)
    gt[..., 0::2].clamp_(0, w)
    gt[..., 1::2].clamp_(0, h)
    gcls = torch.randint(0, 20, (n, args.gt), generator=g)
    gcnt = torch.full((n,), args.gt, dtype=torch.int32)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = {k: torch.rand(s, generator=g).to(dev) for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + args.gt)))}
    a = (images, hw, 800, 1344, gt.to(dev), gcls.to(dev), gcnt.to(dev), keys)
    for _ in range(args.warmup):
        losses = tr.step(*a)
    torch.cuda.synchronize()
    step = lambda: tr.step(*a)  # noqa: E731
    if args.graph:
        if world > 1:
            raise SystemExit("--graph: single GPU only (the all-reduce is not captured)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                glosses = tr.step(*a)
        torch.cuda.current_stream().wait_stream(side)

        def step():  # noqa: F811
            graph.replay()
            return glosses
        step()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    out = dict(metric="train images/sec at 3x800x1333, R50-FPN (forward + backward + all-reduce + SGD)", value=round(n * world * args.steps / el, 2),
               unit="images/sec", n_gpus=world, steps=args.steps, ms_per_step=round(el / args.steps * 1e3, 2), batch_per_gpu=n,
               trainable_parameters=tr.num_params, losses={k: round(float(v), 4) for k, v in losses.items()}, dtype="f16", data="synthetic")
    if args.phases:
        def timed(fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r_ = fn()
            torch.cuda.synchronize()
            return r_, (time.perf_counter() - t) * 1e3
        (_, saved), tf = timed(lambda: tr._forward(*a))
        _, tb = timed(lambda: tr._backward(saved, n))
        _, tu = timed(lambda: tr._update(tr.all_reduce_grads()))
        out["phases_ms"] = dict(forward=round(tf, 2), backward=round(tb, 2), allreduce_update=round(tu, 2))
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
