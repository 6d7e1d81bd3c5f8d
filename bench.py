#!/usr/bin/env python3
"""Headline benchmark: images/sec of the Openset R-CNN inference hot path (VOC-COCO openset_rcnn_R50_FPN_128k,
batch 16 per GPU, synthetic 3x800x1333 uint8 BGR images, random-init weights) on N MI355X of one node.

One "step" = one full pass of the hot path over one batch per GPU: preprocess -> R50+FPN -> CF-RPN head ->
proposal selection -> RoIAlign -> box head -> predictor -> PLN -> softmax classifier -> NMS, inputs already
resident in HBM. Images shard across ranks with no data-path collective (weak scaling).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank/GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel family (MFMA implicit-GEMM conv/FC): algorithmic FLOPs per step divided by
                  the summed duration of its launches, measured with HIP events on the launch stream.
  cpu_baseline -- the CPU oracle ("port") timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

METRIC = "images/sec at 3x800x1333, R50-FPN, 1/2/4/8 MI355X; mAP_k vs ref"
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0}  # dense, /opt/skills/guides/MI355X_MICROARCH.md


def host_cores() -> int:
    """CPUs this process may actually use: min(online CPUs, affinity mask, cgroup v2 cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(params, batch: int, iters: int):
    """The oracle (a port: the reference itself cannot run here) timed on the host cores."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # the C oracle's OpenMP runtime reads it when the library loads
    from oracle import c_binding as CO
    from oracle import osr_oracle as O
    g = torch.Generator().manual_seed(0)
    images = [torch.randint(0, 256, (3, 800, 1333), generator=g, dtype=torch.uint8) for _ in range(batch)]
    torch.set_num_threads(cores)
    with torch.no_grad():
        O.detector_inference(images[:1], params, params, roi_align_fn=CO.roi_align)  # warm-up (1 image)
        t0 = time.perf_counter()
        for _ in range(iters):
            O.detector_inference(images, params, params, roi_align_fn=CO.roi_align)
        dt = time.perf_counter() - t0
    return dict(value=batch * iters / dt, unit="images/sec", cores=cores, kind="port",
                sample=f"{iters} pass(es) of the fp32 torch-CPU/C oracle over {batch} synthetic 3x800x1333 images "
                       f"(same seeded weights), {torch.get_num_threads()} threads, after a 1-image warm-up")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (config: 16)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--streams", type=int, default=2, help="micro-batch streams inside one GPU (1 = single stream)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage breakdown to stderr")
    ap.add_argument("--layers", action="store_true", help="also print every MFMA launch (time, TFLOP/s, GB/s) to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (never set by the driver): OSR_DIST_BACKEND=gloo lets several ranks share one GPU on a 1-GPU box
    backend = os.environ.get("OSR_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)

    pkg = ge.load_package()
    pkg._lib.load()
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params

    params = random_params(0)  # identical weights on every rank (data parallel)
    tdt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    eng = OpensetRCNNEngine(params, dtype=tdt, device=f"cuda:{local_rank}")
    g = torch.Generator().manual_seed(1234 + rank)  # each rank has its own shard of images
    images = torch.randint(0, 256, (args.batch, 3, 800, 1333), generator=g, dtype=torch.uint8).to(eng.device)
    image_hw = torch.tensor([(800, 1333)] * args.batch, dtype=torch.int32, device=eng.device)

    def step():
        if args.streams > 1:
            return eng.forward_device_streams(images, image_hw, 800, 1344, args.streams)
        return eng.forward_device(images, image_hw, 800, 1344)

    if args.graph:
        graph, gout = eng.capture(images, image_hw, 800, 1344, args.streams)

        def step():  # noqa: F811  one hipGraph launch replays the whole pass
            graph.replay()
            return gout

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_det = int(out[3].sum().item())

    # ---- roofline of the dominant kernel family, measured live with HIP events on the launch stream ----
    # (the single-stream full-batch pass picks other tile configurations than the micro-batched one: run it once untimed so
    # that no launch of the bracketed pass is the first use of its kernel)
    eng.forward_device(images, image_hw, 800, 1344)
    torch.cuda.synchronize()
    eng.profile = []
    eng.forward_device(images, image_hw, 800, 1344)  # attribution pass: one stream, so each launch can be bracketed
    torch.cuda.synchronize()
    prof = eng.profile
    eng.profile = None
    mfma_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1, _ in prof)
    mfma_flops = sum(f for _, f, _, _, _ in prof)
    achieved = mfma_flops / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    # HBM-side traffic of the same kernel family: PMC counters cannot be read from inside the process, so the value is
    # the one collected with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2 per the gfx950
    # guide) on this very command and committed under profiles/; null when that file is absent.
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "r01_j_kernel_times_and_traffic.json")
    if os.path.exists(tfile):
        with open(tfile) as fh:
            cf = json.load(fh).get("conv_family", {})
        traffic = round(cf.get("fetch_GB_per_step_x2corrected", 0.0) + cf.get("write_GB_per_step", 0.0), 2)
        traffic_source = "profiles/r01_j_kernel_times_and_traffic.json (scripts/profile_round.sh: --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    algo_bytes = sum(nb for _, _, _, _, nb in prof)
    roofline = dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic,
                    traffic_unit="GB of HBM traffic per step of the same kernel family (all its launches)", traffic_source=traffic_source,
                    algorithmic_GB_per_step=round(algo_bytes / 1e9, 2),
                    kernel="conv_igemm64_kernel family (implicit-GEMM conv + FC)", launches_per_step=len(prof),
                    flops_per_step=mfma_flops, kernel_ms_per_step=round(mfma_ms, 3))
    if args.layers and rank == 0:
        for name, f, e0, e1, nb in prof:
            ms_ = e0.elapsed_time(e1)
            print(f"  {name:48s} {ms_ * 1e3:8.1f} us {f / ms_ / 1e9:8.1f} TFLOP/s {nb / ms_ / 1e6:8.1f} GB/s", file=sys.stderr)
    if args.stages and rank == 0:
        agg = {}
        for name, f, e0, e1, _ in prof:
            key = name.split(".")[1] if name.startswith("backbone.bottom_up") else name.split(".")[0] + "." + name.split(".")[1]
            key = name.split(".")[2] if name.startswith("backbone.bottom_up") else key
            a = agg.setdefault(key, [0.0, 0.0])
            a[0] += e0.elapsed_time(e1)
            a[1] += f
        for k, (ms, f) in agg.items():
            print(f"  {k:40s} {ms:8.3f} ms  {f / ms / 1e9 if ms else 0:8.1f} TFLOP/s", file=sys.stderr)
        print(f"  MFMA kernels total {mfma_ms:.3f} ms of {elapsed / args.steps * 1e3:.3f} ms/step", file=sys.stderr)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": METRIC, "value": round(world * args.batch * args.steps / elapsed, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "VOC-COCO openset_rcnn_R50_FPN_128k.yaml, inference-only, 3x800x1333 uint8 BGR -> padded 800x1344, "
                                   "1000 proposals/level (4273/img), 1000 dets/img, 50+50 final",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world} (images sharded, no collective)",
                       "weights": "random-init (seed 0), FrozenBN folded", "detections_last_step": n_det,
                       "micro_batch_streams": args.streams, "hipgraph": bool(args.graph)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(params, args.cpu_batch, args.cpu_iters)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
