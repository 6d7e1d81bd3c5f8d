#!/usr/bin/env python3
"""Headline benchmark: images/sec of the Openset R-CNN inference hot path (VOC-COCO openset_rcnn_R50_FPN_128k,
batch 16 per GPU, synthetic 3x800x1333 uint8 BGR images, random-init weights) on N MI355X of one node.

One "step" = one full pass of the hot path over one batch per GPU: preprocess -> R50+FPN -> CF-RPN head ->
proposal selection -> RoIAlign -> box head -> predictor -> PLN -> softmax classifier -> NMS, inputs already
resident in HBM. Images shard across ranks with no data-path collective (weak scaling).
The K timed steps are K such passes, each captured as one hipGraph; consecutive passes alternate over --passes-in-flight lanes
(default 4: every lane has its own batch of images, its own activation / output buffers and its own HIP stream), so pass i + 1
starts while pass i is still in its RoI heads -- a serving loop with four batches in flight. All K passes have completed when the
closing synchronize returns; ms_per_step = elapsed / K (the latency of one pass alone is what --passes-in-flight 1 reports).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank/GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     -- the dominant kernel family (MFMA implicit-GEMM conv/FC): algorithmic FLOPs per step divided by
                  the summed duration of its launches, measured with HIP events on the launch stream.
  roofline_hbm -- the HBM group of SURVEY.md 8d (RoIAlign + proposal selection + NMS): algorithmic bytes per step over
                  the summed HIP-event duration, fraction of the 8 TB/s HBM peak; per kernel, and NMS as time + IoU pairs/s.
  train_step   -- BASELINE.json config 3 (train step with the PLN contrastive loss, batch 16 per GPU), measured after the
                  headline (one GPU: in a child process of this command, see train_step_child; several GPUs: in this process, all
                  ranks): ms per iteration, images/s, forward / data-gradient / weight-gradient TFLOP/s.
                  With N > 1 every rank trains on its own images and the gradient buckets are all-reduced over RCCL from inside
                  the backward (configs 4 / 5's pattern); a watchdog prints the headline line should that path not finish.
  cpu_baseline -- the CPU oracle ("port") timed on this box's host cores on a bounded sample of the same workload
                  (3 warm-ups + median of 5 at all cores; a 1-thread point on a smaller sample).
  single_pass  -- the same K steps with ONE pass in flight (one lane, one stream): the figure comparable with the reference's
                  one-batch-at-a-time loop; `value` is the throughput with --passes-in-flight batches resident.
  parity       -- what the benchmarked fp16 path delivers against the fp32 arithmetic the reference runs in: agreement of its final
                  detections with the engine's fp32 PARITY MODE on 4 seeded 256x384 images (tests/test_e2e_parity.py pins that mode to
                  the fp32 oracle), and the parity mode's own images/s at the benchmark batch.
`python bench.py --gpus N` without a launcher starts the N ranks itself (python -m torch.distributed.run, one rank per GPU,
rendezvous on 127.0.0.1) before anything touches the GPU, relays rank 0's line and returns the job's exit code.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

METRIC = "images/sec at 3x800x1333, R50-FPN, 1/2/4/8 MI355X; mAP_k vs ref"
MFMA_PEAK_TFLOPS = {"f16": 2500.0}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                                # HBM3E spec peak (same guide; 6290 GB/s is its measured copy ceiling)


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


def cpu_baseline(params, batch: int, one_thread: bool):
    """The oracle (a port: the reference itself cannot run here) timed on the host cores, BASELINE.md section 3:
    3 warm-ups + 5 timed passes, median, at all cores; plus a 1-thread point on a smaller sample (1 image, 1 warm-up + median
    of 3: a full 3 + 5 protocol at one thread would take minutes)."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # the C oracle's OpenMP runtime reads it when the library loads
    from oracle import c_binding as CO
    from oracle import osr_oracle as O
    g = torch.Generator().manual_seed(0)
    images = [torch.randint(0, 256, (3, 800, 1333), generator=g, dtype=torch.uint8) for _ in range(batch)]

    def timed(threads, imgs, warm, reps):
        torch.set_num_threads(threads)
        CO.set_threads(threads)
        ts = []
        with torch.no_grad():
            for i in range(warm + reps):
                t0 = time.perf_counter()
                O.detector_inference(imgs, params, params, roi_align_fn=CO.roi_align)
                if i >= warm:
                    ts.append(time.perf_counter() - t0)
        return len(imgs) / statistics.median(ts)

    allc = timed(cores, images, 3, 5)
    points = [dict(threads=cores, images_per_sec=round(allc, 4), sample=f"{batch} images, 3 warm-ups + median of 5")]
    if one_thread:
        one = timed(1, images[:1], 1, 3)
        points.append(dict(threads=1, images_per_sec=round(one, 4), sample="1 image, 1 warm-up + median of 3"))
        torch.set_num_threads(cores)
    return dict(value=allc, unit="images/sec", cores=cores, kind="port", os_cpu_count=os.cpu_count(),
                sample=f"fp32 torch-CPU/C oracle, whole inference path on {batch} synthetic 3x800x1333 images (same seeded weights), "
                       f"{cores} threads, 3 warm-ups + median of 5 passes (BASELINE.md section 3)",
                points=points)


def kill_group(proc) -> None:
    """End a child started with start_new_session=True together with everything it started (exact process group, by id)."""
    import signal
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=30)
    except Exception:  # noqa: BLE001
        pass


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`bench.py --gpus N` from a plain command line (the reference: train.py:287-294 `launch(main, num_gpus, ...)`): become the
    parent of a torch.distributed.run job, one rank per GPU over RCCL, rendezvous on 127.0.0.1. Nothing has touched the GPU in
    this process. The ranks inherit stdout, so rank 0's JSON line is this command's line; the job's exit code is returned."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def rank_identity(dist, rank: int, world: int, local_elapsed: float, steps: int, batch: int, gpu_uuid: str, device_name: str) -> dict:
    """What proves the N-rank line came from N ranks on N GPUs (the reference launches one process per GPU, train.py:287-294):
    the world size the process group reports, every rank's GPU uuid (must be N distinct ones on a real node; a rehearsal that
    shares one card shows one) and every rank's own images/s over the timed region. One all_gather_object of small tuples,
    outside the timed region."""
    mine = (int(rank), str(gpu_uuid), str(device_name), float(local_elapsed))
    rows = [None] * world
    if dist is not None and world > 1:
        dist.all_gather_object(rows, mine)
    else:
        rows = [mine]
    rows = sorted(rows)
    uuids = [r[1] for r in rows]
    return dict(ranks=int(dist.get_world_size()) if dist is not None and world > 1 else 1, gpu_uuids=uuids, distinct_gpus=len(set(uuids)),
                device_names=sorted(set(r[2] for r in rows)),
                per_rank_images_per_sec=[round(batch * steps / r[3], 2) if r[3] > 0 else None for r in rows],
                per_rank_ms_per_step=[round(r[3] / steps * 1e3, 3) for r in rows])


def measure_traffic(timeout_s: int = 150):
    """HBM-side traffic of the two kernel groups, measured in this run: two child passes of this very script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (counters only, separate passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE
    doubled for gfx950), eager single stream so that every launch is one dispatch. Returns (conv family GB per step, roi_align GB per
    step, source text) or None when rocprofv3 is unavailable or a pass fails (the committed profile is quoted instead)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import pmc_summary as PS
        steps, warm = 2, 1
        passes = steps + warm + 2 + 0.25  # + the un-timed and the bracketed attribution pass + the 4-image calibration pass
        out = {}
        with tempfile.TemporaryDirectory(prefix="osr_pmc_", dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable,
                       os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--no-train-step", "--no-pmc",
                       "--no-parity", "--no-pcie", "--streams", "1", "--no-graph"]
                env = dict(os.environ, TMPDIR="/tmp")
                # a session of its own: on a time-out the WHOLE group is killed (rocprofv3 and the profiled python under it) and
                # waited for, so that nothing of it still runs on the GPU beside the legs that follow
                proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd="/tmp", start_new_session=True)
                try:
                    rc = proc.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    rc = None
                if rc is None or rc != 0:
                    kill_group(proc)
                    return None
                out[counter] = PS.counter_sum(d, passes)
        conv = lambda c, f: sum(v[0] for k, v in out[c].items() if k.startswith("conv_igemm") or k.startswith("splitk_reduce") or k.startswith("bottleneck64") or k.startswith("stem_pool")) * 1024 * f / 1e9  # noqa: E731
        roi = lambda c, f: sum(v[0] for k, v in out[c].items() if k.startswith("roi_align")) * 1024 * f / 1e9  # noqa: E731
        return (round(conv("FETCH_SIZE", 2) + conv("WRITE_SIZE", 1), 2), round(roi("FETCH_SIZE", 2) + roi("WRITE_SIZE", 1), 2),
                "measured in this run: two child passes of `bench.py --steps 2 --warmup 1 --streams 1 --no-graph` under rocprofv3 --pmc FETCH_SIZE "
                "(x2, gfx950) / --pmc WRITE_SIZE, separate passes")
    except Exception:  # noqa: BLE001  (profiler missing pieces, time-out, parse error: fall back to the committed profile)
        return None


def synthetic_gt(n: int, h: int, w: int, per_image: int = 8, seed: int = 0):
    """BASELINE.md section 3 / SURVEY.md 8d: 8 boxes per image, sizes uniform in 32-512 px, classes uniform in [0, 20), seed 0."""
    g = torch.Generator().manual_seed(seed)
    size = torch.rand(n, per_image, 2, generator=g) * (512 - 32) + 32
    size[..., 0].clamp_(max=w - 1)
    size[..., 1].clamp_(max=h - 1)
    x1 = torch.rand(n, per_image, generator=g) * (w - size[..., 0])
    y1 = torch.rand(n, per_image, generator=g) * (h - size[..., 1])
    boxes = torch.stack((x1, y1, x1 + size[..., 0], y1 + size[..., 1]), dim=-1)
    classes = torch.randint(0, 20, (n, per_image), generator=g)
    return boxes, classes, torch.full((n,), per_image, dtype=torch.int32)


def train_step_leg(params, tdt, device, images, image_hw, steps: int, warmup: int, dist=None, rank: int = 0, world: int = 1, dense_rpn_bwd: bool = False,
                   no_chain: bool = False):
    """BASELINE.json config 3: forward + explicit backward + SGD, batch 16 per GPU at 800x1333. With several ranks (configs 4 / 5's
    pattern) every rank trains on its own images and the flat gradient buffer is all-reduced over RCCL in >= 25 MB buckets issued
    from inside the backward (parallel.GradBuckets); the time is the maximum over the ranks, bracketed by barriers."""
    from openset_rcnn_amd.host import ops
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    n = images.shape[0]
    tr = OpensetRCNNTrainer(params, dtype=tdt, device=device, lr=1e-4, loss_scale=1024.0 if tdt == torch.float16 else 1.0)
    tr.sparse_rpn_bwd = not dense_rpn_bwd
    tr.chain_forward = not no_chain
    gt, gcls, gcnt = synthetic_gt(n, 800, 1333)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    g = torch.Generator().manual_seed(rank)
    keys = {k: torch.rand(s, generator=g).to(device) for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + gt.shape[1])))}
    args = (images, image_hw, 800, 1344, gt.to(device), gcls.to(device), gcnt.to(device), keys)
    for _ in range(warmup):
        losses = tr.step(*args)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.step(*args)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the same loop with the loader's next batch handed to step(): its frozen prefix (stem + res2: weights no update touches) runs under
    # this iteration's backward. Reported beside ms_per_iter, not in its place (one rank only: a secondary figure)
    dt_pipe = None
    if dist is None:
        for _ in range(warmup):
            tr.step(*args, next_images=images)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            losses_p = tr.step(*args, next_images=images)
        torch.cuda.synchronize()
        dt_pipe = (time.perf_counter() - t0) / steps
        tr._prefetched = None
    # one instrumented iteration: algorithmic FLOPs of the MFMA launches per phase, HIP events around the phases
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ops.FLOP_COUNT = dict(conv=0.0, wgrad=0.0)
    ev[0].record()
    _, saved = tr._forward(*args)
    ev[1].record()
    fwd = dict(ops.FLOP_COUNT)
    tr._backward(saved, n, overlap=False)
    ev[2].record()
    bwd = {k: ops.FLOP_COUNT[k] - fwd[k] for k in fwd}
    ops.FLOP_COUNT = None
    tr._update(tr.all_reduce_grads())
    ev[3].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    total = float(sum(float(v) for v in losses.values()))
    mfma_tflop = (fwd["conv"] + bwd["conv"] + bwd["wgrad"]) / 1e12
    roofline = dict(bound="mfma", achieved=round(mfma_tflop * world / dt, 1), peak=MFMA_PEAK_TFLOPS["f16"] * world, unit="TFLOP/s",
                    frac=round(mfma_tflop / dt / MFMA_PEAK_TFLOPS["f16"], 4), traffic=None,
                    note="algorithmic FLOPs of every MFMA launch of the iteration (forward incl. the fused CF-RPN head, data-gradient and weight-gradient "
                         "convolutions; RoI rows = the 512 sampled proposals per image) over the WHOLE iteration's wall time (all kernels, update included)",
                    forward_frac=round(fwd["conv"] / ms[0] / 1e9 / MFMA_PEAK_TFLOPS["f16"], 4),
                    backward_frac=round((bwd["conv"] + bwd["wgrad"]) / ms[1] / 1e9 / MFMA_PEAK_TFLOPS["f16"], 4))
    where = "1xMI355X (BASELINE.json config 3)" if world == 1 else \
        f"{world}xMI355X, one process per GPU, bucketed gradient all-reduce over RCCL overlapped with the backward (BASELINE.json configs 4 / 5's pattern)"
    return dict(config=f"VOC-COCO openset_rcnn_R50_FPN_128k.yaml, train step with PLN contrastive loss, batch 16 per GPU, {where}",
                ms_per_iter=round(dt * 1e3, 3), images_per_sec=round(n * world / dt, 2), n_gpus=world, steps=steps, warmup=warmup,
                forward_ms=round(ms[0], 3), backward_ms=round(ms[1], 3), update_ms=round(ms[2], 3),
                update_ms_note="all-reduce of the whole flat gradient buffer (not overlapped in this instrumented iteration) + SGD" if world > 1 else "SGD",
                forward_TFLOP=round(fwd["conv"] / 1e12, 3), dgrad_TFLOP=round(bwd["conv"] / 1e12, 3), wgrad_TFLOP=round(bwd["wgrad"] / 1e12, 3),
                forward_TFLOPs=round(fwd["conv"] / ms[0] / 1e9, 1), backward_TFLOPs=round((bwd["conv"] + bwd["wgrad"]) / ms[1] / 1e9, 1),
                whole_iteration_TFLOPs=round((fwd["conv"] + bwd["conv"] + bwd["wgrad"]) * world / dt / 1e12, 1), roofline=roofline,
                trainable_params=tr.num_params, gradient_bytes_all_reduced=tr.num_params * 4 if world > 1 else 0, gt_boxes_per_image=8,
                proposals_per_image_train=cap, rois_sampled_per_image=512, loss_total_last=round(total, 4), overflow_skipped_steps=tr.overflow_steps,
                pipelined_ms_per_iter=round(dt_pipe * 1e3, 3) if dt_pipe is not None else None,
                pipelined_note="step(next_images=...): the NEXT batch's frozen prefix (preprocessing + stem + res2, 1.3 ms, a function of that batch and of "
                               "frozen weights only) is enqueued between this iteration's forward and backward and runs under the chain of small head / loss "
                               "launches; every iteration still computes one prefix -- software pipelining, not caching; ms_per_iter above is WITHOUT it",
                rpn_head_backward="on the anchors the loss samples (<= 512 per image; osr_rpn_sparse_rows / gather_cols / scatter_cols_add): hidden state "
                                  "recomputed, dW and the per-tap data gradient as GEMMs over the listed rows -- the FLOPs above count those, not the "
                                  "dense launches' 2 x 1.7 TFLOP on zero rows" if tr.sparse_rpn_bwd else "dense (every anchor row)")


def config4_leg(tdt, device, steps: int, warmup: int, dist=None, rank: int = 0, world: int = 1):
    """BASELINE.json config 4: GraspNet openset_rcnn_R50_FPN_128k.yaml (28 known of 88 classes, its loss weights and thresholds, from
    configs/graspnet.yaml through the same yaml -> engine mapping run_net.py uses), 1280x720 frames, batch 8 per GPU (64 over 8 GPUs), one
    training iteration = host frames -> upload + on-device ResizeShortestEdge(800, 1333) = 750x1333 -> pad 768x1344 -> forward + backward +
    bucketed gradient all-reduce (RCCL, issued from inside the backward) + SGD. Also times ONE all-reduce of the whole flat gradient
    buffer alone (166.5 MB fp32; /root/reference/train.py:201-205's DDP moves the same bytes), so that the fraction of it the
    backward hides can be read off: hidden = 1 - (iteration time with - without the collective) / all-reduce time alone."""
    import numpy as np
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    from openset_rcnn_amd.host.data import DeviceResizer
    from openset_rcnn_amd.host.modeling import engine_cfg_from
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "graspnet.yaml"))
    ecfg = engine_cfg_from(cfg)
    K = ecfg["num_known"]
    class_map = (torch.arange(0, ecfg["num_classes"], 3)[:K].to(torch.int64) + 1)  # a sparse known-id table, like GraspNet's class_id (PLN :80-95)
    n, fh, fw, h, w, hp, wp = 8, 720, 1280, 750, 1333, 768, 1344
    params = random_params(0, num_known=K)
    tr = OpensetRCNNTrainer(params, cfg=ecfg, dtype=tdt, device=device, lr=1e-5, loss_scale=512.0, class_map=class_map)
    g = np.random.default_rng(100 + rank)
    frames = [g.integers(0, 256, (fh, fw, 3), dtype=np.uint8) for _ in range(n)]  # decoded frames in host memory (each rank its own shard)
    rz = DeviceResizer(device)
    # two batch buffers in turn: the copy stream fills one while the step launched before still reads the other; a buffer is refilled only
    # after the step that consumed it has been ENQUEUED AND its event reached (the copy stream waits for that step's end-of-step event)
    batches = [torch.empty((n, 3, h, w), dtype=torch.uint8, device=device) for _ in range(2)]
    consumed = [None, None]
    turn = [0]
    gt_g = torch.Generator().manual_seed(rank)
    ngt = 6
    ctr = torch.rand(n, ngt, 2, generator=gt_g) * torch.tensor([w * 0.8, h * 0.8]) + 40
    size = torch.rand(n, ngt, 2, generator=gt_g) * 300 + 32
    gt = torch.cat((ctr - size / 2, ctr + size / 2), dim=2)
    gt[..., 0::2].clamp_(0, w); gt[..., 1::2].clamp_(0, h)
    gcls = class_map[torch.randint(0, K, (n, ngt), generator=gt_g)]
    gcnt = torch.full((n,), ngt, dtype=torch.int32)
    shapes = tr.eng.pyramid_shapes(hp, wp)
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = {k: torch.rand(sz, generator=gt_g).to(device) for k, sz in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + ngt)))}
    hw = torch.tensor([(h, w)] * n, dtype=torch.int32, device=device)
    dev_args = (hw, hp, wp, gt.to(device), gcls.to(device), gcnt.to(device), keys)

    def load():
        slot = turn[0] & 1
        turn[0] += 1
        if consumed[slot] is not None:
            rz.stream.wait_event(consumed[slot])
        for i, f in enumerate(frames):
            rz(f, (h, w), out=batches[slot][i], wait=False)
        torch.cuda.current_stream().wait_event(rz.done)
        return slot, batches[slot]

    def run(k, with_collective=True):
        for _ in range(k):
            slot, images = load()
            if with_collective:
                losses = tr.step(images, *dev_args)
            else:
                # the same iteration without the gradient exchange. It must leave the job as it found it: an update from the LOCAL
                # gradient would make the ranks' weights diverge in the middle of the benchmark and queue a rank-local verdict in the
                # loss scaler every rank is supposed to apply in lockstep. So: forward + backward with no bucket issued,
                # then the update's launches on a ZEROED gradient with learning rate, momentum and weight decay 0 -- the same kernels
                # over the same bytes (what is being timed), parameters and momentum bit for bit unchanged, no verdict recorded.
                losses = tr.step(images, *dev_args, update=False)
                tr.buckets.reset()
                tr.grad_flat.zero_()
                tr.null_update()
            consumed[slot] = torch.cuda.current_stream().record_event()
        return losses

    def timed_iters(k, **kw):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(k, **kw)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    run(warmup)
    dt, losses = timed_iters(steps)
    out = dict(config=f"GraspNet openset_rcnn_R50_FPN_128k.yaml (configs/graspnet.yaml), 1280x720 frames -> device resize 750x1333 -> pad 768x1344, batch 8 per GPU, "
                      f"global batch {8 * world}, {world}xMI355X, train step (BASELINE.json config 4)",
               ms_per_iter=round(dt * 1e3, 3), images_per_sec=round(n * world / dt, 2), n_gpus=world, steps=steps, warmup=warmup, batch_per_gpu=n,
               num_known=K, num_classes=ecfg["num_classes"], unk_thr=ecfg["unk_thr"], pln_loss_weight=ecfg["pln_loss_weight"],
               loss_total_last=round(float(sum(float(v) for v in losses.values())), 4), overflow_skipped_steps=tr.overflow_steps,
               gradient_bytes=tr.num_params * 4, input="host frames (uint8 HWC) -> pinned staging -> device, resized on the device inside the timed region (copy stream, two batch buffers: the upload of a batch overlaps the step before it)")
    if dist is not None and world > 1:
        # the collective alone: one all-reduce of the whole flat buffer, nothing else on the GPU
        flat = tr.grad_flat
        for _ in range(2):
            dist.all_reduce(flat)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dist.all_reduce(flat)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        ar = (time.perf_counter() - t0) / reps
        flat.zero_()
        dt_local, _ = timed_iters(steps, with_collective=False)
        exposed = max(0.0, dt - dt_local)
        out.update(all_reduce_ms=round(ar * 1e3, 3), all_reduce_GBps_per_rank=round(tr.num_params * 4 / ar / 1e9, 1),
                   ms_per_iter_without_all_reduce=round(dt_local * 1e3, 3), all_reduce_exposed_ms=round(exposed * 1e3, 3),
                   all_reduce_hidden_fraction=round(max(0.0, 1.0 - exposed / ar), 3) if ar > 0 else None,
                   all_reduce_note="whole flat fp32 gradient buffer in one call, idle GPU (the in-step exchange is the same bytes in >= 25 MB buckets issued "
                                   "from inside the backward, host/parallel.py GradBuckets); hidden = 1 - (iteration with - without the collective) / this")
    return out


def train_step_child(args) -> dict:
    """The single-GPU train-step leg runs in a child process after the inference measurement. In one process the two disturb each
    other: the leg is an eager stream of ~500 launches per iteration and ran 8 % slower after the four-lane inference loop (38.0
    against 35.1 ms; likewise with GPU_MAX_HW_QUEUES raised: every hardware queue a process has used stays in the command
    processor's rotation), and run first it cost the inference loop 2.5 %. The child rebuilds the same weights (same seeds)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--train-only", "--train-steps", str(args.train_steps), "--dtype", args.dtype, "--batch", str(args.batch),
           "--no-cpu-baseline", "--no-pmc", "--no-parity", "--no-pcie"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=os.path.dirname(os.path.abspath(__file__)))
        last = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not last:
            return {"error": f"train-step child exited {r.returncode}: {r.stderr.decode()[-300:]}"}
        return json.loads(last[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def pcie_leg(lanes, dev, batch: int, steps: int, warm: int = 2):
    """The PCIe-inclusive rate: every step's 16 frames start in HOST memory as decoded 600x1000x3 uint8 images, go through a pinned
    staging buffer and an asynchronous H2D copy on a copy stream, are resized ON THE DEVICE to 800x1333 (osr_resize_bilinear_u8:
    Pillow's BILINEAR, bit-exact -- what [d2] ResizeShortestEdge does on the host in the reference's loader, train.py:129) straight
    into a lane's input batch, and the lane's captured pass runs behind an event. Uploads / resizes of the next batch overlap the
    passes in flight (the lanes of the headline). `value` never includes this; the boundary takes device buffers."""
    import numpy as np
    from openset_rcnn_amd.host.data import DeviceResizer
    rz = DeviceResizer(dev)
    rng = np.random.RandomState(7)
    frames = [rng.randint(0, 256, (600, 1000, 3)).astype(np.uint8) for _ in range(batch)]
    turn = [0]

    def step():
        g_, o_, st_, imgs_ = lanes[turn[0] % len(lanes)]
        turn[0] += 1
        rz.stream.wait_stream(st_)  # the lane's previous pass has read its input batch before it is overwritten
        for i, f in enumerate(frames):
            rz(f, (800, 1333), out=imgs_[i], wait=False)
        st_.wait_event(rz.done)
        with torch.cuda.stream(st_):
            g_.replay()
        return o_

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return dict(images_per_sec=round(batch * steps / el, 2), ms_per_step=round(el / steps * 1e3, 3), steps=steps, passes_in_flight=len(lanes),
                source=f"{batch} decoded frames of 600x1000x3 uint8 in host memory per step (28.8 MB)",
                path="host frame -> pinned staging -> async H2D (copy stream) -> osr_resize_bilinear_u8 to 3x800x1333 (Pillow BILINEAR, bit-exact) "
                     "into the lane's input batch -> the captured pass; overlapped with the passes in flight",
                note="single host thread fills the pinned buffers (a memcpy per frame); image decode (JPEG) is not included")


F32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, /opt/skills/guides/MI355X_MICROARCH.md


def graph_rate(eng, images, image_hw, steps: int, warm: int = 2):
    """images/s of one engine over `steps` replays of its captured pass (one lane, one stream): how the headline's single_pass
    figure is measured."""
    graph, out = eng.capture(images, image_hw, 800, 1344, 1)
    for _ in range(warm):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del graph, out
    return images.shape[0] / dt, dt


def full_lists_leg(params, tdt, dev, images, image_hw, batch: int, passes: int, steps: int = 10):
    """The headline's schedule with EVERY slot of the proposal lists real: with random-init weights about a
    fifth of the 4273 slots per image hold degenerate boxes the selection drops (empty after clipping), so RoIAlign and the box head do
    ~79 % of the nominal work; a trained model fills all of them. Here the CF-RPN's delta branch is set to predict every anchor itself
    (weights 0, bias 0.5: l = t = r = b = half the anchor), so nothing is dropped, and the same `passes` captured lanes are timed."""
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    p2 = dict(params)
    p2["proposal_generator.rpn_head.anchor_deltas.weight"] = torch.zeros_like(params["proposal_generator.rpn_head.anchor_deltas.weight"])
    p2["proposal_generator.rpn_head.anchor_deltas.bias"] = torch.full_like(params["proposal_generator.rpn_head.anchor_deltas.bias"], 0.5)
    eng = OpensetRCNNEngine(p2, dtype=tdt, device=dev)
    keep = {}
    eng.forward(images[:2], [(800, 1333)] * 2, keep=keep)
    real = int(keep["sel"]["counts"].sum()) // 2
    lane_images = [images]
    for i in range(1, passes):
        gi = torch.Generator().manual_seed(1000 + i)
        lane_images.append(torch.randint(0, 256, (batch, 3, 800, 1333), generator=gi, dtype=torch.uint8).to(dev))
    lanes, _ = make_lanes(eng, lane_images, image_hw, 1, 1)
    turn = [0]
    for _ in range(2 * passes):
        step_lanes(lanes, turn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_lanes(lanes, turn)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    torch.cuda.synchronize()
    g1 = lanes[0][0]
    t1 = time.perf_counter()
    for _ in range(steps):
        g1.replay()
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t1
    out = dict(images_per_sec=round(batch * steps / el, 2), ms_per_step=round(el / steps * 1e3, 3), passes_in_flight=passes, steps=steps,
               single_pass_ms=round(el1 / steps * 1e3, 3), real_proposals_per_image=real,
               note="CF-RPN deltas set to reproduce every anchor (no proposal is dropped: all 4273 slots per image real), same lanes and graphs as the "
                    "headline; the headline's weights leave ~21 % of the slots as padding")
    del lanes, eng
    torch.cuda.empty_cache()
    return out


def vendor_gemm_leg(dev, reps: int = 8, rounds: int = 3):
    """Calibration, not the product: what the vendor's hand-written GEMM (hipBLASLt behind torch.matmul: `Custom_Cijk_..._MT256x256x64_MI16x16x1`,
    the same 256 x 256 x 64 macro tile as conv_igemm64_kernel's) reaches on THIS box on the plain-GEMM restatement of the product's three largest
    MFMA-bound layers, beside the product's kernels on the same operands. A library GEMM has no im2col gather, no halo and no fused epilogue (it
    would need the 3 x 3 layers' im2col matrix materialised: 9x the activation bytes), so it is a reference for what the K loop can reach at the
    clock the part holds under a dense MFMA load -- the practical roof next to the 2.5 PFLOP/s datasheet figure."""
    import math
    from openset_rcnn_amd.host import ops
    g = torch.Generator().manual_seed(0)

    def best_us(fn):
        best = 1e30
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    rows = []
    for name, m, k, n, conv in (("roi_heads.box_head.fc1 (68368 x 12544 -> 1024)", 68368, 12544, 1024, None),
                                ("backbone.fpn_output2 (3x3, 256 -> 256, 16 x 200 x 336; as a GEMM: 1075200 x 2304 -> 256)", 16 * 200 * 336, 2304, 256, (16, 200, 336, 256, 3)),
                                ("backbone.fpn_output3 (16 x 100 x 168; as a GEMM: 268800 x 2304 -> 256)", 16 * 100 * 168, 2304, 256, (16, 100, 168, 256, 3))):
        a = (torch.randn(m, k, generator=g) * 0.5).half().to(dev)
        b = (torch.randn(n, k, generator=g) / math.sqrt(k)).half().to(dev)
        out = torch.empty(m, n, dtype=torch.float16, device=dev)
        bias = torch.zeros(n, device=dev)
        t_lib = best_us(lambda: torch.matmul(a, b.t(), out=out))
        if conv is None:
            t_own = best_us(lambda: ops.linear(a, b, bias, relu=True))
        else:
            nb, h, w, cin, kk = conv
            del a
            x = (torch.randn(nb, h, w, cin, generator=g) * 0.5).half().to(dev)
            wt = b.reshape(n, kk, kk, cin).contiguous()
            t_own = best_us(lambda: ops.conv2d(x, wt, bias, 1, kk // 2, relu=True))
            del x
        fl = 2.0 * m * k * n
        rows.append(dict(layer=name, hipblaslt_us=round(t_lib, 1), hipblaslt_tflops=round(fl / t_lib / 1e6, 1), product_us=round(t_own, 1),
                         product_tflops=round(fl / t_own / 1e6, 1), product_over_hipblaslt=round(t_lib / t_own, 3)))
        del b, out
        torch.cuda.empty_cache()
    return dict(rows=rows, timing=f"best of {rounds} rounds of {reps} back-to-back launches, HIP events, random operands",
                note="hipBLASLt's kernel on these shapes is its hand-written 256x256x64 MFMA-16x16 macro-tile kernel (stream-K variant on the first two: "
                     "scripts/exp_gemm_names.py under rocprofv3); the product's kernels compute the layer itself (implicit im2col + bias + ReLU)")


def parity_leg(tdt, dev, images, image_hw, flops_per_step: float, steps: int = 10):
    """What the benchmarked fp16 path gives up against fp32, and what fp32 costs. (1) agreement of the fast path's final
    detections with the engine's fp32 PARITY MODE (every tensor and product in fp32; tests/test_e2e_parity.py pins that mode to
    the fp32 oracle at 400 / 400) on the tests' 4 seeded 256x384 images: same class, IoU >= 0.99, |score difference| <= 1e-2,
    matched one to one. (2) images/s of the parity mode at the benchmark batch, measured like the headline's single_pass figure:
    the captured pass replayed `steps` times, with its roofline against the 157.3 TFLOP/s fp32 matrix peak. (3) `config5_mode`:
    BASELINE.json config 5 / SURVEY.md section 7 taken literally -- fp16 MFMA operands in the backbone, FPN and CF-RPN head, fp32
    storage AND fp32 products from the RoIAlign output on (the box head's FC layers on the exact-f32 matrix instruction) -- its
    agreement with the parity mode and its images/s: the price of staying inside what north_star sanctions, as a number."""
    from openset_rcnn_amd.host.agreement import detection_agreement
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params, with_known_unknown_mix
    n, h, w = 4, 256, 384
    g = torch.Generator().manual_seed(2024)
    small = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).to(dev)
    sizes = [(h, w), (h, w), (h - 16, w - 40), (h - 6, w)]
    base = random_params(0)
    keep = {}
    e32 = OpensetRCNNEngine(base, dtype=torch.float32, device=dev)
    e32.forward(small, sizes, keep=keep)
    cnt = keep["cnt1"].cpu()
    emb = torch.cat([keep["emb"].view(n, 1000, -1)[i, :int(cnt[i])] for i in range(n)]).cpu()
    params = with_known_unknown_mix(base, emb)
    del e32, keep

    def dets(dtype, **kw):
        eng = OpensetRCNNEngine(params, dtype=dtype, device=dev, **kw)
        out = eng.forward(small, sizes)
        torch.cuda.synchronize()
        return eng, [(d["pred_boxes"], d["scores"], d["pred_classes"]) for d in eng.to_instances(out, n)]

    e32, ref = dets(torch.float32)
    _, got = dets(tdt)
    agree = detection_agreement(got, ref)
    loose = detection_agreement(got, ref, iou_thr=0.9, score_tol=5e-2)
    # parity-mode throughput at the benchmark batch: captured pass, `steps` replays
    rate32, dt32 = graph_rate(e32, images, image_hw, steps)
    del e32
    torch.cuda.empty_cache()
    tf32 = flops_per_step / dt32 / 1e12
    out = dict(fast_mode_agreement=round(agree["fraction"], 4), matched=agree["matched"], reference_detections=agree["reference_detections"],
               returned_detections=agree["returned_detections"], same_class_on_matches=agree["same_class"],
               agreement_at_iou_0p9_dscore_5e_2=round(loose["fraction"], 4),
               criterion="same class, IoU >= 0.99, |score difference| <= 1e-2, one-to-one; reference = this engine's fp32 parity mode "
                         "(fp32 storage and products in every layer), which tests/test_e2e_parity.py holds to the fp32 oracle",
               images="4 seeded 256x384 uint8 images (seed 2024), random-init weights with a calibrated known / unknown mix",
               fast_mode_storage="fp16 activations from the stem to h1 (RoIAlign output and FC1 output included), fp32 accumulation; fp32 from the box features on",
               parity_mode_images_per_sec=round(rate32, 2), parity_mode_ms_per_step=round(dt32 * 1e3, 2), parity_mode_steps=steps,
               parity_mode_roofline=dict(bound="mfma", achieved=round(tf32, 2), peak=F32_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                                         frac=round(tf32 / F32_MATRIX_PEAK_TFLOPS, 4), flops_per_step=flops_per_step,
                                         note="algorithmic FLOPs of the MFMA launches (real proposal rows) over the whole captured pass's time"),
               parity_mode_note=f"fp32 engine (osr_conv_f32.hip, exact-f32 MFMA), batch {images.shape[0]} at 800x1333, captured pass (hipGraph) replayed "
                                f"{steps} times on one stream after 2 warm-up replays")
    try:
        e5, got5 = dets(tdt, fp32_points=("pooled", "h1"))
        a5 = detection_agreement(got5, ref)
        rate5, dt5 = graph_rate(e5, images, image_hw, steps)
        del e5
        out["config5_mode"] = dict(images_per_sec=round(rate5, 2), ms_per_step=round(dt5 * 1e3, 3), steps=steps,
                                   agreement_with_fp32=round(a5["fraction"], 4), matched=a5["matched"], reference_detections=a5["reference_detections"],
                                   storage="fp16 MFMA operands and storage in the backbone, FPN and CF-RPN head; fp32 storage and exact-fp32 products from the RoIAlign "
                                           "output on (pooled rows, FC1, FC2, predictor, PLN, classifier) -- BASELINE.json config 5 / SURVEY.md section 7",
                                   note="captured pass, one lane; FC1 / FC2 run on v_mfma_f32_32x32x2_f32 at 1/16 of the fp16 matrix rate")
    except Exception as e:  # noqa: BLE001
        out["config5_mode"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    # (4) `trained`: the same three precision modes scored by the open-set VOC evaluator (AP@K, pascal_voc_evaluation.py:192) on a checkpoint
    # TRAINED here (tests/trained_parity.py: synthetic learnable VOC-layout set, 1000 iterations of the HIP training step): what the
    # fast mode costs in accuracy where scores are spread by training, not decided by the near-ties of random-init weights
    try:
        from tests.trained_parity import run as trained_run
        t = trained_run(str(dev))
        out["trained"] = dict(APk_fp32=t["APk_fp32"], APk_fast=t["APk_fast"], APk_config5=t["APk_config5"],
                              agreement_fast_vs_fp32=t["agreement_fast_vs_fp32"], agreement_config5_vs_fp32=t["agreement_config5_vs_fp32"],
                              detections_fp32=t["detections_fp32"], known_detections_fp32=t["known_detections_fp32"],
                              metrics_fp32=t["metrics_fp32"], metrics_fast=t["metrics_fast"], train=t["train"], delta_vs_fp32=t.get("delta_vs_fp32"),
                              hard_split={k: t["hard"][k] for k in ("APk_fp32", "APk_fast", "APk_config5", "agreement_fast_vs_fp32", "agreement_config5_vs_fp32",
                                                                    "known_detections_fp32", "ground_truth", "metrics_fp32", "delta_vs_fp32")}
                              if isinstance(t.get("hard"), dict) else None,
                              hard_split_note="the same checkpoint on 128 crowded / occluded test images (5-10 objects, IoU up to 0.35 between them, 40 % unknown kinds of "
                                              "which half share a known class's colour): AP@K, WI and A-OSE with hundreds of decision points (tests/trained_parity.py)",
                              note="AP@K of host/evaluation.py (restatement of openset_rcnn/evaluation/pascal_voc_evaluation.py) on 64 synthetic test images with "
                                   "known and unknown objects; the released checkpoint / VOC-COCO images needed for README.md:98's 59.12 are not reachable offline")
    except Exception as e:  # noqa: BLE001
        out["trained"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    return out


def make_lanes(eng, lane_images, image_hw, streams_per_pass: int = 1, concurrency_hint: int = 1):
    """The headline's schedule: one captured pass (hipGraph) per lane, each lane with its own batch of images, its own activation /
    output buffers (the capture's private pool) and its own HIP stream. Returns ([(graph, outputs, stream, images)], GB of one lane).
    The tile model's concurrency hint = how many launch streams share the GPU (lanes x micro-batch streams)."""
    from openset_rcnn_amd.host import ops as _ops
    dev = eng.device
    lanes, lane_gb = [], None
    with _ops.concurrent_streams(concurrency_hint):
        for li, imgs in enumerate(lane_images):
            torch.cuda.synchronize()
            mem0 = torch.cuda.memory_reserved(dev)
            g_, o_ = eng.capture(imgs, image_hw, 800, 1344, streams_per_pass)
            torch.cuda.synchronize()
            if li == 0:
                lane_gb = (torch.cuda.memory_reserved(dev) - mem0 + imgs.numel()) / 1e9  # one lane: its graph's private pool + its images
            lanes.append((g_, o_, torch.cuda.Stream(device=dev), imgs))
    for g_, _, st_, _ in lanes:  # first replay of every lane's graph (one-time upload of the executable graph), untimed and outside the W warm-up steps
        with torch.cuda.stream(st_):
            g_.replay()
    torch.cuda.synchronize()
    return lanes, lane_gb


def step_lanes(lanes, turn):
    """One step of the headline loop: pass i is launched on lane i % P without waiting for pass i - 1 (its own buffers, its own
    stream); a lane's next pass queues behind its previous one. The caller ends the timed region with a device-wide synchronize."""
    if len(lanes) == 1:
        lanes[0][0].replay()
        return lanes[0][1]
    g_, o_, st_, _ = lanes[turn[0] % len(lanes)]
    turn[0] += 1
    with torch.cuda.stream(st_):
        g_.replay()
    return o_


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (config: 16)")
    ap.add_argument("--dtype", default="f16", choices=["f16"],
                    help="storage dtype of the benchmarked path. (bf16 kernels exist for the training step; as an inference storage type "
                         "bf16 reproduces a quarter of the fp32 detections, tests/test_e2e_parity.py, so it is not offered here)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the config-3 train-step leg")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure HBM traffic with rocprofv3 child passes (quote the committed profile)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg (host frames -> upload -> device resize -> pass)")
    ap.add_argument("--no-parity", action="store_true", help="skip the fast-mode-vs-fp32 agreement / parity-mode throughput leg")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--dense-rpn-bwd", action="store_true", help="train step: the CF-RPN head's backward over every anchor row (rounds 1-3), for A/B")
    ap.add_argument("--train-only", action="store_true", help="(internal) run only the train-step leg (and config 4's) and print its object")
    ap.add_argument("--no-config4", action="store_true", help="with --train-only: skip the config-4 leg (profiles of the config-3 train step alone)")
    ap.add_argument("--streams", type=int, default=1, help="micro-batch streams inside one pass (1 = the pass is one stream of launches)")
    ap.add_argument("--passes-in-flight", type=int, default=4,
                    help="hipGraph mode: consecutive passes (steps) alternate over this many lanes, each with its own images, buffers and stream; "
                         "1 = one pass at a time")
    ap.add_argument("--lane-hint", type=int, default=1, help="1: pass the number of lanes to the conv tile model as its concurrency hint")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-chain", action="store_true", help="(A/B) run res3's conv2 and conv3 as separate launches instead of osr_conv2d_chain_fwd")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage breakdown to stderr")
    ap.add_argument("--layers", action="store_true", help="also print every MFMA launch (time, TFLOP/s, GB/s) to stderr")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)  # (before any GPU call: this process only relays the job)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (never set by the driver): OSR_DIST_BACKEND=gloo lets several ranks share one GPU on a 1-GPU box
    backend = os.environ.get("OSR_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    out_stream = sys.stdout
    if world > 1:
        # stdout carries exactly ONE line, the JSON record: native libraries (c10d's "[Gloo] Rank ..." banners, RCCL's notices) print to
        # file descriptor 1 behind Python's back, so descriptor 1 is pointed at stderr and the record goes to a duplicate of the original
        sys.stdout.flush()
        out_stream = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import datetime
        import torch.distributed as dist
        # Collective timeout 600 s, stated rather than inherited: the longest stretch in which ranks 1..N-1 sit in a collective while
        # rank 0 works alone is the host-only CPU baseline at the very end (~20-30 s, behind the last collective: they wait in
        # destroy_process_group, not in a collective), and the train-step / config-4 legs carry their own 300-s watchdogs (below), which
        # fire first. DESIGN.md section 6 has the arithmetic of the N = 8 run against the driver's limit.
        pg_timeout = datetime.timedelta(seconds=600)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)  # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend, timeout=pg_timeout)

    pkg = ge.load_package()
    pkg._lib.load()
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params, with_known_unknown_mix

    tdt = torch.float16
    dev = f"cuda:{local_rank}"
    g = torch.Generator().manual_seed(1234 + rank)  # each rank has its own shard of images
    images = torch.randint(0, 256, (args.batch, 3, 800, 1333), generator=g, dtype=torch.uint8).to(dev)
    image_hw = torch.tensor([(800, 1333)] * args.batch, dtype=torch.int32, device=dev)
    # Synthetic weights, identical on every rank (data parallel). Random prototypes alone make every detection "unknown" and
    # leave the known-class leg (softmax over <= 20 000 candidates per image + per-class NMS) idle: calibrate the PLN encoder
    # bias on rank 0's images so that about half of the first-stage detections fall within UNK_THR (0.23, the yaml's value)
    # of a prototype (weights.with_known_unknown_mix), un-timed, then build the engine that is measured.
    params = random_params(0)
    cal = OpensetRCNNEngine(params, dtype=tdt, device=dev)
    keep = {}
    # (the calibration images are the same on every rank -- seed 1234, rank 0's stream -- so that every rank derives the same
    # bias without a collective: identical weights by construction)
    cal_images = torch.randint(0, 256, (4, 3, 800, 1333), generator=torch.Generator().manual_seed(1234), dtype=torch.uint8).to(dev)
    cal.forward_device(cal_images, image_hw[:4], 800, 1344, keep)
    cnt = keep["cnt1"].cpu()
    emb = torch.cat([keep["emb"].view(4, -1, keep["emb"].shape[-1])[i, :int(cnt[i])] for i in range(4)]).cpu()
    del cal_images
    params = with_known_unknown_mix(params, emb)
    del cal, keep
    if args.train_only:  # child of the single-GPU run (see train_step_child): the train-step leg in a process of its own
        obj = train_step_leg(params, tdt, dev, images, image_hw, args.train_steps, 2, dense_rpn_bwd=args.dense_rpn_bwd, no_chain=args.no_chain)
        # BASELINE.json config 4's per-GPU workload in the same fresh process (run behind the four-lane inference loop in the
        # parent it inherited that process's hardware-queue rotation, the very thing this child exists to avoid for the train step)
        del images
        torch.cuda.empty_cache()
        if not args.no_config4:
            try:
                obj["config4"] = config4_leg(tdt, dev, max(args.train_steps, 8), 4)
            except Exception as e:  # noqa: BLE001
                obj["config4"] = {"error": repr(e)[:400]}
        print(json.dumps(obj), flush=True)
        return 0
    eng = OpensetRCNNEngine(params, dtype=tdt, device=dev)
    if args.no_chain:
        eng.chain_res3 = False

    def step():
        if args.streams > 1:
            return eng.forward_device_streams(images, image_hw, 800, 1344, args.streams)
        return eng.forward_device(images, image_hw, 800, 1344)

    lanes, lane_gb = [], None
    if args.graph:
        npass = max(1, args.passes_in_flight)
        lane_images = [images]
        for li in range(1, npass):  # every lane has its own batch of images (and, through its capture, its own activations and outputs)
            gi = torch.Generator().manual_seed(1234 + rank + 1000 * li)
            lane_images.append(torch.randint(0, 256, (args.batch, 3, 800, 1333), generator=gi, dtype=torch.uint8).to(dev))
        lanes, lane_gb = make_lanes(eng, lane_images, image_hw, args.streams, npass if args.lane_hint else 1)
        graph, gout = lanes[0][0], lanes[0][1]
        turn = [0]

        def step():  # noqa: F811  one hipGraph launch replays the whole pass
            return step_lanes(lanes, turn)

    def timed(fn, warm, steps):
        """The contract's timed region: W untimed steps, then exactly K steps between barrier + synchronize, MAX over the ranks."""
        for _ in range(warm):
            o = fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        timed.local = el  # this rank's own time (the returned value is the maximum over the ranks)
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=eng.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, o

    elapsed, out = timed(step, args.warmup, args.steps)
    props = torch.cuda.get_device_properties(local_rank)
    identity = rank_identity(dist, rank, world, timed.local, args.steps, args.batch, str(getattr(props, "uuid", "unknown")), props.name)
    n_det = int(out[3].sum().item())
    n_unknown = int((out[2] == eng.cfg["unknown_id"]).logical_and(torch.arange(out[2].shape[1], device=out[2].device)[None, :] < out[3][:, None]).sum().item())

    # ---- one pass at a time: the same K steps on lane 0 alone (its graph, its stream; 16 images in flight, as in config 2) ----
    single_pass = None
    if args.graph and len(lanes) > 1:
        g0, o0, st0, _ = lanes[0]

        def step_single():
            with torch.cuda.stream(st0):
                g0.replay()
            return o0
        el1, _ = timed(step_single, 1, args.steps)
        single_pass = dict(ms_per_step=round(el1 / args.steps * 1e3, 3), images_per_sec=round(world * args.batch * args.steps / el1, 2), passes_in_flight=1,
                           note="the same captured pass, one lane, one stream: every step starts when the previous one has finished on the device queue "
                                "(ms_per_step is then the latency of a pass)")

    # ---- rooflines of the two kernel groups, measured live with HIP events on the launch stream ----
    # (the single-stream full-batch pass picks other tile configurations than the micro-batched one: run it once untimed so
    # that no launch of the bracketed pass is the first use of its kernel)
    eng.forward_device(images, image_hw, 800, 1344)
    # (no host sync between the two passes: the attribution pass is enqueued right behind the untimed one, so its first launches -- the
    # stem -- do not start on a GPU that has idled and clocked down while the host prepared the pass; the events are read after the sync below)
    eng.profile, eng.profile_hbm = [], []
    eng.forward_device(images, image_hw, 800, 1344)  # attribution pass: one stream, so each launch can be bracketed
    torch.cuda.synchronize()
    prof, prof_hbm = eng.resolve_profile()  # (device-side counts are read only now: no host sync inside the bracketed pass)
    eng.profile = eng.profile_hbm = None
    # ---- the same loop fed from host memory: upload + on-device resize overlapped with the passes in flight (rank 0, one GPU).
    # AFTER the attribution pass: this leg overwrites the lanes' input batches with its own frames, and the proposal distribution
    # (hence RoIAlign's time: 1.2 vs 1.6 ms) follows the image content ----
    pcie = None
    if args.graph and world == 1 and not args.no_pcie:
        try:
            pcie = pcie_leg(lanes, dev, args.batch, args.steps)
        except Exception as e:  # noqa: BLE001  (reported in the line; the headline stands)
            pcie = {"error": repr(e)[:300]}

    mfma_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1, _, _ in prof)
    mfma_flops = sum(f for _, f, _, _, _, _ in prof)          # algorithmic: real rows of the proposal lists only
    mfma_flops_nominal = sum(f for _, _, _, _, _, f in prof)  # every row of the fixed-capacity lists, padding included
    achieved = mfma_flops / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    # HBM-side traffic of the same kernel family: PMC counters cannot be read from inside the process, so the value is the
    # one collected with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2 per the gfx950 guide) on this
    # very command by scripts/profile_round.sh and committed under profiles/ (newest round first); null when absent.
    traffic, traffic_source, hbm_traffic = None, None, None
    profile_ms, profile_tag = None, None  # the same family's time per step in the newest committed rocprofv3 --stats summary
    import glob
    for tfile in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_times_and_traffic.json")), reverse=True):  # newest round / letter first
        tag = os.path.basename(tfile)[: -len("_kernel_times_and_traffic.json")]
        if os.path.exists(tfile):
            with open(tfile) as fh:
                blob = json.load(fh)
            cf = blob.get("conv_family", {})
            traffic = round(cf.get("fetch_GB_per_step_x2corrected", 0.0) + cf.get("write_GB_per_step", 0.0), 2)
            ra = blob.get("roi_align", {})
            if ra:
                hbm_traffic = round(ra.get("fetch_GB_per_step_x2corrected", 0.0) + ra.get("write_GB_per_step", 0.0), 2)
            traffic_source = f"profiles/{tag}_kernel_times_and_traffic.json (scripts/profile_round.sh: --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            profile_ms, profile_tag = cf.get("ms_per_step"), tag
            break
    if world == 1 and rank == 0 and not args.no_pmc:
        live = measure_traffic()
        if live is not None:
            traffic, hbm_traffic, traffic_source = live
    algo_bytes = sum(nb for _, _, _, _, nb, _ in prof)
    real_rois = next((info["real_rois"] for name, _, _, _, info in prof_hbm if name == "roi_align" and info), None)
    list_rows = next((info["list_rows"] for name, _, _, _, info in prof_hbm if name == "roi_align" and info), None)
    roofline = dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic,
                    traffic_unit="GB of HBM traffic per step of the same kernel family (all its launches)", traffic_source=traffic_source,
                    algorithmic_GB_per_step=round(algo_bytes / 1e9, 2),
                    kernel="MFMA implicit-GEMM conv / FC family (conv_igemm64_kernel + the fused bottleneck and stem kernels)", launches_per_step=len(prof),
                    flops_per_step=mfma_flops, kernel_ms_per_step=round(mfma_ms, 3),
                    flops_note="algorithmic: 2*M*K*N of every layer definition with M = real rows -- the box head's FC layers are credited for the "
                               "proposals that exist, not for the padding rows of the fixed-capacity lists",
                    flops_per_step_nominal=mfma_flops_nominal, frac_nominal=round(mfma_flops_nominal / (mfma_ms * 1e-3) / 1e12 / peak, 4) if mfma_ms > 0 else 0.0,
                    real_proposals=real_rois, proposal_list_rows=list_rows)
    if profile_ms:
        # the committed profiler summary of the same command gives the family a few percent more time than the HIP events of this run
        # (another box, the tracer's per-dispatch cost): both are printed so that the fraction reads as a range, not a point
        roofline["kernel_ms_per_step_profile"] = profile_ms
        roofline["frac_profile"] = round(mfma_flops / (profile_ms * 1e-3) / 1e12 / peak, 4)
        roofline["profile_source"] = f"profiles/{profile_tag}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --streams 1 --no-graph`)"
    # the two figures north_star's targets are worded in: (1) the R50 + FPN backbone alone (its MFMA launches' FLOPs over their
    # own time); (2) SURVEY.md 8d's dense-conv group formula, 392.9 GFLOP (ResNet-50 + FPN + CF-RPN head at 800 x 1344) x img/s
    bb = [(f, e0.elapsed_time(e1)) for name, f, e0, e1, _, _ in prof if name.startswith("backbone.")]
    bb_flops, bb_ms = sum(f for f, _ in bb), sum(t for _, t in bb)
    roofline["backbone_frac"] = round(bb_flops / (bb_ms * 1e-3) / 1e12 / peak, 4) if bb_ms > 0 else 0.0
    roofline["backbone"] = dict(flops_per_step=bb_flops, kernel_ms_per_step=round(bb_ms, 3), launches=len(bb),
                                scope="ResNet-50 (stem .. res5) + FPN laterals / output convolutions: north_star's '>= 40 % of MFMA peak on the R50 backbone'")
    per_gpu_rate = args.batch * args.steps / elapsed
    roofline["survey_8d_frac"] = round(392.9e9 * per_gpu_rate / 1e12 / peak, 4)
    roofline["survey_8d_note"] = "SURVEY.md 8d: 392.9 GFLOP of dense convolutions per image (ResNet-50 + FPN + CF-RPN head) x images/s per GPU / peak; target 0.40 = 2545 img/s"
    # HBM group: RoIAlign + proposal selection + the three NMS passes. NMS is not HBM-bound (n <= a few thousand
    # boxes per segment, a serial greedy scan): its bytes are folded in, so the aggregate is dominated by RoIAlign, and its time
    # and upper bound of IoU pairs (sum over segments of n^2 / 2) are reported per pass.
    hbm_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1, _ in prof_hbm)
    hbm_bytes = sum(nb for _, nb, _, _, _ in prof_hbm)
    kernels = []
    for name, nb, e0, e1, info in prof_hbm:
        ms_ = e0.elapsed_time(e1)
        k = dict(kernel=name, ms=round(ms_, 4), algorithmic_GB=round(nb / 1e9, 4), GBps=round(nb / ms_ / 1e6, 1), frac_of_hbm_peak=round(nb / ms_ / 1e6 / HBM_PEAK_GBS, 4))
        if name == "roi_align" and info:
            k.update(real_rois=info["real_rois"], list_rows=info["list_rows"], nominal_GB=round(info["nominal_bytes"] / 1e9, 4),
                     frac_of_hbm_peak_nominal=round(info["nominal_bytes"] / ms_ / 1e6 / HBM_PEAK_GBS, 4))
        if name.startswith("nms_topk") and info is not None:
            seg = info.detach().cpu().double()
            pairs = float((seg * seg / 2).sum())
            k.update(boxes=int(seg.sum()), max_segment=int(seg.max()), iou_pairs_upper_bound=pairs, Gpairs_per_s=round(pairs / ms_ / 1e6, 3))
        kernels.append(k)
    hbm_gbs = hbm_bytes / hbm_ms / 1e6 if hbm_ms > 0 else 0.0
    roofline_hbm = dict(bound="hbm", group="RoIAlign + proposal selection + NMS (SURVEY.md 8d)", achieved=round(hbm_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(hbm_gbs / HBM_PEAK_GBS, 4), algorithmic_GB_per_step=round(hbm_bytes / 1e9, 3), kernel_ms_per_step=round(hbm_ms, 3),
                        traffic=hbm_traffic, traffic_unit="GB of HBM traffic per step of the RoIAlign kernel (FETCH_SIZE x2 + WRITE_SIZE)",
                        traffic_source=traffic_source, bytes_note="RoIAlign: the pyramid once + the pooled rows of the real RoIs + their boxes",
                        kernels=kernels)
    if args.layers and rank == 0:
        for name, f, e0, e1, nb, _ in prof:
            ms_ = e0.elapsed_time(e1)
            print(f"  {name:48s} {ms_ * 1e3:8.1f} us {f / ms_ / 1e9:8.1f} TFLOP/s {nb / ms_ / 1e6:8.1f} GB/s", file=sys.stderr)
    if args.stages and rank == 0:
        agg = {}
        for name, f, e0, e1, _, _ in prof:
            key = name.split(".")[1] if name.startswith("backbone.bottom_up") else name.split(".")[0] + "." + name.split(".")[1]
            key = name.split(".")[2] if name.startswith("backbone.bottom_up") else key
            a = agg.setdefault(key, [0.0, 0.0])
            a[0] += e0.elapsed_time(e1)
            a[1] += f
        for k, (ms, f) in agg.items():
            print(f"  {k:40s} {ms:8.3f} ms  {f / ms / 1e9 if ms else 0:8.1f} TFLOP/s", file=sys.stderr)
        for k in kernels:
            print(f"  {k['kernel']:40s} {k['ms']:8.3f} ms  {k['GBps']:8.1f} GB/s", file=sys.stderr)
        print(f"  MFMA kernels total {mfma_ms:.3f} ms, HBM group {hbm_ms:.3f} ms of {elapsed / args.steps * 1e3:.3f} ms/step", file=sys.stderr)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": METRIC, "value": round(world * args.batch * args.steps / elapsed, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "VOC-COCO openset_rcnn_R50_FPN_128k.yaml, inference-only, 3x800x1333 uint8 BGR -> padded 800x1344, "
                                   "1000 proposals/level (4273/img), 1000 dets/img, 50+50 final",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world} (images sharded, no collective)",
                       "weights": "random-init (seed 0), FrozenBN folded; PLN encoder bias calibrated for a known/unknown mix at UNK_THR 0.23",
                       "detections_last_step": n_det, "unknown_detections_last_step": n_unknown, "known_detections_last_step": n_det - n_unknown,
                       "micro_batch_streams": args.streams, "hipgraph": bool(args.graph),
                       "ranks": identity["ranks"], "gpu_uuids": identity["gpu_uuids"], "distinct_gpus": identity["distinct_gpus"],
                       "device_names": identity["device_names"], "per_rank_images_per_sec": identity["per_rank_images_per_sec"],
                       "per_rank_ms_per_step": identity["per_rank_ms_per_step"],
                       "passes_in_flight": max(1, len(lanes)),
                       "images_in_flight": args.batch * max(1, len(lanes)),
                       "lane_memory_GB": round(lane_gb, 2) if lane_gb is not None else None,
                       "value_note": "throughput with passes_in_flight batches of 16 resident (each lane: own images, buffers, stream); the one-batch-at-a-time "
                                     "figure is `single_pass`; the fp16 path's agreement with fp32 is `parity`"},
            "roofline": roofline, "roofline_hbm": roofline_hbm,
        }
        if single_pass is not None:
            line["single_pass"] = single_pass
        if pcie is not None:
            line["pcie_inclusive"] = pcie
    else:
        line = None

    del eng, out
    if args.graph:
        del graph, gout, lanes, lane_images
        lanes = []
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_parity:
        try:
            line["parity"] = parity_leg(tdt, dev, images, image_hw, mfma_flops)
        except Exception as e:  # noqa: BLE001  (reported in the line; the headline stands)
            line["parity"] = {"error": repr(e)[:400]}
        torch.cuda.empty_cache()
        try:
            line["roofline"]["vendor_gemm"] = vendor_gemm_leg(dev)
        except Exception as e:  # noqa: BLE001
            line["roofline"]["vendor_gemm"] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
        if args.graph:
            try:
                line["full_lists"] = full_lists_leg(params, tdt, dev, images, image_hw, args.batch, max(1, args.passes_in_flight))
            except Exception as e:  # noqa: BLE001
                line["full_lists"] = {"error": repr(e)[:400]}
    # ---- train step leg: every rank takes part (the gradient all-reduce is a collective). With several ranks a watchdog makes
    # sure the headline line is printed even if that collective path hangs on the node -- and then the job FAILS (exit code 3).
    rc = 0
    if world == 1 and not args.no_train_step:
        ts = train_step_child(args)
        line["config4"] = ts.pop("config4", {"error": "the train-step child did not report it"}) if isinstance(ts, dict) else {"error": "no train-step child"}
        line["train_step"] = ts
    elif not args.no_train_step:
        import threading

        def give_up():
            if rank == 0:
                line.setdefault("train_step", {"error": "the multi-rank train step did not finish within 300 s (hung collective?); headline unaffected, exit code 3"})
                line.setdefault("config4", {"error": "not reached, or did not finish within its own 300 s (hung collective?); exit code 3"})
                print(json.dumps(line), file=out_stream, flush=True)
            os._exit(3)  # every rank: a hung collective must not read as a successful run (no restart, no exec: the GPU is initialised)
        guard = threading.Timer(300.0, give_up)
        guard.daemon = True
        guard.start()
        try:
            ts = train_step_leg(params, tdt, dev, images, image_hw, args.train_steps, 2, dist, rank, world)
        except Exception as e:  # noqa: BLE001  (reported in the line, the headline stands; the job fails)
            ts = {"error": repr(e)[:400]}
            rc = 3
        if rank == 0:
            line["train_step"] = ts
        # BASELINE.json config 4 (GraspNet heads, 1280x720 frames, batch 8 per GPU): every rank, with a timer of ITS OWN -- a slow box that
        # spent most of the first 300 s on a good train-step result must not trip the guard in the middle of this leg
        guard.cancel()
        guard = threading.Timer(300.0, give_up)
        guard.daemon = True
        guard.start()
        try:
            c4 = config4_leg(tdt, dev, max(args.train_steps, 8), 4, dist, rank, world)
        except Exception as e:  # noqa: BLE001
            c4 = {"error": repr(e)[:400]}
            rc = 3
        guard.cancel()
        if rank == 0:
            line["config4"] = c4
    if rank == 0:
        if not args.no_cpu_baseline:  # host-only: rank 0 times it whatever N is (the other ranks wait at the closing barrier)
            line["cpu_baseline"] = cpu_baseline(params, args.cpu_batch, one_thread=world == 1)
        print(json.dumps(line), file=out_stream, flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
