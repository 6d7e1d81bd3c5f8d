#!/usr/bin/env python3
"""Command-line driver with the reference's train.py interface (/root/reference/train.py:211-306) on top of the HIP path.

Same flags and meaning: --config-file, --resume, --eval-only, --num-gpus, --resume_test, --test_iter, --eval_type,
--opendet-benchmark and trailing `KEY VALUE` config overrides; it reads the reference's `configs/*.yaml` unchanged.
What differs, by construction of this build:
  * one process per GPU. `--num-gpus N` (N > 1) starts `python -m torch.distributed.run --nproc-per-node N` as a child
    process BEFORE anything touches the GPU and exits with its code; under torchrun (WORLD_SIZE set) it joins the job.
    Rendezvous is always 127.0.0.1 (single node; --num-machines / --machine-rank / --dist-url are accepted and rejected
    when they ask for more than one machine).
  * training records no autograd graph, yet the loop body of train.py:135-147 runs as written: `model(data)` returns the losses,
    `losses.backward()` runs the explicit HIP backward (one autograd Function stands for the whole step), `optimizer.step()`
    finishes the bucketed RCCL all-reduce that the backward started and applies SGD (host/solver.py).
  * datasets live under $DETECTRON2_DATASETS (default ./datasets) in the reference's layout (datasets/README of the
    reference: voc_coco/{Annotations,ImageSets,JPEGImages}, graspnet_os/{annotations,images}).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import subprocess
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
logger = logging.getLogger("openset_rcnn")


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    ap.add_argument("--resume", action="store_true", help="resume from the last checkpoint in OUTPUT_DIR")
    ap.add_argument("--eval-only", action="store_true", help="perform evaluation only")
    ap.add_argument("--num-gpus", type=int, default=1, help="number of gpus (one process each)")
    ap.add_argument("--num-machines", type=int, default=1)
    ap.add_argument("--machine-rank", type=int, default=0)
    ap.add_argument("--dist-url", default="tcp://127.0.0.1:29533")
    ap.add_argument("--resume_test", action="store_true", help="evaluate the detections a previous run left in OUTPUT_DIR")
    ap.add_argument("--test_iter", default=0, type=int, help="with --resume_test: iteration whose detections to score, 0 for Final")
    ap.add_argument("--eval_type", default="openset", type=str)
    ap.add_argument("--opendet-benchmark", action="store_true", help="unknown class id 80 and the VOC-COCO class list of OpenDet")
    ap.add_argument("--test-batch", type=int, default=1,
                    help="images per inference launch. 1 = the reference's evaluation (each image padded to a multiple of 32 on its own); larger "
                         "batches are faster but pad every image to the largest one of its batch, which changes features near the padded border")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"],
                    help="storage dtype of activations / MFMA operands; f32 = parity mode (inference only: the reference's own arithmetic, ~10x slower)")
    ap.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    args = ap.parse_args(argv)
    if args.resume_test and args.opendet_benchmark:
        ap.error("opendet benchmark does not support resume_test")
    if args.test_iter and args.opendet_benchmark:
        ap.error("opendet benchmark does not support test_iter")
    if args.num_machines != 1 or args.machine_rank != 0:
        ap.error("single node only: one process per GPU over RCCL/xGMI inside one machine")
    if args.eval_type != "openset":
        ap.error("--eval_type: only 'openset' is implemented (the evaluators here are the open-set ones the reference's yaml files use)")
    return args


def relaunch_under_torchrun(args) -> int:
    """--num-gpus N without a launcher: become the parent of a torchrun job (nothing has touched the GPU yet)."""
    port = args.dist_url.rsplit(":", 1)[-1] if ":" in args.dist_url else "29533"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.num_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def setup(args):
    """train.py:167-182: defaults + added keys + yaml + overrides + benchmark flag, frozen; OUTPUT_DIR created; logger."""
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list([o for o in (args.opts or []) if o != "--"])
    if args.opendet_benchmark:
        cfg.OPENDET_BENCHMARK = True
    cfg.freeze()
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
        with open(os.path.join(cfg.OUTPUT_DIR, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1, default=str)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="[%(asctime)s] %(name)s %(levelname)s: %(message)s")
    return cfg


def register_datasets(cfg):
    """openset_rcnn/data/custom.py:48-52 + detectron2's built-in PASCAL VOC names, under $DETECTRON2_DATASETS; a second call
    in one process (tests) re-points the catalog at the current root."""
    from openset_rcnn_amd.host import datasets as D
    root = os.path.expanduser(os.environ.get("DETECTRON2_DATASETS", "datasets"))
    D.DatasetCatalog.clear()
    D.register_graspnet_os(root)
    D.register_opendet_voc_coco(root)
    D.register_builtin_pascal_voc(root)
    return D


def class_id_for(cfg, D):
    """GraspNet: the sorted contiguous ids of the known categories (prototype_learning_network.py:80-86); VOC-COCO: None."""
    names = list(cfg.DATASETS.TRAIN) + list(cfg.DATASETS.TEST)
    g = [n for n in names if "graspnet" in n]
    if not g:
        return None
    D.DatasetCatalog[g[0]]()  # fills thing_classes
    return D.graspnet_class_map(D.MetadataCatalog.get(g[0]).thing_classes)


def do_test(cfg, args, model, D, iteration: int = 0):
    """train.py:81-106."""
    from openset_rcnn_amd.host.data import DatasetMapper, build_detection_test_loader
    from openset_rcnn_amd.host.evaluation import inference_on_dataset
    results = OrderedDict()
    for name in cfg.DATASETS.TEST:
        folder = os.path.join(cfg.OUTPUT_DIR, "inference", name, str(iteration) if iteration else "Final")
        evaluator = D.get_evaluator(cfg, name, folder)
        if args.resume_test:
            if "resume" not in evaluator.evaluate.__code__.co_varnames:
                raise NotImplementedError(f"--resume_test: the evaluator of {name} keeps no detections file (COCO-style datasets only, as in the reference)")
            res = evaluator.evaluate(resume=True)
        else:
            dicts = D.DatasetCatalog[name]()
            loader = build_detection_test_loader(dicts, DatasetMapper(cfg, is_train=False), batch_size=args.test_batch)
            # the loader already yields this rank's shard only: no second split inside inference_on_dataset
            res = inference_on_dataset(model, loader, evaluator, rank=0, world=1)
        results[name] = res
        if res is not None:
            logger.info("Evaluation results for %s:", name)
            flat = res.get("bbox", res) if isinstance(res, dict) else res
            logger.info("  %s", ", ".join(f"{k}={v}" for k, v in flat.items()))
    return list(results.values())[0] if len(results) == 1 else results


def checkpoint_path(cfg, iteration=None):
    return os.path.join(cfg.OUTPUT_DIR, "model_final.pth" if iteration is None else f"model_{iteration:07d}.pth")


def save_checkpoint(cfg, model, trainer, iteration: int, final: bool = False, write: bool = True):
    """Never writes a state that an overflowed update may have left half-judged: waits for the overflow verdict of the last
    update, and validates the masters (an fp32 inf / NaN in the weights would otherwise only surface at the next iteration)."""
    import torch
    trainer.poll_overflow(wait=True)
    bad = [k for k, v in trainer.master.items() if not bool(torch.isfinite(v).all())]
    if bad:
        raise FloatingPointError(f"refusing to checkpoint iteration {iteration}: non-finite values in {bad[:4]}{' ...' if len(bad) > 4 else ''}")
    if not write:
        return
    model.load_trainer_state(trainer, keep_trainer=True)
    path = checkpoint_path(cfg, None if final else iteration)
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "iteration": iteration,
                "momentum": trainer.export_optimizer_state(), "loss_scale": trainer.loss_scale}, path)
    with open(os.path.join(cfg.OUTPUT_DIR, "last_checkpoint"), "w") as f:
        f.write(os.path.basename(path))
    logger.info("saved %s", path)


def resume_or_load(cfg, model, resume: bool):
    """[d2] DetectionCheckpointer.resume_or_load: with --resume and a last_checkpoint file continue from it, otherwise load
    MODEL.WEIGHTS (.pth state dict or the MSRA R-50.pkl); returns (iteration to start from, optimizer state {momentum buffers,
    loss scale} or None)."""
    import torch
    from openset_rcnn_amd.host.checkpoint import load_checkpoint, load_into
    last = os.path.join(cfg.OUTPUT_DIR, "last_checkpoint")
    if resume and os.path.exists(last):
        with open(last) as f:
            path = os.path.join(cfg.OUTPUT_DIR, f.read().strip())
        blob = torch.load(path, map_location="cpu", weights_only=False)
        load_into(model, blob["model"], strict=False)
        logger.info("resumed from %s (iteration %d)", path, blob.get("iteration", -1))
        return int(blob.get("iteration", -1)) + 1, {"momentum": blob.get("momentum"), "loss_scale": blob.get("loss_scale")}
    w = cfg.MODEL.WEIGHTS
    if w:
        if w.startswith("detectron2://"):
            raise FileNotFoundError(f"MODEL.WEIGHTS {w}: no network here; download the file and pass its path (MODEL.WEIGHTS /path/R-50.pkl)")
        missing, unexpected = load_into(model, load_checkpoint(w), strict=False)
        logger.info("loaded %s (%d keys missing, %d unexpected)", w, len(missing), len(unexpected))
    return 0, None


def do_train(cfg, args, model, D, start_iter: int, opt_state=None):
    """train.py:109-162. The loop body is the reference's (train.py:135-147) line for line: `model(data)` returns the loss dict,
    `losses.backward()` runs the explicit HIP backward, `optimizer.step()` all-reduces and applies SGD."""
    import torch
    from openset_rcnn_amd.host import parallel as comm
    from openset_rcnn_amd.host.data import DatasetMapper, build_detection_train_loader
    from openset_rcnn_amd.host.solver import build_lr_scheduler, build_optimizer
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    model.train()
    optimizer = build_optimizer(cfg, model)
    scheduler = build_lr_scheduler(cfg, optimizer, last_iter=start_iter - 1)
    trainer = model.trainer()
    if opt_state and opt_state.get("momentum") is not None:
        trainer.load_optimizer_state(opt_state["momentum"])
    if opt_state and opt_state.get("loss_scale"):
        trainer.loss_scale = float(opt_state["loss_scale"])
    seed = cfg.SEED if cfg.SEED >= 0 else 0
    dicts = [d for n in cfg.DATASETS.TRAIN for d in D.DatasetCatalog[n]()]
    data_loader = build_detection_train_loader(dicts, DatasetMapper(cfg, is_train=True, seed=seed), cfg.SOLVER.IMS_PER_BATCH, seed=seed,
                                               start_iter=start_iter)  # a resumed run continues the data stream (no image is decoded to skip)
    max_iter = cfg.SOLVER.MAX_ITER
    logger.info("Starting training from iteration %d", start_iter)
    for data, iteration in zip(data_loader, range(start_iter, max_iter)):
        # sampler keys (the randomness of both subsample_labels calls): a function of (seed, rank, iteration), so a resumed run
        # draws what the uninterrupted one would have (seed + rank)
        model.sampler_generator.manual_seed((seed * 1000003 + iteration) * 64 + rank)

        loss_dict = model(data)
        losses = sum(loss_dict.values())
        assert torch.isfinite(losses).all(), loss_dict

        loss_dict_reduced = {k: v.item() for k, v in comm.reduce_dict(loss_dict).items()}
        losses_reduced = sum(loss for loss in loss_dict_reduced.values())

        optimizer.zero_grad()
        losses.backward()
        optimizer.step()
        lr = optimizer.param_groups[0]["lr"]
        scheduler.step()

        if rank == 0 and ((iteration + 1) % 20 == 0 or iteration == max_iter - 1):
            logger.info("iter %d  total_loss %.4f  %s  lr %.6f%s", iteration + 1, losses_reduced,
                        "  ".join(f"{k} {v:.4f}" for k, v in loss_dict_reduced.items()), lr,
                        f"  (overflow-skipped steps: {trainer.overflow_steps}, loss scale {trainer.loss_scale:g})" if trainer.overflow_steps else "")
            # the ten EventStorage scalars of the reference's training iteration (rpn/*, roi_head/*, softmax_classifier/*; train.py's
            # CommonMetricPrinter / JSONWriter show them), this rank's values of the iteration just run
            logger.info("        %s", "  ".join(f"{k} {v:.3f}" for k, v in trainer.event_scalars().items()))
        if cfg.TEST.EVAL_PERIOD > 0 and (iteration + 1) % cfg.TEST.EVAL_PERIOD == 0 and iteration != max_iter - 1:
            model.eval()  # writes the trained masters back into the module
            do_test(cfg, args, model, D, iteration=iteration + 1)
            model.train()
            comm.barrier()
        if cfg.SOLVER.CHECKPOINT_PERIOD > 0 and (iteration + 1) % cfg.SOLVER.CHECKPOINT_PERIOD == 0:
            save_checkpoint(cfg, model, trainer, iteration, write=rank == 0)
    save_checkpoint(cfg, model, trainer, max_iter - 1, final=True, write=rank == 0)
    model.eval()


def main(argv=None) -> int:
    args = parse_args(argv)
    if args.num_gpus > 1 and "WORLD_SIZE" not in os.environ:
        return relaunch_under_torchrun(args)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    cfg = setup(args)
    D = register_datasets(cfg)
    if args.resume_test:
        if int(os.environ.get("RANK", "0")) == 0:  # re-scores a detections file: host-only work for one process
            do_test(cfg, args, None, D, iteration=args.test_iter)
        return 0
    import torch
    world, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("run_net.py needs a GPU: the HIP path has no CPU fallback")
    backend = os.environ.get("OSR_DIST_BACKEND", "nccl")  # rehearsal knob: gloo lets several ranks share one GPU
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    pkg._lib.load()
    from openset_rcnn_amd.host.modeling import build_model
    if cfg.SEED >= 0:
        torch.manual_seed(cfg.SEED)
    cfgm = cfg.clone()
    cfgm.defrost()
    cfgm.MODEL.DEVICE = f"cuda:{local_rank}"
    cfgm.freeze()
    model = build_model(cfgm, class_id_for(cfg, D))
    model.kernel_dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    if args.dtype == "f32" and not args.eval_only:
        raise SystemExit("--dtype f32 is the inference parity mode; train with f16 or bf16 (fp32 masters, fp32 accumulation)")
    start, opt_state = resume_or_load(cfg, model, args.resume)
    model.eval()
    if not args.eval_only:
        do_train(cfg, args, model, D, start, opt_state)
    do_test(cfg, args, model, D)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
