"""One training step of Openset R-CNN on the HIP path: forward with saved activations, the six losses, backward through every
trainable layer, SGD with momentum -- the loop body of the reference's trainer (train.py:132-148: `loss_dict = model(data);
losses.backward(); optimizer.step()`) for META_ARCHITECTURE GeneralizedRCNN with [d2] defaults: FREEZE_AT 2 (stem and res2
frozen), FrozenBN everywhere, SGD momentum 0.9, weight decay 1e-4 on weights and biases.

What runs where: every convolution / FC forward, data gradient and weight gradient is an MFMA kernel launch (osr_conv2d_fwd /
osr_conv2d_wgrad), targets, sampling, losses and their gradients, RoIAlign forward/backward, the CF-RPN tail and the update are
the kernels of osr_train_fwd.hip / osr_train_bwd.hip / osr_roi_align.hip. torch is used for memory, views, zero-fills and a few
layout copies of small fp32 matrices (transposes / column padding for the exact-fp32 GEMM of the 5-, 21- and 256-wide heads).

Mixed precision: fp16 (or bf16) activations and activation gradients, fp32 accumulation, fp32 master weights and momentum,
static loss scaling (gradient tensors are multiplied by `loss_scale`, the update divides it out).
Data parallel: one process per GPU; `all_reduce_grads()` sums the single flat fp32 gradient buffer over RCCL (train.py:201-205
wraps the model in DDP) and the update divides by the world size."""
from __future__ import annotations

import collections
from typing import Dict, List, Optional, Tuple

import torch

from . import ops, parallel
from .engine import OpensetRCNNEngine, loss_types_of
from .weights import R50_BLOCKS, pack_conv_weight, pack_fc1_weight


def warmup_multistep_lr(iteration: int, base_lr: float, steps: Tuple[int, ...], gamma: float = 0.1, warmup_iters: int = 1000,
                        warmup_factor: float = 1.0 / 1000) -> float:
    """[d2] WarmupMultiStepLR (SOLVER.LR_SCHEDULER_NAME default; VOC-COCO yaml: BASE_LR 0.005, STEPS (84000, 116000),
    WARMUP_ITERS 400): linear warm-up of the factor from warmup_factor to 1, then x gamma at every milestone passed."""
    lr = base_lr * gamma ** sum(1 for s in steps if iteration >= s)
    if iteration < warmup_iters:
        alpha = iteration / warmup_iters
        lr *= warmup_factor * (1 - alpha) + alpha
    return lr


class DynamicLossScale:
    """Host half of the overflow guard: one (event, pinned slot) per update, drained IN ORDER. The device half is a flag that
    osr_check_finite clears and every osr_sgd_step launch of the iteration reads (a poisoned update changes nothing, no host
    sync). `record(flag)` queues that flag's copy into a slot of its own right after an update; `poll(wait)` applies the verdicts
    of the updates that have finished -- an overflow halves the scale (floor 1.0), `growth_interval` consecutive clean updates
    double it again, never past the configured scale (what torch's GradScaler does) -- so no verdict is ever overwritten however
    far the host runs ahead of the GPU, and the scale is not a one-way ratchet.

    The same slot carries the iteration's PROPOSAL STATUS word (osr_rpn_select's status_flags: non-zero when a predicted box or
    score was Inf / NaN). The reference raises FloatingPointError for that in training (find_top_proposals.py:96-101); here the
    word rides with the overflow verdict -- no extra host sync -- and `poll` raises the same exception when it drains the slot."""

    DIVERGED = "Predicted boxes or scores contain Inf/NaN. Training has diverged."  # find_top_proposals.py:99-101

    def __init__(self, scale: float, growth_interval: int = 2000, device: Optional[torch.device] = None):
        self.scale, self.scale_max, self.growth_interval = float(scale), float(scale), int(growth_interval)
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.queue: collections.deque = collections.deque()
        self.free: List[torch.Tensor] = []
        self.overflow_steps = 0
        self.clean_steps = 0  # consecutive clean updates since the last overflow / growth

    def record(self, ok_flag: torch.Tensor, proposal_status: Optional[torch.Tensor] = None) -> None:
        """ok_flag: (1,) int32 on the trainer's device, 1 = the update just enqueued was applied, 0 = skipped.
        proposal_status: (1,) int32, osr_rpn_select's status word of the same iteration (0 = every prediction finite)."""
        if self.free:
            slot = self.free.pop()
        else:
            slot = torch.ones((2,), dtype=torch.int32)
            if self.cuda:
                slot = slot.pin_memory()
        slot[0:1].copy_(ok_flag, non_blocking=True)
        if proposal_status is not None:
            slot[1:2].copy_(proposal_status, non_blocking=True)
        else:
            slot[1] = 0
        ev = None
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
        self.queue.append((ev, slot, self.scale))  # the scale this update was issued with

    def poll(self, wait: bool = False, lag: int = 0) -> bool:
        """True when at least one of the drained updates had been skipped. lag: leave the newest `lag` updates in the queue
        whatever their state -- with wait=True the call then applies EXACTLY the verdicts of all updates but the newest `lag`,
        a deterministic set (the same on every rank of a data-parallel job), while the host keeps `lag` iterations of run-ahead."""
        any_overflow = False
        while len(self.queue) > lag:
            ev, slot, issued_scale = self.queue[0]
            if ev is not None:
                if wait:
                    ev.synchronize()
                elif not ev.query():
                    break
            self.queue.popleft()
            ok, diverged = int(slot[0]) == 1, int(slot[1]) != 0
            self.free.append(slot)
            if diverged:
                raise FloatingPointError(self.DIVERGED)
            if ok:
                self.clean_steps += 1
                if self.growth_interval > 0 and self.clean_steps >= self.growth_interval and self.scale < self.scale_max:
                    self.scale = min(self.scale_max, self.scale * 2.0)
                    self.clean_steps = 0
            else:
                any_overflow = True
                self.overflow_steps += 1
                self.clean_steps = 0
                # One back-off per overflow EPISODE (GradScaler's behaviour): with a verdict lag the updates issued between the overflow
                # and its verdict ran at the same, too-large scale and overflow as well; they count as skipped steps but do not halve the
                # scale again (it is already below the scale they were issued with).
                if issued_scale <= self.scale:
                    self.scale = max(1.0, self.scale * 0.5)
        return any_overflow


class OpensetRCNNTrainer:
    def __init__(self, params: Dict[str, torch.Tensor], cfg: Optional[dict] = None, dtype: torch.dtype = torch.float16, device: str = "cuda",
                 lr: float = 0.005, momentum: float = 0.9, weight_decay: float = 1e-4, loss_scale: float = 1024.0, freeze_at: int = 2,
                 frozen_bn: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None, class_map: Optional[torch.Tensor] = None,
                 bucket_bytes: int = 25 << 20, scale_growth_interval: int = 2000):
        """params: BN-folded parameters under detectron2 names (what the engine reads). frozen_bn (optional): for convs followed by
        FrozenBatchNorm, name -> (un-folded weight (cout,cin,kh,kw), per-channel scale gamma/sqrt(var+eps)): the trainable parameter
        is the un-folded weight (weight decay acts on it, the chain rule multiplies the kernel's gradient by the scale)."""
        self.frozen_bn = frozen_bn or {}
        self.row_scale: Dict[str, torch.Tensor] = {}
        self.eng = OpensetRCNNEngine(params, cfg, dtype, device, class_map)
        # the CF-RPN head's backward runs on the sampled anchors only and recomputes their hidden state (osr_rpn_sparse.hip); False:
        # the dense launches of rounds 1-3 (the fused head kernel then also writes the hidden state of every anchor: 0.7 GB)
        self.sparse_rpn_bwd = True
        # None: the list's own capacity (n x 2 x RPN.BATCH_SIZE_PER_IMAGE). A smaller number makes an iteration whose list is longer
        # count as not fitting (tests: the skipped-update path without a pathological sampler)
        self.sparse_rows_cap: Optional[int] = None
        self.dtype, self.device = dtype, self.eng.device
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.scaler = DynamicLossScale(loss_scale, scale_growth_interval, self.eng.device)
        self.freeze_at = freeze_at
        e, dev = self.eng, self.eng.device
        f32 = lambda t: t.detach().float().contiguous().to(dev)  # noqa: E731
        # ---- trainable parameters: fp32 masters (conv / FC weights in the kernels' packed layout) ----
        self.master: Dict[str, torch.Tensor] = {}
        self.lowp: Dict[str, Optional[torch.Tensor]] = {}  # working copy the forward kernels read (None: the master itself is read)
        self.conv_names: List[str] = []
        for si, nb in enumerate(R50_BLOCKS):
            if si + 2 <= freeze_at:
                continue
            for b in range(nb):
                pre = f"backbone.bottom_up.res{si + 2}.{b}"
                for cname in (["shortcut"] if b == 0 else []) + ["conv1", "conv2", "conv3"]:
                    self._add_conv(f"{pre}.{cname}", params, bias=False)  # FrozenBN: the folded shift is not a parameter
        # (the masters are laid out in REVERSE order of gradient completion -- backbone bottom-up, FPN coarse to fine, RPN head, box
        # head, the small fp32 heads last -- so that the finished part of the flat gradient buffer grows from its end and the
        # all-reduce buckets of parallel.GradBuckets are contiguous)
        for lvl in (5, 4, 3, 2):
            self._add_conv(f"backbone.fpn_lateral{lvl}", params, bias=True)
            self._add_conv(f"backbone.fpn_output{lvl}", params, bias=True)
        self._add_conv("proposal_generator.rpn_head.conv", params, bias=True)
        self.master["rpn_tail.w"], self.master["rpn_tail.b"] = e.rpn_wtail, e.rpn_btail  # (5,256): 4 ltrb rows + centerness
        e.rpn_wd, e.rpn_wc, e.rpn_bd, e.rpn_bc = e.rpn_wtail[:4], e.rpn_wtail[4:5], e.rpn_btail[:4], e.rpn_btail[4:5]  # views: one storage
        self.master["fc1.w"] = pack_fc1_weight(params["roi_heads.box_head.fc1.weight"], 256, e.cfg["pooler_resolution"], torch.float32).to(dev)
        self.lowp["fc1.w"] = e.fc1_w
        self.master["fc1.b"] = e.fc1_b
        self.master["fc2.w"] = f32(params["roi_heads.box_head.fc2.weight"])
        self.lowp["fc2.w"] = e.fc2_w
        self.master["fc2.b"] = e.fc2_b
        self.master["pred.w"], self.master["pred.b"] = e.pred_w, e.pred_b
        self.master["enc.w"], self.master["enc.b"] = e.enc_w, e.enc_b
        self.master["dec.w"], self.master["dec.b"] = e.dec_w, e.dec_b
        self.master["cls.w"], self.master["cls.b"] = e.cls_w, e.cls_b
        self.master["protos"] = f32(params["roi_heads.dml.representatives"])
        # ---- one flat gradient buffer (the all-reduce operand), momentum buffers ----
        al = lambda x: (x + 3) // 4 * 4  # noqa: E731  every view starts 16-byte aligned (the kernels use 16-byte accesses)
        total = sum(al(t.numel()) for t in self.master.values())
        self.grad_flat = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.grad: Dict[str, torch.Tensor] = {}
        off, layout = 0, []
        for k, t in self.master.items():
            self.grad[k] = self.grad_flat[off:off + t.numel()].view(t.shape)
            layout.append((k, off, al(t.numel())))
            off += al(t.numel())
        self.buckets = parallel.GradBuckets(self.grad_flat, layout, bucket_bytes)
        self.mom = {k: torch.zeros_like(t) for k, t in self.master.items()}
        self.num_params = sum(t.numel() for t in self.master.values())
        # overflow guard: device flag read by every osr_sgd_step launch; its host half is self.scaler (DynamicLossScale)
        self._ok = torch.ones((1,), dtype=torch.int32, device=dev)
        self._cside: Optional[torch.cuda.Stream] = None  # stream the gradient buckets' collectives are issued from (see _done)
        self._overlap = False
        self.grads_ready = False
        self._side: Optional[torch.cuda.Stream] = None  # stream of the ground-truth-only part of the forward (anchor targets)
        self.overlap_targets = True
        self._wside: Optional[torch.cuda.Stream] = None  # stream of the weight / bias gradient launches (see _wg)
        self.side_wgrad = True
        self.chain_forward = True  # res3 blocks: conv2 -> conv3 in one launch that also stores conv2's output (osr_conv2d_chain_fwd_ex)
        self.multi_tensor_update = True  # the update as two launches (ops.sgd_step_multi_, ops.pack_dgrad_weight_multi_); False: one launch per tensor
        self._sgd_plan = None
        self._pack_plan = None
        self._pre: Optional[torch.cuda.Stream] = None  # stream of the next batch's frozen prefix (_prefetch_frozen)
        self.backward_concurrency_hint = 2  # launch streams of the backward (data gradients + weight gradients): see step(); 24.0 -> 23.7 ms
        self._prefetched = None                        # (images, (hp, wp), (x, feats), event, images._version)
        # blocks whose weight gradients ride on the main stream (measured with res3.0 / res3.0-1 / all of res3: 25.4-25.5 against 25.5-25.7 ms,
        # inside the run-to-run spread: the backward is bound by the sum of its kernels, not by which stream ends last) -- left empty
        self.wgrad_on_main: set = set()
        self._refresh_derived()

    @property
    def loss_scale(self) -> float:
        return self.scaler.scale

    @loss_scale.setter
    def loss_scale(self, v: float) -> None:  # (load_optimizer_state restores the scale a checkpoint was written with)
        self.scaler.scale = float(v)

    @property
    def overflow_steps(self) -> int:
        return self.scaler.overflow_steps

    def _add_conv(self, name: str, params, bias: bool):
        e = self.eng
        if name in self.frozen_bn:
            w_unfolded, scale = self.frozen_bn[name]
            self.master[name + ".w"] = pack_conv_weight(w_unfolded, torch.float32).to(e.device)
            self.row_scale[name + ".w"] = scale.detach().float().contiguous().to(e.device)
        else:
            self.master[name + ".w"] = pack_conv_weight(params[name + ".weight"], torch.float32).to(e.device)
        self.lowp[name + ".w"] = e.w[name + ".w"]
        if bias:
            self.master[name + ".b"] = e.w[name + ".b"]
        self.conv_names.append(name)

    def _fan(self, jobs) -> None:
        """Independent in-place launches dealt round-robin over the trainer's streams: all start after the main stream's current
        point, the main stream continues when all have finished. (No allocation inside the jobs: nothing changes stream.)"""
        if not self.side_wgrad or not torch.cuda.is_available() or self.device.type != "cuda":
            for job in jobs:
                job()
            return
        cur = torch.cuda.current_stream(self.device)
        if self._wside is None:
            self._wside = torch.cuda.Stream(device=self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        sts = [cur, self._wside, self._side]
        for st in sts[1:]:
            st.wait_stream(cur)
        for i, job in enumerate(jobs):
            with torch.cuda.stream(sts[i % len(sts)]):
                job()
        for st in sts[1:]:
            cur.wait_stream(st)

    def _refresh_derived(self):
        """Everything that is a function of the parameters and read by a kernel: backward-data weights, transposed fp32 heads,
        normalised prototypes."""
        e = self.eng
        wd = getattr(self, "wd", None)
        if wd is None:  # first call: allocate the buffers (on the main stream); they are refilled in place every step (osr_pack_dgrad_weight)
            wd = {n: ops.pack_dgrad_weight(e.w[n + ".w"]) for n in self.conv_names}
            wd["fc1"] = ops.pack_dgrad_weight(e.fc1_w).view(e.fc1_w.shape[1], 1, 1, e.fc1_w.shape[0])
            wd["fc2"] = ops.pack_dgrad_weight(e.fc2_w).view(e.fc2_w.shape[1], 1, 1, e.fc2_w.shape[0])
            self.wd = wd
        elif self.multi_tensor_update:
            pairs = [(e.w[n + ".w"], wd[n]) for n in self.conv_names]
            pairs += [(e.fc1_w, wd["fc1"].view(e.fc1_w.shape[1], e.fc1_w.shape[0])), (e.fc2_w, wd["fc2"].view(e.fc2_w.shape[1], e.fc2_w.shape[0]))]
            sig = tuple(t.data_ptr() for pr in pairs for t in pr)
            if self._pack_plan is None or self._pack_plan[0] != sig:
                self._pack_plan = (sig, ops.pack_dgrad_multi_plan(pairs, self.device))
            # The backward-data weights are read by the NEXT BACKWARD only (the next forward reads the low-precision copies the update
            # kernel has just written): with the second stream in use the repack (0.15 ms, one launch) leaves the end of the iteration
            # -- it runs on that stream, idle until the next backward, behind the update, and the next backward waits for its event
            if self.side_wgrad and self._wside is not None and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                self._wside.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._wside):
                    ops.pack_dgrad_weight_multi_(self._pack_plan[1])
                    self._pack_done = self._wside.record_event()
            else:
                ops.pack_dgrad_weight_multi_(self._pack_plan[1])
                self._pack_done = None
        else:
            jobs = [lambda n=n: ops.pack_dgrad_weight(e.w[n + ".w"], wd[n]) for n in self.conv_names]
            jobs.append(lambda: ops.pack_dgrad_weight(e.fc1_w, wd["fc1"].view(e.fc1_w.shape[1], e.fc1_w.shape[0])))
            jobs.append(lambda: ops.pack_dgrad_weight(e.fc2_w, wd["fc2"].view(e.fc2_w.shape[1], e.fc2_w.shape[0])))
            self._fan(jobs)
        if not hasattr(self, "t_cls"):  # zero-padded transposes of the narrow fp32 heads: the padding rows are written once
            self.t_cls = torch.zeros((e.cls_w.shape[1], 32), dtype=torch.float32, device=e.device)    # (1024, 32): d rec = d logits(padded to 32) . W_cls
            self.t_pred = torch.zeros((e.pred_w.shape[1], 16), dtype=torch.float32, device=e.device)  # (1024, 16)
            self.t_dec = torch.empty((e.dec_w.shape[1], e.dec_w.shape[0]), dtype=torch.float32, device=e.device)  # (256, 1024): d emb = d rec . W_dec
            self.t_enc = torch.empty((e.enc_w.shape[1], e.enc_w.shape[0]), dtype=torch.float32, device=e.device)  # (1024, 256): d box_feats = d emb . W_enc
        self.t_cls[:, : e.cls_w.shape[0]].copy_(e.cls_w.t())
        self.t_pred[:, : e.pred_w.shape[0]].copy_(e.pred_w.t())
        ops.pack_dgrad_weight(e.dec_w, self.t_dec)
        ops.pack_dgrad_weight(e.enc_w, self.t_enc)
        e.protos = ops.l2_normalize_rows(self.master["protos"])

    # ---- forward with saved activations -------------------------------------------------------------------------
    def _forward(self, images, image_hw, hp, wp, gt_boxes, gt_classes, gt_count, keys):
        e, c = self.eng, self.eng.cfg
        n = images.shape[0]
        s: dict = {}
        # anchor labels / sampling / targets depend on the ground truth only: a few small-grid launches (one workgroup per image)
        # that would otherwise sit in the stream between the RPN head and its loss; they run beside the backbone on a side stream
        cur = torch.cuda.current_stream(self.device)
        shapes = e.pyramid_shapes(hp, wp)
        lv = e._levels(shapes, n)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        if self.overlap_targets:
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                rpn_targets = e.rpn_targets_forward(lv, n, gt_boxes, gt_count, keys)
                targets_ready = self._side.record_event()
        pref, self._prefetched = self._prefetched, None
        if pref is not None and pref[0] is images and pref[1] == (hp, wp) and pref[4] == images._version:
            # the frozen prefix of THIS batch was computed under the previous iteration's backward (step(next_images=...)): take it
            x, frozen_feats = pref[2]
            cur.wait_event(pref[3])
        else:
            x, frozen_feats = self._frozen_prefix(images, hp, wp)
        blocks = []
        feats = {}
        for si, nb in enumerate(R50_BLOCKS):
            for b in range(nb):
                pre = f"backbone.bottom_up.res{si + 2}.{b}"
                stride = 2 if (b == 0 and si > 0) else 1
                if si + 2 <= self.freeze_at:  # frozen stage: part of the prefix above (nothing of it is needed by the backward)
                    break
                pair = e._shortcut_conv1_one_launch(x, pre, stride) if b == 0 and e.fuse_levels else None  # (one launch for the two readers of x)
                if pair is not None:
                    sc, o1 = pair
                else:
                    sc = e._conv(x, pre + ".shortcut", stride) if b == 0 else x
                    o1 = e._conv(x, pre + ".conv1", stride, relu=True)
                # res3: conv2 -> conv3 + shortcut as ONE launch that also stores conv2's output for the backward (bit-identical to the two)
                ch = ops.conv2d_chain(o1, e.w[pre + ".conv2.w"], e.w[pre + ".conv2.b"], e.w[pre + ".conv3.w"], e.w[pre + ".conv3.b"], sc, 1, 1,
                                      keep_mid=True) if (self.chain_forward and e.w[pre + ".conv2.w"].shape[0] == 128 and e.w[pre + ".conv3.w"].shape[0] == 512) else None
                if ch is not None:
                    y, o2 = ch
                else:
                    o2 = e._conv(o1, pre + ".conv2", 1, 1, relu=True)
                    y = e._conv(o2, pre + ".conv3", relu=True, residual=sc, res_mode=1)
                blocks.append(dict(pre=pre, x=x, o1=o1, o2=o2, y=y, stride=stride, first=b == 0, stage=si + 2))
                x = y
            feats[f"res{si + 2}"] = frozen_feats[f"res{si + 2}"] if si + 2 <= self.freeze_at else x
        s["blocks"], s["res"] = blocks, feats
        out = {}
        lat = {5: e._conv(feats["res5"], "backbone.fpn_lateral5")}
        for lvl in (4, 3, 2):
            lat[lvl] = e._conv(feats[f"res{lvl}"], f"backbone.fpn_lateral{lvl}", residual=lat[lvl + 1], res_mode=2)
        outs = e._fpn_outputs_one_launch([lat[l] for l in (2, 3, 4, 5)]) if e.fuse_levels else None  # (the four output convs as one launch: engine._backbone)
        for i, lvl in enumerate((2, 3, 4, 5)):
            out[f"p{lvl}"] = outs[i] if outs is not None else e._conv(lat[lvl], f"backbone.fpn_output{lvl}", 1, 1)
        out["p6"] = ops.subsample2(out["p5"])
        s["lat"], s["p"] = lat, out
        # CF-RPN head (unfused: the hidden state t is kept), targets, losses
        keep: dict = {}
        e.rpn_keep_hidden = not self.sparse_rpn_bwd
        sel = e._rpn(out, image_hw, keep, topk=c["pre_nms_topk_train"])
        s["rpn_t"], s["rpn_shapes"], s["sel"] = keep["rpn_t"], keep["rpn_shapes"], sel
        self._proposal_status = sel["status_flags"]  # read with this iteration's overflow verdict (DynamicLossScale.record)
        assert list(keep["rpn_shapes"]) == list(shapes), "pyramid_shapes disagrees with the backbone"
        if self.overlap_targets:
            cur.wait_event(targets_ready)
            if not torch.cuda.is_current_stream_capturing():
                for t in rpn_targets.values():
                    t.record_stream(cur)
        else:
            rpn_targets = None
        rpn, rpn_state = e.rpn_losses_forward(sel, n, gt_boxes, gt_count, keys, targets=rpn_targets)
        s.update(rpn_state)
        # RoI heads on the sampled proposals
        roi, roi_state = e.roi_losses_forward(out, sel["boxes"], sel["scores"], sel["counts"], gt_boxes, gt_classes, gt_count, keys["roi"])
        s.update(roi_state)
        losses = dict(loss_rpn_loc=rpn[0], loss_rpn_ctr=rpn[1], loss_box_reg=roi["loss_box_reg"], loss_iou=roi["loss_iou"],
                      loss_dml=roi["loss_dml"], loss_cls=roi["loss_cls"])
        # what event_scalars() reads (references only: nothing is computed or copied unless a logger asks)
        self._last_forward = dict(n=n, rpn_counts=rpn[2:6], prop_counts=sel["counts"], roi_counts=roi["roi_counts"], logits=roi_state["logits"],
                                  cls_k=roi_state["cls_k"], nck=roi_state["nck"], batch_idx=roi_state["smp"]["batch_idx"])
        return losses, s

    def event_scalars(self) -> Dict[str, float]:
        """The ten scalars the reference puts into detectron2's EventStorage during a training iteration, for the LAST forward of this
        trainer (one small D2H copy; call it on logging iterations only):
          rpn/num_pos_anchors, rpn/num_neg_anchors, rpn/obj_num_pos_anchors, rpn/obj_num_neg_anchors (classification_free_rpn.py:459-463),
          rpn/num_proposals (:553-554), roi_head/num_fg_samples, roi_head/num_bg_samples (osrcnn_roi_heads.py:226-228: means over the
          images of the sampled foreground / background proposals), softmax_classifier/cls_accuracy, /fg_cls_accuracy, /false_negative
          (softmax_classifier.py:18-45 on the classifier's logits and the id-mapped targets; the last two only when a foreground row exists)."""
        lf = getattr(self, "_last_forward", None)
        if lf is None:
            return {}
        n, K = lf["n"], self.eng.cfg["num_known"]
        t = lf["cls_k"]
        if self.eng.id_map is None:  # VOC-COCO: known ids are 0..K-1 already, background = NUM_CLASSES -> K, every other class -> -1 (id_map, :224-229)
            t = torch.where(t < K, t, torch.where(t == lf["nck"], torch.full_like(t, K), torch.full_like(t, -1)))
        valid = lf["batch_idx"].view(-1) >= 0
        pred = lf["logits"][:, :K + 1].argmax(dim=1)
        fg = valid & (t >= 0) & (t < K)
        stats = torch.stack([valid.sum(), (valid & (pred == t)).sum(), fg.sum(), (fg & (pred == t)).sum(), (fg & (pred == K)).sum()]).to(torch.float32)
        host = torch.cat([lf["rpn_counts"].to(torch.float32).view(-1), lf["prop_counts"].to(torch.float32).view(-1), lf["roi_counts"].to(torch.float32).view(-1),
                          stats]).cpu().tolist()
        rc, pc, roi, st = host[:4], host[4:4 + n], host[4 + n:4 + n + 3 * n], host[4 + 4 * n:]
        out = {"rpn/num_pos_anchors": rc[0] / n, "rpn/num_neg_anchors": rc[1] / n, "rpn/obj_num_pos_anchors": rc[2] / n, "rpn/obj_num_neg_anchors": rc[3] / n,
               "rpn/num_proposals": sum(pc) / max(n, 1),
               "roi_head/num_fg_samples": sum(roi[1::3]) / n, "roi_head/num_bg_samples": sum(roi[2::3]) / n}
        if st[0] > 0:
            out["softmax_classifier/cls_accuracy"] = st[1] / st[0]
            if st[2] > 0:
                out["softmax_classifier/fg_cls_accuracy"] = st[3] / st[2]
                out["softmax_classifier/false_negative"] = st[4] / st[2]
        return out

    def _frozen_prefix(self, images, hp, wp):
        """Stem (+ preprocessing) and the frozen residual stages: what depends on the batch and on frozen weights only -- the output of
        the last frozen stage (the stem's max pool output when only the stem is frozen)."""
        e, c = self.eng, self.eng.cfg
        if e.fuse_stem and self.freeze_at >= 1:  # (the stem is frozen: nothing of it is needed by the backward)
            x = ops.stem_maxpool_raw(images, hp, wp, c["pixel_mean"], c["pixel_std"], e.w["backbone.bottom_up.stem.conv1.w"], e.w["backbone.bottom_up.stem.conv1.b"])
        else:
            xpad = ops.preprocess(images, hp, wp, c["pixel_mean"], c["pixel_std"], self.dtype)
            x = ops.stem_conv(xpad, e.w["backbone.bottom_up.stem.conv1.w"], e.w["backbone.bottom_up.stem.conv1.b"], hp, wp, relu=True)
            x = ops.maxpool3x3s2(x)
        feats = {}
        for si, nb in enumerate(R50_BLOCKS):
            if si + 2 > self.freeze_at:
                break
            for b in range(nb):
                x = e._bottleneck(x, f"backbone.bottom_up.res{si + 2}.{b}", b == 0, 2 if (b == 0 and si > 0) else 1)
            feats[f"res{si + 2}"] = x
        return x, feats

    def _prefetch_frozen(self, images, hp, wp) -> None:
        """Software pipelining across iterations: the next batch's frozen prefix (a function of that batch and of weights no update
        touches) is enqueued on a stream of its own behind the main stream's CURRENT point -- called between the forward and the
        backward, it runs under the chain of small head / loss launches that leaves the GPU nearly idle there. The next step() that is
        handed the same tensor object picks the result up (and waits for it); any other batch recomputes. Same values either way."""
        if self.freeze_at < 2 or self.device.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.device)
        if self._pre is None:
            self._pre = torch.cuda.Stream(device=self.device)
        self._pre.wait_event(cur.record_event())
        with torch.cuda.stream(self._pre):
            x, feats = self._frozen_prefix(images, hp, wp)
            done = self._pre.record_event()
        for t in [x] + list(feats.values()):
            t.record_stream(cur)
        images.record_stream(self._pre)  # (the caching allocator must not hand the batch's memory out while the side stream reads it)
        # Keyed on the tensor OBJECT and its version counter: a loader that refills one device buffer in place hands the same object with
        # a bumped version (recompute); next_images must not be written again before the step that consumes it. load_state_dict /
        # set_freeze_at drop the prefetch (the prefix is a function of the frozen weights).
        self._prefetched = (images, (hp, wp), (x, feats), done, images._version)

    # ---- backward -------------------------------------------------------------------------------------------------
    def _f32_linear_bwd(self, x, dy, wt, name, dy_pad=None):
        """y = x W^T + b with fp32 operands on the exact-f32 GEMM: returns dx; writes dW, db. wt = W^T padded (k_in, n_pad)."""
        g = self.grad
        dyp = dy if dy_pad is None else torch.nn.functional.pad(dy, (0, dy_pad - dy.shape[1]))
        dx = ops.gemm_f32(dyp, wt, None)                                                    # (m, k_in)
        # dW / db only consume dy and the saved input: off the chain of small launches between the forward and the heavy backward, which
        # sits alone on the critical stream (the weight-gradient stream has nothing else to do at that point)
        self._wg(lambda: (ops.gemm_f32_tn(dy, x, out=g[name + ".w"]),                      # dW = dy^T x, (n_out, k_in)
                          ops.bias_grad(dy, g[name + ".b"])), dy, x)
        self._done(name + ".w", name + ".b")
        return dx

    def _done(self, *names: str) -> None:
        """The gradients of these parameters are final (their last launch is enqueued): a bucket they complete starts its
        all-reduce now, under the rest of the backward. The collective is issued from a stream of its own that waits for two
        EVENTS -- the main stream's and the weight-gradient stream's current points -- so it is ordered behind everything enqueued
        so far on both, and neither of the two compute streams waits for the other or for the collective (a wait_stream of the
        weight-gradient stream on the main one at every call re-serialised the two streams the single-GPU step gains 2 ms from)."""
        if not self._overlap:
            return
        ready = self.buckets.mark_ready(names)
        if not ready:
            return
        if self.device.type != "cuda":
            for b in ready:
                self.buckets.issue(b)
            return
        if self._cside is None:
            self._cside = torch.cuda.Stream(device=self.device)
        self._cside.wait_event(torch.cuda.current_stream(self.device).record_event())
        if self.side_wgrad and self._wside is not None:
            self._cside.wait_event(self._wside.record_event())
        with torch.cuda.stream(self._cside):
            for b in ready:
                self.buckets.issue(b)

    def _wg(self, fn, *reads: torch.Tensor, main: bool = False) -> None:
        """Run a weight / bias gradient launch group off the critical path: the chain of data gradients (dy of a layer -> dy of the
        layer below) stays on the main stream, the launches that only consume a layer's dy and its saved input (wgrad, its split
        reduction, the bias gradient) go to a second stream, where they fill the sparse last rounds of the data-gradient kernels
        and vice versa. `reads`: the tensors of the main stream the launches read (kept alive for the second stream)."""
        if not self.side_wgrad or main:
            fn()
            return
        cur = torch.cuda.current_stream(self.device)
        if self._wside is None:
            self._wside = torch.cuda.Stream(device=self.device)
        self._wside.wait_stream(cur)
        with torch.cuda.stream(self._wside):
            fn()
        if not torch.cuda.is_current_stream_capturing():
            for t in reads:
                t.record_stream(self._wside)

    def _backward(self, s, n, grad_scale: float = 1.0, overlap: bool = True, prefetch=None):
        """Gradients of grad_scale * (sum of the six losses), times the loss scale, into self.grad. overlap: start each gradient
        bucket's all-reduce as soon as the backward has passed it (several ranks only; all_reduce_grads() then just waits)."""
        e, c, g, S = self.eng, self.eng.cfg, self.grad, self.loss_scale * grad_scale
        self._scale_used = self.loss_scale  # the update divides out the scale THIS backward multiplied in, whatever a poll does in between
        self._overlap = overlap and parallel.is_dist()
        self.buckets.reset()
        if getattr(self, "_pack_done", None) is not None:  # the repack of the backward-data weights behind the previous update (_refresh_derived)
            torch.cuda.current_stream(self.device).wait_event(self._pack_done)
            self._pack_done = None
        dt = self.dtype
        p = s["p"]
        # --- CF-RPN: losses -> tail -> weight gradient of the 3x3 conv (weights shared by the five levels). This chain depends on
        #     the forward only and meets the RoI heads' chain at the 3x3 conv's data gradient: it runs first, on the second stream,
        #     under the RoI heads' backward (whose RoIAlign scatter is bound by the atomic rate, not by the matrix cores) ---
        sel = s["sel"]
        rn = "proposal_generator.rpn_head.conv"

        def rpn_chain():
            lt_ = loss_types_of(c)
            d5 = ops.rpn_losses_bwd(sel["levels"], e.cell_anchors, n, sel["pred_deltas"], sel["pred_ctr"], s["labels"], s["obj_labels"], s["matched_boxes"],
                                    s["ctr_target"], c["rpn_loc_weight"], c["rpn_ctr_weight"], c["rpn_batch_size"], S, box_loss=lt_["rpn_box"],
                                    ctr_beta=lt_["rpn_ctr"][1])
            if self.sparse_rpn_bwd:
                # the loss touches the sampled anchors only: list the rows with a gradient, gather their im2col rows, and run the
                # head's backward on that list (csrc/osr_rpn_sparse.hip) -- hidden state recomputed, weight gradient and per-tap data
                # gradient as three small GEMMs; the dense launches spent 2 x 1.7 TFLOP on zeros
                lvl_keys = ("p2", "p3", "p4", "p5", "p6")
                cap = n * 2 * int(c["rpn_batch_size"])
                ids, rmap, cnt2 = ops.rpn_sparse_rows(d5, cap)
                # count2 = {listed, found}: rows beyond the cap are dropped by the list (the reference sampler bounds them by 2 x 256 per image;
                # another sampler, or NaN spilling into unsampled rows, could exceed it). The iteration's update is then skipped like a
                # gradient overflow, instead of applying an update that misses the coarsest levels' gradients -- on EVERY rank: the verdict
                # is rank-local (this rank's list), so it is folded into the gradient itself (an inf in this layer's bias gradient, below,
                # in front of its bucket's all-reduce) and reaches the other ranks through the sum; _update() then sees a non-finite
                # all-reduced buffer everywhere. (the flag used to gate this rank's update only; the peers applied theirs.)
                rows_fit = (cnt2[1:2] <= min(cap, self.sparse_rows_cap or cap)).to(torch.int32)
                cols, d5r = ops.rpn_gather_cols(sel["levels"], [p[k_] for k_ in lvl_keys], n, ids, d5)
                w3 = e.w[rn + ".w"].view(256, 9 * 256)
                t_rows = ops.linear(cols, w3, e.w[rn + ".b"], relu=True)
                dt_rows, dw_tail, db_tail = ops.cfrpn_tail_bwd(t_rows, e.rpn_wtail, d5r)
                g["rpn_tail.w"].copy_(dw_tail)
                g["rpn_tail.b"].copy_(db_tail)
                ops.conv2d_wgrad(cols.view(1, cap, 1, 9 * 256), dt_rows.view(1, cap, 1, 256), 1, 1, dw=g[rn + ".w"].view(256, 1, 1, 9 * 256))
                ops.bias_grad(dt_rows, g[rn + ".b"])
                parallel.poison_unless_(rows_fit, g[rn + ".b"].view(-1)[:1])
                y_rows = ops.linear(dt_rows, w3.t().contiguous(), ops._zero_bias(9 * 256, self.device), out_dtype=torch.float32)
                return (rmap, y_rows), torch.cuda.current_stream(self.device).record_event()
            dta, dw_tail, db_tail = ops.cfrpn_tail_bwd(s["rpn_t"], e.rpn_wtail, d5)
            g["rpn_tail.w"].copy_(dw_tail)
            g["rpn_tail.b"].copy_(db_tail)
            ready = torch.cuda.current_stream(self.device).record_event()
            off_ = 0
            for li_, (k_, (h_, w_)) in enumerate(zip(("p2", "p3", "p4", "p5", "p6"), s["rpn_shapes"])):
                rows_ = n * h_ * w_
                dtl_ = dta[off_:off_ + rows_].view(n, h_, w_, 256)
                off_ += rows_
                ops.conv2d_wgrad(p[k_], dtl_, 3, 3, 1, 1, dw=g[rn + ".w"], accumulate=li_ > 0)
                ops.bias_grad(dtl_, g[rn + ".b"], accumulate=li_ > 0)
            return (dta,), ready
        if self.side_wgrad:
            cur0 = torch.cuda.current_stream(self.device)
            if self._wside is None:
                self._wside = torch.cuda.Stream(device=self.device)
            self._wside.wait_stream(cur0)
            with torch.cuda.stream(self._wside):
                rpn_grad, rpn_ready = rpn_chain()
        else:
            rpn_grad, rpn_ready = rpn_chain()
        self._done("rpn_tail.w", "rpn_tail.b", rn + ".w", rn + ".b")
        # --- RoI-head losses -> predictor / PLN / classifier (fp32 heads) ---
        lt = loss_types_of(c)
        d_pred = ops.roi_box_losses_bwd(s["pred"], s["boxes"], s["smp"]["gt_boxes"].view(-1, 4), s["cls"], s["ious"], c["num_classes"],
                                        c["bbox_reg_weights"], c["box_reg_weight"], c["iou_reg_weight"], S, box_loss=lt["roi_box"],
                                        iou_beta=lt["roi_iou"][1])
        d_logits = ops.softmax_ce_loss_bwd(s["logits"], s["cls_k"], s["nck"], c["cls_loss_weight"], S)
        d_emb_pln, d_protos = ops.pln_loss_bwd(s["emb"], self.master["protos"], s["cls_k"], s["ious"], c["pln_iou_threshold"], c["pln_alpha"],
                                               c["pln_beta"], c["pln_loss_weight"], S, reps=c["reps_per_class"], distance=c["pln_distance"])
        g["protos"].copy_(d_protos)
        self._done("protos")
        d_rec = self._f32_linear_bwd(s["rec"], d_logits, self.t_cls, "cls", dy_pad=32)
        d_emb = self._f32_linear_bwd(s["emb"], d_rec, self.t_dec, "dec")
        d_emb = ops.add_cast(d_emb, d_emb_pln, torch.float32)
        d_bf = self._f32_linear_bwd(s["box_feats"], d_emb, self.t_enc, "enc")
        d_bf2 = self._f32_linear_bwd(s["box_feats"], d_pred, self.t_pred, "pred", dy_pad=16)
        d_bf = ops.add_cast(d_bf, d_bf2, torch.float32)
        ops.relu_mask_(d_bf, s["box_feats"])
        dy2 = ops.add_cast(d_bf, None, dt)                                                  # (m,1024) low precision
        m = dy2.shape[0]
        # --- box head: FC2, FC1 on the MFMA kernels ---
        d_h1 = ops.conv2d_dgrad(dy2.view(1, m, 1, -1), self.wd["fc2"], (m, 1), mask=s["h1"].view(1, m, 1, -1)).view(m, -1)
        self._wg(lambda: (ops.conv2d_wgrad(s["h1"].view(1, m, 1, -1), dy2.view(1, m, 1, -1), 1, 1, dw=g["fc2.w"].view(-1, 1, 1, g["fc2.w"].shape[1])),
                          ops.bias_grad(dy2, g["fc2.b"])), dy2)
        self._done("fc2.w", "fc2.b")
        pooled2 = s["pooled"].view(1, m, 1, -1)
        d_pooled = ops.conv2d_dgrad(d_h1.view(1, m, 1, -1), self.wd["fc1"], (m, 1))
        self._wg(lambda: (ops.conv2d_wgrad(pooled2, d_h1.view(1, m, 1, -1), 1, 1, dw=g["fc1.w"].view(-1, 1, 1, g["fc1.w"].shape[1])),
                          ops.bias_grad(d_h1, g["fc1.b"])), d_h1)
        self._done("fc1.w", "fc1.b")
        P = c["pooler_resolution"]
        shapes = [(p[k].shape[1], p[k].shape[2]) for k in ("p2", "p3", "p4", "p5")]
        d_feat = ops.roi_align_bwd(d_pooled.view(m, P, P, -1), shapes, n, c["pooler_scales"], s["boxes"], s["smp"]["batch_idx"], c["canonical_level"],
                                   c["canonical_size"], 2, rois_per_image=m // n if m % n == 0 else None,  # (the sampled list is (n, S))
                                   out_dtype=dt if d_pooled.dtype == dt else None)
        # --- CF-RPN 3x3 conv: data gradient per level, joined with the RoI heads' feature gradient (the chain above it ran on the
        #     second stream, see the top of this function) ---
        if self.side_wgrad:
            torch.cuda.current_stream(self.device).wait_event(rpn_ready)
            if not torch.cuda.is_current_stream_capturing():
                for t_ in rpn_grad:
                    t_.record_stream(torch.cuda.current_stream(self.device))
        dP = {}
        lvl_shapes = list(zip(("p2", "p3", "p4", "p5", "p6"), s["rpn_shapes"]))
        roi_part = [(d_feat[li] if d_feat[li].dtype == dt else ops.add_cast(d_feat[li], None, dt)) for li in range(4)] + [None]
        if self.sparse_rpn_bwd:
            # col2im of the listed anchors' per-tap gradients straight into the RoI heads' feature gradient (p6 has none: into zeros)
            rmap, y_rows = rpn_grad
            h6, w6 = s["rpn_shapes"][4]
            glist = roi_part[:4] + [torch.zeros((n, h6, w6, 256), dtype=dt, device=self.device)]
            ops.rpn_scatter_cols_add_(sel["levels"], n, rmap, y_rows, glist)
            dP = {k: gl for (k, _), gl in zip(lvl_shapes, glist)}
        else:
            (dt_all,), off = rpn_grad, 0
            for li, (k, (h, w)) in enumerate(lvl_shapes):
                rows = n * h * w
                dP[k] = ops.conv2d_dgrad(dt_all[off:off + rows].view(n, h, w, 256), self.wd[rn], (h, w), 1, 1, add=roi_part[li])
                off += rows
        h5, w5 = p["p5"].shape[1], p["p5"].shape[2]
        dP["p5"] = ops.pool_bwd(dP["p6"], (h5, w5), dP["p5"], 1)  # p6 = p5[::2, ::2]
        # --- FPN: output convs, top-down adds, laterals (finest level first: its gradient flows up to the coarser sums) ---
        d_ls_prev = None
        d_res = {}
        for lvl in (2, 3, 4, 5):
            on, ln = f"backbone.fpn_output{lvl}", f"backbone.fpn_lateral{lvl}"
            ls = s["lat"][lvl]
            h, w = ls.shape[1], ls.shape[2]
            dpl = dP[f"p{lvl}"]
            up = ops.pool_bwd(d_ls_prev, (h, w), None, 0) if d_ls_prev is not None else None
            d_ls = ops.conv2d_dgrad(dpl, self.wd[on], (h, w), 1, 1, add=up)
            res = s["res"][f"res{lvl}"]
            self._wg(lambda ls=ls, dpl=dpl, res=res, d_ls=d_ls, on=on, ln=ln: (
                ops.conv2d_wgrad(ls, dpl, 3, 3, 1, 1, dw=g[on + ".w"]), ops.bias_grad(dpl, g[on + ".b"]),
                ops.conv2d_wgrad(res, d_ls, 1, 1, dw=g[ln + ".w"]), ops.bias_grad(d_ls, g[ln + ".b"])), dpl, d_ls)
            self._done(on + ".w", on + ".b", ln + ".w", ln + ".b")
            if lvl > self.freeze_at:
                d_res[lvl] = (d_ls, ln)  # the lateral's data gradient is formed together with the next stage's (see below)
            d_ls_prev = d_ls
        # --- backbone res5 -> res3: bottlenecks in reverse; G = gradient w.r.t. a block's output ---
        # G always arrives already masked by the ReLU that produced y (`post_mask=` of the launch that formed it): the join of
        # the two gradient branches and the ReLU below it are one epilogue (osr_conv2d_fwd_masked)
        G = None
        for bi in range(len(s["blocks"]) - 1, -1, -1):
            blk = s["blocks"][bi]
            pre, x, o1, o2, y, stride = blk["pre"], blk["x"], blk["o1"], blk["o2"], blk["y"], blk["stride"]
            hy, wy = y.shape[1], y.shape[2]
            last_of_stage = blk is s["blocks"][-1] or pre.endswith(f".{R50_BLOCKS[blk['stage'] - 2] - 1}")
            if last_of_stage:  # the stage output also feeds its FPN lateral
                d_ls, ln = d_res[blk["stage"]]
                G = ops.conv2d_dgrad(d_ls, self.wd[ln], (hy, wy), 1, 0, add=G, post_mask=y)
            # (the last blocks of the backward: the weight-gradient stream is behind the data-gradient chain by then and nothing waits for
            # the chain's end, so their weight gradients ride on the main stream, beside the other stream's backlog)
            on_main = pre in self.wgrad_on_main
            d_o2 = ops.conv2d_dgrad(G, self.wd[pre + ".conv3"], (hy, wy), 1, 0, mask=o2)
            self._wg(lambda o2=o2, G=G, pre=pre: ops.conv2d_wgrad(o2, G, 1, 1, dw=g[pre + ".conv3.w"]), G, main=on_main)
            d_o1 = ops.conv2d_dgrad(d_o2, self.wd[pre + ".conv2"], (o1.shape[1], o1.shape[2]), 1, 1, mask=o1)
            self._wg(lambda o1=o1, d_o2=d_o2, x=x, d_o1=d_o1, pre=pre, stride=stride: (
                ops.conv2d_wgrad(o1, d_o2, 3, 3, 1, 1, dw=g[pre + ".conv2.w"]),
                ops.conv2d_wgrad(x, d_o1, 1, 1, stride, 0, dw=g[pre + ".conv1.w"])), d_o2, d_o1, main=on_main)
            if blk["first"]:
                self._wg(lambda x=x, G=G, pre=pre, stride=stride: ops.conv2d_wgrad(x, G, 1, 1, stride, 0, dw=g[pre + ".shortcut.w"]), G, main=on_main)
                self._done(pre + ".shortcut.w")
            self._done(pre + ".conv3.w", pre + ".conv2.w", pre + ".conv1.w")
            if blk["first"] and blk["stage"] == self.freeze_at + 1:
                break  # the block's input comes from frozen layers
            hx, wx = x.shape[1], x.shape[2]
            # x is the output of the block below (post-ReLU): its mask goes into the launch that completes G -- unless that block
            # is the last of its stage, whose G is completed (and masked) by the lateral's launch at the top of the next turn
            below_last = bi > 0 and s["blocks"][bi - 1]["pre"].endswith(f".{R50_BLOCKS[s['blocks'][bi - 1]['stage'] - 2] - 1}")
            pm = None if below_last else x
            if blk["first"]:
                # (conv1's share is read back by the shortcut's launch at the strided pixels only: no zero fill of the others)
                dx = ops.conv2d_dgrad(d_o1, self.wd[pre + ".conv1"], (hx, wx), stride, 0, strided_only=stride > 1)
                G = ops.conv2d_dgrad(G, self.wd[pre + ".shortcut"], (hx, wx), stride, 0, add=dx, post_mask=pm)
            else:
                G = ops.conv2d_dgrad(d_o1, self.wd[pre + ".conv1"], (hx, wx), 1, 0, add=G, post_mask=pm)
        if prefetch is not None:  # behind the last data gradient: the main stream is done, the weight-gradient stream still has its backlog
            self._prefetch_frozen(*prefetch)
        if self.side_wgrad and self._wside is not None:  # join: the update (and any collective issued from here on) sees every weight gradient
            torch.cuda.current_stream(self.device).wait_stream(self._wside)

    # ---- optimiser ------------------------------------------------------------------------------------------------
    def all_reduce_grads(self) -> int:
        """Sum the flat gradient buffer over the data-parallel ranks (RCCL over xGMI); returns the world size. The buckets whose
        parameters the backward has already marked are in flight since then (overlapped with the rest of the backward); this
        issues the remainder and waits for all of them."""
        return self.buckets.finish()

    def _update(self, world: int):
        """SGD on every master, gated on the device by the overflow flag: an iteration whose (all-reduced) gradients hold an inf or
        NaN changes neither parameters nor momentum (the reference trains in fp32 and cannot overflow; fp16 gradients can). The
        flag is read back lazily by `poll_overflow()` -- no host sync here."""
        gs = 1.0 / (getattr(self, "_scale_used", self.loss_scale) * world)
        self._ok.fill_(1)
        ops.check_finite_(self.grad_flat, self._ok)
        self._apply_sgd(self.lr, self.momentum, self.weight_decay, gs)
        self.scaler.record(self._ok, getattr(self, "_proposal_status", None))
        self._proposal_status = None

    def null_update(self) -> None:
        """The update's launches with nothing to apply (bench.py's no-collective timing): the caller has zeroed the gradient; learning
        rate 0, weight decay 0 and momentum factor 1 make p' = p - 0 * (1 * v + 0) and v' = 1 * v + 0 -- parameters, momentum buffers,
        low-precision copies and backward-data weights come out bit for bit as they went in, no verdict is queued in the loss scaler."""
        self._ok.fill_(1)
        ops.check_finite_(self.grad_flat, self._ok)
        self._apply_sgd(0.0, 1.0, 0.0, 1.0)

    def _apply_sgd(self, lr: float, momentum: float, weight_decay: float, gs: float) -> None:
        if self.multi_tensor_update:
            # every parameter tensor in ONE launch (osr_sgd_step_multi over a device-resident table), then every backward-data weight in
            # one more (_refresh_derived): ~145 launches of a few microseconds of work each became two; same bits per element
            sig = tuple(t.data_ptr() for k in self.master for t in (self.master[k], self.grad[k], self.mom[k]) + ((self.lowp[k],) if self.lowp.get(k) is not None else ()))
            if self._sgd_plan is None or self._sgd_plan[0] != sig:
                self._sgd_plan = (sig, ops.sgd_multi_plan([(pm, self.grad[k], self.mom[k], self.row_scale.get(k), self.lowp.get(k))
                                                           for k, pm in self.master.items()], self.device))
            ops.sgd_step_multi_(self._sgd_plan[1], lr, momentum, weight_decay, gs, self._ok)
        else:
            # ~75 in-place launches of a few microseconds each (one per parameter tensor), then ~70 repacking launches: dealt over the
            # three streams of the trainer they run three abreast instead of one behind the other
            self._fan([lambda k=k, pm=pm: ops.sgd_step_(pm, self.grad[k], self.mom[k], lr, momentum, weight_decay, gs, self.row_scale.get(k),
                                                       self.lowp.get(k), self._ok) for k, pm in self.master.items()])
        self._refresh_derived()

    # several ranks: step k applies the verdicts of the updates up to k - 1 - MULTI_RANK_LAG, waited for -- the same set on every rank
    # (they are functions of the all-reduced gradient), so the scales cannot drift apart, and the host still runs two iterations ahead
    # of the GPU (waiting for update k - 1 at the top of step k exposed the ~150 small update launches and the next forward's launch
    # latency on every iteration)
    MULTI_RANK_LAG = 2

    def poll_overflow(self, wait: bool = False, lag: int = 0) -> bool:
        """Drain, IN ORDER, the verdicts of the updates that have finished (wait=True: of every update issued so far). True when
        at least one of them was skipped because of non-finite gradients. Dynamic loss scaling as GradScaler does it: an
        overflow halves the scale (floor 1.0) and counts in `overflow_steps`; `scale_growth_interval` consecutive clean updates
        double it again, never past the configured scale. Every update has a pinned slot of its own, so a verdict is never lost
        however far the host runs ahead. With several ranks step() calls this with wait=True and lag=MULTI_RANK_LAG: every rank then applies the same
        verdicts (they are functions of the all-reduced gradient) at the same iteration, and the scales cannot drift apart.
        (`wait=True` is also what the checkpoint writer uses, so that no skipped or half-applied state is written blind.)"""
        return self.scaler.poll(wait, lag)

    def step(self, images, image_hw, hp, wp, gt_boxes, gt_classes, gt_count, keys, update: bool = True, next_images=None) -> Dict[str, torch.Tensor]:
        """One iteration: returns the loss dict (GPU scalars). update=False leaves the parameters untouched (gradients stay in
        self.grad, scaled by loss_scale). next_images (optional): the NEXT iteration's image batch, if the loader already has it: its
        frozen prefix (stem + frozen stages) is computed under this iteration's backward (_prefetch_frozen) and used by the next
        step() that is given that same tensor."""
        # earlier iterations' verdicts adjust the loss scale here, at one deterministic point of the iteration; with several ranks
        # the call waits for them, so that all ranks change the scale at the same iteration (see poll_overflow)
        self.poll_overflow(wait=parallel.is_dist(), lag=self.MULTI_RANK_LAG if parallel.is_dist() else 0)
        losses, saved = self._forward(images, image_hw, hp, wp, gt_boxes, gt_classes, gt_count, keys)
        # the data-gradient chain shares the GPU with the weight-gradient stream: tell the conv tile model (osr_conv_params.concurrency; a
        # tile-selection hint only -- results do not depend on it) so that it may take the 256 x 256 tile where a lone stream would not
        with ops.concurrent_streams(self.backward_concurrency_hint):
            self._backward(saved, images.shape[0], overlap=update, prefetch=(next_images, hp, wp) if next_images is not None else None)
        if update:
            self._update(self.all_reduce_grads())
        return losses

    def export_optimizer_state(self) -> Dict[str, torch.Tensor]:
        """Momentum buffers (CPU copies) for a checkpoint that is resumed exactly ([d2] checkpoints carry the optimizer state)."""
        out = {k: v.detach().cpu().clone() for k, v in self.mom.items()}
        # One rank: every issued update's verdict goes into the scale that is written. Several ranks: the queue is left ALONE -- the ranks apply
        # verdicts in lockstep inside step() (MULTI_RANK_LAG), and a drain here, on whichever ranks happen to write a checkpoint, would change
        # one rank's scale 2-3 steps before its peers' (different scales multiplied into the same all-reduce). A collective checkpoint
        # writer that wants the newest verdicts in the file calls poll_overflow(wait=True) on EVERY rank first (run_net.save_checkpoint).
        if not parallel.is_dist():
            self.poll_overflow(wait=True)
        out[self.SCALE_KEY] = torch.tensor([self.scaler.scale, float(self.scaler.clean_steps), float(self.scaler.overflow_steps), self.scaler.scale_max],
                                           dtype=torch.float64)
        return out

    SCALE_KEY = "__dynamic_loss_scale__"  # (scale, consecutive clean updates, skipped updates so far, configured scale)

    def load_optimizer_state(self, state: Dict[str, torch.Tensor]) -> None:
        state = dict(state)
        sc = state.pop(self.SCALE_KEY, None)
        if sc is not None:  # a run that had backed off resumes at the scale it was written with, not at the configured one
            self.scaler.scale, self.scaler.clean_steps, self.scaler.overflow_steps = float(sc[0]), int(sc[1]), int(sc[2])
        for k, v in state.items():
            if k not in self.mom or tuple(self.mom[k].shape) != tuple(v.shape):
                raise KeyError(f"optimizer state {k}: not a momentum buffer of this trainer")
            self.mom[k].copy_(v.to(self.mom[k].device))

    def export_state_dict(self) -> Dict[str, torch.Tensor]:
        """The trainable parameters under detectron2 names and layouts (un-folded conv weights), for load_state_dict /
        checkpointing ([d2] DetectionCheckpointer writes these keys under "model")."""
        e = self.eng
        out: Dict[str, torch.Tensor] = {}
        for n in self.conv_names:
            out[n + ".weight"] = self.master[n + ".w"].permute(0, 3, 1, 2).contiguous().cpu()
            if n + ".b" in self.master:
                out[n + ".bias"] = self.master[n + ".b"].cpu().clone()
        pr = e.cfg["pooler_resolution"]
        fc1 = self.master["fc1.w"]  # (out, ph, pw, c) flattened -> the reference's (out, c*ph*pw)
        out["roi_heads.box_head.fc1.weight"] = fc1.view(fc1.shape[0], pr, pr, -1).permute(0, 3, 1, 2).reshape(fc1.shape[0], -1).contiguous().cpu()
        out["roi_heads.box_head.fc1.bias"] = self.master["fc1.b"].cpu().clone()
        out["roi_heads.box_head.fc2.weight"] = self.master["fc2.w"].cpu().clone()
        out["roi_heads.box_head.fc2.bias"] = self.master["fc2.b"].cpu().clone()
        t_w, t_b = self.master["rpn_tail.w"].cpu(), self.master["rpn_tail.b"].cpu()
        out["proposal_generator.rpn_head.anchor_deltas.weight"] = t_w[:4].reshape(4, -1, 1, 1).clone()
        out["proposal_generator.rpn_head.anchor_deltas.bias"] = t_b[:4].clone()
        out["proposal_generator.rpn_head.centerness.weight"] = t_w[4:5].reshape(1, -1, 1, 1).clone()
        out["proposal_generator.rpn_head.centerness.bias"] = t_b[4:5].clone()
        p_w, p_b = self.master["pred.w"].cpu(), self.master["pred.b"].cpu()
        out["roi_heads.box_predictor.bbox_pred.weight"], out["roi_heads.box_predictor.bbox_pred.bias"] = p_w[:4].clone(), p_b[:4].clone()
        out["roi_heads.box_predictor.iou_pred.weight"], out["roi_heads.box_predictor.iou_pred.bias"] = p_w[4:5].clone(), p_b[4:5].clone()
        for short, long in (("enc", "roi_heads.dml.encoder"), ("dec", "roi_heads.dml.decoder"), ("cls", "roi_heads.softmaxcls.cls_score")):
            out[long + ".weight"], out[long + ".bias"] = self.master[short + ".w"].cpu().clone(), self.master[short + ".b"].cpu().clone()
        out["roi_heads.dml.representatives"] = self.master["protos"].cpu().clone()
        return out
