"""Mirror of the two detectron2 solver factories the reference's trainer calls (train.py:110-111): `build_optimizer(cfg, model)`
and `build_lr_scheduler(cfg, optimizer)`, so that the loop body of train.py:135-146 runs unchanged on the HIP path --

    loss_dict = model(data); losses = sum(loss_dict.values()); assert torch.isfinite(losses).all(), loss_dict
    loss_dict_reduced = {k: v.item() for k, v in comm.reduce_dict(loss_dict).items()}
    optimizer.zero_grad(); losses.backward(); optimizer.step(); scheduler.step()

`losses.backward()` runs the explicit HIP backward into the trainer's flat fp32 gradient buffer (modeling._ExplicitBackward);
`optimizer.step()` sums that buffer over the ranks (RCCL, bucketed, overlapped with the tail of the backward when several ranks
run) and applies SGD with momentum and weight decay on the fp32 masters (osr_sgd_step), gated by the overflow guard."""
from __future__ import annotations

from typing import List

from .train import warmup_multistep_lr


class HipSGD:
    """[d2] build_optimizer -> torch.optim.SGD(momentum, weight_decay on weights and biases alike: WEIGHT_DECAY_BIAS and
    WEIGHT_DECAY_NORM default to WEIGHT_DECAY, BIAS_LR_FACTOR 1). One parameter group; `param_groups[0]["lr"]` is what the
    scheduler writes and what train.py:147 logs."""

    def __init__(self, model, lr: float, momentum: float, weight_decay: float):
        self.model = model
        self.param_groups: List[dict] = [dict(lr=float(lr), momentum=float(momentum), weight_decay=float(weight_decay), initial_lr=float(lr))]

    def zero_grad(self, set_to_none: bool = True) -> None:
        """No-op: every backward overwrites the whole gradient buffer (there is no accumulation across iterations)."""

    def step(self) -> None:
        t = self.model.trainer()
        if not getattr(t, "grads_ready", False):
            raise RuntimeError("optimizer.step() before losses.backward(): the gradient buffer holds no gradients of this iteration")
        g = self.param_groups[0]
        t.lr, t.momentum, t.weight_decay = g["lr"], g["momentum"], g["weight_decay"]
        t._update(t.all_reduce_grads())
        t.grads_ready = False

    def state_dict(self) -> dict:
        return dict(param_groups=[dict(g) for g in self.param_groups], momentum=self.model.trainer().export_optimizer_state())

    def load_state_dict(self, state: dict) -> None:
        self.param_groups = [dict(g) for g in state["param_groups"]]
        self.model.trainer().load_optimizer_state(state["momentum"])


def build_optimizer(cfg, model) -> HipSGD:
    s = cfg.SOLVER
    # (every norm layer on this path is a FrozenBN, so WEIGHT_DECAY_NORM has no parameter to act on)
    wd_bias = s.get("WEIGHT_DECAY_BIAS", None)
    if float(s.get("BIAS_LR_FACTOR", 1.0)) != 1.0 or (wd_bias is not None and float(wd_bias) != float(s.WEIGHT_DECAY)):
        raise NotImplementedError("SOLVER.BIAS_LR_FACTOR / WEIGHT_DECAY_BIAS other than the [d2] defaults: osr_sgd_step uses one group")
    clip = s.get("CLIP_GRADIENTS", None)
    if clip is not None and clip.get("ENABLED", False):
        raise NotImplementedError("SOLVER.CLIP_GRADIENTS: not on the hot path (both Openset yaml files leave it off)")
    return HipSGD(model, s.BASE_LR, s.MOMENTUM, s.WEIGHT_DECAY)


class WarmupMultiStepLR:
    """[d2] build_lr_scheduler for SOLVER.LR_SCHEDULER_NAME "WarmupMultiStepLR" (the default both yaml files use)."""

    def __init__(self, optimizer: HipSGD, base_lr, steps, gamma, warmup_iters, warmup_factor, last_iter: int = -1):
        self.optimizer, self.base_lr, self.steps, self.gamma = optimizer, float(base_lr), tuple(steps), float(gamma)
        self.warmup_iters, self.warmup_factor = int(warmup_iters), float(warmup_factor)
        self.last_iter = last_iter
        self.step()  # like torch's schedulers: construction sets the learning rate of iteration last_iter + 1

    def get_lr(self) -> float:
        return warmup_multistep_lr(self.last_iter, self.base_lr, self.steps, self.gamma, self.warmup_iters, self.warmup_factor)

    def step(self) -> None:
        self.last_iter += 1
        self.optimizer.param_groups[0]["lr"] = self.get_lr()

    def state_dict(self) -> dict:
        return dict(last_iter=self.last_iter)

    def load_state_dict(self, state: dict) -> None:
        self.last_iter = int(state["last_iter"]) - 1
        self.step()


def build_lr_scheduler(cfg, optimizer: HipSGD, last_iter: int = -1) -> WarmupMultiStepLR:
    s = cfg.SOLVER
    if s.LR_SCHEDULER_NAME != "WarmupMultiStepLR" or s.WARMUP_METHOD != "linear":
        raise NotImplementedError(f"SOLVER.LR_SCHEDULER_NAME {s.LR_SCHEDULER_NAME} / WARMUP_METHOD {s.WARMUP_METHOD}: the yaml files use WarmupMultiStepLR, linear")
    return WarmupMultiStepLR(optimizer, s.BASE_LR, s.STEPS, s.GAMMA, s.WARMUP_ITERS, s.WARMUP_FACTOR, last_iter)
