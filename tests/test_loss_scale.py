"""Host half of the fp16 overflow guard (openset_rcnn_amd.host.train.DynamicLossScale): every update's verdict is applied, in
order, however late it is polled; the scale grows back after a run of clean updates. CPU: the device flag is a CPU tensor."""
import torch


def _flag(ok: bool) -> torch.Tensor:
    return torch.tensor([1 if ok else 0], dtype=torch.int32)


def test_no_verdict_is_lost_when_the_host_runs_ahead(osr):
    from openset_rcnn_amd.host.train import DynamicLossScale
    sc = DynamicLossScale(1024.0, growth_interval=0)
    # two updates issued back to back, polled only afterwards: the first one overflowed, the second did not. A single shared
    # host flag (round 2) would have been overwritten by the second update and the overflow would never have been seen.
    sc.record(_flag(False))
    sc.record(_flag(True))
    assert sc.poll() is True
    assert sc.scale == 512.0 and sc.overflow_steps == 1 and sc.clean_steps == 1
    assert sc.poll() is False and sc.scale == 512.0  # reported once
    # three more, two of them poisoned, all ISSUED at 512 before any of their verdicts was read: one overflow episode, one back-off
    # (GradScaler's behaviour; the updates issued between an overflow and its verdict ran at the same, too-large scale)
    for ok in (False, True, False):
        sc.record(_flag(ok))
    assert sc.poll(wait=True) is True and sc.scale == 256.0 and sc.overflow_steps == 3
    # an overflow issued AT the backed-off scale is a new episode
    sc.record(_flag(False))
    assert sc.poll(wait=True) is True and sc.scale == 128.0 and sc.overflow_steps == 4
    sc.poll()
    # the pinned slots are recycled, not re-allocated per update
    assert len(sc.free) == 3 and not sc.queue


def test_scale_grows_back_after_clean_updates_and_stops_at_the_configured_scale(osr):
    from openset_rcnn_amd.host.train import DynamicLossScale
    sc = DynamicLossScale(1024.0, growth_interval=4)
    sc.record(_flag(False))
    sc.poll()
    sc.record(_flag(False))
    sc.poll()
    assert sc.scale == 256.0
    for _ in range(3):
        sc.record(_flag(True))
    sc.poll()
    assert sc.scale == 256.0 and sc.clean_steps == 3
    sc.record(_flag(True))
    sc.poll()
    assert sc.scale == 512.0 and sc.clean_steps == 0
    for _ in range(12):
        sc.record(_flag(True))
    sc.poll()
    assert sc.scale == 1024.0  # never past the configured scale
    # an overflow in the middle of a clean run restarts the count
    for ok in (True, True, True, False, True):
        sc.record(_flag(ok))
    sc.poll()
    assert sc.scale == 512.0 and sc.clean_steps == 1


def test_floor_of_the_scale(osr):
    from openset_rcnn_amd.host.train import DynamicLossScale
    sc = DynamicLossScale(2.0, growth_interval=0)
    for _ in range(3):
        sc.record(_flag(False))
        sc.poll()
    assert sc.scale == 1.0 and sc.overflow_steps == 3


def test_non_finite_proposals_raise_when_the_iteration_is_drained(osr):
    """find_top_proposals.py:96-101: `FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")` in
    training. The selection kernel's status word rides in the overflow verdict's slot and raises when the host drains it."""
    import pytest
    from openset_rcnn_amd.host.train import DynamicLossScale
    sc = DynamicLossScale(1024.0, growth_interval=0)
    sc.record(_flag(True), torch.tensor([0], dtype=torch.int32))
    assert sc.poll() is False
    sc.record(_flag(True), torch.tensor([1], dtype=torch.int32))
    sc.record(_flag(True))
    with pytest.raises(FloatingPointError, match="Predicted boxes or scores contain Inf/NaN. Training has diverged."):
        sc.poll()
    assert sc.poll() is False and not sc.queue  # the iteration behind the diverged one is drained normally; slots are recycled clean
    sc.record(_flag(True))
    assert sc.poll() is False


def test_lagged_poll_applies_a_deterministic_set_of_verdicts(osr):
    """with several ranks the step waits for the verdicts up to two updates back (not for the newest one): the
    applied set is the same on every rank and the host keeps two iterations of run-ahead."""
    from openset_rcnn_amd.host.train import DynamicLossScale
    sc = DynamicLossScale(1024.0, growth_interval=0)
    for ok in (False, True, False, True):
        sc.record(_flag(ok))
    assert sc.poll(wait=True, lag=2) is True          # updates 0 and 1 only
    assert sc.scale == 512.0 and sc.overflow_steps == 1 and len(sc.queue) == 2
    assert sc.poll(wait=True, lag=2) is False         # nothing older than the newest two
    sc.record(_flag(True))
    # update 2 (overflowed) is now old enough: skipped and counted, but it was ISSUED at 1024, above the current scale -- the same
    # overflow episode as update 0, no second back-off (one episode used to cost 3 skipped updates and an 8x drop)
    assert sc.poll(wait=True, lag=2) is True and sc.scale == 512.0 and sc.overflow_steps == 2 and len(sc.queue) == 2
    assert sc.poll(wait=True) is False and not sc.queue
    sc.record(_flag(False))                           # issued at 512: a new episode
    assert sc.poll(wait=True) is True and sc.scale == 256.0
