"""The schedules that are BENCHMARKED, tested as they are benchmarked: the two concurrency bugs of round 3
(a write-after-read race of an LDS ring; a 16-byte store whose data register was overwritten) were invisible to single-stream
tests and showed up only when kernels of different passes ran side by side.
  * bench.py's headline loop: four lanes, each a captured pass (hipGraph) over its own 16 x 3 x 800 x 1333 batch on its own stream,
    replayed interleaved -- every lane's outputs must equal that lane's eager single-stream pass, bit for bit;
  * the training step's three-stream schedule (weight gradients and the CF-RPN chain on a second stream, ground-truth-only
    targets on a third) at BASELINE config 3's size -- the flat gradient buffer must equal the single-stream run's, bit for bit
    (every kernel of the step is deterministic since the RoIAlign backward gathers instead of scattering)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DEV = "cuda:0"


def test_four_lanes_in_flight_give_each_lane_its_eager_result(osr):
    import bench
    from openset_rcnn_amd.host.engine import OpensetRCNNEngine
    from openset_rcnn_amd.host.weights import random_params
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    eng = OpensetRCNNEngine(random_params(0), dtype=torch.float16, device=DEV)
    batch, npass = 16, 4
    hw = torch.tensor([(800, 1333)] * batch, dtype=torch.int32, device=DEV)
    lane_images = [torch.randint(0, 256, (batch, 3, 800, 1333), generator=torch.Generator().manual_seed(1234 + 1000 * li), dtype=torch.uint8).to(DEV)
                   for li in range(npass)]
    # each lane's reference: the eager pass on one stream, alone on the GPU
    refs = []
    for imgs in lane_images:
        out = eng.forward_device(imgs, hw, 800, 1344)
        torch.cuda.synchronize()
        refs.append([t.clone() for t in out])
    assert sum(int(r[3].sum()) for r in refs) > 0  # the passes produce detections
    lanes, lane_gb = bench.make_lanes(eng, lane_images, hw, 1, npass)
    assert len(lanes) == npass
    turn = [0]
    for rnd in range(3):  # 3 x 4 = 12 steps in flight, checked after every round of four
        for _ in range(npass):
            bench.step_lanes(lanes, turn)
        torch.cuda.synchronize()
        for li, (_, out, _, _) in enumerate(lanes):
            for k, (got, want) in enumerate(zip(out, refs[li])):
                assert torch.equal(got, want), f"round {rnd}, lane {li}, output {k}: the interleaved replay differs from the lane's eager pass"


def test_three_stream_training_step_equals_the_single_stream_step_at_full_size(osr):
    import bench
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params
    n = 16
    images = torch.randint(0, 256, (n, 3, 800, 1333), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).to(DEV)
    hw = torch.tensor([(800, 1333)] * n, dtype=torch.int32, device=DEV)
    gt, gcls, gcnt = bench.synthetic_gt(n, 800, 1333)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    g = torch.Generator().manual_seed(0)
    keys = {k: torch.rand(s, generator=g).to(DEV) for k, s in (("rpn_reg", (n, r)), ("rpn_obj", (n, r)), ("roi", (n, cap + gt.shape[1])))}
    args = (images, hw, 800, 1344, gt.to(DEV), gcls.to(DEV), gcnt.to(DEV), keys)
    tr = OpensetRCNNTrainer(random_params(0), dtype=torch.float16, device=DEV, lr=1e-4, loss_scale=1024.0)

    def grads(side: bool):
        tr.side_wgrad = tr.overlap_targets = side
        tr.grad_flat.zero_()
        losses = tr.step(*args, update=False)
        torch.cuda.synchronize()
        return tr.grad_flat.clone(), {k: float(v) for k, v in losses.items()}

    g1, l1 = grads(False)
    g3, l3 = grads(True)
    g3b, _ = grads(True)
    assert l1 == l3
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(g3, g3b), "the three-stream step is not reproducible"
    assert torch.equal(g1, g3), "the three-stream schedule changes the gradients"
