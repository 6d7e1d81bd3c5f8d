"""Where in the trainer's backward each gradient bucket's all-reduce is issued (the DDP overlap of
/root/reference/train.py:201-205).

tests/test_sharding_gloo.py checks GradBuckets itself (nb - 1 buckets in flight before finish()); nothing checked that the TRAINER
calls mark_done where it should, so a reordering of the backward could silently serialise the exchange behind the whole backward.
Here one real OpensetRCNNTrainer step runs on the GPU inside a ONE-rank process group (is_dist() is true, so the overlap path is the one
that runs; world size 1: no collective is actually sent) with recorders on the data-gradient launches, the weight-gradient launches and
GradBuckets.issue. Asserted: every bucket is issued exactly once, in the buffer's reverse-completion order, BEFORE all_reduce_grads();
a bucket is issued only after the last weight-gradient launch of every parameter in it; and each bucket goes out UNDER the rest of the
backward -- behind the head's bucket come the FPN's data gradients, behind res5's the whole of res4 / res3, ... -- i.e. the number of
data-gradient launches still to come after each issue is what the layer order says, not zero."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def one_rank_group():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_every_bucket_is_issued_under_the_rest_of_the_backward(osr, one_rank_group):
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from openset_rcnn_amd.host import ops, parallel
    from openset_rcnn_amd.host.train import OpensetRCNNTrainer
    from openset_rcnn_amd.host.weights import random_params
    from oracle import osr_oracle as O

    assert parallel.is_dist()
    tr = OpensetRCNNTrainer(random_params(0), dtype=torch.float16, device=DEV, lr=0.002, loss_scale=512.0)
    g = torch.Generator().manual_seed(23)
    n, h, w, gmax = 2, 128, 160, 4
    images = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    gt = torch.tensor([[[20., 24., 90., 100.], [60., 30., 150., 120.], [0, 0, 0, 0], [0, 0, 0, 0]],
                       [[10., 10., 70., 60.], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]])
    gcls = torch.tensor([[3, 7, 0, 0], [11, 0, 0, 0]])
    shapes = O.level_shapes(h, w)
    r = sum(a * b for a, b in shapes)
    cap = sum(min(2000, a * b) for a, b in shapes)
    keys = dict(rpn_reg=torch.rand(n, r, generator=g), rpn_obj=torch.rand(n, r, generator=g), roi=torch.rand(n, cap + gmax, generator=g))
    args = (images.to(DEV), torch.tensor([(h, w)] * n, dtype=torch.int32).to(DEV), h, w, gt.to(DEV), gcls.to(DEV),
            torch.tensor([2, 1], dtype=torch.int32).to(DEV), {k: v.to(DEV) for k, v in keys.items()})

    name_of = {t.data_ptr(): k for k, t in tr.grad.items()}
    log = []  # ("dgrad", None) | ("wgrad", parameter) | ("issue", bucket) | ("finish", None)
    real = dict(dgrad=ops.conv2d_dgrad, wgrad=ops.conv2d_wgrad, tn=ops.gemm_f32_tn, issue=tr.buckets.issue, finish=tr.buckets.finish)

    def dgrad(*a, **k):
        log.append(("dgrad", None))
        return real["dgrad"](*a, **k)

    def wgrad(*a, **k):
        dw = k.get("dw")
        log.append(("wgrad", name_of.get(dw.data_ptr()) if dw is not None else None))
        return real["wgrad"](*a, **k)

    def tn(*a, **k):
        out = k.get("out")
        log.append(("wgrad", name_of.get(out.data_ptr()) if out is not None else None))
        return real["tn"](*a, **k)

    def issue(b):
        log.append(("issue", b))
        return real["issue"](b)

    def finish():
        log.append(("finish", None))
        return real["finish"]()

    ops.conv2d_dgrad, ops.conv2d_wgrad, ops.gemm_f32_tn = dgrad, wgrad, tn
    tr.buckets.issue, tr.buckets.finish = issue, finish
    try:
        tr.step(*args)
        torch.cuda.synchronize()
    finally:
        ops.conv2d_dgrad, ops.conv2d_wgrad, ops.gemm_f32_tn = real["dgrad"], real["wgrad"], real["tn"]

    nb = len(tr.buckets.buckets)
    assert nb >= 5, "166.5 MB of fp32 gradients in >= 25 MB buckets"
    fin = [i for i, e in enumerate(log) if e[0] == "finish"]
    assert len(fin) == 1
    issues = [(i, e[1]) for i, e in enumerate(log) if e[0] == "issue"]
    assert sorted(b for _, b in issues) == list(range(nb)), "every bucket exactly once"
    assert issues[0][1] == 0, "bucket 0 (the end of the buffer: the heads, whose gradients the backward finishes first) goes out first"
    assert all(i < fin[0] for i, _ in issues), "no bucket may be left for all_reduce_grads() to issue: the backward marks every parameter"
    # a bucket goes out only behind the last weight-gradient launch of each of its parameters
    last_wgrad = {}
    for i, e in enumerate(log):
        if e[0] == "wgrad" and e[1] is not None:
            last_wgrad[e[1]] = i
    for i, b in issues:
        for nm in tr.buckets.buckets[b]["names"]:
            if nm in last_wgrad:
                assert last_wgrad[nm] < i, f"bucket {b} was issued before the last weight-gradient launch of {nm}"
    # ... and UNDER the rest of the backward: data-gradient launches still to come after each issue
    dg = [i for i, e in enumerate(log) if e[0] == "dgrad"]
    after = {b: sum(1 for j in dg if j > i) for i, b in issues}
    owner = tr.buckets.owner
    b_fc1, b_res5, b_res4, b_res3 = (owner[k] for k in ("fc1.w", "backbone.bottom_up.res5.0.conv1.w", "backbone.bottom_up.res4.0.conv1.w",
                                                         "backbone.bottom_up.res3.0.conv1.w"))
    assert b_fc1 <= b_res5 <= b_res4 <= b_res3
    total = len(dg)
    assert total >= 45, f"{total} data-gradient launches: box head, FPN, res5 .. res3"
    # the box head's bucket (FC1: 51 MB on its own) leaves before the FPN and the whole backbone: nearly every data gradient is still to come
    assert after[b_fc1] >= total - 6, (after, total)
    # res5's bucket leaves while res4 and res3 (6 + 4 bottlenecks x 3-4 data gradients) are still to run; res4's while res3 is
    assert after[b_res5] >= 25, after
    if b_res4 != b_res3:
        assert after[b_res4] >= 8, after
    # only the bucket that holds the LAST parameters of the backward (res2 is frozen: res3.0's) may have nothing behind it
    assert sum(1 for b in range(nb) if after[b] == 0) <= 1, after
