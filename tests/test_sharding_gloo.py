"""N>1 control path on CPU: 2 gloo ranks shard a global batch exactly like bench.py / the evaluator path does
(no data-path collective; MAX-reduced timing; rank-0 gather), SURVEY.md 8e."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.load_package()
    from openset_rcnn_amd.host import parallel as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = P.shard_range(global_n, rank, world)
    # stand-in for the per-image engine output: one record per image of this rank's shard
    mine = [{"image_id": i, "n_det": (i * 7) % 5} for i in range(lo, hi)]
    P.barrier()
    t = P.max_over_ranks(1.0 + rank)  # slowest rank defines the step time
    gathered = P.gather_to_rank0(mine)
    if rank == 0:
        q.put((t, P.merge_sharded(gathered)))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, global_n = 2, 33  # odd on purpose: ragged shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0
    assert [m["image_id"] for m in merged] == list(range(global_n))  # every image exactly once, global order kept


def test_shard_range_partitions():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.load_package()
    from openset_rcnn_amd.host.parallel import shard_range
    for n in (0, 1, 16, 33, 64, 127):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _eval_worker(rank, world, port, voc_dir, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.load_package()
    from openset_rcnn_amd.host.evaluation import PascalVOCDetectionEvaluator, inference_on_dataset
    from openset_rcnn_amd.host.structures import Boxes, Instances
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def model(batch):  # stand-in detector: one perfect "aeroplane" box per image
        out = []
        for x in batch:
            i = Instances((100, 200))
            i.pred_boxes = Boxes(torch.tensor([[10.0, 10.0, 60.0, 60.0]]))
            i.scores = torch.tensor([0.9])
            i.pred_classes = torch.tensor([0])
            out.append({"instances": i})
        return out

    ev = PascalVOCDetectionEvaluator(voc_dir, "toy", ["aeroplane", "unknown"], 1)
    res = inference_on_dataset(model, [[{"image_id": k}] for k in ("a", "b", "c", "d")], ev)
    if rank == 0:
        q.put(res)
    else:
        assert res is None
    dist.destroy_process_group()


def test_two_rank_evaluation(tmp_path):
    d = tmp_path / "voc"
    (d / "Annotations").mkdir(parents=True)
    (d / "ImageSets" / "Main").mkdir(parents=True)
    for k in "abcd":
        (d / "Annotations" / f"{k}.xml").write_text(
            "<annotation><size><width>200</width><height>100</height></size><object><name>aeroplane</name><difficult>0</difficult>"
            "<bndbox><xmin>11</xmin><ymin>11</ymin><xmax>60</xmax><ymax>60</ymax></bndbox></object></annotation>")
    (d / "ImageSets" / "Main" / "toy.txt").write_text("a\nb\nc\nd\n")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, str(d), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["AP@K"] == 100.0 and res["R@K"] == 100.0 and res["AOSE"] == 0.0  # all four images counted exactly once


def _grad_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.load_package()
    from openset_rcnn_amd.host import parallel as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the trainer's exchange step: one flat fp32 gradient buffer per rank, summed in place; the update divides by the world size
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    w = P.all_reduce_sum_(flat)
    if rank == 0:
        q.put((w, flat))
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    w, flat = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert w == 2 and torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 3)


def _bucket_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.load_package()
    from openset_rcnn_amd.host import parallel as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a flat gradient buffer with the trainer's structure: parameter views at 16-byte aligned offsets, completion order = descending offset
    g = torch.Generator().manual_seed(100 + rank)
    sizes = [4096, 12, 70000, 256, 33000, 5, 120000, 1024, 20]
    layout, off = [], 0
    for i, n in enumerate(sizes):
        al = (n + 3) // 4 * 4
        layout.append((f"p{i}", off, al))
        off += al
    flat = torch.randn(off, generator=g)
    single = flat.clone()
    P.all_reduce_sum_(single)  # the single-shot exchange of round 1
    buckets = P.GradBuckets(flat, layout, bucket_bytes=200_000)
    assert len(buckets.buckets) >= 3 and buckets.buckets[0]["hi"] == off and buckets.buckets[-1]["lo"] == 0
    assert all(a["lo"] == b["hi"] for a, b in zip(buckets.buckets, buckets.buckets[1:]))  # contiguous, from the end towards the start
    for name, _, _ in reversed(layout[1:]):  # the backward marks parameters as their gradients complete; p0 is never marked
        buckets.mark_done(name)
    in_flight = sum(buckets.issued)
    w = buckets.finish()  # issues the bucket the backward did not complete, waits for all
    # a second iteration on the same object (re-armed by finish)
    flat2 = flat.clone()
    for name, _, _ in reversed(layout):
        buckets.mark_done(name)
    w2 = buckets.finish()
    # the 6-scalar loss reduce of train.py:139
    red = P.reduce_dict({"loss_b": torch.tensor(1.0 + rank), "loss_a": torch.tensor(10.0 * (rank + 1))})
    if rank == 0:
        q.put((w, w2, in_flight, len(buckets.buckets), torch.equal(flat2, single), flat, single, {k: float(v) for k, v in red.items()}))
    dist.destroy_process_group()


def test_bucketed_all_reduce_equals_single_shot_and_loss_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    w, w2, in_flight, nb, first_ok, flat, single, red = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert w == 2 and w2 == 2 and first_ok
    assert in_flight == nb - 1  # every bucket but the one holding the unmarked parameter was already in flight before finish()
    # the second pass reduced the already-summed buffer again: (a+b) + (a+b)
    assert torch.equal(flat, single * 2)
    assert red == {"loss_a": 15.0, "loss_b": 1.5}  # averaged on rank 0, sorted keys


def _poison_worker(rank, world, port, bad_rank, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.load_package()
    from openset_rcnn_amd.host import parallel as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [4096, 12, 70000, 256, 33000]
    layout, off = [], 0
    for i, n in enumerate(sizes):
        layout.append((f"p{i}", off, n))
        off += n
    flat = torch.randn(off, generator=torch.Generator().manual_seed(7 + rank))
    buckets = P.GradBuckets(flat, layout, bucket_bytes=100_000)
    # the trainer's rpn_chain: the rank-local verdict ("my sparse row list fitted") goes into one element of the layer's bias
    # gradient BEFORE that bucket is marked done
    fit = torch.tensor([0 if rank == bad_rank else 1], dtype=torch.int32)
    name, o, _ = layout[3]
    P.poison_unless_(fit, flat[o:o + 1])
    local_finite = bool(torch.isfinite(flat).all())
    for nm, _, _ in reversed(layout):
        buckets.mark_done(nm)
    buckets.finish()
    q.put((rank, local_finite, bool(torch.isfinite(flat).all())))
    dist.destroy_process_group()


def test_a_rank_local_verdict_reaches_every_rank_through_the_gradient():
    """ADVICE r05 (host/train.py: the sparse CF-RPN row list's 'did not fit' flag): only rank 1's list overflows; after the bucketed
    all-reduce BOTH ranks hold a non-finite gradient, so both skip the update and halve the loss scale together. With no rank
    overflowing the buffer stays finite."""
    ctx = mp.get_context("spawn")
    for bad, want in ((1, False), (-1, True)):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_poison_worker, args=(r, 2, port, bad, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert [g[0] for g in got] == [0, 1]
        assert got[0][1] is True  # rank 0's own list fitted: its local gradient was clean ...
        assert got[1][1] is (bad != 1)
        assert got[0][2] is want and got[1][2] is want  # ... and the reduced buffer tells both ranks the same thing
