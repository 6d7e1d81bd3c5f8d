"""bench.py --gpus N from a plain command line starts the N ranks itself: the parent relays a
torch.distributed.run job before anything touches the GPU and returns its exit code. CPU: the launcher is monkeypatched."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_n_without_a_launcher_starts_one_rank_per_gpu(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 5

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    # (no GPU here: reaching torch.cuda.is_available() would raise SystemExit -- the relaunch happens before it)
    assert bench.main(["--gpus", "8", "--steps", "7", "--warmup", "2"]) == 5
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd[cmd.index("--master-port") + 2]
    assert os.path.basename(script) == "bench.py" and os.path.isabs(script)
    assert cmd[-6:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]  # the ranks see the same flags (and report n_gpus = 8)
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # RCCL's dmabuf IPC on this pool


def test_under_a_launcher_the_world_size_must_match(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert "WORLD_SIZE=4" in str(e.value)


def test_single_gpu_run_does_not_relaunch(monkeypatch):
    called = []
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: called.append(a) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:  # no GPU in the CPU suite: the run refuses instead of falling back
        bench.main(["--gpus", "1"])
    assert not called and "needs a GPU" in str(e.value)


def test_bf16_is_not_offered_as_a_benchmark_dtype():
    with pytest.raises(SystemExit):
        bench.main(["--dtype", "bf16"])


def _identity_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = bench.rank_identity(dist, rank, world, local_elapsed=0.5 + 0.25 * rank, steps=10, batch=16, gpu_uuid=f"GPU-{rank:04d}", device_name="AMD Instinct MI355X")
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_n_rank_line_names_its_ranks_and_devices():
    """the N > 1 line carries what proves N ranks on N devices -- the process group's world size, every
    rank's GPU uuid (N distinct on a real node) and every rank's own rate. Two gloo ranks on CPU with stand-in uuids."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_identity_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["ranks"] == 2 and out["gpu_uuids"] == ["GPU-0000", "GPU-0001"] and out["distinct_gpus"] == 2
    assert out["per_rank_images_per_sec"] == [320.0, 213.33] and out["per_rank_ms_per_step"] == [50.0, 75.0]
    assert out["device_names"] == ["AMD Instinct MI355X"]
    one = bench.rank_identity(None, 0, 1, 0.2, 4, 16, "GPU-x", "dev")
    assert one["ranks"] == 1 and one["distinct_gpus"] == 1 and one["per_rank_images_per_sec"] == [320.0]


CONFIG4_FIELDS = {"config", "ms_per_iter", "images_per_sec", "n_gpus", "batch_per_gpu", "num_known", "num_classes", "gradient_bytes", "all_reduce_ms",
                  "all_reduce_GBps_per_rank", "ms_per_iter_without_all_reduce", "all_reduce_exposed_ms", "all_reduce_hidden_fraction"}


def check_n_rank_line(line: dict, world: int) -> None:
    """What the first multi-GPU run must answer: BASELINE configs 2 / 3 / 4 in one record."""
    assert line["n_gpus"] == world and line["config"]["ranks"] == world and line["scaling"] == "weak"
    assert "roofline" in line and "train_step" in line and "config4" in line
    c4 = line["config4"]
    assert CONFIG4_FIELDS <= set(c4), sorted(CONFIG4_FIELDS - set(c4))
    assert c4["n_gpus"] == world and c4["batch_per_gpu"] == 8 and c4["num_known"] == 28 and c4["num_classes"] == 88
    assert c4["gradient_bytes"] == line["train_step"]["trainable_params"] * 4 or c4["gradient_bytes"] > 150e6
    assert 0.0 <= c4["all_reduce_hidden_fraction"] <= 1.0 and c4["all_reduce_ms"] > 0
    assert "1280x720" in c4["config"] and "GraspNet" in c4["config"]


def test_committed_two_rank_rehearsal_line_has_the_config4_object():
    """The record of the 2-rank rehearsal (gloo, both ranks on one MI355X; profiles/r05_rehearsal2_line.json, written by the GPU test
    below on the GPU box) parses and answers configs 2, 3 and 4 -- a CPU-side check of the N > 1 record's schema."""
    import json
    path = os.path.join(ROOT, "profiles", "r05_rehearsal2_line.json")
    line = json.loads(open(path).read())
    check_n_rank_line(line, 2)
    assert line["config"]["distinct_gpus"] == 1  # honest: a rehearsal, both ranks on one card


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_prints_exactly_one_json_line(tmp_path):
    """`bench.py --gpus 2` with two gloo ranks sharing this GPU: stdout is ONE line (c10d's banners go to stderr), and that line carries
    the headline, the multi-rank train step and the config-4 leg with the all-reduce measured alone."""
    import json
    import subprocess
    env = dict(os.environ, OSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--train-steps", "2", "--passes-in-flight", "1",
           "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    line = json.loads(lines[0])
    check_n_rank_line(line, 2)
    out = os.environ.get("OSR_REHEARSAL_OUT")
    if out:
        with open(out, "w") as fh:
            fh.write(lines[0] + "\n")
