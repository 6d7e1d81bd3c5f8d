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
