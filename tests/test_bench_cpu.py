"""bench.py launch forms (CPU): the bare `python bench.py --gpus N` command must turn itself into N RCCL ranks
(counterpart of the reference's `torchrun --nproc_per_node=N ... --num_gpus N`, gen3c_single_image.py:248-255)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_self_launch_argv_is_a_one_node_torchrun_of_this_script():
    import bench
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "2", "--warmup", "1"], port=29999)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29999"
    i = argv.index(str(ROOT / "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"]  # the user's flags travel unchanged
    # a free port is picked when none is given (no hard-coded rendezvous port)
    p1 = bench.self_launch_argv(2, [])
    assert 1024 < int(p1[p1.index("--master-port") + 1]) < 65536


def test_bare_multi_gpu_command_reexecs_under_the_launcher(monkeypatch):
    import bench
    calls = {}

    def fake_call(argv, env=None):
        calls["argv"], calls["env"] = argv, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("G3_BENCH_SHARE_GPU", "1")  # no GPU in the CPU container: skip the device-count gate
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert "--nproc-per-node=4" in calls["argv"] and calls["argv"][-4:] == ["--gpus", "4", "--steps", "3"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bare_multi_gpu_command_fails_loudly_without_enough_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "G3_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 8 but only" in (r.stderr + r.stdout)
