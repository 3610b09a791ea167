"""GPU: context-parallel denoise step with the real HIP kernels == the non-CP step, 2 ranks sharing cuda:0 over gloo
(tools/cp_check.py; RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU bench)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> str:
    import socket
    with socket.socket() as s:  # hard-coded rendezvous ports collide on shared boxes
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_context_parallel_step_matches_single_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(ROOT / "tools" / "cp_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "[cp_check] OK" in r.stdout


def test_context_parallel_code_path_over_rccl_single_rank():
    """The same check with backend "nccl" (= RCCL) and ONE rank: RCCL refuses two ranks on one GPU, but a 1-rank group still runs
    every collective call of the context-parallel path (async all_gather_into_tensor of K and the V^T shards, Work.wait(), the
    NCCL-stream / compute-stream hand-off) through the real RCCL process group."""
    import os
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(ROOT / "tools" / "cp_check.py")]
    env = dict(os.environ, G3_CP_CHECK_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=env)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "[cp_check] OK" in r.stdout
