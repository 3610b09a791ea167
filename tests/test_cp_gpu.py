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


def test_cp_attention_two_stream_one_wave_kernel_single_rank():
    """ContextParallelAttention at a size where the head groups together fill the chip evenly (14 080 local tokens, 32 heads): the groups
    alternate between two streams and run the one-wave-per-SIMD kernel on segmented V^T. With a 1-rank group the gathered K / V are the
    local ones, so the result must equal one plain attention call over all heads (same kernel arithmetic per head => bitwise)."""
    import os
    import torch
    import torch.distributed as dist
    from gen3c_amd import ops
    from gen3c_amd.parallel import ContextParallelAttention
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", _free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        group = dist.new_group([0])
        S, B, H = 14080, 1, 32
        g = torch.Generator(device=dev).manual_seed(21)
        q, k, v = (torch.randn(S * B, H * 128, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
        # gloo moves the (CPU-staged) tensors of the 1-rank all-gather; the kernels and streams are the product ones
        cpa = ContextParallelAttention(group, head_groups=4)
        out = cpa(q, k, v, S, B, H)
        ref = ops.flash_attn(q, k, ops.transpose_v(v, S, B, H), S, S, B, H, variant=11)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        # the kernel choice is per call: explicit "wave8" gives the 8-wave kernel's bits, and the library's process-wide option is untouched
        from gen3c_amd import _lib
        name0 = _lib.load().g3_flash_attn_kernel_name(S, S, B, H)
        out8 = ContextParallelAttention(group, head_groups=4, kernel="wave8")(q, k, v, S, B, H)
        assert torch.equal(out8, ops.flash_attn(q, k, ops.transpose_v(v, S, B, H), S, S, B, H, variant=4))
        assert _lib.load().g3_flash_attn_kernel_name(S, S, B, H) == name0
        # local-first schedule with one rank: the local part is everything, merged alone (fp32 partial -> bf16)
        cpl = ContextParallelAttention(group, head_groups=4, schedule="local_first")
        cpl.stats = []
        outl = cpl(q, k, v, S, B, H)
        torch.cuda.synchronize()
        assert float((outl.float() - ref.float()).norm() / ref.float().norm()) < 1e-3
        assert len(cpl.stats) == 4 and all(kind == "wait" and tm.elapsed_ms() >= 0 for kind, _g, tm in cpl.stats)
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_multi_gpu_path_autotunes_and_prints_cp_object():
    """`python bench.py --gpus 2` through its own launcher with both ranks on cuda:0 over gloo (plumbing run of the driver's N > 1 command,
    reduced size): the context-parallel autotune runs, the timed region runs on the winner, and the JSON line carries the `cp` diagnosis."""
    import json
    import os
    env = dict(os.environ, G3_BENCH_BACKEND="gloo", G3_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--blocks", "4", "--latent", "8,16,24",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["output_finite"] and out["scaling"] == "strong"
    cp = out["cp"]
    assert cp["chosen"]["head_groups"] in (1, 2, 4, 8) and cp["chosen"]["kernel"] in ("w4b", "wave8") and cp["chosen"]["schedule"] in ("gather_first", "local_first")
    assert len(cp["autotune_ms"]) == 16 and min(r_["ms"] for r_ in cp["autotune_ms"]) == cp["chosen"]["ms"]
    for key in ("attention_ms_per_step", "gemm_ms_per_step", "exposed_collective_wait_ms_per_step", "gathered_bytes_per_step", "rccl_ranks"):
        assert key in cp, key
    # 4 blocks x (K + V shard of the other rank): 2 forwards batched as B = 2 -> rows = 384 * 2, 4096 features, bf16
    assert cp["gathered_bytes_per_step"] == 4 * 2 * (384 * 2 * 4096 * 2)
    print(json.dumps(cp)[:1500])


def _run_bench_share(extra_env, timeout=900):
    import os
    env = dict(os.environ, G3_BENCH_BACKEND="gloo", G3_BENCH_SHARE_GPU="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--blocks", "4", "--latent", "8,16,24",
           "--no-cpu-baseline", "--no-extras"]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT), env=env)


def test_bench_multi_gpu_survives_a_failing_autotune_candidate():
    """candidate (4, w4b, local_first) raises on rank 1 only - both ranks must drop it together (the failure flag rides on the
    agreement all_reduce), finish the autotune on the other 15, run the timed region and print the normal line, with the casualty listed."""
    import json
    r = _run_bench_share({"G3_BENCH_INJECT": "autotune:4,w4b,local_first:1"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cp = out["cp"]
    assert out["value"] is not None and out["output_finite"] and len(cp["autotune_ms"]) == 15
    assert [(f["head_groups"], f["kernel"], f["schedule"]) for f in cp["autotune_failed"]] == [(4, "w4b", "local_first")]
    assert all("ran" in row for row in cp["autotune_ms"])
    rk = cp["rank_own_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["min"] <= rk["max"]
    print(json.dumps(cp["autotune_failed"]), json.dumps(rk))


def test_bench_multi_gpu_prints_a_null_line_when_the_timed_region_fails_on_another_rank():
    """Rank 1 raises inside the timed region (rank 0 is then stuck in a collective): rank 0 must still print ONE JSON line - value null, the
    phase, the chosen configuration, rank 1's exception (delivered through the process group's key-value store) - and the run must end at once
    with a non-zero exit code, not after the process-group timeout."""
    import json
    import time
    t0 = time.time()
    r = _run_bench_share({"G3_BENCH_INJECT": "timed:1"}, timeout=600)
    took = time.time() - t0
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["value"] is None and out["n_gpus"] == 2 and out["failed_phase"] == "timed"
    assert "rank 1" in out["error"] and "injected failure" in out["error"]
    assert out["progress"]["cp"]["chosen"]["head_groups"] in (1, 2, 4, 8)
    assert took < 300, f"the failed run took {took:.0f} s to end"
