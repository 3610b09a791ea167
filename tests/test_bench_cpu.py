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


def test_cp_autotune_picks_the_fastest_candidate_and_reports(monkeypatch):
    """bench.py --gpus N tunes (head groups x attention kernel x collective schedule) by measurement and reports a `cp` object
    (VERDICT r2 next #2). Here with stand-ins for the net / denoiser / process group: the candidate grid, the all-ranks agreement
    (all_reduce MAX), the restore of the block limit and the keys of the report."""
    import torch
    import bench

    class FakeCPA:
        def __init__(self):
            self.cfg, self.stats, self.bytes_gathered = None, [], 12345678

        def configure(self, head_groups=None, kernel=None, schedule=None):
            self.cfg = (head_groups, kernel, schedule)

    class FakeNet:
        def __init__(self):
            self._cp_attn, self._tune_blocks = FakeCPA(), None

    class FakeDist:
        class ReduceOp:
            MAX = "max"
        reduced = 0

        @staticmethod
        def barrier():
            pass

        @classmethod
        def all_reduce(cls, t, op=None):
            cls.reduced += 1

    net = FakeNet()
    seen = []

    class FakeDen:
        def denoise_step(self, xt, step, c, u, g, aug, seed):
            assert net._tune_blocks == bench.CP_TUNE_BLOCKS and step == 0
            seen.append(net._cp_attn.cfg)

    # deterministic "timings": the winner is (2, "w4b", "local_first")
    clock = {"t": 0.0}
    cost = lambda cfg: 1.0 if cfg == (2, "w4b", "local_first") else 2.0 + cfg[0] * 0.01

    def fake_perf_counter():
        return clock["t"]

    def fake_sync():
        if seen:
            clock["t"] += cost(seen[-1]) / 2

    monkeypatch.setattr(bench.time, "perf_counter", fake_perf_counter)
    monkeypatch.setattr(torch.cuda, "synchronize", fake_sync)
    best, table, failed = bench.autotune_cp(net, FakeDen(), None, None, None, torch.device("cpu"), FakeDist)
    assert failed == [] and len(table) == 16 and {r["schedule"] for r in table} == {"gather_first", "local_first"} and {r["head_groups"] for r in table} == {1, 2, 4, 8}
    assert (best["head_groups"], best["kernel"], best["schedule"]) == (2, "w4b", "local_first")
    assert net._cp_attn.cfg == (2, "w4b", "local_first") and net._tune_blocks is None and FakeDist.reduced == 16 * 3  # per candidate: 2 barrier-agreements + the result

    class T:
        def __init__(self, ms):
            self.ms = ms

        def elapsed_ms(self):
            return self.ms

    net._cp_attn.stats = [("wait", 0, T(0.5)), ("wait", 1, T(0.1))] * 2
    rep = bench.cp_report(net._cp_attn, [({}, 10.0)] * 4, [({}, 3.0)] * 8, steps=2, rccl_ranks=8)
    for key in ("rccl_ranks", "attention_ms_per_step", "gemm_ms_per_step", "exposed_collective_wait_ms_per_step", "collective_waits_per_step",
                "worst_single_wait_ms", "gathered_bytes_per_step", "attention_launches_per_step"):
        assert key in rep, key
    assert rep["exposed_collective_wait_ms_per_step"] == 0.6 and rep["attention_ms_per_step"] == 20.0 and rep["gemm_ms_per_step"] == 12.0
    assert rep["gathered_bytes_per_step"] == 12345678 // 2 and rep["worst_single_wait_ms"] == 0.5


def _autotune_fakes(monkeypatch, fail_on=None, fail_everything=False):
    import torch
    import bench

    class FakeCPA:
        def __init__(self):
            self.cfg, self.stats, self.bytes_gathered, self.effective = None, [], 0, None

        def configure(self, head_groups=None, kernel=None, schedule=None):
            self.cfg = (head_groups, kernel, schedule)

    class FakeNet:
        def __init__(self):
            self._cp_attn, self._tune_blocks = FakeCPA(), None

    class FakeDist:
        class ReduceOp:
            MAX = "max"

        @staticmethod
        def barrier():
            pass

        @staticmethod
        def all_reduce(t, op=None):
            pass

    net = FakeNet()

    class FakeDen:
        def denoise_step(self, xt, step, c, u, g, aug, seed):
            if fail_everything or net._cp_attn.cfg == fail_on:
                raise RuntimeError("HIP error: invalid configuration argument")
            net._cp_attn.effective = dict(schedule="gather_first", kernel="wave8", head_groups=net._cp_attn.cfg[0])  # e.g. a silent fallback

    monkeypatch.setattr(torch.cuda, "synchronize", lambda: None)
    return bench, net, FakeDen(), FakeDist


def test_cp_autotune_drops_a_raising_candidate_and_keeps_going(monkeypatch):
    """one raising candidate must not cost the run. It is dropped (and listed), the others are still timed, the table says what each
    candidate actually RAN (ContextParallelAttention.effective - ADVICE r3: a requested local_first / w4b may fall back silently)."""
    import torch
    bench, net, den, fdist = _autotune_fakes(monkeypatch, fail_on=(8, "wave8", "local_first"))
    progress = {}
    best, table, failed = bench.autotune_cp(net, den, None, None, None, torch.device("cpu"), fdist, rank=0, progress=progress)
    assert len(table) == 15 and len(failed) == 1
    assert (failed[0]["head_groups"], failed[0]["kernel"], failed[0]["schedule"]) == (8, "wave8", "local_first") and "invalid configuration" in failed[0]["error"]
    assert best is not None and net._tune_blocks is None
    assert all(r["ran"] == dict(kernel="wave8", schedule="gather_first", head_groups=r["head_groups"]) for r in table)
    assert progress["candidate"] == (8, "wave8", "local_first")  # the watchdog's line would name the last candidate entered


def test_cp_autotune_with_no_survivor_configures_the_fixed_fallback(monkeypatch):
    import torch
    bench, net, den, fdist = _autotune_fakes(monkeypatch, fail_everything=True)
    best, table, failed = bench.autotune_cp(net, den, None, None, None, torch.device("cpu"), fdist)
    assert best is None and table == [] and len(failed) == 16
    assert net._cp_attn.cfg == bench.CP_FALLBACK and net._tune_blocks is None


def test_injection_hook_parses(monkeypatch):
    import bench
    monkeypatch.setenv("G3_BENCH_INJECT", "autotune:4,w4b,local_first:1")
    assert bench._injected("autotune", 1, (4, "w4b", "local_first")) and not bench._injected("autotune", 0, (4, "w4b", "local_first"))
    assert not bench._injected("autotune", 1, (2, "w4b", "local_first")) and not bench._injected("timed", 1)
    monkeypatch.setenv("G3_BENCH_INJECT", "timed:1")
    assert bench._injected("timed", 1) and not bench._injected("timed", 0)
    monkeypatch.delenv("G3_BENCH_INJECT")
    assert not bench._injected("timed", 1)


def test_run_guard_prints_a_null_value_line_when_a_phase_overruns():
    """RunGuard in a child process: no process group (store = None), a 'timed' phase with a 1 s deadline that never finishes -> rank 0 prints a
    JSON line with value null, the phase and the progress it was given, and the process leaves with code 1."""
    import json
    code = (
        "import sys, time; sys.path.insert(0, %r); import bench\n"
        "g = bench.RunGuard(0, 8, {'metric': 'm', 'value': None, 'n_gpus': 8})\n"
        "g.progress.update(candidate=(4, 'w4b', 'local_first'))\n"
        "g.enter('timed', 1.0)\n"
        "time.sleep(30)\n" % str(ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] is None and line["n_gpus"] == 8 and line["failed_phase"] == "timed" and "deadline" in line["error"]
    assert line["progress"]["candidate"] == [4, "w4b", "local_first"]


def _json_objects(text):
    """Every top-level JSON object found anywhere in `text`, in order - robust against two processes' output sharing a line."""
    import json
    dec, out, i = json.JSONDecoder(), [], text.find("{")
    while i >= 0:
        try:
            obj, end = dec.raw_decode(text, i)
            if isinstance(obj, dict):
                out.append(obj)
            i = text.find("{", end)
        except ValueError:
            i = text.find("{", i + 1)
    return out


def test_json_objects_helper_survives_interleaved_output():
    assert _json_objects('noise {"a": 1}WORKER {"b": {"c": 2}}\n{bad {"d": 3}') == [{"a": 1}, {"b": {"c": 2}}, {"d": 3}]


def test_autotune_failure_agreement_over_gloo_world2(tmp_path):
    """The N > 1 agreement protocol of bench.autotune_cp over a REAL 2-rank gloo process group (CPU): candidate (4, w4b, local_first) raises on rank 1
    only, BEFORE its step (rank 0 would otherwise enter the step's exchange alone and hang); candidate (2, wave8, gather_first) raises on both ranks
    after the exchange. Both ranks must finish, drop exactly these two candidates, time the other 14 and configure the same winner."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, G3_BENCH_INJECT="autotune:4,w4b,local_first:1", WORKER_FAIL_IN_STEP="1", OMP_NUM_THREADS="1", WORKER_OUT_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        str(ROOT / "tests" / "_autotune_gloo_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    # each worker writes its own file (the two children share one stdout pipe: their lines may interleave)
    files = sorted(tmp_path.glob("worker*.json"))
    assert len(files) == 2, r.stdout[-2000:] + r.stderr[-3000:]
    outs = sorted((json.loads(f.read_text()) for f in files), key=lambda o: o["rank"])
    for o in outs:
        assert o["n_table"] == 14 and sorted(map(tuple, o["failed"])) == [(2, "wave8", "gather_first"), (4, "w4b", "local_first")]
    assert outs[0]["best"] == outs[1]["best"] and outs[0]["cfg"] == outs[1]["cfg"] and outs[0]["best"] is not None


def test_run_guard_relays_a_remote_failure_over_gloo_world2():
    """RunGuard over a real 2-rank gloo group (CPU): rank 1 raises in its timed phase, rank 0 is stuck in an all-reduce rank 1 never joins. Rank 0 must
    print ONE JSON line - value null, the phase, rank 1's exception, the chosen configuration - and the job must end within seconds, not at a timeout."""
    import json
    import socket
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        str(ROOT / "tests" / "_runguard_gloo_worker.py")], capture_output=True, text=True, timeout=300, env=env)
    took = time.time() - t0
    assert r.returncode != 0 and "UNREACHABLE" not in r.stdout
    objs = _json_objects(r.stdout)  # (not line-split: the two children share one pipe)
    assert len(objs) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    out = objs[0]
    assert out["value"] is None and out["failed_phase"] == "timed" and "rank 1" in out["error"] and "HIP error on rank 1" in out["error"]
    assert out["progress"]["cp"]["chosen"]["head_groups"] == 4 and took < 120


def test_pmc_csv_parser_means_the_named_kernels_counter():
    import bench
    hdr = '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
    row = lambda name, c, v: f'1,1,"Agent 2",1,2,2,64,5,"{name}",256,0,0,128,256,32,"{c}",{v},10,20\n'
    text = hdr + row("void (anonymous namespace)::flash_attn_fwd_w4b_kernel<true>(AttnParams)", "FETCH_SIZE", 7.0e6) + \
        row("void (anonymous namespace)::flash_attn_fwd_w4b_kernel<true>(AttnParams)", "FETCH_SIZE", 8.0e6) + \
        row("void transpose_v_kernel(x)", "FETCH_SIZE", 1.0) + row("void (anonymous namespace)::flash_attn_fwd_w4b_kernel<true>(AttnParams)", "WRITE_SIZE", 4.0e6)
    assert bench.parse_pmc_csv(text, "flash_attn_fwd_w4b", "FETCH_SIZE") == (7.5e6, 2)
    assert bench.parse_pmc_csv(text, "flash_attn_fwd_w4b", "WRITE_SIZE") == (4.0e6, 1)
    assert bench.parse_pmc_csv(text, "gemm", "FETCH_SIZE") == (None, 0)
