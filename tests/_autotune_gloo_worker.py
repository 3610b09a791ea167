"""Worker of tests/test_bench_cpu.py::test_autotune_failure_agreement_over_gloo_world2: bench.autotune_cp over a REAL 2-rank gloo process group on CPU.
The net / denoiser are stand-ins (no GPU here), the collectives are real: every "denoise step" all-reduces, like the K / V exchange of a real step, so a
rank that skipped a step while its peer entered it would hang the run - which is what the barrier-plus-agreement protocol of autotune_cp prevents."""
import json
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


class CPA:
    def __init__(self):
        self.cfg, self.effective = None, None

    def configure(self, head_groups=None, kernel=None, schedule=None):
        self.cfg = (head_groups, kernel, schedule)


class Net:
    def __init__(self):
        self._cp_attn, self._tune_blocks = CPA(), None


def main():
    dist.init_process_group("gloo", init_method="env://")
    rank = dist.get_rank()
    torch.cuda.synchronize = lambda *a, **k: None  # no GPU in this test
    net = Net()

    class Den:
        def denoise_step(self, xt, step, c, u, g, aug, seed):
            t = torch.ones(4)
            dist.all_reduce(t)  # the step's exchange: both ranks must be in the same step
            assert float(t[0]) == dist.get_world_size()
            if os.environ.get("WORKER_FAIL_IN_STEP") and net._cp_attn.cfg == (2, "wave8", "gather_first"):
                raise RuntimeError("symmetric failure after the exchange (both ranks raise at the same point)")
            net._cp_attn.effective = dict(schedule=net._cp_attn.cfg[2], kernel=net._cp_attn.cfg[1], head_groups=net._cp_attn.cfg[0])

    best, table, failed = bench.autotune_cp(net, Den(), None, None, None, torch.device("cpu"), dist, rank=rank, progress={})
    out = dict(rank=rank, n_table=len(table), failed=[(f["head_groups"], f["kernel"], f["schedule"]) for f in failed],
               best=None if best is None else (best["head_groups"], best["kernel"], best["schedule"]), cfg=net._cp_attn.cfg)
    # one file per rank: two torch.distributed.run children share the parent's stdout pipe and their lines can land on one line
    Path(os.environ["WORKER_OUT_DIR"], f"worker{rank}.json").write_text(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
