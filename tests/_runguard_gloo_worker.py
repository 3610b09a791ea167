"""Worker of tests/test_bench_cpu.py::test_run_guard_relays_a_remote_failure_over_gloo_world2: rank 1 fails in its "timed" phase while rank 0 sits in a
collective that can no longer complete; rank 0's RunGuard thread must learn of it through the process group's store and print the value-null line."""
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    dist.init_process_group("gloo", init_method="env://")
    rank = dist.get_rank()
    guard = bench.RunGuard(rank, 2, {"metric": "m", "value": None, "n_gpus": 2})
    guard.progress.update(cp=dict(chosen=dict(head_groups=4)))
    guard.enter("timed", 120)
    if rank == 1:
        time.sleep(1.0)
        guard.fail("timed", RuntimeError("HIP error on rank 1"))
    t = torch.ones(1)
    dist.all_reduce(t)  # rank 1 never joins
    print("UNREACHABLE", flush=True)


if __name__ == "__main__":
    main()
