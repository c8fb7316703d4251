"""The REAL `python bench.py --gpus N` code path with N > 1 processes -- spawn -> torch group -> ctx_dp_init -> warm-up and timed
ctx_dp_train_step's -> sustained and sampled legs -> the comm block -> ONE JSON line -- executed on the one GPU a gpurun box has
(the first time an 8-GPU node runs the SCALE command must not be the first time that code runs).

BENCH_ONE_GPU=1 puts every rank on device 0; two ranks cannot share a device under RCCL, so the torch group is gloo and the
collectives behind the C ABI go through tests/fake_rccl (CTX_RCCL_LIB), as in tests/test_gpu_dp_two_ranks.py.  Everything else is the
shipped bench.  No scaling claim follows from the numbers: N processes time-share one GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")

pytestmark = pytest.mark.gpu


def run_bench(n, extra=(), timeout=900):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(FAKE), "tests/fake_rccl/libfakerccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_ONE_GPU="1", CTX_RCCL_LIB=FAKE, FAKE_RCCL_TIMEOUT_S="300")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-split-leg", "--no-secondary", "--sustained-s", "0.2", "--kernel-iters", "1", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                        # rank chatter and librccl banners went to stderr: ONE line on stdout
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_runs_end_to_end_on_one_device(n):
    line, err = run_bench(n)
    assert line["n_gpus"] == n and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["dtype"] == "f32"
    assert line["config"]["per_gpu_batch"] == 8 and line["config"]["global_batch"] == 8 * n
    assert line["config"]["parallelism"].startswith(f"dp{n}")
    assert line["one_gpu_stand_in"] is True              # marked: N processes on ONE device, never a scaling figure
    # whole-job value from the max-over-ranks time of exactly K steps
    assert abs(line["value"] - 3 * 8 * n / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    comm = line["comm"]
    assert "error" not in comm, comm
    assert comm["client"] == "cabi" and comm["overlap"] is True
    assert comm["ctx_dp_world_by_rank"] == [n] * n       # every rank's handle joined a group of N behind the C ABI
    assert comm["payload_MB"] > 100 and comm["allreduce_ms"] > 0 and comm["compute_ms_per_step"] > 0
    assert line["sampled"] and "error" not in line["sampled"], line["sampled"]
    assert line["sampled"]["global_batch"] == 8 * n and line["sampled"]["entry"] == "ctx_dp_train_step_sampled"
    assert line["sustained"]["steps"] >= 3
    assert line["roofline"]["bound"] == "mfma" and line.get("cpu_baseline") is None     # (rank 0 at N = 1 only)
    import math
    assert math.isfinite(line["loss_after"])
