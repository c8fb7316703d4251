"""`python bench.py --gpus N` must start its own ranks (the driver launches N = 1 that way): the
launcher re-executes under torch.distributed.run and relays ONE JSON line as the last line of stdout.  Exercised here on
CPU with a gloo group and a sleep in place of the train step (BENCH_SPAWN_SELFTEST=1); the torchrun form is run too."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_SPAWN_SELFTEST"] = "1"
    return env


def _check(out, n):
    lines = [ln for ln in out.strip().splitlines() if ln.strip()]
    line = json.loads(lines[-1])                       # the JSON line is the LAST line of stdout
    assert line["n_gpus"] == n and line["steps"] == 3 and line["warmup"] == 1 and line["selftest"] is True
    return lines


def test_plain_invocation_spawns_its_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _check(r.stdout, 2)
    assert len(lines) == 1                              # rank chatter went to stderr
    assert "rank 1 of 2 done" in r.stderr


def test_torchrun_invocation_still_works():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(r.stdout, 2)
