"""`python bench.py --gpus N` without a launcher starts its own ranks (BASELINE config 4's entry point). Checked here
without a GPU through --dry-run: the parent spawns N rank processes (RANK / WORLD_SIZE / MASTER_* set, rendezvous on
127.0.0.1), they gather one row block over gloo, rank 0's JSON line - and only that - reaches stdout, and a failing
rank fails the whole call."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_two_ranks_and_relay_one_json_line():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # gloo's connection notes etc. go to stderr
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["gathered_rows"] == 16
    assert d["row_sum"] == float(sum(range(16)))           # both shards arrived, in env-id order


def test_a_failing_rank_fails_the_call():
    r = _run("--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1")
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr


def test_under_a_launcher_the_ranks_are_the_launchers():
    """RANK / WORLD_SIZE already set (torch.distributed.run's job): no second generation of processes."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
