"""`bench.py --gpus N` without a launcher around it starts its own N ranks.  Checked here on CPU with the
`--rendezvous-only` mode over gloo: the ranks are fresh child processes, they form one process group, the line reports
how many took part, a disagreement between `--gpus` and WORLD_SIZE is refused, and a rank that dies takes the run down
with a non-zero status instead of leaving the others waiting.  (The rendering of a two-rank split against the single
frame is tests/test_distributed_gloo.py, with the oracle standing in for the GPU kernels.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PT_BENCH_BACKEND"] = "gloo"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_counts_them():
    r = _run(["--gpus", "2", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE line on stdout, whatever the libraries print
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2
    # the ranks build their trees at the same time: each gets its share of the cores, not all of them
    assert int(line["build_threads_per_rank"]) >= 1
    assert int(line["build_threads_per_rank"]) <= max(1, min(16, os.cpu_count() or 1) // 2)


def test_gpus_must_agree_with_world_size():
    r = _run(["--gpus", "2", "--rendezvous-only"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr
    assert not r.stdout.strip()


def test_a_failing_rank_fails_the_run():
    r = _run(["--gpus", "3", "--rendezvous-only"], {"PT_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert "rank 1 of 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_one_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["n_ranks_seen"] == 1
