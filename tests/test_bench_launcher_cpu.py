"""bench.py's own N-rank launcher: `python bench.py --gpus N` without RANK in the environment starts N
fresh child processes, waits, relays rank 0's JSON line and fails when a rank fails.  `--mode launchcheck` does no GPU work
(gloo process group, empty timed step), so this runs on the CPU box; the GPU modes run through the same launcher on the
GPU box (tests/test_bench_modes_gpu.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(PANN_BENCH_BACKEND="gloo", **(env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def test_launcher_starts_n_ranks_and_relays_rank0_line():
    p = _run("--gpus", "3", "--mode", "launchcheck", "--steps", "4", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # ONE JSON line, rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == 3 and j["ranks_seen"] == [0, 1, 2] and j["steps"] == 4 and j["warmup"] == 1
    assert j["launcher_pid"] is not None and j["backend"] == "gloo"


def test_launcher_fails_when_a_rank_fails():
    p = _run("--gpus", "2", "--mode", "launchcheck", env={"PANN_BENCH_FAIL_RANK": "1"})
    assert p.returncode != 0


def test_single_rank_needs_no_process_group():
    p = _run("--gpus", "1", "--mode", "launchcheck")
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["ranks_seen"] == [0] and j["launcher_pid"] is None


def test_world_size_must_match_gpus_flag():
    # the shape of the driver's launch: ranks exist already (RANK set); --gpus has to agree with the group
    e = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29733"}
    p = _run("--gpus", "2", "--mode", "launchcheck", env=e)
    assert p.returncode != 0 and "--gpus 2" in p.stderr
    p = _run("--gpus", "1", "--mode", "launchcheck", env=e)
    assert p.returncode == 0, p.stderr[-2000:]
