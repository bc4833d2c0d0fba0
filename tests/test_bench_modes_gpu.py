"""bench.py's multi-rank modes through bench.py's OWN launcher on the GPU box: two ranks share the
one visible GPU (gloo rendezvous; RCCL cannot put two ranks on one device), small tables.  What is checked is the launch
contract (n_gpus, ranks_seen, one JSON line), that every mode runs its sharded path, and that the sharded builds give the
graph of the one-rank build (checksum) -- kernel parity itself is tests/test_sharded_gpu.py's job."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e["PANN_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *[str(a) for a in argv]], env=e, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_query_mode_two_ranks():
    j = _bench("--gpus", 2, "--n", 20000, "--nq", 500, "--steps", 3, "--warmup", 1, "--R", 32, "--L", 64)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1] and j["scaling"] == "weak" and j["unit"] == "queries/s"
    assert j["recall_at_10"] > 0.9 and j["roofline"]["kernel"] == "beam_search_b64_kernel" and j["roofline"]["achieved"] > 0
    assert abs(j["value"] - 2 * 500 / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]       # whole-job rate: both ranks' queries


def test_c4_mode_two_ranks_one_collective_per_step():
    j = _bench("--gpus", 2, "--mode", "c4", "--n", 20000, "--nq", 400, "--steps", 3, "--warmup", 1, "--R", 32, "--L", 64)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1] and j["config"]["n_total"] == 40000
    assert j["collectives_per_step"] == 1.0 and j["recall_at_10"] > 0.9


def test_c3build_mode_graph_does_not_depend_on_the_rank_count():
    a = _bench("--gpus", 1, "--mode", "c3build", "--n", 30000, "--nq", 300, "--steps", 1, "--warmup", 0, "--R", 32, "--L", 64)
    b = _bench("--gpus", 2, "--mode", "c3build", "--n", 30000, "--nq", 300, "--steps", 2, "--warmup", 1, "--R", 32, "--L", 64)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["ranks_seen"] == [0, 1] and b["scaling"] == "strong"
    assert a["graph_checksum"] == b["graph_checksum"] and b["collectives_per_build"] > 0
    assert a["recall_at_10"] == b["recall_at_10"] and b["recall_at_10"] > 0.9


def test_c5build_mode_graph_does_not_depend_on_the_rank_count():
    a = _bench("--gpus", 1, "--mode", "c5build", "--n", 20000, "--nq", 300, "--trees", 5, "--steps", 1, "--warmup", 0)
    b = _bench("--gpus", 2, "--mode", "c5build", "--n", 20000, "--nq", 300, "--trees", 5, "--steps", 2, "--warmup", 1)
    assert b["n_gpus"] == 2 and b["ranks_seen"] == [0, 1]
    assert a["graph_checksum"] == b["graph_checksum"] and a["avg_degree"] == b["avg_degree"]
