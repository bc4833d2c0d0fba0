"""world_size-2 gloo tests of the N>1 path (bench timing contract, query sharding, sharded index
fan-out + all-gather + merge, graph stitch).  CPU only: the per-shard searcher is the oracle."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle_api
    from parlayann_amd import datasets, distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_api.load()
    X = datasets.sift_like(4000, 32, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(60, 32, seed=2, dtype=np.uint8)

    def build(shard):
        G, _ = o.vamana_build(shard, 16, 32, 1.2, seed=3, threads=2)
        return (shard, G)

    def search(state, queries, k, beam):
        r = o.batch_search(state[0], state[1], queries=queries, k=k, beam=beam, threads=2)
        return r["ids"], r["dists"]

    sh = D.ShardedIndex(X, build, search)
    ids, dists = sh.search(Q, 10, 32)
    # timing contract: max over ranks (rank 1 is the slow one)
    import time
    el = D.timed_steps(lambda: time.sleep(0.01 * (rank + 1)), steps=3, warmup=1)
    # stitch: every rank contributes its rows (global ids), all ranks end with the same full graph
    per = (len(X) + world - 1) // world
    rows = np.zeros((per, 17), np.uint32)
    G = sh.state[1].copy(); G[:, 1:] += np.uint32(sh.lo)
    rows[:len(G)] = G
    full = D.stitch_graph(rows, len(X))
    # HCNNG with the trees split over the ranks == the single-process build
    Xh = X[:1500]
    Gh = D.hcnng_build_tree_parallel(lambda t: o.hcnng_build(Xh, 1, 100, 3, seed=11 + t, threads=2), len(Xh), 5, 3)
    q.put((rank, ids, dists, el, full[:, 0].sum(), D.shard_range(len(X), rank, world), Gh))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharded_search_timing_and_stitch(oracle):
    from parlayann_amd import datasets, distributed as D
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation: per-shard oracle results merged by (dist, id)
    X = datasets.sift_like(4000, 32, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(60, 32, seed=2, dtype=np.uint8)
    per_ids, per_d, degsum = [], [], 0
    for r in range(world):
        lo, hi = D.shard_range(len(X), r, world)
        assert res[r][5] == (lo, hi)
        G, _ = oracle.vamana_build(X[lo:hi], 16, 32, 1.2, seed=3, threads=2)
        degsum += int(G[:, 0].sum())
        rr = oracle.batch_search(X[lo:hi], G, queries=Q, k=10, beam=32, threads=2)
        per_ids.append(rr["ids"] + np.uint32(lo)); per_d.append(rr["dists"])
    exp_i, exp_d = D.merge_topk(np.stack(per_ids), np.stack(per_d), 10)
    for r in range(world):
        np.testing.assert_array_equal(res[r][1], exp_i)
        np.testing.assert_array_equal(res[r][2], exp_d)
        assert res[r][4] == degsum
    assert abs(res[0][3] - res[1][3]) < 1e-9 and res[0][3] >= 3 * 0.02      # both ranks report the slow rank's time
    Gh = oracle.hcnng_build(X[:1500], 5, 100, 3, seed=11, threads=2)
    for r in range(world):
        np.testing.assert_array_equal(res[r][6], Gh)
    # the merged answer is a real top-k: close to brute force over the whole set
    gt, gd = oracle.bruteforce_knn(X, Q, 50)
    assert oracle.recall(exp_i, gt, gd, 10) > 0.9


def test_merge_topk_orders_by_dist_then_id():
    from parlayann_amd import distributed as D
    ids = np.array([[[5, 9, 1]], [[7, 2, 8]]], np.uint32)
    d = np.array([[[1.0, 2.0, 3.0]], [[1.0, 2.0, 2.5]]], np.float32)
    i, dd = D.merge_topk(ids, d, 4)
    assert i.tolist() == [[5, 7, 2, 9]] and dd.tolist() == [[1.0, 1.0, 2.0, 2.0]]


# ---- sharded Vamana build: every batch split over the ranks, ONE all-gather of the batch's rows ----

def _connected_from(G, start=0):
    n = len(G)
    seen = np.zeros(n, bool); seen[start] = True
    frontier = [start]
    while frontier:
        nxt = []
        for v in frontier:
            for w in G[v, 1:1 + G[v, 0]]:
                if not seen[w]:
                    seen[w] = True; nxt.append(int(w))
        frontier = nxt
    return int(seen.sum())


def _build_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import oracle_api
    from parlayann_amd import datasets, distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_api.load()
    X = datasets.sift_like(3000, 32, seed=1, dtype=np.uint8)
    R, L = 16, 32
    G = np.zeros((len(X), R + 1), np.uint32)                       # this rank's replica of the graph
    calls = {"a": 0, "a_points": 0}

    def phase_a(ids, alpha):                                       # the oracle as the per-rank worker (no GPU here)
        calls["a"] += 1; calls["a_points"] += ids.numel()
        rows = o.vamana_phase_a(X, G, ids.numpy().view(np.uint32), R, L, alpha, threads=2)
        return torch.from_numpy(rows.view(np.int32))

    def phase_b(ids, rows, alpha):
        o.vamana_phase_b(X, G, ids.numpy().view(np.uint32), rows.numpy().view(np.uint32), R, alpha, threads=2)

    info = D.vamana_build_sharded(len(X), R, L, 1.2, 2, 9, phase_a, phase_b, finish=lambda: o.sort_neighbors(X, G), min_split=8)
    q.put((rank, G, info, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharded_vamana_build_equals_single_process(oracle):
    from parlayann_amd import datasets, distributed as D
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_build_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = datasets.sift_like(3000, 32, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(200, 32, seed=2, dtype=np.uint8)
    Gs, _ = oracle.vamana_build(X, 16, 32, 1.2, num_passes=2, seed=9, threads=4)          # the single-process build
    for r in range(world):
        np.testing.assert_array_equal(res[r][1], Gs)                                      # replicas identical, == single build
    # the work really was split: each rank searched + pruned about half of the 2 x 3000 inserts (the small first batches whole)
    pts = [res[r][3]["a_points"] for r in range(world)]
    assert all(3000 <= p <= 3300 for p in pts), pts
    perm, bounds = D.build_schedule(len(X), 9)
    np.testing.assert_array_equal(perm, oracle.permutation(len(X), 9))
    big = [(a, b) for a, b in bounds if b - a >= 8]
    assert res[0][2]["collectives"] == 2 * len(big)
    assert res[0][2]["bytes_gathered"] == 2 * sum(world * ((b - a + world - 1) // world) * 16 * 4 for a, b in big)
    # searchable: connected from the start vertex, recall that of the single-process graph (identical graph => identical)
    # (a directed Vamana graph need not reach every vertex from the start: the single-process graph reaches the same set)
    reach = _connected_from(res[0][1])
    assert reach == _connected_from(Gs) and reach > 0.98 * len(X)
    gt, gd = oracle.bruteforce_knn(X, Q, 50)
    rs = oracle.recall(oracle.batch_search(X, res[0][1], queries=Q, k=10, beam=64)["ids"], gt, gd, 10)
    r1 = oracle.recall(oracle.batch_search(X, Gs, queries=Q, k=10, beam=64)["ids"], gt, gd, 10)
    assert abs(rs - r1) <= 0.001 and rs > 0.9


def test_batch_schedule_matches_the_reference_rule():
    """vamana/index.h:206-209,223-234: prefix doubling up to max_batch = min(.02 n, 1e6), then fixed batches"""
    from parlayann_amd import distributed as D
    perm, bounds = D.build_schedule(10000, 1)
    assert sorted(perm.tolist()) == list(range(10000))
    assert bounds[:4] == [(0, 1), (1, 3), (3, 7), (7, 15)]
    sizes = [b - a for a, b in bounds]
    assert max(sizes) == 200 and bounds[-1][1] == 10000 and all(bounds[i][1] == bounds[i + 1][0] for i in range(len(bounds) - 1))
    # doubling batches while 2^inc <= max_batch (so 128 is the last doubling size), then 200 per batch
    assert sizes[:8] == [1, 2, 4, 8, 16, 32, 64, 128] and set(sizes[8:-1]) == {200}
