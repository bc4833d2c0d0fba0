"""Sharded index on the GPU box: 2 processes (gloo rendezvous, both using the one visible GPU for
compute), each builds + searches its id range with the product's DeviceIndex; the all-gathered,
merged top-k must equal the single-process merge of the same per-shard device results."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(shard):
    from parlayann_amd import DeviceIndex
    ix = DeviceIndex(shard, max_degree=32)
    ix.vamana_build(32, 64, 1.2, num_passes=1, seed=3)
    return ix


def _search(ix, queries, k, beam):
    r = ix.batch_search(queries, k=k, beam=beam)
    return r["ids"], r["dists"]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from parlayann_amd import datasets, distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = datasets.sift_like(20000, 128, seed=1, dtype=np.float16)
    Q = datasets.sift_like(200, 128, seed=2, dtype=np.float16)
    sh = D.ShardedIndex(X, _build, _search)
    ids, dists = sh.search(Q, 10, 64)
    q.put((rank, ids, dists))
    dist.barrier()
    sh.state.close()
    dist.destroy_process_group()


def test_two_rank_sharded_index(oracle):
    from parlayann_amd import datasets, distributed as D
    world, port = 2, 29600 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = datasets.sift_like(20000, 128, seed=1, dtype=np.float16)
    Q = datasets.sift_like(200, 128, seed=2, dtype=np.float16)
    per_i, per_d = [], []
    for r in range(world):
        lo, hi = D.shard_range(len(X), r, world)
        ix = _build(X[lo:hi])
        i, d = _search(ix, Q, 10, 64)
        ix.close()
        per_i.append(i + np.uint32(lo)); per_d.append(d)
    exp_i, exp_d = D.merge_topk(np.stack(per_i), np.stack(per_d), 10)
    for r in range(world):
        np.testing.assert_array_equal(res[r][1], exp_i)
        np.testing.assert_array_equal(res[r][2], exp_d)
    gt, gd = oracle.bruteforce_knn(X, Q, 50)
    assert oracle.recall(exp_i, gt, gd, 10) > 0.94
