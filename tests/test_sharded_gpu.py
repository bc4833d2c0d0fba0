"""The multi-GPU paths on the GPU box with the PRODUCT's device code on every rank: 2 processes (gloo rendezvous, both
computing on the one visible GPU -- RCCL cannot put two ranks on one device; under gloo the collectives' device tensors are
staged through the host, under nccl they are not).  Every expectation comes from the ORACLE, not from the device code:

  * sharded index (C4 shape): merged top-k of the two shards == per-shard oracle searches merged by (dist, id), bit for bit
  * sharded Vamana build (one all-gather of the batch's rows per batch) == the oracle's single-process build, bit for bit
  * tree-parallel HCNNG on ONE resident DeviceIndex per rank == the oracle's single-process build, bit for bit
"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D_, NQ = 20000, 128, 400


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from parlayann_amd import DeviceIndex, datasets, distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = datasets.sift_like(N, D_, seed=1, dtype=np.float16)          # integer-valued: fp16 distances exact in any order
    Q = datasets.sift_like(NQ, D_, seed=2, dtype=np.float16)
    # ---- sharded index: every shard built and searched on the device, ids/dists merged on the device ----
    sh = D.DeviceShardedIndex(X, 32, lambda ix: ix.vamana_build(32, 64, 1.2, num_passes=1, seed=3))
    d_q = torch.from_numpy(Q.view(np.uint8).reshape(NQ, -1)).to(sh.dev)
    oi, od = sh.search(d_q, 10, 64)
    torch.cuda.synchronize()
    ids, dists = oi.cpu().numpy().view(np.uint32), od.cpu().numpy()
    sh_collectives = sh.collectives
    sh.close()
    # ---- sharded Vamana build: points + graph replicated, every batch split over the ranks ----
    Xb = X[:8000]
    ix = DeviceIndex(Xb, max_degree=32)
    info = D.device_vamana_build_sharded(ix, 32, 64, 1.2, num_passes=2, seed=5, min_split=16)
    Gv = ix.get_graph()
    ix.close()
    # ---- tree-parallel HCNNG on one resident index per rank ----
    Xh = datasets.sift_like(6000, 64, seed=7, dtype=np.uint8)
    ih = DeviceIndex(Xh, max_degree=15)
    D.device_hcnng_build_tree_parallel(ih, 5, 300, 3, seed=11)
    Gh = ih.get_graph()
    ih.close()
    q.put((rank, ids, dists, Gv, info["collectives"], info["bytes_gathered"], Gh, sh_collectives))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_ranks():
    world, port = 2, 29600 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < world:                       # a rank that died must fail the test at once, not after a long silent wait
        try:
            res.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"a rank exited with {dead} (its traceback is on stderr)"
            assert time.time() - t0 < 400, "ranks did not finish in 400 s"
            print(f"[two_ranks] waiting, {time.time() - t0:.0f}s", flush=True)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def test_sharded_index_equals_per_shard_oracle_merged(two_ranks, oracle):
    from parlayann_amd import datasets, distributed as D
    X = datasets.sift_like(N, D_, seed=1, dtype=np.float16)
    Q = datasets.sift_like(NQ, D_, seed=2, dtype=np.float16)
    per_i, per_d = [], []
    for r in range(2):
        lo, hi = D.shard_range(N, r, 2)
        G, _ = oracle.vamana_build(X[lo:hi], 32, 64, 1.2, num_passes=1, seed=3)
        rr = oracle.batch_search(X[lo:hi], G, queries=Q, k=10, beam=64)
        per_i.append(rr["ids"] + np.uint32(lo)); per_d.append(rr["dists"])
    exp_i, exp_d = D.merge_topk(np.stack(per_i), np.stack(per_d), 10)
    for r in range(2):
        np.testing.assert_array_equal(two_ranks[r][1], exp_i)
        np.testing.assert_array_equal(two_ranks[r][2], exp_d)
        assert two_ranks[r][7] == 1                    # ONE all-gather per search: packed [nq, 2k] rows (ids | distance bits)
    # Recall bar.  Round 1 asserted 0.97, measured 0.968 (200 queries) and lowered the bar to 0.94; the bar is back at 0.97
    # (400 queries: 0.9705, deterministic -- the results above are bit-exact) and the sharded answer is also held against
    # ONE graph over all points searched with the same beam (0.978: two 10K-point graphs are each a little worse than one
    # 20K-point graph at R = 32, L = 64, one pass; the gap must stay under one point).  Both numbers from the oracle:
    gt, gd = oracle.bruteforce_knn(X, Q, 50)
    Gall, _ = oracle.vamana_build(X, 32, 64, 1.2, num_passes=1, seed=3)
    r_single = oracle.recall(oracle.batch_search(X, Gall, queries=Q, k=10, beam=64)["ids"], gt, gd, 10)
    r_sharded = oracle.recall(exp_i, gt, gd, 10)
    assert r_sharded > 0.97 and r_sharded >= r_single - 0.01, (r_sharded, r_single)


def test_sharded_vamana_build_equals_oracle_single_process_build(two_ranks, oracle):
    from parlayann_amd import datasets, distributed as D
    X = datasets.sift_like(N, D_, seed=1, dtype=np.float16)[:8000]
    Q = datasets.sift_like(NQ, D_, seed=2, dtype=np.float16)
    Gs, _ = oracle.vamana_build(X, 32, 64, 1.2, num_passes=2, seed=5)
    cols = np.arange(32)[None, :]
    for r in range(2):
        G = two_ranks[r][3]
        np.testing.assert_array_equal(G[:, 0], Gs[:, 0])
        np.testing.assert_array_equal(np.where(cols < G[:, :1], G[:, 1:], 0), np.where(cols < Gs[:, :1], Gs[:, 1:], 0))
    _, bounds = D.build_schedule(len(X), 5)
    big = [(a, b) for a, b in bounds if b - a >= 16]
    assert two_ranks[0][4] == 2 * len(big) and two_ranks[0][5] == 2 * sum(2 * ((b - a + 1) // 2) * 32 * 4 for a, b in big)
    gt, gd = oracle.bruteforce_knn(X, Q, 50)
    assert oracle.recall(oracle.batch_search(X, Gs, queries=Q, k=10, beam=64)["ids"], gt, gd, 10) > 0.97


def test_tree_parallel_hcnng_on_resident_index_equals_oracle(two_ranks, oracle):
    from parlayann_amd import datasets
    Xh = datasets.sift_like(6000, 64, seed=7, dtype=np.uint8)
    Gh = oracle.hcnng_build(Xh, 5, 300, 3, seed=11)
    for r in range(2):
        np.testing.assert_array_equal(two_ranks[r][6], Gh)


def test_merge_topk_kernel_against_numpy(oracle):
    import ctypes as C
    import torch
    from parlayann_amd import _capi, distributed as D
    lib = _capi.load()
    rng = np.random.default_rng(5)
    for W, nq, k in ((2, 300, 10), (8, 1000, 100), (3, 7, 1)):
        ids = rng.integers(0, 1 << 31, (W, nq, k)).astype(np.uint32)
        d = np.sort(rng.integers(0, 50, (W, nq, k)).astype(np.float32), axis=2)       # many ties: the id decides
        ids[0, ::5, k - 1] = 0xFFFFFFFF; d[0, ::5, k - 1] = np.inf                     # short lists
        dev = torch.device("cuda", 0)
        ti = torch.from_numpy(ids.view(np.int32)).to(dev); td = torch.from_numpy(d).to(dev)
        oi = torch.empty((nq, k), dtype=torch.int32, device=dev); od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(lib.pann_merge_topk_dev(ti.data_ptr(), td.data_ptr(), W, nq, k, k, None, k, oi.data_ptr(), od.data_ptr(), st))
        torch.cuda.synchronize()
        ei, ed = D.merge_topk(ids, d, k)
        np.testing.assert_array_equal(oi.cpu().numpy().view(np.uint32), ei)
        np.testing.assert_array_equal(od.cpu().numpy(), ed)
        # the packed form DeviceShardedIndex gathers: [W][nq][ids | distance bits], ids local to their shard + a base per list
        base = (np.arange(W, dtype=np.uint32) * 1000)
        loc = ids.copy(); used = ids != 0xFFFFFFFF
        loc[used] = (ids - base[:, None, None])[used]
        packed = np.concatenate([loc.view(np.int32), d.view(np.int32)], axis=2)
        tp = torch.from_numpy(packed).to(dev); tb = torch.from_numpy(base.view(np.int32)).to(dev)
        oi.fill_(0); od.fill_(0)
        _capi.check(lib.pann_merge_topk_dev(tp.data_ptr(), tp.data_ptr() + 4 * k, W, nq, k, 2 * k, tb.data_ptr(), k, oi.data_ptr(),
                                            od.data_ptr(), st))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(oi.cpu().numpy().view(np.uint32), ei)
        np.testing.assert_array_equal(od.cpu().numpy(), ed)


def test_sharded_search_reads_the_status_word_and_relaunches_on_dropped_overflow(oracle):
    """a _dev launch that reports PANN_STATUS_DROPPED_OVERFLOW has not produced valid lists; the sharded
    path must grow the scratch and search again before it gathers (one rank here: the exchange is the identity).  Path-like
    graph, cut = 1.0, k = 1: 600+ visited vertices are cut from a frontier that never fills (test_edge_cases_gpu.py)."""
    import torch
    from parlayann_amd import distributed as D
    n = 1500
    X = np.zeros((n, 8), np.float32); X[:, 0] = np.arange(n)
    G = np.zeros((n, 5), np.uint32)
    for i in range(n):
        nb = [j for j in (i - 2, i - 1, i + 1, i + 2) if 0 <= j < n]
        G[i, 0] = len(nb); G[i, 1:1 + len(nb)] = nb
    Q = np.zeros((3, 8), np.float32); Q[:, 0] = [1499.0, 1400.5, 700.0]
    o = oracle.batch_search(X, G, queries=Q, k=1, beam=16, cut=1.0, out_k=1)
    assert o["visited_count"].max() > 600
    sh = D.DeviceShardedIndex(X, 4, lambda ix: ix.set_graph(G))
    assert sh.ix.dropped_capacity == 256
    d_q = torch.from_numpy(Q.view(np.uint8).reshape(3, -1)).to(sh.dev)
    oi, od = sh.search(d_q, 1, 16, cut=1.0, counters=True)
    torch.cuda.synchronize()
    assert sh.ix.dropped_capacity > 256 and sh.collectives == 1
    np.testing.assert_array_equal(oi.cpu().numpy().view(np.uint32), o["ids"])
    np.testing.assert_array_equal(od.cpu().numpy(), o["dists"])
    np.testing.assert_array_equal(sh.last_counters[0].cpu().numpy().view(np.uint32), o["visited_count"])
    sh.close()
