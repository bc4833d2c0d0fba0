"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL
(backend "nccl") on the GPU box, gloo in CPU tests.  The reference is single-process; nothing here
translates reference code.

  * replicated index, sharded queries  : no collective on the data path (bench.py); only a barrier
    and a max-reduce of the elapsed time (`timed_steps`).
  * sharded index (C4)                 : every rank searches ALL queries on its own id range; one
    all-gather of the per-rank top-k (k*8 B per query per rank) and a merge by (dist,id).
  * sharded build stitch               : ONE all-gather of the adjacency rows (n/W x (R+1) x 4 B per rank).
"""
import time

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """contiguous id range [lo, hi) of `rank`"""
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def timed_steps(step, steps, warmup, sync=lambda: None, device=None):
    """bench.py's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier + device sync on both sides; returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(); barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def merge_topk(ids, dists, k):
    """ids/dists: [W, nq, k] per-shard results with GLOBAL ids -> [nq, k] smallest by (dist, id)."""
    ids = np.asarray(ids); dists = np.asarray(dists)
    W, nq, kk = ids.shape
    allid = np.transpose(ids, (1, 0, 2)).reshape(nq, W * kk)
    alld = np.transpose(dists, (1, 0, 2)).reshape(nq, W * kk)
    out_i = np.empty((nq, k), np.uint32); out_d = np.empty((nq, k), np.float32)
    for i in range(nq):
        order = np.lexsort((allid[i], alld[i]))[:k]
        out_i[i] = allid[i][order]; out_d[i] = alld[i][order]
    return out_i, out_d


def all_gather_array(a, device=None):
    """all-gather equally-shaped numpy arrays (through GPU tensors when `device` is given: RCCL)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a[None]
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1))
    if device is not None:
        t = t.to(device)
    W = dist.get_world_size()
    out = torch.empty(W * t.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)          # one collective; concatenated layout [W * bytes]
    return out.cpu().numpy().view(a.dtype).reshape((W,) + a.shape)


class ShardedIndex:
    """Base points split into `world` contiguous id ranges, one sub-index (own sub-graph, local ids)
    per rank.  `local_build(points_shard) -> state` and `local_search(state, queries, k, beam) ->
    (local ids [nq,k], dists [nq,k])` are injected (the product passes DeviceIndex methods; CPU
    tests pass the oracle)."""

    def __init__(self, points, local_build, local_search, rank=None, world=None, device=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.n = len(points)
        self.lo, self.hi = shard_range(self.n, self.rank, self.world)
        self.device = device
        self.local_search = local_search
        self.state = local_build(points[self.lo:self.hi])

    def search(self, queries, k, beam):
        lid, d = self.local_search(self.state, queries, k, beam)
        gid = (lid.astype(np.int64) + self.lo).astype(np.uint32)      # ids offset by the shard base
        gid[lid == 0xFFFFFFFF] = 0xFFFFFFFF
        ids = all_gather_array(gid, self.device)
        dists = all_gather_array(np.ascontiguousarray(d, dtype=np.float32), self.device)
        return merge_topk(ids, dists, k)


def stitch_graph(local_rows, n, device=None):
    """ONE all-gather of adjacency rows: local_rows is this rank's [per, R+1] slab (reference row
    layout, ids already global), padded to the common shard height; returns the full [n, R+1] graph."""
    rows = all_gather_array(local_rows, device)
    return rows.reshape(-1, local_rows.shape[1])[:n]


def hcnng_build_tree_parallel(build_tree, n, num_clusters, mst_deg, device=None):
    """HCNNG over ranks: the cluster trees are independent
    (clusterEdge.h:146-153), so rank r builds trees r, r+W, r+2W, ...; `build_tree(t)` returns tree
    t's edges as an [n, mst_deg+1] slab (reference row layout: count, then <= mst_deg neighbours).
    ONE all-gather of the slabs, then every rank concatenates each vertex's lists in tree order
    t = 0, 1, 2, ... -- exactly the adjacency the single-process build appends (hcnng_index.h:117-131)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    per = (num_clusters + world - 1) // world
    w = mst_deg + 1
    slab = np.zeros((n, per * w), np.uint32)
    for j, t in enumerate(range(rank, num_clusters, world)):
        g = np.asarray(build_tree(t), dtype=np.uint32)
        assert g.shape == (n, w)
        slab[:, j * w:(j + 1) * w] = g
    slabs = all_gather_array(slab, device)                        # [W, n, per*(mst_deg+1)]
    out = np.zeros((n, num_clusters * mst_deg + 1), np.uint32)
    for t in range(num_clusters):
        g = slabs[t % world][:, (t // world) * w:(t // world + 1) * w]
        cnt = g[:, 0].astype(np.int64)
        for j in range(mst_deg):
            sel = np.nonzero(cnt > j)[0]
            out[sel, 1 + out[sel, 0].astype(np.int64)] = g[sel, 1 + j]
            out[sel, 0] += 1
    return out
