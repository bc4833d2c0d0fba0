"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL
(backend "nccl") on the GPU box, gloo in CPU tests.  The reference is single-process; nothing here
translates reference code.

  * replicated index, sharded queries  : no collective on the data path (bench.py); only a barrier
    and a max-reduce of the elapsed time (`timed_steps`).
  * sharded index (C4)                 : every rank searches ALL queries on its own id range; ONE
    all-gather of the per-rank packed top-k rows (k*8 B per query per rank) and a merge by (dist,id).
  * sharded Vamana build               : points and graph replicated, every BATCH of batch_insert split over the ranks;
    ONE all-gather of the batch's new adjacency rows (m x R x 4 B) per batch stitches the replicas back together
    (`vamana_build_sharded`).  The graph is bit-identical to the single-GPU build.
  * sharded build stitch (primitive)   : ONE all-gather of adjacency rows (n/W x (R+1) x 4 B per rank).
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
    """ids/dists: [W, nq, k] per-shard results with GLOBAL ids -> [nq, k] smallest by (dist, id).  Host (numpy) form,
    the checker of the device kernel `pann_merge_topk_dev` and the merge of the CPU tests; the product path
    (`DeviceShardedIndex`) merges on the GPU."""
    ids = np.asarray(ids, dtype=np.uint32); dists = np.asarray(dists, dtype=np.float32)
    W, nq, kk = ids.shape
    allid = np.transpose(ids, (1, 0, 2)).reshape(nq, W * kk)
    alld = np.transpose(dists, (1, 0, 2)).reshape(nq, W * kk)
    # one integer key per pair that orders like (dist, id): the float's bits made monotone, then the id
    u = (alld + np.float32(0.0)).view(np.uint32).astype(np.uint64)
    ordd = np.where(u & 0x80000000, u ^ 0xFFFFFFFF, u ^ 0x80000000)
    key = (ordd << np.uint64(32)) | allid.astype(np.uint64)
    key[allid == 0xFFFFFFFF] = np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(key, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(allid, order, 1), np.take_along_axis(alld, order, 1)


def all_gather_tensor(t):
    """ONE all-gather of equally shaped tensors -> [W, *t.shape] on t's device.  Under nccl (= RCCL) device tensors go
    GPU to GPU; under gloo (CPU tests, rehearsals of several ranks on one GPU) a device tensor is staged through the host."""
    rank, world = _world()
    if world == 1:
        return t.unsqueeze(0)
    staged = t.is_cuda and dist.get_backend() != "nccl"
    src = (t.cpu() if staged else t).contiguous()
    flat = src.reshape(-1)                        # the concatenated form: accepted by gloo and nccl alike
    out = torch.empty(world * flat.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, flat)
    out = out.reshape((world,) + tuple(src.shape))
    return out.to(t.device) if staged else out


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


class DeviceShardedIndex:
    """The sharded index on the product path, nothing on the host between the kernels: rank r holds base points
    [lo, hi) as a DeviceIndex with its own sub-graph (local ids); a query batch (a device tensor, the same on every rank)
    is searched on every shard (pann_batch_search_dev); ONE all-gather moves every rank's packed [nq, 2k] rows (k local ids |
    k distance bits: k * 8 bytes per query per rank) and pann_merge_topk_dev adds each shard's base to its ids and keeps the
    k smallest by (dist, id).  Results stay in HBM (torch tensors)."""

    def __init__(self, points, max_degree, build, device_ordinal=0, metric="Euclidian", n_total=None):
        """points: the whole base (every rank slices its own range out of it), or -- with n_total given -- only this rank's
        slice [lo, hi) of a base of n_total points (100M-point runs: nobody holds the whole base)"""
        from .index import DeviceIndex
        self.rank, self.world = _world()
        self.n = len(points) if n_total is None else int(n_total)
        self.lo, self.hi = shard_range(self.n, self.rank, self.world)
        shard = points[self.lo:self.hi] if n_total is None else points
        assert len(shard) == self.hi - self.lo
        self.ix = DeviceIndex(shard, max_degree=max_degree, device=device_ordinal, metric=metric)
        build(self.ix)
        self.dev = torch.device("cuda", device_ordinal)
        self._starts = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        bases = [shard_range(self.n, r, self.world)[0] for r in range(self.world)]
        self._bases = torch.from_numpy(np.array(bases, dtype=np.uint32).view(np.int32)).to(self.dev)     # uint32 bit patterns
        self.collectives = 0            # all-gathers issued so far (tests assert ONE per search)
        self.last_counters = None       # (visited_count, dist_cmps, degree_sum) device tensors of the last local search

    def _merge(self, packed, k):
        """packed: local [nq, 2*kk] int32 rows (kk local ids | kk distance bits) -> global top-k over all ranks, on the device;
        ONE all-gather"""
        import ctypes as C
        from ._capi import check
        nq, kk = packed.shape[0], packed.shape[1] // 2
        allp = all_gather_tensor(packed)                                    # [W, nq, 2*kk]
        self.collectives += 1
        oi = torch.empty((nq, k), dtype=torch.int32, device=self.dev)
        od = torch.empty((nq, k), dtype=torch.float32, device=self.dev)
        st = torch.cuda.current_stream(self.dev)
        check(self.ix._lib.pann_merge_topk_dev(allp.data_ptr(), allp.data_ptr() + 4 * kk, allp.shape[0], nq, kk, 2 * kk,
                                               self._bases.data_ptr(), k, oi.data_ptr(), od.data_ptr(), C.c_void_p(st.cuda_stream)))
        return oi, od

    def bruteforce(self, queries, k):
        """exact ground truth over all shards (not a timed path: the per-shard brute force takes host arrays)"""
        li, ld = self.ix.bruteforce_knn(queries, k)
        packed = torch.from_numpy(np.concatenate([li.view(np.int32), ld.view(np.int32)], axis=1)).to(self.dev)
        return self._merge(packed, k)

    def search(self, d_queries, k, beam, cut=1.35, limit=None, counters=False):
        """d_queries: [nq, row bytes] uint8 device tensor (raw rows of the index dtype).  Returns (ids, dists) device tensors
        [nq, k] (int32 holding uint32 ids, float32).  The launch's status word (include/pann.h PANN_STATUS_*) is read before the
        exchange: a dropped-list overflow grows the scratch and runs the local search again (as pann_batch_search does on the
        host path) -- a rank never gathers lists the kernel has declared invalid."""
        import ctypes as C
        from ._capi import PANN_STATUS_DROPPED_OVERFLOW, QueryParams, SearchOut, check
        lib = self.ix._lib
        nq = d_queries.shape[0]
        packed = torch.empty((nq, 2 * k), dtype=torch.int32, device=self.dev)
        ids = torch.empty((nq, k), dtype=torch.int32, device=self.dev)
        dists = torch.empty((nq, k), dtype=torch.float32, device=self.dev)
        lim = self.ix.n if limit is None else int(limit)
        qp = QueryParams(k=k, beam=beam, cut=cut, limit=lim, degree_limit=self.ix.max_degree, rerank_factor=100, pad=1.0)
        out = SearchOut(ids=ids.data_ptr(), dists=dists.data_ptr(), out_k=k, status=self._status.data_ptr())
        if counters:
            self.last_counters = tuple(torch.empty(nq, dtype=torch.int32, device=self.dev) for _ in range(3))
            out.visited_count, out.dist_cmps, out.degree_sum = (t.data_ptr() for t in self.last_counters)
        st = torch.cuda.current_stream(self.dev)
        for attempt in range(2):
            check(lib.pann_batch_search_dev(self.ix.handle, d_queries.data_ptr(), None, nq, d_queries.shape[1], self._starts.data_ptr(),
                                            1, C.byref(qp), C.byref(out), C.c_void_p(st.cuda_stream)))
            status = int(self._status.item())                                 # one host read per search; the exchange follows anyway
            if not (status & PANN_STATUS_DROPPED_OVERFLOW):
                break
            if attempt == 1:
                raise RuntimeError("DeviceShardedIndex.search: dropped-list overflow persists after reserving min(limit, n) entries")
            self.ix.reserve_dropped(max(1, min(lim, self.ix.n)))              # a query drops at most one entry per visit
        if status:
            raise RuntimeError(f"DeviceShardedIndex.search: launch status {status} (include/pann.h PANN_STATUS_*)")
        packed[:, :k] = ids
        packed[:, k:] = dists.view(torch.int32)
        return self._merge(packed, k)

    def close(self):
        self.ix.close()


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


# ---------------------------------------------------------------------------------------------------------------------
# Sharded Vamana build (vamana/index.h:188-316 is what the ranks run between the collectives)
# ---------------------------------------------------------------------------------------------------------------------

def build_schedule(n, seed):
    """(permutation [n] uint32, batch bounds [(floor, ceiling), ...]) of one pass of build_index -- from libpann.so's
    host-only helpers, so every rank (and the single-GPU pann_vamana_build) derives the same schedule."""
    import ctypes as C
    from . import _capi
    lib = _capi.load()
    perm = np.empty(n, np.uint32)
    lib.pann_build_permutation(n, seed, perm.ctypes.data_as(C.c_void_p))
    nb = int(lib.pann_vamana_batch_schedule(n, n, None, 0))
    bounds = np.zeros((nb, 2), np.uint64)
    lib.pann_vamana_batch_schedule(n, n, bounds.ctypes.data_as(C.c_void_p), nb)
    return perm, [(int(a), int(b)) for a, b in bounds]


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_rows(local, per):
    """ONE all-gather of equally sized row blocks: `local` is this rank's [<=per, R] int32 tensor (a device tensor on the
    product path), padded here to [per, R]; returns [W * per, R] on the same device."""
    R = local.shape[1]
    if local.shape[0] < per:
        pad = torch.full((per - local.shape[0], R), -1, dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    return all_gather_tensor(local).reshape(-1, R)


def vamana_build_sharded(n, R, L, alpha, num_passes, seed, phase_a, phase_b, finish=None, device="cpu", min_split=None):
    """build_index (vamana/index.h:150-186) with every batch of batch_insert (:223-300) split over the ranks.

    Every rank holds ALL points and a replica of the graph.  Per batch (m ids of the shared insertion order):
      1. rank r runs `phase_a(ids[r*per : (r+1)*per])` -> rows [., R] (beam search + robustPrune, :247-266: reads the graph only)
      2. ONE all-gather of the rows (m x R x 4 bytes over all ranks; C3: 200 000 x 64 x 4 = 51 MB per batch, 134 batches)
      3. every rank runs `phase_b(ids, rows)` for the whole batch (:268-300: deterministic), so the replicas stay identical
    Batches of fewer than `min_split` points (the first prefix-doubling batches) are run whole by every rank, without a
    collective.  The result equals the single-GPU build bit for bit (phase A of a point does not depend on who runs it).

    phase_a(ids: int32 tensor on `device`, alpha) -> int32 tensor [len(ids), R] on `device` (unused slots -1 == 0xFFFFFFFF)
    phase_b(ids, rows, alpha); finish() = the final neighbour sort (:180-185), run by every rank.
    Returns {"collectives": count, "bytes_gathered": total bytes received per rank}."""
    rank, world = _world()
    if min_split is None:
        min_split = 64 * world
    perm, bounds = build_schedule(n, seed)
    d_perm = torch.from_numpy(perm.view(np.int32)).to(device)
    info = {"collectives": 0, "bytes_gathered": 0}
    for p in range(num_passes):
        a = alpha if p == num_passes - 1 else 1.0                      # :173-178
        for lo, hi in bounds:
            ids = d_perm[lo:hi]
            m = hi - lo
            if world == 1 or m < min_split:
                rows = phase_a(ids, a)
            else:
                per = (m + world - 1) // world
                s0, s1 = min(m, rank * per), min(m, (rank + 1) * per)
                mine = phase_a(ids[s0:s1], a) if s1 > s0 else torch.empty((0, R), dtype=torch.int32, device=device)
                rows = all_gather_rows(mine, per)[:m]
                info["collectives"] += 1
                info["bytes_gathered"] += world * per * R * 4
            phase_b(ids, rows, a)
    if finish is not None:
        finish()
    return info


def device_vamana_build_sharded(ix, R, L, alpha, num_passes=1, seed=1, sort_neighbors=True, min_split=None):
    """`vamana_build_sharded` on a DeviceIndex (replicated on every rank's GPU): the phases are the C-ABI's
    pann_vamana_search_prune_dev / pann_vamana_apply_rows_dev on torch device tensors, the collective is RCCL's.  For the
    duration of the build the handle runs on torch's current stream (pann_index_set_stream), so the phases, the all-gather
    and torch's small kernels between them are ordered by ONE stream: no host synchronisation is added here."""
    dev = torch.device("cuda", ix._lib.pann_index_device(ix._h))
    stats = _new_build_stats()
    ix.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def phase_a(ids, a):
        ids = ids.contiguous()
        rows = torch.empty((ids.numel(), R), dtype=torch.int32, device=dev)
        ix.vamana_search_prune_dev(ids.data_ptr(), ids.numel(), R, L, a, rows.data_ptr(), stats=stats)
        return rows

    def phase_b(ids, rows, a):
        ids = ids.contiguous(); rows = rows.contiguous()
        ix.vamana_apply_rows_dev(ids.data_ptr(), ids.numel(), rows.data_ptr(), R, a, stats=stats)

    try:
        info = vamana_build_sharded(ix.n, R, L, alpha, num_passes, seed, phase_a, phase_b,
                                    finish=ix.vamana_sort_neighbors if sort_neighbors else None, device=dev, min_split=min_split)
    finally:
        ix.set_stream(0, private=True)
    info["stats"] = stats
    return info


def _new_build_stats():
    from ._capi import BuildStats
    return BuildStats()


def device_hcnng_build_tree_parallel(ix, num_clusters, cluster_size, mst_deg, seed=1):
    """HCNNG with the trees split over the ranks on ONE resident DeviceIndex per rank (all points, graph replica): rank r builds
    trees r, r + W, ... into a device slab (pann_hcnng_build_trees_dev), ONE all-gather of the slabs (n x ceil(T/W) x mst_deg x 4
    bytes per rank), every rank interleaves them in tree order into its graph (pann_hcnng_assemble_dev) -- the single-GPU
    graph, bit for bit.  Returns the seconds {tree, leaf kNN, MST} this rank spent."""
    import ctypes as C
    from ._capi import check
    rank, world = _world()
    dev = torch.device("cuda", ix._lib.pann_index_device(ix._h))
    per = (num_clusters + world - 1) // world
    mine = len(range(rank, num_clusters, world))
    stride = per * mst_deg
    slab = torch.empty((ix.n, stride), dtype=torch.int32, device=dev)
    times = np.zeros(3, np.float64)
    ix.set_stream(torch.cuda.current_stream(dev).cuda_stream)     # trees, all-gather and assembly ordered by ONE stream
    try:
        check(ix._lib.pann_hcnng_build_trees_dev(ix._h, rank, world, mine, cluster_size, mst_deg, seed, C.c_void_p(slab.data_ptr()), stride,
                                                 times.ctypes.data_as(C.c_void_p)))
        slabs = all_gather_tensor(slab)                                      # [W, n, stride]
        check(ix._lib.pann_hcnng_assemble_dev(ix._h, C.c_void_p(slabs.data_ptr()), slabs.shape[0], stride, num_clusters, mst_deg))
    finally:
        ix.set_stream(0, private=True)
    return {"tree_s": times[0], "leaf_knn_s": times[1], "mst_s": times[2], "bytes_gathered": int(slabs.numel()) * 4}
