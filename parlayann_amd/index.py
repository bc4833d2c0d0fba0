"""DeviceIndex: thin Python owner of a pann_index handle (device mirror of PointRange + Graph).

Mirrors what python/graph_index.cpp:82-118 holds (points + graph) and the batched seam of
:192-216; numpy arrays in, numpy arrays out.  Every method goes through the C-ABI.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (PANN_BF16, PANN_F16, PANN_F32, PANN_I8, PANN_L2, PANN_MIPS, PANN_U8, BuildStats, QueryParams, SearchOut,
                    check)
from .bf16 import bfloat16

_DT = {np.dtype(np.uint8): PANN_U8, np.dtype(np.int8): PANN_I8, np.dtype(np.float32): PANN_F32,
       np.dtype(np.float16): PANN_F16, bfloat16: PANN_BF16}


def dtype_code(dt):
    return _DT[np.dtype(dt)]


def _metric_code(metric):
    if metric in (PANN_L2, PANN_MIPS):
        return metric
    m = str(metric).lower()
    if m in ("euclidian", "euclidean", "l2"):
        return PANN_L2
    if m in ("mips", "ip"):
        return PANN_MIPS
    raise ValueError(f"unknown metric {metric!r}")


def _row_stride(a):
    """row stride of a C-contiguous 2-D array (numpy may report anything for a length-1 axis)"""
    return a.shape[1] * a.itemsize


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def host_graph(n, max_deg):
    """An empty graph slab in the reference layout (graph.h:134-141): n x (max_deg+1), slot 0 = degree."""
    return np.zeros((n, max_deg + 1), dtype=np.uint32)


class DeviceIndex:
    def __init__(self, points, graph=None, max_degree=None, metric="Euclidian", device=0, exact_float_order=False):
        lib = _capi.load()
        points = np.ascontiguousarray(points)
        if points.ndim != 2 or points.dtype not in _DT:
            raise ValueError("points must be a 2-D uint8/int8/float32/float16/bfloat16 (parlayann_amd.bfloat16) array")
        n, d = points.shape
        if graph is not None:
            graph = np.ascontiguousarray(graph, dtype=np.uint32)
            if graph.ndim != 2 or graph.shape[0] != n:
                raise ValueError("graph must be n x (max_deg+1) uint32 (reference layout)")
            max_degree = graph.shape[1] - 1
        if max_degree is None:
            raise ValueError("give a graph or max_degree")
        self.n, self.d, self.max_degree = n, d, int(max_degree)
        self.dtype, self.metric = points.dtype, _metric_code(metric)
        h = C.c_void_p()
        check(lib.pann_index_create(C.byref(h), _ptr(points), n, d, _DT[points.dtype], _row_stride(points),
                                    self.metric, _ptr(graph), self.max_degree, device))
        self._h, self._lib = h, lib
        if exact_float_order:      # validation mode: bit-identical float results on real-valued data
            check(lib.pann_index_set_exact_float_order(h, 1))

    @classmethod
    def from_files(cls, points_path, dtype, graph_path=None, max_degree=None, metric="Euclidian", device=0, rows=None,
                   chunk_bytes=256 << 20, exact_float_order=False):
        """Stream a reference-format vector file (point_range.h:74-117: [n:u32][d:u32][rows]) -- and, optionally, a graph file
        (graph.h:147-232: [n:u32][maxDeg:u32][deg[n]][edges]) -- onto the device in chunks of about chunk_bytes, without
        holding either whole on the host.  rows = (lo, hi): only that range of the vector file becomes the index (row lo is
        vertex 0: one shard of a sharded index; the graph file, if given, must describe exactly these hi - lo vertices)."""
        lib = _capi.load()
        dt = np.dtype(dtype)
        if dt not in _DT:
            raise ValueError("dtype must be uint8/int8/float32/float16/bfloat16 (parlayann_amd.bfloat16)")
        with open(points_path, "rb") as f:
            n_file, d = (int(v) for v in np.fromfile(f, dtype=np.uint32, count=2))
        lo, hi = (0, n_file) if rows is None else (int(rows[0]), int(rows[1]))
        if not (0 <= lo < hi <= n_file):
            raise ValueError(f"rows {rows} outside the {n_file} rows of {points_path}")
        n = hi - lo
        gf = None
        if graph_path is not None:
            gf = open(graph_path, "rb")
            gn, gmax = (int(v) for v in np.fromfile(gf, dtype=np.uint32, count=2))
            if gn != n:
                gf.close()
                raise ValueError(f"graph file has {gn} vertices, the index {n}")
            max_degree = gmax
        if max_degree is None:
            raise ValueError("give a graph file or max_degree")
        self = cls.__new__(cls)
        self.n, self.d, self.max_degree = n, d, int(max_degree)
        self.dtype, self.metric = dt, _metric_code(metric)
        h = C.c_void_p()
        check(lib.pann_index_create_empty(C.byref(h), n, d, _DT[dt], self.metric, self.max_degree, device))
        self._h, self._lib = h, lib
        try:
            row_bytes = d * dt.itemsize
            step = max(1, int(chunk_bytes) // row_bytes)
            mm = np.memmap(points_path, dtype=np.uint8, mode="r", offset=8 + lo * row_bytes, shape=(n, row_bytes))
            for a in range(0, n, step):
                chunk = np.ascontiguousarray(mm[a:min(a + step, n)])
                check(lib.pann_index_upload_points(h, a, _ptr(chunk), len(chunk), row_bytes))
            del mm
            if gf is not None:
                deg = np.fromfile(gf, dtype=np.uint32, count=n)
                if deg.size != n or (deg > self.max_degree).any():
                    raise ValueError("graph file: bad degree array")
                gstep = max(1, int(chunk_bytes) // (4 * (self.max_degree + 1)))
                cols = np.arange(self.max_degree)[None, :]
                for a in range(0, n, gstep):
                    b = min(a + gstep, n)
                    dg = deg[a:b]
                    edges = np.fromfile(gf, dtype=np.uint32, count=int(dg.sum(dtype=np.int64)))
                    blk = np.zeros((b - a, self.max_degree + 1), dtype=np.uint32)
                    blk[:, 0] = dg
                    blk[:, 1:][cols < dg[:, None]] = edges
                    self.update_rows(np.arange(a, b, dtype=np.uint32), blk)
            if exact_float_order:
                check(lib.pann_index_set_exact_float_order(h, 1))
        except Exception:
            self.close()
            raise
        finally:
            if gf is not None:
                gf.close()
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pann_index_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    def _queries(self, queries):
        """external query rows: C-contiguous nq x d of the INDEX dtype (a float32 array handed to an f16 index would be
        reinterpreted byte-wise by the C-ABI, which only sees a pointer and a stride)"""
        q = np.ascontiguousarray(queries)
        if q.dtype != self.dtype or q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"queries must be nq x {self.d} of dtype {self.dtype}, got {q.shape} {q.dtype}")
        return q

    def reserve_dropped(self, cap):
        """pann_index_reserve_dropped: per-query scratch of the cut-prune bookkeeping (include/pann.h)"""
        check(self._lib.pann_index_reserve_dropped(self._h, int(cap)))

    @property
    def dropped_capacity(self):
        return int(self._lib.pann_index_dropped_capacity(self._h))

    # ---- graph ----
    def set_graph(self, graph):
        graph = np.ascontiguousarray(graph, dtype=np.uint32)
        assert graph.shape == (self.n, self.max_degree + 1)
        check(self._lib.pann_index_set_graph(self._h, _ptr(graph)))

    def update_rows(self, row_ids, rows):
        row_ids = np.ascontiguousarray(row_ids, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        assert rows.shape == (len(row_ids), self.max_degree + 1)
        check(self._lib.pann_index_update_rows(self._h, _ptr(row_ids), _ptr(rows), len(row_ids)))

    def clear_graph(self):
        """empty graph again (Graph(maxDeg, n), graph.h:145-147), on the device"""
        check(self._lib.pann_index_clear_graph(self._h))

    def set_stream(self, stream_ptr, private=False):
        """pann_index_set_stream: run the handle's calls on the caller's stream (a hipStream_t as an integer; 0 = the device's
        default stream, torch's usual current stream); private=True: back to the handle's own stream"""
        check(self._lib.pann_index_set_stream(self._h, C.c_void_p(stream_ptr or None), 1 if private else 0))

    def get_option(self, name):
        return int(self._lib.pann_index_get_option(self._h, name.encode()))

    def set_option(self, name, value):
        """pann_index_set_option: per-handle tuning knobs ("forest_group", "gt_pieces"); results never depend on them"""
        check(self._lib.pann_index_set_option(self._h, name.encode(), int(value)))

    def get_graph(self):
        g = np.empty((self.n, self.max_degree + 1), dtype=np.uint32)
        check(self._lib.pann_index_get_graph(self._h, _ptr(g)))
        return g

    # ---- batched beam search: the searchAll / qsearchAll seam (beamSearch.h:374,556) ----
    def batch_search(self, queries=None, k=10, beam=64, cut=1.35, limit=None, degree_limit=None, starts=(0,),
                     query_ids=None, out_k=None, visited_cap=0, want_dists=True):
        nq = len(queries) if queries is not None else len(query_ids)
        qp = QueryParams(k=k, beam=beam, cut=cut, limit=self.n if limit is None else limit,
                         degree_limit=self.max_degree if degree_limit is None else degree_limit,
                         rerank_factor=100, pad=1.0)
        out_k = k if out_k is None else out_k
        res = {
            "ids": np.empty((nq, out_k), dtype=np.uint32),
            "dists": np.empty((nq, out_k), dtype=np.float32) if want_dists else None,
            "frontier_size": np.empty(nq, dtype=np.uint32),
            "visited_count": np.empty(nq, dtype=np.uint32),
            "dist_cmps": np.empty(nq, dtype=np.uint32),
            "degree_sum": np.empty(nq, dtype=np.uint32),
            "visited_ids": np.empty((nq, visited_cap), dtype=np.uint32) if visited_cap else None,
            "visited_dists": np.empty((nq, visited_cap), dtype=np.float32) if visited_cap else None,
            "status": np.zeros(1, dtype=np.uint32),
        }
        out = SearchOut(ids=_ptr(res["ids"]), dists=_ptr(res["dists"]), out_k=out_k,
                        frontier_size=_ptr(res["frontier_size"]), visited_count=_ptr(res["visited_count"]),
                        dist_cmps=_ptr(res["dist_cmps"]), degree_sum=_ptr(res["degree_sum"]),
                        visited_ids=_ptr(res["visited_ids"]), visited_dists=_ptr(res["visited_dists"]),
                        visited_cap=visited_cap, status=_ptr(res["status"]))
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        per_query = starts.ndim == 2          # nq x nstarts: beamSearchRandom-style, one start set per query
        if per_query and starts.shape[0] != nq:
            raise ValueError("per-query starts must be nq x nstarts")
        q = qid = None
        stride = 0
        if queries is not None:
            q = self._queries(queries)
            stride = _row_stride(q)
        else:
            qid = np.ascontiguousarray(query_ids, dtype=np.uint32)
        fn = self._lib.pann_batch_search_per_query_starts if per_query else self._lib.pann_batch_search
        check(fn(self._h, _ptr(q), _ptr(qid), nq, stride, _ptr(starts), starts.shape[-1], C.byref(qp), C.byref(out)))
        return res

    # ---- robustPrune (vamana/index.h:63-137), batched ----
    def robust_prune_batch(self, owners, cand_ids, cand_offsets, alpha, R, cand_dists=None, add_out_nbrs=True):
        owners = np.ascontiguousarray(owners, dtype=np.uint32)
        cand_ids = np.ascontiguousarray(cand_ids, dtype=np.uint32)
        off = np.ascontiguousarray(cand_offsets, dtype=np.uint64)
        cd = None if cand_dists is None else np.ascontiguousarray(cand_dists, dtype=np.float32)
        m = len(owners)
        rows = np.zeros((m, R + 1), np.uint32)
        dc = np.zeros(m, np.uint32)
        check(self._lib.pann_robust_prune_batch(self._h, _ptr(owners), m, _ptr(cand_ids), _ptr(cd), _ptr(off),
                                                float(alpha), R, 1 if add_out_nbrs else 0, _ptr(rows), _ptr(dc)))
        return rows, dc

    # ---- Vamana (vamana/index.h:150-316) ----
    def vamana_insert_batch(self, batch_ids, R, L, alpha, start=0):
        b = np.ascontiguousarray(batch_ids, dtype=np.uint32)
        st = BuildStats()
        check(self._lib.pann_vamana_insert_batch(self._h, _ptr(b), len(b), start, R, L, float(alpha), C.byref(st)))
        return st

    def vamana_build(self, R, L, alpha, num_passes=1, seed=1, sort_neighbors=True, single_batch=0, point_stats=None):
        """build_index (vamana/index.h:150-186).  single_batch = degree != 0: BuildParams::single_batch -- `degree` random
        start edges per vertex, then every pass is one batch of all points (:156-170,236-240).
        point_stats = (visited[n], dist_cmps[n]) uint32 arrays: accumulated per point like the reference's BuildStats (stats.h:63-73)."""
        st = BuildStats()
        if point_stats is not None:
            vis, dc = point_stats
            if not all(a.dtype == np.uint32 and a.flags.c_contiguous and a.shape == (self.n,) for a in (vis, dc)):
                raise ValueError("point_stats must be two C-contiguous uint32 arrays of n entries")
            st.per_point_visited = vis.ctypes.data_as(C.c_void_p)
            st.per_point_dist_cmps = dc.ctypes.data_as(C.c_void_p)
        if single_batch:
            check(self._lib.pann_vamana_build_single_batch(self._h, R, L, float(alpha), num_passes, int(single_batch), seed,
                                                           1 if sort_neighbors else 0, C.byref(st)))
        else:
            check(self._lib.pann_vamana_build(self._h, R, L, float(alpha), num_passes, seed, 1 if sort_neighbors else 0,
                                              C.byref(st)))
        return st

    # ---- the two phases of a batch on device pointers (multi-GPU build, parlayann_amd/distributed.py) ----
    def vamana_search_prune_dev(self, d_batch_ptr, m, R, L, alpha, d_rows_ptr, start=0, stats=None):
        """phase A (vamana/index.h:247-266) for m batch ids at device address d_batch_ptr -> rows [m, R] at d_rows_ptr"""
        st = BuildStats() if stats is None else stats
        check(self._lib.pann_vamana_search_prune_dev(self._h, C.c_void_p(d_batch_ptr), m, start, R, L, float(alpha),
                                                     C.c_void_p(d_rows_ptr), C.byref(st)))
        return st

    def vamana_apply_rows_dev(self, d_batch_ptr, m, d_rows_ptr, R, alpha, stats=None):
        """phase B (vamana/index.h:268-300) for the whole batch: rows [m, R] at device address d_rows_ptr"""
        st = BuildStats() if stats is None else stats
        check(self._lib.pann_vamana_apply_rows_dev(self._h, C.c_void_p(d_batch_ptr), m, C.c_void_p(d_rows_ptr), R, float(alpha),
                                                   C.byref(st)))
        return st

    def vamana_sort_neighbors(self):
        check(self._lib.pann_vamana_sort_neighbors(self._h))

    # ---- distances / dense all-pairs ----
    def hcnng_build(self, num_clusters, cluster_size, mst_deg, seed=1):
        """hcnng_index.h:273-281 on the device (trees, leaf kNN, Kruskal); returns {tree_s, leaf_knn_s, mst_s}."""
        times = np.zeros(3, np.float64)
        check(self._lib.pann_hcnng_build(self._h, num_clusters, cluster_size, mst_deg, seed, _ptr(times)))
        return {"tree_s": times[0], "leaf_knn_s": times[1], "mst_s": times[2]}

    def range_search(self, starts, radius_2, max_results, queries=None, query_ids=None):
        """beamSearch.h:245-306: BFS from the starts that lie within radius_2 over all vertices within radius_2.
        starts: (ns,) shared or (nq, ns) per query, 0xFFFFFFFF = padding.  Rows of `ids` are in BFS order."""
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        per_query = starts.ndim == 2
        if (queries is None) == (query_ids is None):
            raise ValueError("exactly one of queries / query_ids must be given")
        if queries is not None:
            q = self._queries(queries); nq = len(q); qp, qs, qi = _ptr(q), _row_stride(q), None
        else:
            qid = np.ascontiguousarray(query_ids, dtype=np.uint32); nq = len(qid); qp, qs, qi = None, 0, _ptr(qid)
        if per_query and len(starts) != nq:
            raise ValueError("per-query starts must have one row per query")
        ids = np.empty((nq, max_results), np.uint32)
        cnt = np.zeros(nq, np.uint32); cmps = np.zeros(nq, np.uint32); trunc = np.zeros(nq, np.uint32)
        check(self._lib.pann_range_search(self._h, qp, qi, nq, qs, _ptr(starts), starts.shape[-1], 1 if per_query else 0,
                                          float(radius_2), max_results, _ptr(ids), _ptr(cnt), _ptr(cmps), _ptr(trunc)))
        w = int(cnt.max()) if nq else 0                      # entries past a row's count are unspecified: pad them
        ids[:, w:] = 0xFFFFFFFF
        if w:
            sub = ids[:, :w]
            sub[np.arange(w, dtype=np.uint32)[None, :] >= cnt[:, None]] = 0xFFFFFFFF
        return {"ids": ids, "counts": cnt, "dist_cmps": cmps, "truncated": trunc}

    def pair_distances(self, a_ids, b_ids):
        a = np.ascontiguousarray(a_ids, dtype=np.uint32); b = np.ascontiguousarray(b_ids, dtype=np.uint32)
        out = np.empty(len(a), np.float32)
        check(self._lib.pann_pair_distances(self._h, _ptr(a), _ptr(b), len(a), _ptr(out)))
        return out

    def query_distances(self, queries, ids):
        q = self._queries(queries); ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(q), len(ids)), np.float32)
        check(self._lib.pann_query_distances(self._h, _ptr(q), len(q), _row_stride(q), _ptr(ids), len(ids), _ptr(out)))
        return out

    def leaf_knn_batch(self, ids, leaf_offsets, m):
        """hcnng_index.h:145-181 for many leaves: per member the m nearest other members."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        off = np.ascontiguousarray(leaf_offsets, dtype=np.uint64)
        oi = np.empty((len(ids), m), np.uint32); od = np.empty((len(ids), m), np.float32)
        check(self._lib.pann_leaf_knn_batch(self._h, _ptr(ids), _ptr(off), len(off) - 1, m, _ptr(oi), _ptr(od)))
        return oi, od

    def leaf_knn(self, ids, m):
        return self.leaf_knn_batch(ids, [0, len(ids)], m)

    def bruteforce_knn(self, queries, k):
        """data_tools/compute_groundtruth.cpp:22-59."""
        q = self._queries(queries)
        oi = np.empty((len(q), k), np.uint32); od = np.empty((len(q), k), np.float32)
        check(self._lib.pann_bruteforce_knn(self._h, _ptr(q), len(q), _row_stride(q), k, _ptr(oi), _ptr(od)))
        return oi, od

    def pivot_split(self, ids, seg_offsets, pivot_a, pivot_b):
        """clusterEdge.h:66-83: side 0 when d(id, pivot_a) <= d(id, pivot_b)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        off = np.ascontiguousarray(seg_offsets, dtype=np.uint64)
        pa = np.ascontiguousarray(pivot_a, dtype=np.uint32); pb = np.ascontiguousarray(pivot_b, dtype=np.uint32)
        side = np.empty(len(ids), np.uint8)
        check(self._lib.pann_pivot_split(self._h, _ptr(ids), _ptr(off), len(off) - 1, _ptr(pa), _ptr(pb), _ptr(side)))
        return side

    def rerank(self, queries, cand_ids, cand_counts, k, resort=True):
        """beamSearch.h:426-452: exact distances of each query's candidates, (re)sorted, first k."""
        q = self._queries(queries)
        cand = np.ascontiguousarray(cand_ids, dtype=np.uint32)
        cnt = None if cand_counts is None else np.ascontiguousarray(cand_counts, dtype=np.uint32)
        oi = np.empty((len(q), k), np.uint32); od = np.empty((len(q), k), np.float32)
        check(self._lib.pann_rerank(self._h, _ptr(q), len(q), _row_stride(q), _ptr(cand), cand.shape[1], _ptr(cnt), k,
                                    1 if resort else 0, _ptr(oi), _ptr(od)))
        return oi, od
