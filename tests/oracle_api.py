"""ctypes wrapper of oracle/libpann_oracle.so -- the CPU restatement used as the CHECKER.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libpann_oracle.so")

DT = {np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.float32): 2, np.dtype(np.float16): 3,
      np.dtype([("bf16", np.uint16)]): 4}          # parlayann_amd.bfloat16 (numpy has no bfloat16)
METRIC = {"l2": 0, "euclidian": 0, "mips": 1, 0: 0, 1: 1}

_lib = None


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def load():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "pann_oracle.cpp")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        _lib = C.CDLL(LIB)
        _lib.pann_oracle_distance.restype = C.c_float
        _lib.pann_oracle_hash64_2.restype = C.c_uint64
        _lib.pann_oracle_hash64_2.argtypes = [C.c_uint64]
        _lib.pann_oracle_recall.restype = C.c_double
        _lib.pann_oracle_mips_i8_maxval.restype = C.c_float
    return Oracle(_lib)


def _m(metric):
    return METRIC[metric.lower() if isinstance(metric, str) else metric]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        self.threads = max(1, lib.pann_oracle_hw_threads())

    def hash64_2(self, x):
        return self.lib.pann_oracle_hash64_2(C.c_uint64(x))

    def distance(self, a, b, metric="l2"):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        return float(self.lib.pann_oracle_distance(C.c_int(DT[a.dtype]), C.c_int(_m(metric)), _p(a), _p(b),
                                                   C.c_uint32(a.shape[0])))

    def batch_search(self, points, graph, queries=None, query_ids=None, k=10, beam=64, cut=1.35, limit=None,
                     degree_limit=None, starts=(0,), metric="l2", out_k=None, visited_cap=0, threads=None):
        points = np.ascontiguousarray(points); graph = np.ascontiguousarray(graph, dtype=np.uint32)
        n, d = points.shape
        maxdeg = graph.shape[1] - 1
        nq = len(queries) if queries is not None else len(query_ids)
        out_k = k if out_k is None else out_k
        limit = n if limit is None else limit
        degree_limit = maxdeg if degree_limit is None else degree_limit
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        q = qid = None
        qstride = 0
        if queries is not None:
            q = np.ascontiguousarray(queries); qstride = q.strides[0]
            assert q.dtype == points.dtype
        else:
            qid = np.ascontiguousarray(query_ids, dtype=np.uint32)
        res = {
            "ids": np.empty((nq, out_k), np.uint32), "dists": np.empty((nq, out_k), np.float32),
            "frontier_size": np.empty(nq, np.uint32), "visited_count": np.empty(nq, np.uint32),
            "dist_cmps": np.empty(nq, np.uint32), "degree_sum": np.empty(nq, np.uint32),
            "visited_ids": np.zeros((nq, visited_cap), np.uint32) if visited_cap else None,
            "visited_dists": np.zeros((nq, visited_cap), np.float32) if visited_cap else None,
            "visit_order_ids": np.zeros((nq, visited_cap), np.uint32) if visited_cap else None,
        }
        rc = self.lib.pann_oracle_batch_search(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(maxdeg), _p(q), _p(qid), C.c_uint64(nq), C.c_uint64(qstride),
            _p(starts), C.c_uint32(len(starts)), C.c_int64(k), C.c_int64(beam), C.c_double(cut), C.c_int64(limit),
            C.c_int64(degree_limit), C.c_uint32(out_k), _p(res["ids"]), _p(res["dists"]), _p(res["frontier_size"]),
            _p(res["visited_count"]), _p(res["dist_cmps"]), _p(res["degree_sum"]), C.c_uint32(visited_cap),
            _p(res["visited_ids"]), _p(res["visited_dists"]), _p(res["visit_order_ids"]),
            C.c_int(threads or self.threads))
        res["rc"] = rc
        return res

    def robust_prune_batch(self, points, graph, owners, cand_ids, cand_dists, cand_offsets, alpha, R, add=True,
                           metric="l2", threads=None):
        points = np.ascontiguousarray(points); graph = np.ascontiguousarray(graph, dtype=np.uint32)
        n, d = points.shape
        owners = np.ascontiguousarray(owners, dtype=np.uint32)
        cand_ids = np.ascontiguousarray(cand_ids, dtype=np.uint32)
        cd = None if cand_dists is None else np.ascontiguousarray(cand_dists, dtype=np.float32)
        off = np.ascontiguousarray(cand_offsets, dtype=np.uint64)
        m = len(owners)
        rows = np.zeros((m, R + 1), np.uint32); dc = np.zeros(m, np.uint32)
        rc = self.lib.pann_oracle_robust_prune_batch(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(graph.shape[1] - 1), _p(owners), C.c_uint64(m), _p(cand_ids),
            _p(cd), _p(off), C.c_double(alpha), C.c_uint32(R), C.c_int(1 if add else 0), _p(rows), _p(dc),
            C.c_int(threads or self.threads))
        assert rc == 0
        return rows, dc

    def permutation(self, m, seed):
        out = np.empty(m, np.uint32)
        self.lib.pann_oracle_permutation(C.c_uint64(m), C.c_uint64(seed), _p(out))
        return out

    def vamana_insert_batch(self, points, graph, batch, R, L, alpha, start=0, metric="l2", threads=None):
        """In-place on `graph` (reference layout)."""
        points = np.ascontiguousarray(points)
        assert graph.dtype == np.uint32 and graph.flags.c_contiguous
        n, d = points.shape
        batch = np.ascontiguousarray(batch, dtype=np.uint32)
        stats = np.zeros(6, np.uint64)
        rc = self.lib.pann_oracle_vamana_insert_batch(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(graph.shape[1] - 1), _p(batch), C.c_uint64(len(batch)),
            C.c_uint32(start), C.c_uint32(R), C.c_uint32(L), C.c_double(alpha), _p(stats),
            C.c_int(threads or self.threads))
        assert rc == 0
        return stats

    def vamana_phase_a(self, points, graph, batch, R, L, alpha, start=0, metric="l2", threads=None):
        """vamana/index.h:247-266 for the batch points given; returns rows [m, R], unused slots 0xFFFFFFFF"""
        points = np.ascontiguousarray(points)
        n, d = points.shape
        batch = np.ascontiguousarray(batch, dtype=np.uint32)
        rows = np.empty((len(batch), R), np.uint32)
        rc = self.lib.pann_oracle_vamana_phase_a(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(graph.shape[1] - 1), _p(batch), C.c_uint64(len(batch)),
            C.c_uint32(start), C.c_uint32(R), C.c_uint32(L), C.c_double(alpha), _p(rows), None, C.c_int(threads or self.threads))
        assert rc == 0
        return rows

    def vamana_phase_b(self, points, graph, batch, rows, R, alpha, metric="l2", threads=None):
        """vamana/index.h:268-300 for the whole batch, in place on `graph`"""
        points = np.ascontiguousarray(points)
        assert graph.dtype == np.uint32 and graph.flags.c_contiguous
        n, d = points.shape
        batch = np.ascontiguousarray(batch, dtype=np.uint32); rows = np.ascontiguousarray(rows, dtype=np.uint32)
        rc = self.lib.pann_oracle_vamana_phase_b(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(graph.shape[1] - 1), _p(batch), C.c_uint64(len(batch)), _p(rows),
            C.c_uint32(R), C.c_double(alpha), None, C.c_int(threads or self.threads))
        assert rc == 0

    def sort_neighbors(self, points, graph, metric="l2"):
        """vamana/index.h:180-185 (ties by id), in place: a zero-pass build with the final sort"""
        points = np.ascontiguousarray(points)
        n, d = points.shape
        rc = self.lib.pann_oracle_vamana_build(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(graph.shape[1] - 1), C.c_uint32(graph.shape[1] - 1), C.c_uint32(1),
            C.c_double(1.0), C.c_int(0), C.c_uint64(0), C.c_int(1), None, C.c_int(self.threads))
        assert rc == 0

    def vamana_build(self, points, R, L, alpha, num_passes=1, seed=1, sort_neighbors=True, metric="l2",
                     max_degree=None, threads=None, point_stats=None, single_batch=0):
        """point_stats: optional (visited[n], dists[n]) uint32 arrays, accumulated like the reference's BuildStats"""
        points = np.ascontiguousarray(points)
        n, d = points.shape
        maxdeg = R if max_degree is None else max_degree
        graph = np.zeros((n, maxdeg + 1), np.uint32)
        stats = np.zeros(6, np.uint64)
        if single_batch:
            rc = self.lib.pann_oracle_vamana_build_single_batch(
                _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
                C.c_int(_m(metric)), _p(graph), C.c_uint32(maxdeg), C.c_uint32(R), C.c_uint32(L), C.c_double(alpha),
                C.c_int(num_passes), C.c_uint32(single_batch), C.c_uint64(seed), C.c_int(1 if sort_neighbors else 0), _p(stats),
                C.c_int(threads or self.threads))
            assert rc == 0
            return graph, stats
        if point_stats is not None:
            self.lib.pann_oracle_set_build_point_stats(_p(point_stats[0]), _p(point_stats[1]))
        rc = self.lib.pann_oracle_vamana_build(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(maxdeg), C.c_uint32(R), C.c_uint32(L), C.c_double(alpha),
            C.c_int(num_passes), C.c_uint64(seed), C.c_int(1 if sort_neighbors else 0), _p(stats),
            C.c_int(threads or self.threads))
        self.lib.pann_oracle_set_build_point_stats(None, None)
        assert rc == 0
        return graph, stats

    def bruteforce_knn(self, points, queries, k, metric="l2", threads=None):
        points = np.ascontiguousarray(points); queries = np.ascontiguousarray(queries)
        n, d = points.shape
        nq = len(queries)
        ids = np.empty((nq, k), np.uint32); dists = np.empty((nq, k), np.float32)
        rc = self.lib.pann_oracle_bruteforce_knn(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(queries), C.c_uint64(nq), C.c_uint64(queries.strides[0]), C.c_uint32(k),
            _p(ids), _p(dists), C.c_int(threads or self.threads))
        assert rc == 0
        return ids, dists

    def leaf_knn(self, points, ids, m, metric="l2", threads=None):
        points = np.ascontiguousarray(points); ids = np.ascontiguousarray(ids, dtype=np.uint32)
        n, d = points.shape
        N = len(ids)
        oi = np.empty((N, m), np.uint32); od = np.empty((N, m), np.float32)
        rc = self.lib.pann_oracle_leaf_knn(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(ids), C.c_uint32(N), C.c_uint32(m), _p(oi), _p(od),
            C.c_int(threads or self.threads))
        assert rc == 0
        return oi, od

    def range_search(self, points, graph, starts, radius_2, cap, queries=None, query_ids=None, metric="l2", threads=None):
        points = np.ascontiguousarray(points); graph = np.ascontiguousarray(graph, dtype=np.uint32)
        n, d = points.shape
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        per_query = starts.ndim == 2
        if queries is not None:
            queries = np.ascontiguousarray(queries); nq = len(queries)
        else:
            query_ids = np.ascontiguousarray(query_ids, dtype=np.uint32); nq = len(query_ids)
        ids = np.full((nq, cap), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(nq, np.uint32); cmps = np.zeros(nq, np.uint32); trunc = np.zeros(nq, np.uint32)
        rc = self.lib.pann_oracle_range_search(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(graph.shape[1] - 1),
            _p(queries) if queries is not None else None, C.c_uint64(queries.strides[0] if queries is not None else 0),
            _p(query_ids) if query_ids is not None else None, C.c_uint64(nq), _p(starts), C.c_uint32(starts.shape[-1]),
            C.c_int(1 if per_query else 0), C.c_float(radius_2), C.c_uint32(cap), _p(ids), _p(cnt), _p(cmps), _p(trunc),
            C.c_int(threads or self.threads))
        assert rc == 0
        return {"ids": ids, "counts": cnt, "dist_cmps": cmps, "truncated": trunc}

    def recall(self, result_ids, gt_ids, gt_dists, k):
        r = np.ascontiguousarray(result_ids, dtype=np.uint32)
        g = np.ascontiguousarray(gt_ids, dtype=np.uint32); gd = np.ascontiguousarray(gt_dists, dtype=np.float32)
        return float(self.lib.pann_oracle_recall(_p(r), C.c_uint32(r.shape[1]), _p(g), _p(gd),
                                                 C.c_uint32(g.shape[1]), C.c_uint64(len(r)), C.c_uint32(k)))

    # ---- quantisation ----
    def euclid_u8_params(self, x):
        x = np.ascontiguousarray(x, np.float32); out = np.zeros(2, np.float32)
        self.lib.pann_oracle_euclid_u8_params(_p(x), C.c_uint64(x.shape[0]), C.c_uint32(x.shape[1]), _p(out))
        return np.float32(out[0]), int(out[1])

    def euclid_u8_translate(self, x, slope, offset):
        x = np.ascontiguousarray(x, np.float32); out = np.empty(x.shape, np.uint8)
        self.lib.pann_oracle_euclid_u8_translate(_p(x), C.c_uint64(x.shape[0]), C.c_uint32(x.shape[1]), C.c_float(slope),
                                                 C.c_int32(offset), _p(out))
        return out

    def normalize(self, x):
        x = np.array(x, np.float32, copy=True, order="C")
        self.lib.pann_oracle_normalize(_p(x), C.c_uint64(x.shape[0]), C.c_uint32(x.shape[1]))
        return x

    def mips_i8_maxval(self, x, trim=True):
        x = np.ascontiguousarray(x, np.float32)
        return np.float32(self.lib.pann_oracle_mips_i8_maxval(_p(x), C.c_uint64(x.shape[0]), C.c_uint32(x.shape[1]),
                                                              C.c_int(1 if trim else 0)))

    def mips_i8_translate(self, x, mv):
        x = np.ascontiguousarray(x, np.float32); out = np.empty(x.shape, np.int8)
        self.lib.pann_oracle_mips_i8_translate(_p(x), C.c_uint64(x.shape[0]), C.c_uint32(x.shape[1]), C.c_float(mv), _p(out))
        return out

    def hcnng_build(self, points, num_clusters, cluster_size, mst_deg, seed=1, metric="l2", threads=None):
        points = np.ascontiguousarray(points)
        n, d = points.shape
        maxdeg = num_clusters * mst_deg
        graph = np.zeros((n, maxdeg + 1), np.uint32)
        rc = self.lib.pann_oracle_hcnng_build(
            _p(points), C.c_uint64(n), C.c_uint32(d), C.c_int(DT[points.dtype]), C.c_uint64(points.strides[0]),
            C.c_int(_m(metric)), _p(graph), C.c_uint32(maxdeg), C.c_long(num_clusters), C.c_long(cluster_size),
            C.c_long(mst_deg), C.c_uint64(seed), C.c_int(threads or self.threads))
        assert rc == 0
        return graph
