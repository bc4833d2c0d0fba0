"""Python mirror of python/graph_index.cpp (GraphIndex<T,Point>, :48-337) and of the six index
classes python/module.cpp registers (:50-57,150-155), over the C-ABI.

    Index(data_path, index_path)                       (positional order of graph_index.cpp:82)
    .batch_search(queries, knn, beam_width, quant=False, visit_limit=-1) -> (uint32[nq,knn], float32[nq,knn])
    .single_search(q, knn, beam_width, quant, visit_limit)               -> uint32[knn]
    .batch_search_from_string(queries_path, knn, beam_width, quant=False, visit_limit=-1)
    .check_recall(queries_file, gt_file, neighbors, k)   prints "Recall: x"
"""
import numpy as np

from . import io, quantize
from .index import DeviceIndex
from .recall import recall_at_k


class GraphIndex:
    T = None
    metric = None

    def __init__(self, data_path, index_path, hnsw=False, device=0):
        if hnsw:
            raise NotImplementedError("HNSW indices are out of scope (SURVEY.md section 2 #17)")
        self.points = io.read_bin(data_path, self.T)
        self.graph = io.read_graph(index_path)
        if len(self.graph) != len(self.points):        # graph_index.cpp:113-116
            raise RuntimeError("graph size and point size do not match")
        self.use_quantization = np.dtype(self.T).itemsize > 1     # :86
        self.q_index = None
        pts = self.points
        if self.use_quantization:
            if self.metric == "Euclidian":
                self.eparams = quantize.euclid_u8_params(pts)                        # EQuantRange(Points) :90
                qpts = quantize.euclid_u8_translate(pts, self.eparams)
                self.q_index = DeviceIndex(qpts, self.graph, metric="Euclidian", device=device)
            else:
                pts = quantize.normalize_rows(pts)                                   # :94-95
                self.points = pts
                self.mmax = quantize.mips_i8_max_val(pts, trim=True)                 # Quantized_Mips_Point<8,true> :69
                qpts = quantize.mips_i8_translate(pts, self.mmax)
                self.q_index = DeviceIndex(qpts, self.graph, metric="mips", device=device)
        self.index = DeviceIndex(pts, self.graph, metric=self.metric, device=device)

    # QueryParams(knn, beam, 1.35, visit_limit, min(maxDeg, 3*visit_limit))   (:198,:222,:242)
    def _qp(self, knn, beam_width, visit_limit):
        return dict(k=knn, beam=beam_width, cut=1.35, limit=visit_limit,
                    degree_limit=min(self.index.max_degree, 3 * visit_limit))

    def _search(self, queries, knn, beam_width, quant, visit_limit):           # search_dispatch :120-190
        queries = np.ascontiguousarray(queries, dtype=self.T)
        qp = self._qp(knn, beam_width, visit_limit)
        if not (quant and self.use_quantization):
            r = self.index.batch_search(queries, out_k=knn, **qp)                # :188
            self._need(r["frontier_size"], knn)
            return r["ids"], r["dists"]
        if self.metric == "Euclidian":
            qq = quantize.euclid_u8_translate(queries, self.eparams)
            if self.eparams.identity:                                            # slope == 1: plain search on the u8 copy (:148-152)
                r = self.q_index.batch_search(qq, out_k=knn, **qp)
                self._need(r["frontier_size"], knn)
                return r["ids"], r["dists"]
            full_q = queries
        else:
            full_q = quantize.normalize_rows(queries)                            # q.normalize() :172
            qq = quantize.mips_i8_translate(full_q, self.mmax)
        # beam_search_rerank (beamSearch.h:390-454): search the quantised copy, re-score the first
        # min(k * rerank_factor, |beam|) with exact distances, sort, keep k
        r = self.q_index.batch_search(qq, out_k=beam_width, **qp)
        self._need(r["frontier_size"], knn)
        counts = np.minimum(r["frontier_size"], knn * 100).astype(np.uint32)     # QP.rerank_factor = 100 (types.h:224)
        return self.index.rerank(full_q, r["ids"], counts, knn, resort=True)

    @staticmethod
    def _need(frontier_size, knn):
        if len(frontier_size) and int(frontier_size.min()) < knn:               # beamSearch.h:416-419
            raise RuntimeError(f"Error: beam search returned {int(frontier_size.min())} elements, which is less than k = {knn}")

    def batch_search(self, queries, knn, beam_width, quant=False, visit_limit=-1):
        return self._search(queries, knn, beam_width, quant, visit_limit)

    def single_search(self, q, knn, beam_width, quant, visit_limit):
        ids, _ = self._search(np.asarray(q)[None, :], knn, beam_width, quant, visit_limit)
        return ids[0]

    def batch_search_from_string(self, queries, knn, beam_width, quant=False, visit_limit=-1):
        return self._search(io.read_bin(queries, self.T), knn, beam_width, quant, visit_limit)

    def check_recall(self, queries_file, graph_file, neighbors, k):             # :259-305
        gt_ids, _ = io.read_ibin(graph_file)
        neighbors = np.asarray(neighbors)
        if neighbors.size and (neighbors[:, :k].max() >= len(self.points)):
            raise RuntimeError("neighbor reported by query out of range")
        # resolve_eq_distances (:263,:275-283): the tie set comes from distances RECOMPUTED between the query file's points and
        # this index's (full-precision) points, not from the distance column of the ground-truth file -- one rerank launch,
        # resort off = the given order with exact distances
        queries = io.read_bin(queries_file, self.T)
        _, gt_d = self.index.rerank(np.ascontiguousarray(queries, dtype=self.T), gt_ids, None, gt_ids.shape[1], resort=False)
        rec = recall_at_k(neighbors, gt_ids, gt_d, k)
        print(f"Recall: {rec:.6g}")
        return rec


def _mk(name, T, metric):
    return type(name, (GraphIndex,), {"T": T, "metric": metric})


FloatEuclidianIndex = _mk("FloatEuclidianIndex", np.float32, "Euclidian")
FloatMipsIndex = _mk("FloatMipsIndex", np.float32, "mips")
UInt8EuclidianIndex = _mk("UInt8EuclidianIndex", np.uint8, "Euclidian")
UInt8MipsIndex = _mk("UInt8MipsIndex", np.uint8, "mips")
Int8EuclidianIndex = _mk("Int8EuclidianIndex", np.int8, "Euclidian")
Int8MipsIndex = _mk("Int8MipsIndex", np.int8, "mips")
