"""GPU parity on REAL-valued float data in the "exact float order" validation mode
(pann_index_set_exact_float_order): every distance is summed left to right with one rounding per
operation, as the reference's scalar loops (euclidian_point.h:83-90, mips_point.h:59-65) and the
oracle do, so everything downstream -- search results, counters, pruned lists, whole graphs --
must be BIT-IDENTICAL to the oracle even though the values are not small integers."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, datasets

pytestmark = pytest.mark.gpu


def _norm(G):
    G = G.copy()
    cols = np.arange(G.shape[1] - 1)[None, :]
    G[:, 1:][cols >= G[:, :1]] = 0
    return G


def _real(n, d, dtype, metric, seed):
    X = datasets.deep_like(n, d, seed=seed) if metric == "l2" else datasets.t2i_like(n, d, seed=seed)
    return np.ascontiguousarray(X.astype(dtype))


@pytest.mark.parametrize("dtype,metric,d", [(np.float32, "l2", 96), (np.float32, "mips", 200), (np.float16, "l2", 128),
                                            (np.float16, "mips", 100), (np.float32, "l2", 32)])
def test_exact_search_bit_identical(oracle, dtype, metric, d):
    n, nq = 8000, 400
    X, Q = _real(n, d, dtype, metric, 1234), _real(nq, d, dtype, metric, 4321)
    G, _ = oracle.vamana_build(X, 32, 64, 1.2 if metric == "l2" else 1.0, seed=5, metric=metric)
    ix = DeviceIndex(X, G, metric=metric, exact_float_order=True)
    for beam, k, cut, limit in ((64, 10, 1.35, None), (20, 10, 0.0, None), (128, 100, 1.35, 90), (300, 10, 1.1, None)):
        o = oracle.batch_search(X, G, queries=Q, k=k, beam=beam, cut=cut, limit=limit, metric=metric)
        g = ix.batch_search(Q, k=k, beam=beam, cut=cut, limit=limit)
        np.testing.assert_array_equal(o["ids"], g["ids"])
        np.testing.assert_array_equal(o["dists"].view(np.uint32), g["dists"].view(np.uint32))
        for f in ("frontier_size", "visited_count", "dist_cmps", "degree_sum"):
            np.testing.assert_array_equal(o[f], g[f], err_msg=f)
    ix.close()


@pytest.mark.parametrize("dtype,metric,d", [(np.float32, "l2", 96), (np.float16, "mips", 128)])
def test_exact_vamana_build_identical_graph(oracle, dtype, metric, d):
    n = 6000
    X = _real(n, d, dtype, metric, 77)
    alpha = 1.2 if metric == "l2" else 1.0
    Go, so = oracle.vamana_build(X, 32, 64, alpha, num_passes=2, seed=11, metric=metric)
    ix = DeviceIndex(X, max_degree=32, metric=metric, exact_float_order=True)
    st = ix.vamana_build(32, 64, alpha, num_passes=2, seed=11)
    np.testing.assert_array_equal(_norm(Go), _norm(ix.get_graph()))
    assert int(so[0]) == st.search_dist_cmps and int(so[1]) == st.prune_dist_cmps
    ix.close()


@pytest.mark.parametrize("dtype,metric,d", [(np.float32, "l2", 96), (np.float16, "l2", 128), (np.float32, "mips", 200)])
def test_exact_dense_paths(oracle, dtype, metric, d):
    n = 3000
    X, Q = _real(n, d, dtype, metric, 9), _real(64, d, dtype, metric, 10)
    ix = DeviceIndex(X, max_degree=8, metric=metric, exact_float_order=True)
    ids = np.random.default_rng(1).choice(n, 700, replace=False).astype(np.uint32)
    oi, od = oracle.leaf_knn(X, ids, 10, metric)
    gi, gd = ix.leaf_knn(ids, 10)
    np.testing.assert_array_equal(oi, gi)
    np.testing.assert_array_equal(od.view(np.uint32), gd.view(np.uint32))
    bi, bd = oracle.bruteforce_knn(X, Q, 20, metric)
    ci, cd = ix.bruteforce_knn(Q, 20)
    np.testing.assert_array_equal(bi, ci)
    np.testing.assert_array_equal(bd.view(np.uint32), cd.view(np.uint32))
    a = np.random.default_rng(2).integers(0, n, 500).astype(np.uint32)
    b = np.random.default_rng(3).integers(0, n, 500).astype(np.uint32)
    want = np.array([oracle.distance(X[i], X[j], metric) for i, j in zip(a, b)], np.float32)
    np.testing.assert_array_equal(want.view(np.uint32), ix.pair_distances(a, b).view(np.uint32))
    qd = ix.query_distances(Q[:4], a[:100])
    want = np.array([[oracle.distance(X[j], q, metric) for j in a[:100]] for q in Q[:4]], np.float32)
    np.testing.assert_array_equal(want.view(np.uint32), qd.view(np.uint32))
    ix.close()


def test_exact_hcnng_build_identical_graph(oracle):
    n = 5000
    X = _real(n, 96, np.float32, "l2", 31)
    Go = oracle.hcnng_build(X, num_clusters=4, cluster_size=200, mst_deg=3, seed=7)
    ix = DeviceIndex(X, max_degree=Go.shape[1] - 1, exact_float_order=True)
    ix.hcnng_build(4, 200, 3, seed=7)
    np.testing.assert_array_equal(_norm(Go), _norm(ix.get_graph()))
    ix.close()


def test_exact_mode_is_a_noop_for_integer_types(oracle):
    X = datasets.sift_like(3000, 128, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(100, 128, seed=2, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, 24, 48, 1.2, seed=5)
    a = DeviceIndex(X, G)
    b = DeviceIndex(X, G, exact_float_order=True)
    ra, rb = a.batch_search(Q, k=10, beam=32), b.batch_search(Q, k=10, beam=32)
    np.testing.assert_array_equal(ra["ids"], rb["ids"])
    np.testing.assert_array_equal(ra["dist_cmps"], rb["dist_cmps"])
    a.close(); b.close()
