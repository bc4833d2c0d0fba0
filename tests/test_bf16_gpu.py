"""PANN_BF16 (north_star: "MFMA ... for the dense fp16/bf16 ... contraction"): the bfloat16 element type through every kernel
family -- gather-distance beam search (register frontier b64 / b128 and the LDS-frontier kernel), robustPrune, Vamana and HCNNG
builds, the dense all-pairs kernels (v_mfma_f32_16x16x32_bf16; VALU in exact-float-order mode) -- bit-exact against the oracle
on integer-valued data (exact in bf16 up to 256; every f32 partial sum is an integer < 2^24, so any summation order agrees),
and in exact-float-order mode on real-valued data."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, bfloat16, datasets, from_bf16, to_bf16

pytestmark = pytest.mark.gpu


def _cmp(o, g, fields=("ids", "dists", "frontier_size", "visited_count", "dist_cmps")):
    for f in fields:
        np.testing.assert_array_equal(o[f], g[f], err_msg=f)


@pytest.mark.parametrize("metric", ["l2", "mips"])
@pytest.mark.parametrize("d", [128, 96, 200])
def test_bf16_search_matches_oracle(oracle, metric, d):
    n, nq = 6000, 150
    X = datasets.sift_like(n, d, seed=11, dtype=bfloat16)
    Q = datasets.sift_like(nq, d, seed=12, dtype=bfloat16)
    assert X.dtype == bfloat16 and from_bf16(X).max() <= 255
    G, _ = oracle.vamana_build(X, 32, 64, 1.2 if metric == "l2" else 1.0, seed=3, metric=metric)
    ix = DeviceIndex(X, G, metric=metric)
    for beam, k in ((20, 5), (64, 10), (100, 10), (300, 20)):
        o = oracle.batch_search(X, G, queries=Q, k=k, beam=beam, metric=metric)
        g = ix.batch_search(Q, k=k, beam=beam)
        _cmp(o, g)
    ix.close()


def test_bf16_vamana_and_hcnng_builds_identical_to_oracle(oracle):
    X = datasets.sift_like(5000, 128, seed=21, dtype=bfloat16)
    ix = DeviceIndex(X, max_degree=32)
    ix.vamana_build(32, 64, 1.2, num_passes=2, seed=4)
    G = ix.get_graph()
    Go, _ = oracle.vamana_build(X, 32, 64, 1.2, num_passes=2, seed=4)
    cols = np.arange(32)[None, :]
    np.testing.assert_array_equal(G[:, 0], Go[:, 0])
    np.testing.assert_array_equal(np.where(cols < G[:, :1], G[:, 1:], 0), np.where(cols < Go[:, :1], Go[:, 1:], 0))
    ix.close()
    ih = DeviceIndex(X, max_degree=12)
    ih.hcnng_build(4, 300, 3, seed=9)                               # leaf kNN on the matrix cores (bf16 MFMA)
    np.testing.assert_array_equal(ih.get_graph(), oracle.hcnng_build(X, 4, 300, 3, seed=9))
    ih.close()


@pytest.mark.parametrize("metric", ["l2", "mips"])
def test_bf16_dense_mfma_matches_oracle(oracle, metric):
    X = datasets.sift_like(9000, 128, seed=31, dtype=bfloat16)
    Q = datasets.sift_like(70, 128, seed=32, dtype=bfloat16)
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    for k in (100, 10, 16, 17):                          # register lists of 8 / 1 registers per row
        gi, gd = ix.bruteforce_knn(Q, k)
        oi, od = oracle.bruteforce_knn(X, Q, k, metric)
        np.testing.assert_array_equal(gi, oi); np.testing.assert_array_equal(gd, od)
    ids = np.random.default_rng(1).choice(len(X), 700, replace=False).astype(np.uint32)
    li, ld = ix.leaf_knn(ids, 10)
    loi, lod = oracle.leaf_knn(X, ids, 10, metric)
    np.testing.assert_array_equal(li, loi); np.testing.assert_array_equal(ld, lod)
    # distances of single pairs through the gather path
    a = np.arange(0, 500, dtype=np.uint32); b = a[::-1].copy()
    want = np.array([oracle.distance(X[i], X[j], metric) for i, j in zip(a, b)], np.float32)
    np.testing.assert_array_equal(ix.pair_distances(a, b), want)
    ix.close()


def test_bf16_real_valued_exact_float_order(oracle):
    """real-valued data: bit-identical only when the device sums left to right like the CPU (validation mode)"""
    Xf = datasets.deep_like(4000, 96, seed=41); Qf = datasets.deep_like(100, 96, seed=42)
    X, Q = to_bf16(Xf), to_bf16(Qf)
    np.testing.assert_allclose(from_bf16(X), Xf, rtol=2 ** -8)
    G, _ = oracle.vamana_build(X, 24, 48, 1.2, seed=5)
    ix = DeviceIndex(X, G, exact_float_order=True)
    for beam in (32, 100):
        _cmp(oracle.batch_search(X, G, queries=Q, k=10, beam=beam), ix.batch_search(Q, k=10, beam=beam))
    gi, gd = ix.bruteforce_knn(Q, 20)
    oi, od = oracle.bruteforce_knn(X, Q, 20)
    np.testing.assert_array_equal(gi, oi); np.testing.assert_array_equal(gd, od)
    ix.close()
    # the fast path agrees to rounding: same neighbours for nearly every query
    fx = DeviceIndex(X, G)
    r = fx.batch_search(Q, k=10, beam=64)
    o = oracle.batch_search(X, G, queries=Q, k=10, beam=64)
    assert np.mean(r["ids"] == o["ids"]) > 0.99
    np.testing.assert_allclose(r["dists"], o["dists"], rtol=1e-5)
    fx.close()
