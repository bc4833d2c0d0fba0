"""The Python mirror of python/wrapper.py + python/graph_index.cpp (names, argument orders,
quantised search + rerank) over the C-ABI, checked against compositions of oracle calls."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, datasets, io, quantize, wrapper

pytestmark = pytest.mark.gpu


def _overlap(a, b):
    return np.mean([len(set(x.tolist()) & set(y.tolist())) / len(x) for x, y in zip(a, b)])


def test_uint8_euclidian_index_end_to_end(tmp_path, oracle, capsys):
    X = datasets.sift_like(6000, 128, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(150, 128, seed=4321, dtype=np.uint8)
    io.write_bin(tmp_path / "b.bin", X); io.write_bin(tmp_path / "q.bin", Q)
    wrapper.build_vamana_index("Euclidian", "uint8", str(tmp_path / "b.bin"), str(tmp_path / "g"), 32, 64, 1.2, False, seed=4)
    Go, _ = oracle.vamana_build(X, 32, 64, 1.2, num_passes=1, seed=4)
    G = io.read_graph(tmp_path / "g")
    np.testing.assert_array_equal(G[:, 0], Go[:, 0])
    Index = wrapper.load_index("Euclidian", "uint8", str(tmp_path / "b.bin"), str(tmp_path / "g"))
    ids, dists = Index.batch_search(Q, 10, 64, True, 1000)          # quant has no effect for 1-byte types (:86)
    o = oracle.batch_search(X, G, queries=Q, k=10, beam=64, cut=1.35, limit=1000, degree_limit=min(32, 3000))
    np.testing.assert_array_equal(ids, o["ids"]); np.testing.assert_array_equal(dists, o["dists"])
    ids2, _ = Index.batch_search_from_string(str(tmp_path / "q.bin"), 10, 64, True, 1000)
    np.testing.assert_array_equal(ids2, ids)
    np.testing.assert_array_equal(Index.single_search(Q[3], 10, 64, False, 1000), ids[3])
    # degree_limit = min(maxDeg, 3*visit_limit) (graph_index.cpp:198)
    ids3, _ = Index.batch_search(Q, 10, 64, False, 5)
    o3 = oracle.batch_search(X, G, queries=Q, k=10, beam=64, cut=1.35, limit=5, degree_limit=15)
    np.testing.assert_array_equal(ids3, o3["ids"])
    gt, gd = oracle.bruteforce_knn(X, Q, 100)
    io.write_ibin(tmp_path / "gt", gt, gd)
    rec = Index.check_recall(str(tmp_path / "q.bin"), str(tmp_path / "gt"), ids, 10)
    assert "Recall: " in capsys.readouterr().out and abs(rec - oracle.recall(ids, gt, gd, 10)) < 1e-9
    # the tie set is recomputed from the points (graph_index.cpp:275-283): a ground-truth file whose distance column is
    # wrong (other tools store sqrt'd / rounded / differently signed distances) gives the same recall
    io.write_ibin(tmp_path / "gt_bad", gt, np.zeros_like(gd))
    assert Index.check_recall(str(tmp_path / "q.bin"), str(tmp_path / "gt_bad"), ids, 10) == rec
    with pytest.raises(Exception):
        wrapper.load_index("cosine", "uint8", "x", "y")


def test_float_euclidian_quantised_search_and_rerank(tmp_path, oracle):
    X = (datasets.deep_like(6000, 96, seed=1) * 2.0).astype(np.float32)
    Q = (datasets.deep_like(120, 96, seed=2) * 2.0).astype(np.float32)
    io.write_bin(tmp_path / "b.bin", X)
    wrapper.build_vamana_index("Euclidian", "float", str(tmp_path / "b.bin"), str(tmp_path / "g"), 32, 64, 1.2, False)
    G = io.read_graph(tmp_path / "g")
    Index = wrapper.load_index("Euclidian", "float", str(tmp_path / "b.bin"), str(tmp_path / "g"))
    assert Index.use_quantization and not Index.eparams.identity
    ids, dists = Index.batch_search(Q, 10, 64, True, 1000)
    # oracle composition of beam_search_rerank (beamSearch.h:390-454): search the u8 copy, exact re-score, sort
    slope, offset = oracle.euclid_u8_params(X)
    Xq = oracle.euclid_u8_translate(X, slope, offset); Qq = oracle.euclid_u8_translate(Q, slope, offset)
    o = oracle.batch_search(Xq, G, queries=Qq, k=10, beam=64, cut=1.35, limit=1000, degree_limit=32, out_k=64)
    exp = []
    for i in range(len(Q)):
        c = o["ids"][i, :min(o["frontier_size"][i], 1000)]
        d = np.array([oracle.distance(Q[i], X[j]) for j in c], np.float32)
        exp.append(c[np.lexsort((c, d))][:10])
    exp = np.array(exp)
    assert _overlap(ids, exp) > 0.995        # real-valued floats: ulp-level ties may reorder (DESIGN.md float order)
    ex_d = np.array([[oracle.distance(Q[i], X[j]) for j in ids[i]] for i in range(len(Q))], np.float32)
    np.testing.assert_allclose(dists, ex_d, rtol=2e-6, atol=1e-7)
    # un-quantised search on the float points
    ids_f, _ = Index.batch_search(Q, 10, 64, False, 1000)
    of = oracle.batch_search(X, G, queries=Q, k=10, beam=64, cut=1.35, limit=1000, degree_limit=32)
    assert _overlap(ids_f, of["ids"]) > 0.99


def test_float_integer_valued_uses_plain_u8_copy(tmp_path, oracle):
    X = datasets.sift_like(5000, 128, seed=1234, dtype=np.float32)
    Q = datasets.sift_like(100, 128, seed=4321, dtype=np.float32)
    io.write_bin(tmp_path / "b.bin", X)
    wrapper.build_vamana_index("Euclidian", "float", str(tmp_path / "b.bin"), str(tmp_path / "g"), 32, 64, 1.2, True)
    G = io.read_graph(tmp_path / "g")
    Index = wrapper.load_index("Euclidian", "float", str(tmp_path / "b.bin"), str(tmp_path / "g"))
    assert Index.eparams.identity                                   # slope 1 -> beam_search on the u8 copy (:148-152)
    ids, dists = Index.batch_search(Q, 10, 64, True, 1000)
    o = oracle.batch_search(X.astype(np.uint8), G, queries=Q.astype(np.uint8), k=10, beam=64, cut=1.35, limit=1000,
                            degree_limit=32)
    np.testing.assert_array_equal(ids, o["ids"]); np.testing.assert_array_equal(dists, o["dists"])
    ids_f, d_f = Index.batch_search(Q, 10, 64, False, 1000)         # integer-valued floats: bit-exact as well
    np.testing.assert_array_equal(ids_f, o["ids"]); np.testing.assert_array_equal(d_f, o["dists"])


def test_float_mips_quantised(tmp_path, oracle):
    X = datasets.t2i_like(6000, 200, seed=1)
    Q = datasets.t2i_like(100, 200, seed=2)
    io.write_bin(tmp_path / "b.bin", X)
    wrapper.build_vamana_index("mips", "float", str(tmp_path / "b.bin"), str(tmp_path / "g"), 40, 80, 1.0, False)
    G = io.read_graph(tmp_path / "g")
    assert G[:, 0].max() <= 40 and G[:, 0].mean() > 5
    Index = wrapper.load_index("mips", "float", str(tmp_path / "b.bin"), str(tmp_path / "g"))
    ids, dists = Index.batch_search(Q, 10, 64, True, 1000)
    Xn = oracle.normalize(X); Qn = oracle.normalize(Q)
    mv = oracle.mips_i8_maxval(Xn, trim=True)
    Xq = oracle.mips_i8_translate(Xn, mv); Qq = oracle.mips_i8_translate(Qn, mv)
    o = oracle.batch_search(Xq, G, queries=Qq, k=10, beam=64, cut=1.35, limit=1000, degree_limit=40, metric="mips", out_k=64)
    exp = []
    for i in range(len(Q)):
        c = o["ids"][i, :min(o["frontier_size"][i], 1000)]
        d = np.array([oracle.distance(Qn[i], Xn[j], "mips") for j in c], np.float32)
        exp.append(c[np.lexsort((c, d))][:10])
    assert _overlap(ids, np.array(exp)) > 0.995
    gt, gd = oracle.bruteforce_knn(Xn, Qn, 100, metric="mips")
    assert oracle.recall(ids, gt, gd, 10) > 0.8


def test_hcnng_wrapper(tmp_path, oracle):
    X = datasets.sift_like(6000, 128, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(100, 128, seed=4321, dtype=np.uint8)
    io.write_bin(tmp_path / "b.bin", X)
    wrapper.build_hcnng_index("Euclidian", "uint8", str(tmp_path / "b.bin"), str(tmp_path / "h"), 3, 12, 300)
    G = io.read_graph(tmp_path / "h")
    assert G.shape[1] - 1 == 36
    Index = wrapper.load_index("Euclidian", "uint8", str(tmp_path / "b.bin"), str(tmp_path / "h"))
    ids, _ = Index.batch_search(Q, 10, 64, False, 1000)
    gt, gd = oracle.bruteforce_knn(X, Q, 100)
    assert oracle.recall(ids, gt, gd, 10) > 0.9
    o = oracle.batch_search(X, G, queries=Q, k=10, beam=64, cut=1.35, limit=1000, degree_limit=36)
    np.testing.assert_array_equal(ids, o["ids"])


def test_rerank_no_resort_keeps_order(oracle):
    from parlayann_amd import DeviceIndex
    X = datasets.sift_like(3000, 64, seed=1, dtype=np.float32)
    Q = datasets.sift_like(20, 64, seed=2, dtype=np.float32)
    ix = DeviceIndex(X, max_degree=8)
    rng = np.random.default_rng(0)
    cand = rng.integers(0, len(X), (len(Q), 50)).astype(np.uint32)
    cnt = rng.integers(5, 51, len(Q)).astype(np.uint32)
    ids, d = ix.rerank(Q, cand, cnt, 5, resort=False)                # beamSearch.h:447-452
    np.testing.assert_array_equal(ids, cand[:, :5])
    np.testing.assert_array_equal(d, np.array([[oracle.distance(Q[i], X[j]) for j in cand[i, :5]] for i in range(len(Q))], np.float32))
    ids, d = ix.rerank(Q, cand, cnt, 5, resort=True)
    for i in range(len(Q)):
        c = cand[i, :cnt[i]]
        dd = np.array([oracle.distance(Q[i], X[j]) for j in c], np.float32)
        # duplicates among random candidates stay duplicated, as std::sort would leave them
        order = np.lexsort((c, dd))[:5]
        np.testing.assert_array_equal(ids[i], c[order]); np.testing.assert_array_equal(d[i], dd[order])
    ix.close()


def test_search_and_parse_sweep(oracle):
    from parlayann_amd import DeviceIndex, sweep
    X = datasets.sift_like(8000, 64, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(200, 64, seed=4321, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, 32, 64, 1.2, seed=2)
    ix = DeviceIndex(X, G)
    gt, gd = ix.bruteforce_knn(Q, 100)
    results, (best, buckets) = sweep.search_and_parse(ix, Q, gt, gd, 10)
    assert len(results) == 43 + 20 + 1
    by_beam = {r["beamQ"]: r for r in results[:43]}
    # every sweep point equals the oracle at the same QueryParams (spot checks incl. beam 1000: HBM filter)
    for beam in (10, 38, 300, 1000):
        o = oracle.batch_search(X, G, queries=Q, k=10, beam=beam, cut=1.35)
        assert abs(by_beam[beam]["recall"] - oracle.recall(o["ids"], gt, gd, 10)) < 1e-9
        assert by_beam[beam]["avg_cmps"] == int(o["dist_cmps"].astype(np.uint64).sum() // len(Q))
    lim = results[43 + 5]                                         # limit 15: beam max(15,10), degree_limit min(32,75)
    o = oracle.batch_search(X, G, queries=Q, k=10, beam=15, cut=1.35, limit=15, degree_limit=32)
    assert abs(lim["recall"] - oracle.recall(o["ids"], gt, gd, 10)) < 1e-9
    best_acc = results[-1]
    o = oracle.batch_search(X, G, queries=Q, k=100, beam=1000, cut=10.0, out_k=10)
    assert abs(best_acc["recall"] - oracle.recall(o["ids"], gt, gd, 10)) < 1e-9
    assert buckets == sorted(buckets) and all(b <= r["recall"] for b, r in zip(buckets, best))
    assert by_beam[1000]["recall"] >= by_beam[10]["recall"]
    ix.close()


@pytest.mark.parametrize("dtype,metric,d", [(np.uint8, "Euclidian", 128), (np.int8, "mips", 200), (np.float16, "Euclidian", 96),
                                            (np.uint8, "Euclidian", 160), (np.int8, "mips", 192)])   # 192-byte row stride
def test_hcnng_build_identical_to_oracle(oracle, dtype, metric, d):
    """host tree + Kruskal around the device calls (pivot split, leaf kNN) vs. the all-CPU oracle:
    same seeding rules, integer-valued data -> the graphs must be identical, slot for slot."""
    X = datasets.sift_like(5000, d, seed=1234, dtype=np.float32)
    X = (X - 128).clip(-127, 127).astype(np.int8) if dtype == np.int8 else X.astype(dtype)
    G = wrapper.hcnng_build(X, metric, 6, 200, 3, seed=9)                       # all on the device
    Gh = wrapper.hcnng_build(X, metric, 6, 200, 3, seed=9, host_mirror=True)    # C++ host mirror around the device calls
    Go = oracle.hcnng_build(X, 6, 200, 3, seed=9, metric="l2" if metric == "Euclidian" else "mips")
    np.testing.assert_array_equal(Gh, Go)
    np.testing.assert_array_equal(G, Go)


@pytest.mark.parametrize("n,clusters,leaf,mst_deg", [(11000, 2, 5000, 3),      # leaves above 4096 members: Kruskal state in HBM scratch
                                                     (3000, 3, 40, 2),          # many small leaves, some smaller than the 10-NN lists
                                                     (2500, 2, 700, 5)])
def test_hcnng_build_leaf_size_extremes(oracle, n, clusters, leaf, mst_deg):
    X = datasets.sift_like(n, 64, seed=77, dtype=np.uint8)
    G = wrapper.hcnng_build(X, "Euclidian", clusters, leaf, mst_deg, seed=3)
    Go = oracle.hcnng_build(X, clusters, leaf, mst_deg, seed=3)
    np.testing.assert_array_equal(G, Go)


def test_hcnng_forest_groups_give_the_same_graph(oracle):
    """trees are split level by level in groups (all of them when group * n < 2^31; pann_index_set_forest_group bounds the
    scratch); the grouping must not matter"""
    X = datasets.sift_like(4000, 64, seed=3, dtype=np.uint8)
    Go = oracle.hcnng_build(X, 5, 150, 3, seed=4)
    for g in (1, 2, 5, 0):
        ix = DeviceIndex(X, max_degree=15)
        ix.set_option("forest_group", g)
        ix.hcnng_build(5, 150, 3, seed=4)
        np.testing.assert_array_equal(ix.get_graph(), Go)
        ix.clear_graph()                                              # pann_index_clear_graph: the rebuild starts from Graph(maxDeg, n)
        assert not ix.get_graph().any()
        ix.hcnng_build(5, 150, 3, seed=4)
        np.testing.assert_array_equal(ix.get_graph(), Go)
        ix.close()


def test_parlayannpy_dropin_names():
    from parlayann_amd import _ParlayANNpy as m
    for cls in ("FloatEuclidianIndex", "FloatMipsIndex", "UInt8EuclidianIndex", "UInt8MipsIndex", "Int8EuclidianIndex",
                "Int8MipsIndex"):
        assert hasattr(m, cls)
    assert sum(1 for n in m.__all__ if n.startswith("build_")) == 24
    assert m.defaults.ALPHA == 1.2 and m.defaults.GRAPH_DEGREE == 64 and m.defaults.BEAMWIDTH == 128
    with pytest.raises(NotImplementedError):
        m.build_hnsw_float_euclidian_index("Euclidian", "a", "b", 1, 2, 3.0, 4.0)


@pytest.mark.parametrize("dtype", [np.uint8, np.float16, np.float32])
def test_streamed_index_from_files_equals_in_memory_index(tmp_path, oracle, dtype):
    """DeviceIndex.from_files (pann_index_create_empty + pann_index_upload_points + pann_index_update_rows in chunks): the whole
    file, and one shard of it with its own graph file, give the searches of the in-memory index bit for bit"""
    from parlayann_amd import DeviceIndex, datasets, io
    n, d = 5000, 96
    X = datasets.sift_like(n, d, seed=1234, dtype=dtype)
    Q = datasets.sift_like(64, d, seed=4321, dtype=dtype)
    G, _ = oracle.vamana_build(X, 24, 48, 1.2, seed=3)
    io.write_bin(tmp_path / "base.bin", X); io.write_graph(tmp_path / "graph", G)
    ref = DeviceIndex(X, G)
    want = ref.batch_search(Q, k=10, beam=40)
    for chunk in (1 << 12, 1 << 30):                      # many small chunks (ragged last one) / a single chunk
        ix = DeviceIndex.from_files(tmp_path / "base.bin", dtype, graph_path=tmp_path / "graph", chunk_bytes=chunk)
        assert (ix.n, ix.d, ix.max_degree) == (n, d, 24)
        np.testing.assert_array_equal(ix.get_graph(), io.read_graph(tmp_path / "graph"))     # (slots past the degree are not in the file)
        got = ix.batch_search(Q, k=10, beam=40)
        for f in ("ids", "dists", "visited_count", "dist_cmps"):
            np.testing.assert_array_equal(got[f], want[f])
        ix.close()
    ref.close()
    # one shard: rows [1000, 3500) of the same file with the shard's own graph
    lo, hi = 1000, 3500
    Gs, _ = oracle.vamana_build(X[lo:hi], 16, 32, 1.2, seed=5)
    io.write_graph(tmp_path / "graph_s", Gs)
    sh = DeviceIndex.from_files(tmp_path / "base.bin", dtype, graph_path=tmp_path / "graph_s", rows=(lo, hi), chunk_bytes=1 << 14)
    o = oracle.batch_search(X[lo:hi], Gs, queries=Q, k=10, beam=32)
    g = sh.batch_search(Q, k=10, beam=32)
    np.testing.assert_array_equal(g["ids"], o["ids"]); np.testing.assert_array_equal(g["dists"], o["dists"])
    sh.close()
    # no graph file: an empty graph of the requested degree, ready for a device build
    e = DeviceIndex.from_files(tmp_path / "base.bin", dtype, max_degree=12, rows=(0, 700))
    assert e.get_graph()[:, 0].max() == 0
    e.vamana_build(12, 24, 1.2, seed=2)
    Go, _ = oracle.vamana_build(X[:700], 12, 24, 1.2, seed=2)
    np.testing.assert_array_equal(e.get_graph()[:, 0], Go[:, 0])
    e.close()
    with pytest.raises(ValueError):
        DeviceIndex.from_files(tmp_path / "base.bin", dtype, graph_path=tmp_path / "graph_s")      # 2500-vertex graph, 5000 rows

