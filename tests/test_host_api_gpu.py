"""tests/host_api_check.cpp calls the C++ host mirror with the REFERENCE's own argument lists -- beam_search(p, G, Points,
starting_points, QP), the 10-argument qsearchAll, knn_index<PR, QPR, indexType>::build_index(G, Points, QPoints, BuildStats,
sort), robustPrune(p, cand, G, Points, alpha, add), hcnng_index::build_index(...), checkRecall(G, Base, Query, ...) -- and
dumps what they return; here every array is compared with the oracle ("give the boundary the reference's real
signatures")."""
import os
import subprocess

import numpy as np
import pytest

from parlayann_amd import datasets, io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_checker():
    exe = os.path.join(ROOT, "tests", "host_api_check")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "parlayann_amd", "host"), "-s"])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-Wno-sign-compare", "-o", exe,
                           os.path.join(ROOT, "tests", "host_api_check.cpp"), "-L" + os.path.join(ROOT, "parlayann_amd", "lib"),
                           "-lpann", "-Wl,-rpath," + os.path.join(ROOT, "parlayann_amd", "lib")])
    return exe


@pytest.fixture(scope="module")
def run(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("hostapi")
    X = datasets.sift_like(6000, 64, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(120, 64, seed=4321, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, 32, 48, 1.2, num_passes=1, seed=5)
    gt, gd = oracle.bruteforce_knn(X, Q, 100)
    io.write_bin(d / "base.bin", X); io.write_bin(d / "query.bin", Q); io.write_graph(d / "g.graph", G); io.write_ibin(d / "gt.ibin", gt, gd)
    nn_i, nn_d = oracle.bruteforce_knn(X, Q[2:3], 40)
    r2 = float(nn_d[0, -1])                                        # radius: the query's 40 nearest base points
    rs = [int(nn_i[0, 3]), int(nn_i[0, 20])]                       # two starts inside it
    out = d / "out"; out.mkdir()
    p = subprocess.run([build_checker(), str(d / "base.bin"), str(d / "query.bin"), str(d / "g.graph"), str(d / "gt.ibin"), str(out), repr(r2),
                        str(rs[0]), str(rs[1])], capture_output=True, text=True)
    assert p.returncode == 0 and "host_api_check done" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]

    def load(name, dt):
        return np.fromfile(out / (name + ".bin"), dtype=dt)
    return dict(X=X, Q=Q, G=G, gt=gt, gd=gd, load=load, out=out, stdout=p.stdout, r2=r2, rs=rs)


def _sorted_visited(o, i, cap):
    c = int(o["visited_count"][i])
    return o["visited_ids"][i, :c], o["visited_dists"][i, :c]       # the oracle returns `visited` sorted by (dist,id), like the reference


def _check_beam(run, oracle, name, **kw):
    ld = run["load"]
    o = oracle.batch_search(run["X"], run["G"], k=10, beam=64, out_k=64, visited_cap=512, **kw)
    f = int(o["frontier_size"][0])
    np.testing.assert_array_equal(ld(name + "_frontier_ids", np.uint32), o["ids"][0, :f])
    np.testing.assert_array_equal(ld(name + "_frontier_dists", np.float32), o["dists"][0, :f])
    vi, vd = _sorted_visited(o, 0, 512)
    np.testing.assert_array_equal(ld(name + "_visited_ids", np.uint32), vi)
    np.testing.assert_array_equal(ld(name + "_visited_dists", np.float32), vd)
    assert int(ld(name + "_cmps", np.uint64)[0]) == int(o["dist_cmps"][0])


def test_single_query_beam_search_forms(run, oracle):
    Q = run["Q"]
    _check_beam(run, oracle, "bs_ext", queries=Q[3:4], starts=(0, 5, 9))
    _check_beam(run, oracle, "bs_base", query_ids=[77], starts=(0, 5, 9))            # same_as: own vertex skipped
    _check_beam(run, oracle, "bs_single", queries=Q[4:5])
    _check_beam(run, oracle, "bs_impl", queries=Q[4:5])
    _check_beam(run, oracle, "bs_filtered", queries=Q[5:6])


def test_build_search_and_robust_prune(run, oracle):
    ld, X, G = run["load"], run["X"], run["G"]
    o = oracle.batch_search(X, G, query_ids=[123], k=0, beam=48, cut=0.0, out_k=0, visited_cap=512)
    vi, vd = _sorted_visited(o, 0, 512)
    np.testing.assert_array_equal(ld("build_visited_ids", np.uint32), vi)
    np.testing.assert_array_equal(ld("build_visited_dists", np.float32), vd)
    assert int(ld("build_visited_cmps", np.uint32)[0]) == int(o["dist_cmps"][0])
    rows, dc = oracle.robust_prune_batch(X, G, [123], vi, vd, [0, len(vi)], 1.2, 32, add=True)
    np.testing.assert_array_equal(ld("prune_pairs_row", np.uint32), rows[0, 1:1 + rows[0, 0]])
    assert int(ld("prune_pairs_cmps", np.uint64)[0]) == int(dc[0])
    rows, dc = oracle.robust_prune_batch(X, G, [123], vi, None, [0, len(vi)], 1.0, 32, add=False)
    np.testing.assert_array_equal(ld("prune_ids_row", np.uint32), rows[0, 1:1 + rows[0, 0]])
    assert int(ld("prune_ids_cmps", np.uint64)[0]) == int(dc[0])


def test_batched_searches(run, oracle):
    ld, X, Q, G = run["load"], run["X"], run["Q"], run["G"]
    o = oracle.batch_search(X, G, queries=Q, k=10, beam=64)
    for name in ("searchAll_ids", "qsearchAll_ids"):
        np.testing.assert_array_equal(ld(name, np.uint32).reshape(len(Q), 10), o["ids"])
    np.testing.assert_array_equal(ld("searchAll_visited", np.uint32), o["visited_count"])
    np.testing.assert_array_equal(ld("searchAll_dists", np.uint32), o["dist_cmps"])
    np.testing.assert_array_equal(ld("qsearchAll_visited", np.uint32), o["visited_count"])
    o3 = oracle.batch_search(X, G, queries=Q, k=10, beam=64, starts=(0, 5, 9))
    np.testing.assert_array_equal(ld("searchAll3_ids", np.uint32).reshape(len(Q), 10), o3["ids"])
    st = ld("random_starts", np.uint32)
    assert st.max() < len(X) and len(np.unique(st)) > len(Q) // 2
    got = ld("random_ids", np.uint32).reshape(len(Q), 10)
    for i in range(len(Q)):                                           # the oracle takes one shared start set per call
        orr = oracle.batch_search(X, G, queries=Q[i:i + 1], k=10, beam=64, starts=st[i:i + 1])
        np.testing.assert_array_equal(got[i], orr["ids"][0])
    # beam_search_rerank with equal ranges: first k frontier ids with their exact distances (beamSearch.h:445-452)
    o6 = oracle.batch_search(X, G, queries=Q[6:7], k=10, beam=64)
    np.testing.assert_array_equal(ld("rerank_ids", np.uint32), o6["ids"][0])
    np.testing.assert_array_equal(ld("rerank_dists", np.float32), o6["dists"][0])


def test_range_search_and_check_recall(run, oracle):
    ld, X, Q, G = run["load"], run["X"], run["Q"], run["G"]
    o = oracle.range_search(X, G, run["rs"] + [0], run["r2"], 4096, queries=Q[2:3])
    c = int(o["counts"][0])
    assert c > 5
    np.testing.assert_array_equal(ld("range_ids", np.uint32), o["ids"][0, :c])
    assert int(ld("range_cmps", np.uint64)[0]) == int(o["dist_cmps"][0])
    ob = oracle.batch_search(X, G, queries=Q, k=10, beam=64)
    rec = ld("recall", np.float64)
    assert abs(rec[0] - oracle.recall(ob["ids"], run["gt"], run["gd"], 10)) < 1e-6
    assert rec[1] == int(ob["visited_count"].astype(np.uint64).sum() // len(Q)) and rec[2] == int(ob["dist_cmps"].astype(np.uint64).sum() // len(Q))
    # tail = the 99th-percentile element of the reference's statistics() (stats.h:84-92)
    assert rec[3] == np.sort(ob["visited_count"])[int(.99 * np.float32(len(Q)))]


def test_build_index_verbatim_signature_and_real_build_stats(run, oracle):
    ld, X, Q = run["load"], run["X"], run["Q"]
    n = len(X)
    pv, pd = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    Go, _ = oracle.vamana_build(X, 32, 48, 1.2, num_passes=2, seed=7, point_stats=(pv, pd))
    G = io.read_graph(run["out"] / "built.graph")
    cols = np.arange(32)[None, :]
    np.testing.assert_array_equal(G[:, 0], Go[:, 0])
    np.testing.assert_array_equal(np.where(cols < G[:, :1], G[:, 1:], 0), np.where(cols < Go[:, :1], Go[:, 1:], 0))
    # BuildStats: per inserted point, like the reference (the averages were smeared before)
    np.testing.assert_array_equal(ld("build_visited", np.uint32), pv)
    np.testing.assert_array_equal(ld("build_dists", np.uint32), pd)
    assert pv.min() > 0 and len(np.unique(pv)) > 20
    o = oracle.batch_search(X, Go, queries=Q, k=10, beam=64)
    np.testing.assert_array_equal(ld("built_search_ids", np.uint32).reshape(len(Q), 10), o["ids"])
    Gi = Go.copy(); Gi[0, 0] = 0
    oi = oracle.batch_search(X, Gi, queries=Q, k=10, beam=64)
    got = ld("isolated_start_ids", np.uint32).reshape(len(Q), 10)
    np.testing.assert_array_equal(got, oi["ids"])                     # vertex 0 without out-edges: only the start is found
    assert np.all(got[:, 0] == 0) and np.all(got[:, 1] == 0xFFFFFFFF)
    # two batch_insert passes (alpha 1.0 then 1.2) == build_index without the final neighbour sort
    Gu, _ = oracle.vamana_build(X, 32, 48, 1.2, num_passes=2, seed=7, sort_neighbors=False)
    Gb = io.read_graph(run["out"] / "batch_insert.graph")
    np.testing.assert_array_equal(Gb[:, 0], Gu[:, 0])
    np.testing.assert_array_equal(np.where(cols < Gb[:, :1], Gb[:, 1:], 0), np.where(cols < Gu[:, :1], Gu[:, 1:], 0))


def test_hcnng_build_index_verbatim_signature(run, oracle):
    Gh = io.read_graph(run["out"] / "hcnng.graph")
    Go = oracle.hcnng_build(run["X"], 8, 200, 3, seed=3)
    np.testing.assert_array_equal(Gh, Go)
    assert "mirrors alive" in run["stdout"]
