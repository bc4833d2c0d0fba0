"""GPU parity of the BFS range search (C-ABI pann_range_search -> range_search.hip) against the oracle's
restatement of range_search (algorithms/utils/beamSearch.h:245-306): result lists in BFS order, counts and
distance-comparison counters must be identical."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, datasets

pytestmark = pytest.mark.gpu
PAD = 0xFFFFFFFF


def _check(o, g, cmps=True):
    np.testing.assert_array_equal(o["counts"], g["counts"])
    np.testing.assert_array_equal(o["truncated"], g["truncated"])
    np.testing.assert_array_equal(o["ids"], g["ids"])
    if cmps:
        ok = o["truncated"] == 0            # a truncated query stops early; where exactly is not part of the contract
        np.testing.assert_array_equal(o["dist_cmps"][ok], g["dist_cmps"][ok])


def _setup(oracle, n, d, dtype, metric="l2", R=32):
    X = datasets.sift_like(n, d, seed=1234, dtype=np.float32)
    Q = datasets.sift_like(200, d, seed=4321, dtype=np.float32)
    if dtype == np.int8:
        X, Q = (X - 128).clip(-127, 127), (Q - 128).clip(-127, 127)
    X, Q = X.astype(dtype), Q.astype(dtype)
    G, _ = oracle.vamana_build(X, R, 2 * R, 1.2 if metric == "l2" else 1.0, seed=5, metric=metric)
    return X, Q, G


@pytest.mark.parametrize("dtype,metric,d", [(np.uint8, "l2", 128), (np.float16, "l2", 128), (np.float32, "l2", 200),
                                            (np.int8, "mips", 100)])
def test_range_search_from_beam_search_results(oracle, dtype, metric, d):
    n = 6000
    X, Q, G = _setup(oracle, n, d, dtype, metric)
    ix = DeviceIndex(X, G, metric=metric)
    s = ix.batch_search(Q, k=10, beam=32)
    starts = s["ids"]                                           # nq x 10: the beam-search answer seeds the BFS
    for rank in (10, 40):                                       # radius = typical distance of the rank-th neighbour
        gt_i, gt_d = oracle.bruteforce_knn(X, Q, rank, metric)
        r2 = float(np.median(gt_d[:, -1]))
        o = oracle.range_search(X, G, starts, r2, 2048, queries=Q, metric=metric)
        g = ix.range_search(starts, r2, 2048, queries=Q)
        assert o["counts"].max() > 5
        _check(o, g)
    ix.close()


def test_range_search_base_point_queries_shared_and_padded_starts(oracle):
    n = 5000
    X, Q, G = _setup(oracle, n, 128, np.uint8)
    ix = DeviceIndex(X, G)
    qid = np.arange(0, 400, dtype=np.uint32)
    r2 = float(np.median(oracle.bruteforce_knn(X, X[:100], 20)[1][:, -1]))
    # the reference's own call site (vamana/neighbors.h:96-99) starts at the query's own vertex, which
    # same_as() skips: empty result, zero comparisons
    own = qid[:, None].copy()
    o = oracle.range_search(X, G, own, r2, 64, query_ids=qid)
    g = ix.range_search(own, r2, 64, query_ids=qid)
    assert o["counts"].sum() == 0 and o["dist_cmps"].sum() == 0
    _check(o, g)
    # starts = the out-neighbours of the query's vertex (the commented `use_existing` branch, :260-262),
    # PAD-filled rows, duplicated entries, the query's own id among them
    st = np.full((len(qid), G.shape[1] + 3), PAD, np.uint32)
    for i, q in enumerate(qid):
        deg = G[q, 0]
        st[i, :deg] = G[q, 1:1 + deg]
        st[i, deg] = q
        st[i, deg + 1] = G[q, 1]
        st[i, deg + 2] = G[q, 1]
    o = oracle.range_search(X, G, st, r2, 256, query_ids=qid)
    g = ix.range_search(st, r2, 256, query_ids=qid)
    assert o["counts"].mean() > 3
    _check(o, g)
    # one shared start list for all queries, more than one wavefront wide, with repeats
    shared = np.concatenate([np.arange(0, 150, dtype=np.uint32), np.arange(100, 130, dtype=np.uint32)])
    o = oracle.range_search(X, G, shared, r2 * 2, 300, queries=Q)
    g = ix.range_search(shared, r2 * 2, 300, queries=Q)
    _check(o, g)
    ix.close()


def test_range_search_truncation_and_large_radius(oracle):
    n = 3000
    X, Q, G = _setup(oracle, n, 128, np.uint8, R=80)            # rows wider than one wavefront
    ix = DeviceIndex(X, G)
    starts = ix.batch_search(Q, k=5, beam=16)["ids"]
    big = 1e12                                                   # everything is in range: BFS walks the component
    o = oracle.range_search(X, G, starts, big, n, queries=Q[:50])
    g = ix.range_search(starts[:50], big, n, queries=Q[:50])
    assert o["counts"].min() > n // 2
    _check(o, g)
    o = oracle.range_search(X, G, starts, big, 100, queries=Q)
    g = ix.range_search(starts, big, 100, queries=Q)
    assert o["truncated"].all() and (o["counts"] == 100).all()
    _check(o, g, cmps=False)
    ix.close()


def test_range_search_exact_float_real_valued(oracle):
    n = 5000
    X = datasets.deep_like(n, 96, seed=1234); Q = datasets.deep_like(100, 96, seed=4321)
    G, _ = oracle.vamana_build(X, 32, 64, 1.2, seed=5)
    ix = DeviceIndex(X, G, exact_float_order=True)
    starts = ix.batch_search(Q, k=10, beam=32)["ids"]
    r2 = float(np.median(oracle.bruteforce_knn(X, Q, 30)[1][:, -1]))
    _check(oracle.range_search(X, G, starts, r2, 400, queries=Q), ix.range_search(starts, r2, 400, queries=Q))
    ix.close()


def test_range_search_errors(oracle):
    X, Q, G = _setup(oracle, 500, 32, np.uint8, R=8)
    ix = DeviceIndex(X, G)
    with pytest.raises(RuntimeError):
        ix.range_search(np.array([600], np.uint32), 10.0, 8, queries=Q)          # start out of range
    with pytest.raises(RuntimeError):
        ix.range_search(np.array([0], np.uint32), 10.0, 8, query_ids=np.array([500], np.uint32))
    with pytest.raises(ValueError):
        ix.range_search(np.array([0], np.uint32), 10.0, 8)
    ix.close()


def test_range_search_large_balls_take_the_worst_case_table_pass(oracle):
    """a radius that covers much of the data set: the seen sets outgrow the small per-wave tables of the first pass (more than
    8192 ids), those queries are redone with worst-case tables; small-ball queries of the same call stay in the first pass"""
    n = 12000
    X, Q, G = _setup(oracle, n, 64, np.uint8, R=24)
    ix = DeviceIndex(X, G)
    s = ix.batch_search(Q, k=10, beam=32)
    gt_i, gt_d = oracle.bruteforce_knn(X, Q, 2000)
    r_big = float(np.median(gt_d[:, -1]))              # about 2000 points in the ball: ~2000 x 24 neighbours > 8192 seen ids
    r_small = float(np.median(gt_d[:, 9]))
    for r2, cap in ((r_big, n), (r_small, n), (r_big, 500)):
        o = oracle.range_search(X, G, s["ids"], r2, cap, queries=Q)
        g = ix.range_search(s["ids"], r2, cap, queries=Q)
        _check(o, g)
    assert oracle.range_search(X, G, s["ids"], r_big, n, queries=Q)["dist_cmps"].max() > 8192
    ix.close()


def test_range_search_random_cases(oracle):
    """seeded sweep: element type, metric, degree (rows longer than one wave), radius from tiny to most of the data set,
    capacity from 1 up, shared / per-query / padded / repeated starts, external and base-point queries"""
    rng = np.random.default_rng(99)
    for it in range(24):
        dtype = [np.uint8, np.int8, np.float16, np.float32][it % 4]
        metric = "mips" if dtype == np.int8 and it % 8 == 1 else "l2"
        d = int(rng.choice([16, 48, 100, 128])); n = int(rng.integers(500, 4000)); R = int(rng.choice([8, 24, 64, 80]))
        X, Q, G = _setup(oracle, n, d, dtype, metric, R=R)
        Q = Q[: int(rng.integers(1, 60))]
        ix = DeviceIndex(X, G, metric=metric)
        rank = int(rng.choice([1, 5, 30, 200, min(n - 1, 1500)]))
        gd = oracle.bruteforce_knn(X, Q, rank, metric)[1]
        r2 = float(np.median(gd[:, -1]))
        cap = int(rng.choice([1, 7, 64, 700, n]))
        ns = int(rng.choice([1, 3, 10, 70]))
        base = it % 3 == 0                                     # base-point queries skip their own vertex
        nq = len(Q)
        qid = rng.choice(n, nq, replace=False).astype(np.uint32)
        if it % 2:
            starts = rng.integers(0, n, (nq, ns)).astype(np.uint32)          # per query, repeats likely
            starts[rng.random((nq, ns)) < 0.2] = PAD
        else:
            starts = rng.integers(0, n, ns).astype(np.uint32)
        kw = dict(query_ids=qid) if base else dict(queries=Q)
        o = oracle.range_search(X, G, starts, r2, cap, metric=metric, **kw)
        g = ix.range_search(starts, r2, cap, **kw)
        np.testing.assert_array_equal(o["counts"], g["counts"], err_msg=f"case {it}")
        np.testing.assert_array_equal(o["truncated"], g["truncated"], err_msg=f"case {it}")
        np.testing.assert_array_equal(o["ids"], g["ids"], err_msg=f"case {it}")
        ok = o["truncated"] == 0
        np.testing.assert_array_equal(o["dist_cmps"][ok], g["dist_cmps"][ok], err_msg=f"case {it}")
        ix.close()

