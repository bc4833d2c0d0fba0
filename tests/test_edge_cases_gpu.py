"""Edge cases of the hot path on the device vs. the oracle: empty and degenerate inputs, ties,
duplicate adjacency entries (the reference sorts+uniques candidates "to be robust for neighbor lists
with duplicates", beamSearch.h:171-175), tiny indices, extreme parameters."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, datasets

pytestmark = pytest.mark.gpu


def _cmp(o, g):
    for f in ("ids", "dists", "frontier_size", "visited_count", "dist_cmps"):
        np.testing.assert_array_equal(o[f], g[f], err_msg=f)


def test_empty_batches_are_noops():
    X = datasets.sift_like(100, 16, seed=1, dtype=np.uint8)
    ix = DeviceIndex(X, max_degree=4)
    r = ix.batch_search(np.zeros((0, 16), np.uint8), k=1, beam=4)
    assert r["ids"].shape == (0, 1)
    rows, dc = ix.robust_prune_batch(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint64), 1.2, 4)
    assert rows.shape == (0, 5)
    ix.vamana_insert_batch(np.zeros(0, np.uint32), 4, 8, 1.2)
    assert ix.pair_distances(np.zeros(0, np.uint32), np.zeros(0, np.uint32)).shape == (0,)
    ix.close()


def test_empty_graph_and_isolated_start(oracle):
    X = datasets.sift_like(50, 16, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(5, 16, seed=2, dtype=np.uint8)
    G = np.zeros((50, 9), np.uint32)
    ix = DeviceIndex(X, G)
    o = oracle.batch_search(X, G, queries=Q, k=1, beam=8, out_k=8)
    g = ix.batch_search(Q, k=1, beam=8, out_k=8)
    _cmp(o, g)
    assert np.all(g["frontier_size"] == 1) and np.all(g["visited_count"] == 1)
    ix.close()


def test_single_point_and_tiny_indices(oracle):
    for n in (1, 2, 3, 9):
        X = datasets.sift_like(n, 8, seed=n, dtype=np.uint8)
        G = np.zeros((n, 5), np.uint32)
        for i in range(n):                      # ring
            nb = [(i + 1) % n, (i + n - 1) % n] if n > 1 else []
            nb = sorted(set(nb) - {i})
            G[i, 0] = len(nb); G[i, 1:1 + len(nb)] = nb
        ix = DeviceIndex(X, G)
        Q = datasets.sift_like(4, 8, seed=77, dtype=np.uint8)
        o = oracle.batch_search(X, G, queries=Q, k=1, beam=16, out_k=16)
        g = ix.batch_search(Q, k=1, beam=16, out_k=16)
        _cmp(o, g)
        ix.close()


def test_all_points_identical_ties_broken_by_id(oracle):
    X = np.full((300, 32), 7, np.uint8)
    G, _ = oracle.vamana_build(X, 8, 16, 1.2, seed=2)
    ix = DeviceIndex(X, max_degree=8)
    ix.vamana_build(8, 16, 1.2, seed=2)
    np.testing.assert_array_equal(ix.get_graph()[:, 0], G[:, 0])
    Q = np.full((3, 32), 9, np.uint8)
    o = oracle.batch_search(X, G, queries=Q, k=5, beam=16, out_k=16)
    ix2 = DeviceIndex(X, G)
    g = ix2.batch_search(Q, k=5, beam=16, out_k=16)
    _cmp(o, g)
    f = g["frontier_size"][0]
    assert np.all(np.diff(g["ids"][0, :f].astype(np.int64)) > 0)      # equal distances: ascending ids
    li, ld = ix2.leaf_knn(np.arange(0, 300, 3, dtype=np.uint32), 10)
    oi, od = oracle.leaf_knn(X, np.arange(0, 300, 3, dtype=np.uint32), 10)
    np.testing.assert_array_equal(li, oi); np.testing.assert_array_equal(ld, od)
    ix.close(); ix2.close()


def test_duplicate_and_self_entries_in_adjacency_rows(oracle):
    X = datasets.sift_like(2000, 32, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(60, 32, seed=2, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, 16, 32, 1.2, seed=3, max_degree=24)
    rng = np.random.default_rng(0)
    for v in rng.choice(2000, 400, replace=False):        # append duplicates of existing neighbours and self loops
        d = int(G[v, 0])
        extra = [G[v, 1 + rng.integers(0, d)], v, G[v, 1]][: 24 - d]
        G[v, 1 + d:1 + d + len(extra)] = extra
        G[v, 0] = d + len(extra)
    ix = DeviceIndex(X, G)
    for beam, k in ((32, 10), (64, 10), (100, 10)):
        o = oracle.batch_search(X, G, queries=Q, k=k, beam=beam, out_k=beam, visited_cap=512)
        g = ix.batch_search(Q, k=k, beam=beam, out_k=beam, visited_cap=512)
        _cmp(o, g)
    qids = rng.integers(0, 2000, 100).astype(np.uint32)    # build-mode search: self is skipped even when listed
    o = oracle.batch_search(X, G, query_ids=qids, k=0, beam=48, cut=0.0, out_k=48)
    g = ix.batch_search(query_ids=qids, k=0, beam=48, cut=0.0, out_k=48)
    _cmp(o, g)
    ix.close()


def test_queries_that_are_base_points_and_extreme_params(oracle):
    X = datasets.sift_like(3000, 64, seed=1, dtype=np.float16)
    G, _ = oracle.vamana_build(X, 16, 32, 1.2, seed=3)
    Q = X[::50].copy()
    ix = DeviceIndex(X, G)
    for kw in (dict(k=1, beam=1), dict(k=10, beam=10), dict(k=64, beam=64), dict(k=10, beam=64, limit=1),
               dict(k=10, beam=64, cut=1.0), dict(k=10, beam=64, cut=1e9), dict(k=10, beam=64, degree_limit=0),
               dict(k=10, beam=64, degree_limit=1), dict(k=3, beam=5, limit=3, degree_limit=2)):
        kw.setdefault("cut", 1.35)
        o = oracle.batch_search(X, G, queries=Q, out_k=kw["beam"], **kw)
        g = ix.batch_search(Q, out_k=kw["beam"], **kw)
        _cmp(o, g)
    g = ix.batch_search(Q, k=1, beam=32)
    assert np.mean(g["dists"][:, 0] == 0) > 0.7              # most queries find themselves at distance 0
    ix.close()


def test_robust_prune_degenerate_candidate_lists(oracle):
    X = datasets.sift_like(500, 16, seed=1, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, 8, 16, 1.2, seed=3)
    owners = np.array([0, 1, 2, 3, 4], np.uint32)
    cands = [np.zeros(0, np.uint32), np.array([1, 1, 1], np.uint32), np.array([2], np.uint32),
             np.arange(100, 140, dtype=np.uint32), np.array([4, 7, 7, 4, 9], np.uint32)]
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.uint64)
    ix = DeviceIndex(X, G)
    for add in (True, False):
        for alpha in (1.0, 1.2, 100.0):
            ro, dco = oracle.robust_prune_batch(X, G, owners, np.concatenate(cands), None, off, alpha, 8, add=add)
            rg, dcg = ix.robust_prune_batch(owners, np.concatenate(cands), off, alpha, 8, add_out_nbrs=add)
            np.testing.assert_array_equal(ro, rg); np.testing.assert_array_equal(dco, dcg)
    ix.close()


def test_zero_dimension_padding_and_odd_dims(oracle):
    for d, dt in ((1, np.uint8), (3, np.float32), (5, np.float16), (63, np.int8), (65, np.uint8), (129, np.float16)):
        X = datasets.sift_like(800, d, seed=d, dtype=np.float32)
        X = (X - 128).clip(-127, 127).astype(np.int8) if dt == np.int8 else X.astype(dt)
        G, _ = oracle.vamana_build(X, 8, 16, 1.2, seed=3)
        ix = DeviceIndex(X, G)
        Q = X[:20] if d < 3 else np.ascontiguousarray(X[5:45])
        o = oracle.batch_search(X, G, queries=Q, k=5, beam=16, out_k=16)
        g = ix.batch_search(Q, k=5, beam=16, out_k=16)
        _cmp(o, g)
        ix.close()


_REPLAY_IDS = [13684, 70653, 68340, 65698, 24635, 61076, 64589, 14896, 6023, 7202, 63156, 4421, 101363, 91014, 104600, 58243, 94466, 14090, 45263, 106329, 10443, 81886, 54875, 99683, 60743, 31593, 26379, 85117, 15397, 53897, 30931, 43329, 45736, 93829, 79979, 104814, 98773, 70120, 84091, 99167, 65109, 61660, 8861, 58549, 3391, 20439, 81582, 41645, 6255, 71749, 96325, 89412, 47683, 77135, 85256, 49658, 26603, 90426]


def test_rows_of_64_bytes_use_the_lds_query_variants(oracle):
    """d*esize <= 64 -> layout lpc 4 / nch 1, which runs the GENERIC kernel variants (query in LDS) although the row
    is a single chunk per lane.  The host once sized the dynamic LDS for the register variants in this case
    (64 bytes short): found through one re-prune whose distance_comps differed (the list below, owner 100748
    of a 110K-point build).  Every kernel family is exercised on 64-byte rows here."""
    n = 110_000
    X = datasets.sift_like(n, 32, seed=1234, dtype=np.uint8)
    G0 = np.zeros((n, 17), np.uint32)
    ix = DeviceIndex(X, G0)
    ids = np.array(_REPLAY_IDS, np.uint32)
    owners = np.array([100748], np.uint32); off = np.array([0, len(ids)], np.uint64)
    for R in (16, 58):
        ro, dco = oracle.robust_prune_batch(X, G0, owners, ids, None, off, 1.2, R, add=False)
        rg, dcg = ix.robust_prune_batch(owners, ids, off, 1.2, R, add_out_nbrs=False)
        np.testing.assert_array_equal(ro, rg); np.testing.assert_array_equal(dco, dcg)
    ix.close()
    for dtype, d in ((np.uint8, 48), (np.float32, 16), (np.float16, 32)):
        m = 5000
        Y = datasets.sift_like(m, d, seed=5, dtype=dtype); Q = datasets.sift_like(200, d, seed=6, dtype=dtype)
        G, _ = oracle.vamana_build(Y, 24, 48, 1.2, seed=3)
        iy = DeviceIndex(Y, G)
        for beam in (20, 64, 100, 200):
            o = oracle.batch_search(Y, G, queries=Q, k=10, beam=beam)
            g = iy.batch_search(Q, k=10, beam=beam)
            for f in ("ids", "visited_count", "dist_cmps"):
                np.testing.assert_array_equal(o[f], g[f], err_msg=f"{np.dtype(dtype).name} d={d} beam={beam} {f}")
        sel = np.arange(0, 700, dtype=np.uint32)
        oi, od = oracle.leaf_knn(Y, sel, 10)
        gi, gd = iy.leaf_knn(sel, 10)
        np.testing.assert_array_equal(oi, gi); np.testing.assert_array_equal(od, gd)
        st = iy.batch_search(Q, k=5, beam=16)["ids"]
        r2 = float(np.median(oracle.bruteforce_knn(Y, Q, 20)[1][:, -1]))
        o = oracle.range_search(Y, G, st, r2, 256, queries=Q); g = iy.range_search(st, r2, 256, queries=Q)
        np.testing.assert_array_equal(o["ids"], g["ids"]); np.testing.assert_array_equal(o["dist_cmps"], g["dist_cmps"])
        iy.close()
        iz = DeviceIndex(Y, max_degree=24)
        iz.vamana_build(24, 48, 1.2, seed=3)
        Gd = iz.get_graph()
        cols = np.arange(24)[None, :]
        np.testing.assert_array_equal(np.where(cols < G[:, :1], G[:, 1:], 0), np.where(cols < Gd[:, :1], Gd[:, 1:], 0))
        iz.close()


def _line_graph(n):
    """points on a line (x_i = i), vertex i linked to i+-1, i+-2: a greedy walk from 0 to a far query visits ~n/2
    vertices while cut = 1.0, k = 1 keeps the frontier at two or three entries, i.e. never full"""
    X = np.zeros((n, 8), np.float32)
    X[:, 0] = np.arange(n)
    G = np.zeros((n, 5), np.uint32)
    for i in range(n):
        nb = [j for j in (i - 2, i - 1, i + 1, i + 2) if 0 <= j < n]
        G[i, 0] = len(nb); G[i, 1:1 + len(nb)] = nb
    return X, G


@pytest.mark.parametrize("beam", [16, 100, 300])       # register-frontier (b64, b128) and LDS-frontier kernels
def test_dropped_list_grows_instead_of_failing(oracle, beam):
    """More than 256 visited vertices cut from a frontier that never fills (`ndrop` near `dcap` was
    untested and an overflow was a hard error where the reference simply succeeds)."""
    X, G = _line_graph(1500)
    Q = np.zeros((3, 8), np.float32); Q[:, 0] = [1499.0, 1400.5, 700.0]
    o = oracle.batch_search(X, G, queries=Q, k=1, beam=beam, cut=1.0, out_k=2)
    assert o["visited_count"].max() > 600
    ix = DeviceIndex(X, G)
    assert ix.dropped_capacity == 256
    g = ix.batch_search(Q, k=1, beam=beam, cut=1.0, out_k=2)
    _cmp(o, g)
    assert ix.dropped_capacity > 256 and g["status"][0] == 0
    ix.close()


def test_dropped_list_growth_is_bounded_by_running_the_batch_in_ranges(oracle):
    """growing the dropped list for a whole large batch could ask for tens of GB (nq x min(limit, n) x 8 bytes).
    Above a 1 GiB budget the batch runs in ranges of queries instead; the handle keeps a modest capacity afterwards.
    100 000 queries x 1536 entries x 8 B = 1.2 GB -> two ranges."""
    X, G = _line_graph(1500)
    rng = np.random.default_rng(3)
    Q = np.zeros((100_000, 8), np.float32); Q[:, 0] = rng.integers(0, 1500, len(Q)) + 0.25
    Q[7, 0] = 1499.0                                             # at least one query that walks the whole line
    o = oracle.batch_search(X, G, queries=Q, k=1, beam=16, cut=1.0, out_k=2)
    assert o["visited_count"].max() > 600
    ix = DeviceIndex(X, G)
    g = ix.batch_search(Q, k=1, beam=16, cut=1.0, out_k=2)
    _cmp(o, g)
    assert g["status"][0] == 0 and 256 < ix.dropped_capacity <= 2048
    g2 = ix.batch_search(Q[:1000], k=1, beam=16, cut=1.0, out_k=2)      # the next call starts from the kept capacity
    _cmp(oracle.batch_search(X, G, queries=Q[:1000], k=1, beam=16, cut=1.0, out_k=2), g2)
    ix.close()


def test_dev_entry_reports_dropped_overflow_in_status_word(oracle):
    """pann_batch_search_dev performs no synchronisation: the status word is copied to pann_search_out::status on the
    launch stream; bit 2 = results invalid, reserve a larger list and launch again"""
    import ctypes as C
    import torch
    from parlayann_amd import _capi
    from parlayann_amd._capi import QueryParams, SearchOut, check
    X, G = _line_graph(1500)
    Q = np.zeros((2, 8), np.float32); Q[:, 0] = [1499.0, 3.0]
    o = oracle.batch_search(X, G, queries=Q, k=1, beam=16, cut=1.0, out_k=2)
    ix = DeviceIndex(X, G)
    lib = _capi.load()
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(Q.view(np.uint8).reshape(2, -1)).to(dev)
    d_s = torch.zeros(1, dtype=torch.int32, device=dev)
    d_ids = torch.empty((2, 2), dtype=torch.int32, device=dev)
    d_dists = torch.empty((2, 2), dtype=torch.float32, device=dev)
    d_vis = torch.empty(2, dtype=torch.int32, device=dev)
    d_status = torch.full((1,), 77, dtype=torch.int32, device=dev)
    qp = QueryParams(k=1, beam=16, cut=1.0, limit=1500, degree_limit=4, rerank_factor=100, pad=1.0)
    out = SearchOut(ids=d_ids.data_ptr(), dists=d_dists.data_ptr(), out_k=2, visited_count=d_vis.data_ptr(),
                    status=d_status.data_ptr())
    st = torch.cuda.current_stream(dev)

    def launch():
        check(lib.pann_batch_search_dev(ix.handle, d_q.data_ptr(), None, 2, 32, d_s.data_ptr(), 1, C.byref(qp), C.byref(out),
                                        C.c_void_p(st.cuda_stream)))
        torch.cuda.synchronize(dev)
        return int(d_status.item())
    assert launch() & _capi.PANN_STATUS_DROPPED_OVERFLOW
    ix.reserve_dropped(1500)
    assert launch() == 0
    np.testing.assert_array_equal(d_ids.cpu().numpy().view(np.uint32), o["ids"])
    np.testing.assert_array_equal(d_dists.cpu().numpy(), o["dists"])
    np.testing.assert_array_equal(d_vis.cpu().numpy().view(np.uint32), o["visited_count"])
    ix.close()


def test_graph_upload_rejects_out_of_range_neighbours():
    """a graph file of another dataset must be PANN_ERR_BAD_ARG, not an out-of-bounds gather"""
    from parlayann_amd import PannError
    X = datasets.sift_like(200, 16, seed=1, dtype=np.uint8)
    G = np.zeros((200, 9), np.uint32)
    G[:, 0] = 2; G[:, 1] = (np.arange(200) + 1) % 200; G[:, 2] = (np.arange(200) + 7) % 200
    bad = G.copy(); bad[17, 2] = 200                       # == n: out of range
    with pytest.raises(PannError) as e:
        DeviceIndex(X, bad)
    assert e.value.code == 1 and "out of range" in str(e.value)
    ix = DeviceIndex(X, G)
    with pytest.raises(PannError):
        ix.set_graph(bad)
    with pytest.raises(PannError):
        ix.update_rows(np.array([5], np.uint32), np.array([[1, 4000000000, 0, 0, 0, 0, 0, 0, 0]], np.uint32))
    # the offending rows are left empty, the rest of the upload stands: searches stay in bounds
    g = ix.batch_search(X[:4], k=1, beam=8)
    assert g["ids"].max() < 200
    ix.set_graph(G)
    ix.close()


def test_query_dtype_and_width_are_checked_everywhere():
    X = datasets.sift_like(300, 16, seed=1, dtype=np.float16)
    ix = DeviceIndex(X, max_degree=8)
    ix.vamana_build(8, 16, 1.2)
    wrong = X[:4].astype(np.float32)
    for call in (lambda q: ix.batch_search(q, k=1, beam=8), lambda q: ix.bruteforce_knn(q, 3),
                 lambda q: ix.query_distances(q, [0, 1]), lambda q: ix.rerank(q, np.zeros((len(q), 4), np.uint32), None, 2),
                 lambda q: ix.range_search([0], 10.0, 8, queries=q)):
        with pytest.raises(ValueError):
            call(wrong)
        with pytest.raises(ValueError):
            call(X[:4, :8])
        call(X[:4])
    ix.close()
