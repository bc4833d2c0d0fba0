"""The 12-bit filter-code table (csrc/filter_codes.hip, round 3): beam-91..128 searches keep the reference's lossy filter
(beamSearch.h:52-59) as class codes in LDS, the codes of the neighbours travel with the graph rows and are maintained by the
builder's row writers.  Everything is compared with the oracle, which runs the plain id table: graphs, counters (dist_cmps is
what a wrong filter decision changes first), visited counts, results."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, datasets

pytestmark = pytest.mark.gpu


def _norm(G):
    G = G.copy()
    cols = np.arange(G.shape[1] - 1)[None, :]
    G[:, 1:][cols >= G[:, :1]] = 0
    return G


@pytest.mark.parametrize("dtype,metric,d,n,R,L,passes", [
    (np.uint8, "l2", 128, 30000, 64, 128, 2),        # R = 64: heavy re-prunes and appended reverse edges carry codes too
    (np.float16, "l2", 128, 20000, 32, 100, 1),
    (np.float32, "l2", 96, 12000, 48, 128, 2),       # 384-byte rows: the generic (query in LDS) layout
    (np.int8, "mips", 200, 9000, 40, 91, 1),         # the smallest beam with a 4 096-slot table
])
def test_build_through_the_code_table_equals_oracle(oracle, dtype, metric, d, n, R, L, passes):
    X = datasets.sift_like(n, d, seed=1234, dtype=np.float32)
    X = (X - 128).clip(-127, 127).astype(np.int8) if dtype == np.int8 else X.astype(dtype)
    alpha = 1.2 if metric == "l2" else 1.0
    Go, so = oracle.vamana_build(X, R, L, alpha, num_passes=passes, seed=11, metric=metric, sort_neighbors=False)
    ix = DeviceIndex(X, max_degree=R, metric=metric)
    st = ix.vamana_build(R, L, alpha, num_passes=passes, seed=11, sort_neighbors=False)
    np.testing.assert_array_equal(_norm(Go), _norm(ix.get_graph()))
    assert int(so[0]) == st.search_dist_cmps and int(so[1]) == st.prune_dist_cmps and int(so[2]) == st.visited_total
    # the codes are still in step with the (unsorted) graph: beam-128 queries and base-point queries go through them
    Q = datasets.sift_like(300, d, seed=4321, dtype=np.float32)
    Q = (Q - 128).clip(-127, 127).astype(np.int8) if dtype == np.int8 else Q.astype(dtype)
    G = ix.get_graph()
    for kw in (dict(queries=Q, k=10, beam=128), dict(queries=Q, k=10, beam=100, cut=1.1), dict(query_ids=np.arange(0, n, 37, dtype=np.uint32), k=0, beam=128, visited_cap=600)):
        g = ix.batch_search(**kw)
        o = oracle.batch_search(X, G, metric=metric, **kw)
        for f in ("ids", "dists", "visited_count", "dist_cmps", "frontier_size"):
            np.testing.assert_array_equal(o[f], g[f], err_msg=f"{f} {list(kw)}")
    # a row upload makes the codes stale: the same searches fall back to the id table and still agree
    ix.update_rows(np.array([5], np.uint32), G[5:6])
    g = ix.batch_search(queries=Q, k=10, beam=128)
    o = oracle.batch_search(X, G, queries=Q, k=10, beam=128, metric=metric)
    for f in ("ids", "dists", "visited_count", "dist_cmps"):
        np.testing.assert_array_equal(o[f], g[f], err_msg=f)
    # the builder rebuilds them from the graph it finds (second pass of a resumed build)
    batch = oracle.permutation(n, 5)[:2000]
    Go2 = G.copy()
    so2 = oracle.vamana_insert_batch(X, Go2, batch, R, L, alpha, metric=metric)
    sg2 = ix.vamana_insert_batch(batch, R, L, alpha)
    np.testing.assert_array_equal(_norm(Go2), _norm(ix.get_graph()))
    assert int(so2[0]) == sg2.search_dist_cmps and int(so2[1]) == sg2.prune_dist_cmps
    ix.close()


def test_sorted_build_and_clear_graph_keep_working(oracle):
    """the final neighbour sort permutes rows (codes stale -> id table); clear_graph empties rows AND codes (still in step)"""
    n, R, L = 15000, 32, 128
    X = datasets.sift_like(n, 64, seed=7, dtype=np.uint8)
    Go, so = oracle.vamana_build(X, R, L, 1.2, num_passes=1, seed=3)
    ix = DeviceIndex(X, max_degree=R)
    for _ in range(2):
        st = ix.vamana_build(R, L, 1.2, num_passes=1, seed=3)
        np.testing.assert_array_equal(_norm(Go), _norm(ix.get_graph()))
        assert int(so[0]) == st.search_dist_cmps
        Q = datasets.sift_like(100, 64, seed=8, dtype=np.uint8)
        g = ix.batch_search(Q, k=5, beam=128); o = oracle.batch_search(X, Go, queries=Q, k=5, beam=128)
        np.testing.assert_array_equal(o["ids"], g["ids"]); np.testing.assert_array_equal(o["dist_cmps"], g["dist_cmps"])
        ix.clear_graph()
    ix.close()


def test_codes_can_be_switched_off_and_graphs_do_not_depend_on_them(oracle):
    n, R, L = 20000, 32, 128
    X = datasets.sift_like(n, 64, seed=21, dtype=np.uint8)
    a = DeviceIndex(X, max_degree=R); b = DeviceIndex(X, max_degree=R)
    b.set_option("filter_codes", 0)
    sa = a.vamana_build(R, L, 1.2, num_passes=2, seed=4, sort_neighbors=False)
    sb = b.vamana_build(R, L, 1.2, num_passes=2, seed=4, sort_neighbors=False)
    assert a.get_option("filter_codes") == 1 and b.get_option("filter_codes") == 0
    np.testing.assert_array_equal(a.get_graph(), b.get_graph())
    assert sa.search_dist_cmps == sb.search_dist_cmps and sa.visited_total == sb.visited_total
    a.vamana_sort_neighbors()
    assert a.get_option("filter_codes") == 0                      # the sort permuted the rows: codes stale until the next build call
    a.close(); b.close()


def test_tables_too_large_for_12_bit_codes_fall_back_to_ids():
    """17.2M points: 4 199 ids per slot class on average, more than a 12-bit code can number -- the builder must notice when it
    builds the codes and keep the id table (a short insert on the big table: no oracle at this size, the assertions are that the
    fallback engaged and that the batch was inserted)"""
    n = 17_200_000
    rng = np.random.default_rng(5)
    X = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    ix = DeviceIndex(X, max_degree=16)
    batch = rng.choice(n, 20000, replace=False).astype(np.uint32)
    st = ix.vamana_insert_batch(batch[:1], 16, 128, 1.2, start=int(batch[0]))
    st = ix.vamana_insert_batch(batch, 16, 128, 1.2, start=int(batch[0]))
    assert ix.get_option("filter_codes") == 0 and st.visited_total > 0
    r = ix.batch_search(queries=X[batch[:100]], k=5, beam=128, starts=(int(batch[0]),))
    assert np.isin(r["ids"][:, 0], batch).all()                  # what a search reaches are inserted points (a one-batch star graph)
    ix.close()


@pytest.mark.parametrize("dtype,d,R,L", [(np.uint8, 128, 32, 64), (np.float32, 96, 48, 128), (np.float16, 128, 32, 200)])
def test_launch_order_of_a_batch_does_not_change_the_graph(oracle, dtype, d, R, L):
    """round 3: on large tables the builder launches the searches of a batch in locality order (nearest of 256 pivots); forced
    here on a small table for the beam-64, beam-128 (code table) and generic kernels: the graph is the oracle's, as before"""
    n = 30000
    X = datasets.sift_like(n, d, seed=31, dtype=np.float32).astype(dtype)
    Go, so = oracle.vamana_build(X, R, L, 1.2, num_passes=2, seed=6)
    ix = DeviceIndex(X, max_degree=R)
    ix.set_option("locality_order", 2)
    st = ix.vamana_build(R, L, 1.2, num_passes=2, seed=6)
    assert ix.get_option("locality_order") == 1
    np.testing.assert_array_equal(_norm(Go), _norm(ix.get_graph()))
    assert int(so[0]) == st.search_dist_cmps and int(so[1]) == st.prune_dist_cmps and int(so[2]) == st.visited_total
    ix.close()
