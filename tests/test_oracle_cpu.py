"""CPU-only checks of the oracle itself (no reference golden vectors exist for this path --
"parity unpinned", see oracle/pann_oracle.cpp header -- so these pin the restatement against
independent numpy statements of the same reference lines and against algebraic properties)."""
import numpy as np
import pytest

from parlayann_amd import datasets


def _hash64_2_py(x):
    M = (1 << 64) - 1
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & M
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & M
    return x ^ (x >> 31)


def test_hash64_2_matches_published_splitmix_finaliser(oracle):
    # parlaylib utilities.h hash64_2 (call site beamSearch.h:55); known splitmix64 finaliser outputs
    assert oracle.hash64_2(0) == 0
    for x in (1, 2, 12345, 0xFFFFFFFF, 999999937):
        assert oracle.hash64_2(x) == _hash64_2_py(x)
    # splitmix64's first output for seed 0 is finalise(0x9e3779b97f4a7c15) = 0xe220a8397b1dcdaf
    assert oracle.hash64_2(0x9e3779b97f4a7c15) == 0xe220a8397b1dcdaf


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.float32, np.float16])
def test_distance_matches_numpy(oracle, dtype):
    rng = np.random.default_rng(0)
    for d in (1, 7, 96, 128, 200):
        if dtype == np.int8:
            a = rng.integers(-128, 128, d).astype(np.int8); b = rng.integers(-128, 128, d).astype(np.int8)
        else:
            a = rng.integers(0, 256, d).astype(dtype); b = rng.integers(0, 256, d).astype(dtype)
        l2 = np.sum((a.astype(np.int64) - b.astype(np.int64)) ** 2)
        ip = np.sum(a.astype(np.int64) * b.astype(np.int64))
        # euclidian_point.h:54-62,74-81: int32 accumulate then one float cast; exact below 2**24
        assert oracle.distance(a, b, "l2") == np.float32(l2)
        assert oracle.distance(a, b, "mips") == -np.float32(ip)      # mips_point.h:43-57


def test_float_distance_is_sequential_unfused(oracle):
    rng = np.random.default_rng(1)
    a = rng.normal(size=96).astype(np.float32); b = rng.normal(size=96).astype(np.float32)
    acc = np.float32(0)
    for i in range(96):   # euclidian_point.h:83-90 evaluated left to right, one rounding per op
        t = np.float32(b[i] - a[i]); acc = np.float32(acc + np.float32(t * t))
    assert oracle.distance(a, b, "l2") == acc


def test_search_reaches_brute_force_on_easy_data(oracle):
    X = datasets.sift_like(5000, 64, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(100, 64, seed=2, dtype=np.uint8)
    G, stats = oracle.vamana_build(X, R=32, L=64, alpha=1.2, seed=3)
    assert G[:, 0].max() <= 32 and G[:, 0].min() >= 1
    gt, gd = oracle.bruteforce_knn(X, Q, 20)
    r = oracle.batch_search(X, G, queries=Q, k=10, beam=64, out_k=64)
    assert oracle.recall(r["ids"], gt, gd, 10) > 0.93
    # frontier rows are sorted by (dist,id) and hold true distances (beamSearch.h:46-48,211)
    for i in range(len(Q)):
        f = r["frontier_size"][i]
        ids, ds = r["ids"][i, :f], r["dists"][i, :f]
        assert np.all((ds[1:] > ds[:-1]) | ((ds[1:] == ds[:-1]) & (ids[1:] > ids[:-1])))
        assert ds[0] == oracle.distance(X[ids[0]], Q[i])
    # threads do not change results
    r1 = oracle.batch_search(X, G, queries=Q, k=10, beam=64, out_k=64, threads=1)
    np.testing.assert_array_equal(r["ids"], r1["ids"])
    np.testing.assert_array_equal(r["dist_cmps"], r1["dist_cmps"])


def test_lossy_filter_is_observable(oracle):
    """the direct-mapped filter re-evaluates evicted ids, so dist_cmps
    exceeds the number of distinct ids touched for at least some queries."""
    X = datasets.sift_like(6000, 32, seed=1, dtype=np.uint8)
    Q = datasets.sift_like(50, 32, seed=2, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, R=32, L=64, alpha=1.2, seed=3)
    r = oracle.batch_search(X, G, queries=Q, k=10, beam=64, visited_cap=1024)
    over = 0
    for i in range(len(Q)):
        nv = r["visited_count"][i]
        touched = {0}
        for v in r["visited_ids"][i, :nv]:
            touched.update(G[v, 1:1 + G[v, 0]].tolist())
        over += int(r["dist_cmps"][i] > len(touched))
        assert r["dist_cmps"][i] >= 1
    assert over > 0


def test_robust_prune_properties(oracle):
    X = datasets.sift_like(3000, 32, seed=1, dtype=np.uint8)
    G, _ = oracle.vamana_build(X, R=16, L=32, alpha=1.2, seed=3)
    rng = np.random.default_rng(0)
    owners = rng.integers(0, len(X), 50).astype(np.uint32)
    cands = [rng.choice(len(X), 40, replace=False).astype(np.uint32) for _ in owners]
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.uint64)
    rows, dc = oracle.robust_prune_batch(X, G, owners, np.concatenate(cands), None, off, 1.2, 16)
    for i, p in enumerate(owners):
        out = rows[i, 1:1 + rows[i, 0]]
        assert len(out) <= 16 and p not in out and len(set(out.tolist())) == len(out)
        pool = set(cands[i].tolist()) | set(G[p, 1:1 + G[p, 0]].tolist())
        assert set(out.tolist()) <= pool
        # the nearest candidate other than p always survives (vamana/index.h:95-103)
        best = min((oracle.distance(X[c], X[p]), c) for c in pool if c != p)
        assert out[0] == best[1]
    # alpha -> infinity prunes everything after the first pick unless dist(p,p') is infinite
    rows_inf, _ = oracle.robust_prune_batch(X, G, owners, np.concatenate(cands), None, off, 1e30, 16)
    assert np.all(rows_inf[:, 0] >= 1)


def test_build_schedule_and_recall_file_formats(oracle):
    perm = oracle.permutation(1000, 42)
    assert sorted(perm.tolist()) == list(range(1000))
    assert not np.array_equal(perm, np.arange(1000))
    np.testing.assert_array_equal(perm, oracle.permutation(1000, 42))
    # tie-aware recall (check_nn_recall.h:83-109)
    gt = np.array([[1, 2, 3, 4]], np.uint32); gd = np.array([[1.0, 2.0, 2.0, 3.0]], np.float32)
    assert oracle.recall(np.array([[1, 3]], np.uint32), gt, gd, 2) == 1.0   # 3 ties with the 2nd
    assert oracle.recall(np.array([[1, 4]], np.uint32), gt, gd, 2) == 0.5


def test_bf16_conversion_and_oracle_distance(oracle):
    """parlayann_amd.bfloat16: round-to-nearest-even from f32, exact widening back; the oracle's bf16 distance is the f32 rule
    on the widened values"""
    from parlayann_amd import bfloat16, from_bf16, to_bf16
    x = np.array([0.0, 1.0, -2.5, 255.0, 256.0, 257.0, 1e-3, 3.1415927, -65504.0], np.float32)
    b = to_bf16(x)
    assert b.dtype == bfloat16 and b.itemsize == 2
    back = from_bf16(b)
    assert back[3] == 255.0 and back[4] == 256.0 and back[5] in (256.0, 258.0)          # 8 significant bits
    assert np.all(np.abs(back - x) <= np.abs(x) * 2.0 ** -8)
    np.testing.assert_array_equal(to_bf16(back).view(np.uint16), b.view(np.uint16))     # idempotent
    rng = np.random.default_rng(0)
    A = to_bf16(rng.integers(0, 256, (4, 64)).astype(np.float32)); B = to_bf16(rng.integers(0, 256, (4, 64)).astype(np.float32))
    for i in range(4):
        a, c = from_bf16(A[i]).astype(np.float64), from_bf16(B[i]).astype(np.float64)
        assert oracle.distance(A[i], B[i], "l2") == np.float32(((a - c) ** 2).sum())
        assert oracle.distance(A[i], B[i], "mips") == np.float32(-(a * c).sum())
