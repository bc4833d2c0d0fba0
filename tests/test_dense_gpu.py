"""GPU parity of the dense all-pairs kernels (HCNNG leaf kNN, brute-force ground truth, plain
distances) against the CPU oracle; integer-valued data, so results are bit-exact."""
import numpy as np
import pytest

from parlayann_amd import DeviceIndex, datasets

pytestmark = pytest.mark.gpu


def _mk(n, d, dtype, seed=1234):
    X = datasets.sift_like(n, d, seed=seed, dtype=np.float32)
    return (X - 128).clip(-127, 127).astype(np.int8) if dtype == np.int8 else X.astype(dtype)


@pytest.mark.parametrize("dtype,metric,d", [(np.uint8, "l2", 128), (np.float16, "l2", 128), (np.float32, "l2", 96),
                                            (np.int8, "mips", 200), (np.float32, "mips", 200), (np.uint8, "l2", 20),
                                            (np.float16, "l2", 96), (np.float16, "mips", 200), (np.float16, "l2", 40)])
def test_leaf_knn(oracle, dtype, metric, d):
    X = _mk(5000, d, dtype)
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    rng = np.random.default_rng(3)
    sizes = [1000, 300, 64, 65, 37, 2, 1, 11]
    leaves = [rng.choice(len(X), s, replace=False).astype(np.uint32) for s in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    gi, gd = ix.leaf_knn_batch(np.concatenate(leaves), off, 10)
    for li, ids in enumerate(leaves):
        oi, od = oracle.leaf_knn(X, ids, 10, metric=metric)
        np.testing.assert_array_equal(oi, gi[off[li]:off[li + 1]], err_msg=f"leaf {li} size {len(ids)}")
        np.testing.assert_array_equal(od.view(np.uint32), gd[off[li]:off[li + 1]].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.int8])
@pytest.mark.parametrize("metric", ["l2", "mips"])
@pytest.mark.parametrize("d", [33, 64, 65, 129, 144, 160, 192, 193, 208, 209, 256])
def test_leaf_knn_one_byte_every_row_stride(oracle, dtype, metric, d):
    """one-byte rows of 129..192 bytes have a 192-byte stride; the lane-owns-row kernel must not read a
    13th chunk there (it belongs to the next row; for the last row it lies past the slab).  n is a multiple of 64 and the
    leaves hold the table's last rows."""
    n = 2048
    X = _mk(n, d, dtype, seed=77 + d)
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    rng = np.random.default_rng(d)
    leaves = [np.arange(n - 300, n, dtype=np.uint32), rng.choice(n, 257, replace=False).astype(np.uint32),
              np.array([n - 1, 0, n - 2], dtype=np.uint32)]
    sizes = [len(l) for l in leaves]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    gi, gd = ix.leaf_knn_batch(np.concatenate(leaves), off, 10)
    for li, ids in enumerate(leaves):
        oi, od = oracle.leaf_knn(X, ids, 10, metric=metric)
        np.testing.assert_array_equal(oi, gi[off[li]:off[li + 1]], err_msg=f"leaf {li} size {len(ids)}")
        np.testing.assert_array_equal(od.view(np.uint32), gd[off[li]:off[li + 1]].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("dtype,metric,d,k", [(np.uint8, "l2", 128, 100), (np.float16, "l2", 128, 10),
                                              (np.int8, "mips", 200, 100), (np.float32, "l2", 96, 37),
                                              (np.float16, "mips", 128, 100), (np.float16, "l2", 200, 64)])
def test_bruteforce_knn(oracle, dtype, metric, d, k):
    X = _mk(20000, d, dtype)
    Q = _mk(130, d, dtype, seed=4321)
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    gi, gd = ix.bruteforce_knn(Q, k)
    oi, od = oracle.bruteforce_knn(X, Q, k, metric=metric)
    np.testing.assert_array_equal(oi, gi)
    np.testing.assert_array_equal(od, gd)
    ix.close()


@pytest.mark.parametrize("metric,d,k,n,nq,nsplit", [
    ("l2", 128, 100, 20000, 130, None), ("l2", 128, 17, 5000, 64, None), ("l2", 128, 64, 5000, 65, 3),
    ("l2", 128, 65, 5000, 1, 2), ("l2", 128, 128, 9000, 200, 1), ("mips", 128, 100, 9000, 33, 4),
    ("l2", 64, 100, 6000, 70, None), ("l2", 100, 50, 6000, 70, None), ("l2", 32, 100, 700, 20, 5),
    ("l2", 128, 100, 50, 10, None), ("l2", 128, 100, 64, 64, 1), ("mips", 96, 30, 129, 3, 2),
    ("l2", 128, 112, 5000, 65, 3), ("l2", 128, 113, 5000, 65, 2), ("mips", 128, 112, 3000, 17, None),      # 7 / 8 list registers per row
])
def test_bruteforce_register_list_kernel(oracle, metric, d, k, n, nq, nsplit):
    """two-byte floats, rows <= 256 bytes, k in 17..128: dense_gt_mfma_kernel (lists in registers, B double-buffered);
    partial tiles, fewer points than k, one query, every piece count"""
    X = _mk(n, d, np.float16)
    Q = _mk(nq, d, np.float16, seed=4321)
    X[n // 2] = X[n // 3]                      # equal distances: the id decides (check_nn_recall-style ties)
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    ix.set_option("gt_pieces", nsplit or 0)
    gi, gd = ix.bruteforce_knn(Q, k)
    oi, od = oracle.bruteforce_knn(X, Q, k, metric=metric)
    np.testing.assert_array_equal(oi, gi)
    np.testing.assert_array_equal(od, gd)
    ix.close()


@pytest.mark.parametrize("dtype,metric,d,k,n,nq,nsplit", [
    (np.uint8, "l2", 128, 100, 20000, 130, None), (np.uint8, "mips", 128, 100, 9000, 65, 2), (np.uint8, "l2", 300, 40, 5000, 33, None),
    (np.int8, "mips", 200, 100, 20000, 130, None), (np.int8, "l2", 100, 17, 6000, 64, 3), (np.int8, "l2", 256, 128, 3000, 10, 1),
    (np.float32, "l2", 96, 100, 9000, 70, None), (np.float32, "mips", 128, 37, 6000, 129, 2), (np.float32, "l2", 20, 100, 3000, 5, None),
    (np.uint8, "l2", 128, 100, 50, 3, None), (np.float32, "l2", 65, 30, 200, 64, 1),
])
def test_bruteforce_register_list_kernel_valu_types(oracle, dtype, metric, d, k, n, nq, nsplit):
    """one-byte types and f32, rows <= 512 bytes, k in 17..128: dense_gt_valu_kernel (v_dot4 / fma register tile in the MFMA's
    output layout, the same register lists); one and two 256-byte segments, rows that end inside a 16-byte chunk"""
    X = _mk(n, d, dtype)
    Q = _mk(nq, d, dtype, seed=4321)
    X[n // 2] = X[n // 3]
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    ix.set_option("gt_pieces", nsplit or 0)
    gi, gd = ix.bruteforce_knn(Q, k)
    oi, od = oracle.bruteforce_knn(X, Q, k, metric=metric)
    np.testing.assert_array_equal(oi, gi)
    np.testing.assert_array_equal(od, gd)
    ix.close()


@pytest.mark.parametrize("dtype,k,nsplit", [(np.float16, 10, 6), (np.float16, 100, 5), (np.uint8, 10, 4), (np.uint8, 100, None),
                                             (np.float16, 16, 3), (np.float16, 17, 3), (np.float32, 10, 4), (np.int8, 1, 5)])
def test_bruteforce_many_workgroups_per_cu(oracle, dtype, k, nsplit):
    """4 000 queries x 60 000 points: several hundred workgroups, i.e. several of them resident on every CU at once and every
    piece of a row racing to publish its bound -- the configuration the small cases above never reach"""
    X = _mk(60000, 64, dtype)
    Q = _mk(4000, 64, dtype, seed=4321)
    ix = DeviceIndex(X, max_degree=8)
    ix.set_option("gt_pieces", nsplit or 0)
    gi, gd = ix.bruteforce_knn(Q, k)
    oi, od = oracle.bruteforce_knn(X, Q, k)
    np.testing.assert_array_equal(oi, gi)
    np.testing.assert_array_equal(od, gd)
    ix.close()


def test_bruteforce_random_shapes(oracle):
    """seeded sweep over (type, metric, d, k, n, nq, pieces): every shape the ground-truth kernels dispatch on -- register
    lists (k > 16), lane lists (k <= 16), one / two segments, the LDS-list fallback for long rows"""
    rng = np.random.default_rng(20260)
    dts = [np.uint8, np.int8, np.float16, np.float32]
    for it in range(40):
        dtype = dts[it % 4]
        metric = "l2" if rng.random() < 0.6 else "mips"
        d = int(rng.choice([8, 20, 64, 96, 100, 128, 200, 256]))
        k = int(rng.choice([1, 10, 16, 17, 33, 64, 100, 128]))
        n = int(rng.integers(1, 3000)); nq = int(rng.integers(1, 150))
        pieces = int(rng.integers(1, 6))
        X = _mk(n, d, dtype, seed=int(rng.integers(1 << 30)))
        Q = _mk(nq, d, dtype, seed=int(rng.integers(1 << 30)))
        ix = DeviceIndex(X, max_degree=8, metric=metric)
        ix.set_option("gt_pieces", pieces)
        gi, gd = ix.bruteforce_knn(Q, k)
        oi, od = oracle.bruteforce_knn(X, Q, k, metric=metric)
        tag = f"case {it}: {np.dtype(dtype).name} {metric} d={d} k={k} n={n} nq={nq}"
        np.testing.assert_array_equal(oi, gi, err_msg=tag)
        np.testing.assert_array_equal(od, gd, err_msg=tag)
        ix.close()


@pytest.mark.parametrize("dtype,metric,d", [(np.uint8, "l2", 128), (np.int8, "l2", 100), (np.float32, "mips", 200),
                                            (np.float16, "mips", 128), (np.uint8, "mips", 32)])
def test_plain_distances(oracle, dtype, metric, d):
    X = _mk(3000, d, dtype)
    Q = _mk(20, d, dtype, seed=4321)
    ix = DeviceIndex(X, max_degree=8, metric=metric)
    rng = np.random.default_rng(0)
    a = rng.integers(0, len(X), 500).astype(np.uint32); b = rng.integers(0, len(X), 500).astype(np.uint32)
    got = ix.pair_distances(a, b)
    want = np.array([oracle.distance(X[i], X[j], metric) for i, j in zip(a, b)], np.float32)
    np.testing.assert_array_equal(got, want)
    ids = rng.integers(0, len(X), 150).astype(np.uint32)
    got = ix.query_distances(Q, ids)
    want = np.array([[oracle.distance(X[j], q, metric) for j in ids] for q in Q], np.float32)
    np.testing.assert_array_equal(got, want)
    ix.close()


def test_real_valued_floats_close_not_exact(oracle):
    """DESIGN.md "float order": on real-valued data the device sums in a different order than the
    reference's left-to-right loop; distances agree to a few ulp, not bit for bit."""
    X = datasets.deep_like(3000, 96, seed=1)
    ix = DeviceIndex(X, max_degree=8)
    rng = np.random.default_rng(0)
    a = rng.integers(0, len(X), 400).astype(np.uint32); b = rng.integers(0, len(X), 400).astype(np.uint32)
    got = ix.pair_distances(a, b)
    want = np.array([oracle.distance(X[i], X[j]) for i, j in zip(a, b)], np.float32)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)   # stated tolerance: 2e-6 relative
    ix.close()
