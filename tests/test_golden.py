"""Regression fixtures (tests/golden/oracle_v1.npz, made by tests/golden/make_golden.py from THIS
repo's oracle -- not reference output, see the script's header).  CPU: the oracle still reproduces
them.  GPU: the device path reproduces them through the C-ABI."""
import os

import numpy as np
import pytest

from golden.make_golden import CASES

G_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(G_PATH)


def _check(gold, name, r):
    for f in ("ids", "dists", "frontier_size", "visited_count", "dist_cmps"):
        np.testing.assert_array_equal(gold[f"{name}_{f}"], r[f], err_msg=f"{name}:{f}")


def test_oracle_reproduces_golden(oracle, gold):
    X, Q, G = gold["X"], gold["Q"], gold["G"]
    G2, _ = oracle.vamana_build(X, 16, 32, 1.2, num_passes=1, seed=5)
    np.testing.assert_array_equal(G, G2)
    for name, k, beam, cut, limit, dl in CASES:
        _check(gold, name, oracle.batch_search(X, G, queries=Q, k=k, beam=beam, cut=cut, limit=limit, degree_limit=dl,
                                               out_k=min(beam, 16)))
    rows, dc = oracle.robust_prune_batch(X, G, gold["prune_owners"], gold["prune_cand"].ravel(), None,
                                         (np.arange(len(gold["prune_owners"]) + 1) * 30).astype(np.uint64), 1.2, 16)
    np.testing.assert_array_equal(rows, gold["prune_rows"]); np.testing.assert_array_equal(dc, gold["prune_dc"])
    li, ld = oracle.leaf_knn(X, gold["leaf_ids"], 10)
    np.testing.assert_array_equal(li, gold["leaf_nn"]); np.testing.assert_array_equal(ld, gold["leaf_d"])


@pytest.mark.gpu
def test_device_reproduces_golden(gold):
    from parlayann_amd import DeviceIndex
    X, Q, G = gold["X"], gold["Q"], gold["G"]
    ix = DeviceIndex(X, G)
    for name, k, beam, cut, limit, dl in CASES:
        _check(gold, name, ix.batch_search(Q, k=k, beam=beam, cut=cut, limit=limit, degree_limit=dl, out_k=min(beam, 16)))
    rows, dc = ix.robust_prune_batch(gold["prune_owners"], gold["prune_cand"].ravel(),
                                     (np.arange(len(gold["prune_owners"]) + 1) * 30).astype(np.uint64), 1.2, 16)
    np.testing.assert_array_equal(rows, gold["prune_rows"]); np.testing.assert_array_equal(dc, gold["prune_dc"])
    li, ld = ix.leaf_knn(gold["leaf_ids"], 10)
    np.testing.assert_array_equal(li, gold["leaf_nn"]); np.testing.assert_array_equal(ld, gold["leaf_d"])
    ix2 = DeviceIndex(X, max_degree=16)
    ix2.vamana_build(16, 32, 1.2, num_passes=1, seed=5)
    G2 = ix2.get_graph()
    cols = np.arange(16)[None, :]
    np.testing.assert_array_equal(G[:, 0], G2[:, 0])
    np.testing.assert_array_equal(np.where(cols < G[:, :1], G[:, 1:], 0), G2[:, 1:])
    ix.close(); ix2.close()


G2_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v2.npz")


def test_oracle_reproduces_golden_v2(oracle):
    """range search, HCNNG, quantisers, real-valued float search: the oracle regenerates tests/golden/oracle_v2.npz"""
    from golden.make_golden_v2 import compute
    gold = np.load(G2_PATH)
    now = compute(oracle)
    assert set(now) == set(gold.files)
    for k in gold.files:
        np.testing.assert_array_equal(gold[k], now[k], err_msg=k)


@pytest.mark.gpu
def test_device_reproduces_golden_v2():
    from parlayann_amd import DeviceIndex, quantize
    gold = np.load(G2_PATH)
    ix = DeviceIndex(gold["rs_X"], gold["rs_G"])
    r = ix.range_search(gold["rs_starts"], float(gold["rs_r2"]), 128, queries=gold["rs_Q"])
    np.testing.assert_array_equal(r["ids"], gold["rs_ids"]); np.testing.assert_array_equal(r["counts"], gold["rs_counts"])
    np.testing.assert_array_equal(r["dist_cmps"], gold["rs_cmps"])
    ix.close()
    ih = DeviceIndex(gold["rs_X"], max_degree=gold["hc_G"].shape[1] - 1)
    ih.hcnng_build(3, 100, 3, seed=9)
    np.testing.assert_array_equal(ih.get_graph(), gold["hc_G"])
    ih.close()
    p = quantize.euclid_u8_params(gold["qz_F"])
    assert p.slope == gold["qz_slope"] and int(p.offset) == int(gold["qz_off"])
    np.testing.assert_array_equal(quantize.euclid_u8_translate(gold["qz_F"], p), gold["qz_u8"])
    for trim in (0, 1):
        mv = quantize.mips_i8_max_val(gold["qz_M"], trim=bool(trim))
        assert np.float32(mv) == gold[f"qz_mv{trim}"]
        np.testing.assert_array_equal(quantize.mips_i8_translate(gold["qz_M"], mv), gold[f"qz_i8_{trim}"])
    fx = DeviceIndex(gold["fl_X"], gold["fl_G"], exact_float_order=True)          # real-valued floats: exact-order mode
    fr = fx.batch_search(gold["fl_Q"], k=10, beam=32, out_k=16)
    np.testing.assert_array_equal(fr["ids"], gold["fl_ids"]); np.testing.assert_array_equal(fr["dists"].view(np.uint32), gold["fl_dists"].view(np.uint32))
    np.testing.assert_array_equal(fr["dist_cmps"], gold["fl_cmps"]); np.testing.assert_array_equal(fr["visited_count"], gold["fl_vis"])
    fx.close()


G3_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v3.npz")


def test_oracle_reproduces_golden_v3(oracle):
    """bfloat16 points, the single-batch build, exact kNN at k = 100 / 10, per-point build statistics: the oracle regenerates
    tests/golden/oracle_v3.npz"""
    from golden.make_golden_v3 import compute
    gold = np.load(G3_PATH)
    now = compute(oracle)
    assert set(now) == set(gold.files)
    for k in gold.files:
        np.testing.assert_array_equal(gold[k], now[k], err_msg=k)


@pytest.mark.gpu
def test_device_reproduces_golden_v3():
    from parlayann_amd import DeviceIndex
    from parlayann_amd.bf16 import bfloat16
    gold = np.load(G3_PATH)
    X = gold["sb_X"]; cols = np.arange(16)[None, :]

    def norm(G):
        return np.concatenate([G[:, :1], np.where(cols < G[:, :1], G[:, 1:], 0)], 1)
    ix = DeviceIndex(X, max_degree=16)
    st = ix.vamana_build(16, 32, 1.2, num_passes=2, seed=5, single_batch=4)
    np.testing.assert_array_equal(norm(ix.get_graph()), norm(gold["sb_G"]))
    assert (st.search_dist_cmps, st.prune_dist_cmps) == tuple(int(v) for v in gold["sb_cmps"])
    Q = gold["gt_Q"]
    for k in (100, 10):                                   # register lists of 8 / 1 registers per row, v_dot4 tile
        gi, gd = ix.bruteforce_knn(Q, k)
        np.testing.assert_array_equal(gi, gold[f"gt_u8_ids{k}"]); np.testing.assert_array_equal(gd, gold[f"gt_u8_d{k}"])
    ix.close()
    ih = DeviceIndex(X.astype(np.float16), max_degree=8, metric="mips")        # the MFMA ground-truth kernel
    gi, gd = ih.bruteforce_knn(Q.astype(np.float16), 100)
    np.testing.assert_array_equal(gi, gold["gt_f16_mips_ids"]); np.testing.assert_array_equal(gd, gold["gt_f16_mips_d"])
    ih.close()
    ib = DeviceIndex(gold["bf_X"].view(bfloat16), gold["bf_G"])
    rb = ib.batch_search(gold["bf_Q"].view(bfloat16), k=10, beam=32, out_k=12)
    np.testing.assert_array_equal(rb["ids"], gold["bf_ids"]); np.testing.assert_array_equal(rb["dists"], gold["bf_dists"])
    np.testing.assert_array_equal(rb["dist_cmps"], gold["bf_cmps"])
    ib.close()
    vis = np.zeros(len(X), np.uint32); dc = np.zeros(len(X), np.uint32)
    ip = DeviceIndex(X, max_degree=16)
    ip.vamana_build(16, 32, 1.2, num_passes=1, seed=9, point_stats=(vis, dc))
    np.testing.assert_array_equal(norm(ip.get_graph()), norm(gold["ps_G"]))
    np.testing.assert_array_equal(vis, gold["ps_visited"]); np.testing.assert_array_equal(dc, gold["ps_dists"])
    ip.close()

