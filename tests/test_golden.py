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
