#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_v1.npz.

PROVENANCE: these vectors are produced by THIS repository's CPU oracle (oracle/pann_oracle.cpp),
NOT by the reference -- the reference cannot be built or imported in this environment (DESIGN.md
section 1, "parity unpinned").  They pin the oracle and the device path against regressions across
rounds; they are not evidence of parity with the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402
from parlayann_amd import datasets  # noqa: E402

CASES = [  # (name, k, beam, cut, limit, degree_limit)
    ("q_b64", 10, 64, 1.35, None, None), ("q_b16", 10, 16, 1.35, None, None), ("q_b100", 10, 100, 1.35, None, None),
    ("q_nocut", 10, 64, 0.0, None, None), ("q_lim", 10, 32, 1.35, 20, 12), ("build_L48", 0, 48, 0.0, None, None),
]


def main():
    o = oracle_api.load()
    X = datasets.sift_like(2000, 32, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(40, 32, seed=4321, dtype=np.uint8)
    G, _ = o.vamana_build(X, 16, 32, 1.2, num_passes=1, seed=5)
    out = {"X": X, "Q": Q, "G": G}
    for name, k, beam, cut, limit, dl in CASES:
        r = o.batch_search(X, G, queries=Q, k=k, beam=beam, cut=cut, limit=limit, degree_limit=dl, out_k=min(beam, 16))
        for f in ("ids", "dists", "frontier_size", "visited_count", "dist_cmps"):
            out[f"{name}_{f}"] = r[f]
    owners = np.arange(0, 2000, 97, dtype=np.uint32)
    cand = np.stack([(owners * 7 + 13 * j) % 2000 for j in range(30)], axis=1).astype(np.uint32)
    off = (np.arange(len(owners) + 1) * 30).astype(np.uint64)
    rows, dc = o.robust_prune_batch(X, G, owners, cand.ravel(), None, off, 1.2, 16)
    out.update(prune_owners=owners, prune_cand=cand, prune_rows=rows, prune_dc=dc)
    ids = np.arange(0, 2000, 9, dtype=np.uint32)
    li, ld = o.leaf_knn(X, ids, 10)
    out.update(leaf_ids=ids, leaf_nn=li, leaf_d=ld)
    np.savez_compressed(os.path.join(HERE, "oracle_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_v1.npz"))


if __name__ == "__main__":
    main()
