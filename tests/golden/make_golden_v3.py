#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_v3.npz: fixtures for what round 2 added -- bfloat16 points, the single-batch build
(BuildParams::single_batch), exact brute-force kNN at k = 100 / 10 (the register-list ground-truth kernels) and per-point build
statistics.

PROVENANCE: produced by THIS repository's CPU oracle, not by the reference (see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402
from parlayann_amd import datasets  # noqa: E402
from parlayann_amd.bf16 import bfloat16  # noqa: E402


def compute(o):
    out = {}
    X = datasets.sift_like(1500, 32, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(40, 32, seed=4321, dtype=np.uint8)
    # single-batch build: start edges + one batch per pass
    Gs, st = o.vamana_build(X, 16, 32, 1.2, num_passes=2, seed=5, single_batch=4)
    out.update(sb_X=X, sb_G=Gs, sb_cmps=np.asarray(st[:2], np.uint64))
    # exact kNN, k = 100 and k = 10, uint8 and fp16 (integer-valued: bit-exact on every path)
    for k in (100, 10):
        i8, d8 = o.bruteforce_knn(X, Q, k)
        out[f"gt_u8_ids{k}"] = i8; out[f"gt_u8_d{k}"] = d8
    Xh = X.astype(np.float16); Qh = Q.astype(np.float16)
    ih, dh = o.bruteforce_knn(Xh, Qh, 100, "mips")
    out.update(gt_Q=Q, gt_f16_mips_ids=ih, gt_f16_mips_d=dh)
    # bfloat16 points: search on an oracle-built graph
    Xb = datasets.sift_like(1500, 32, seed=77, dtype=bfloat16); Qb = datasets.sift_like(40, 32, seed=78, dtype=bfloat16)
    Gb, _ = o.vamana_build(Xb, 16, 32, 1.2, num_passes=1, seed=3)
    rb = o.batch_search(Xb, Gb, queries=Qb, k=10, beam=32, out_k=12)
    out.update(bf_X=Xb.view(np.uint16), bf_Q=Qb.view(np.uint16), bf_G=Gb, bf_ids=rb["ids"], bf_dists=rb["dists"], bf_cmps=rb["dist_cmps"])
    # per-point build statistics (stats.h:57-94)
    vis = np.zeros(len(X), np.uint32); dc = np.zeros(len(X), np.uint32)
    Gp, _ = o.vamana_build(X, 16, 32, 1.2, num_passes=1, seed=9, point_stats=(vis, dc))
    out.update(ps_G=Gp, ps_visited=vis, ps_dists=dc)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_v3.npz"), **compute(oracle_api.load()))
    print("wrote", os.path.join(HERE, "oracle_v3.npz"))
