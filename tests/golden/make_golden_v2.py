#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_v2.npz: fixtures for the paths added after oracle_v1 -- BFS range search, the HCNNG
builder, the scalar quantisers, and real-valued float search (pinned by the exact-float-order mode on the device).

PROVENANCE: produced by THIS repository's CPU oracle, not by the reference (see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402
from parlayann_amd import datasets  # noqa: E402


def compute(o):
    out = {}
    X = datasets.sift_like(1500, 32, seed=1234, dtype=np.uint8)
    Q = datasets.sift_like(30, 32, seed=4321, dtype=np.uint8)
    G, _ = o.vamana_build(X, 16, 32, 1.2, num_passes=1, seed=5)
    starts = o.batch_search(X, G, queries=Q, k=5, beam=16)["ids"][:, :5].copy()
    r2 = float(np.median(o.bruteforce_knn(X, Q, 15)[1][:, -1]))
    rs = o.range_search(X, G, starts, r2, 128, queries=Q)
    out.update(rs_X=X, rs_Q=Q, rs_G=G, rs_starts=starts, rs_r2=np.float32(r2), rs_ids=rs["ids"], rs_counts=rs["counts"],
               rs_cmps=rs["dist_cmps"])
    out["hc_G"] = o.hcnng_build(X, 3, 100, 3, seed=9)
    F = datasets.deep_like(1200, 24, seed=7) * 3.0 - 0.2
    slope, off = o.euclid_u8_params(F)
    out.update(qz_F=F.astype(np.float32), qz_slope=np.float32(slope), qz_off=np.int32(off), qz_u8=o.euclid_u8_translate(F, slope, off))
    M = datasets.t2i_like(1200, 20, seed=8)
    for trim in (0, 1):
        mv = o.mips_i8_maxval(M, trim=bool(trim))
        out[f"qz_mv{trim}"] = np.float32(mv); out[f"qz_i8_{trim}"] = o.mips_i8_translate(M, mv)
    out["qz_M"] = M.astype(np.float32)
    R = datasets.deep_like(1500, 24, seed=11); RQ = datasets.deep_like(30, 24, seed=12)
    GR, _ = o.vamana_build(R, 16, 32, 1.2, num_passes=1, seed=5)
    fr = o.batch_search(R, GR, queries=RQ, k=10, beam=32, out_k=16)
    out.update(fl_X=R, fl_Q=RQ, fl_G=GR, fl_ids=fr["ids"], fl_dists=fr["dists"], fl_cmps=fr["dist_cmps"], fl_vis=fr["visited_count"])
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_v2.npz"), **compute(oracle_api.load()))
    print("wrote", os.path.join(HERE, "oracle_v2.npz"))
