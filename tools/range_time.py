#!/usr/bin/env python3
"""range_search (beamSearch.h:245-306) throughput the way its live caller uses it (vamana/neighbors.h:86-104): a beam search
per query, its results within the radius are the start points, then the BFS over everything within the radius.
1M x 128 fp16 SIFT-1M-shaped, 10K queries, radius = the median distance of the query's 50th neighbour (about 50 results per
query).  usage: range_time.py [n=1000000] [nq=10000] [rank=50] [max_results=512]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parlayann_amd import DeviceIndex, datasets  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 50
X = datasets.sift1m_like(n, 128, seed=1234, dtype=np.float16)
Q = datasets.sift1m_like(nq, 128, seed=4321, dtype=np.float16)
ix = DeviceIndex(X, max_degree=64)
ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1)
gt, gd = ix.bruteforce_knn(Q, 100)
radius = float(np.median(gd[:, rank - 1]))
t0 = time.perf_counter(); b = ix.batch_search(Q, k=10, beam=64); tb = time.perf_counter() - t0
starts = np.where(b["dists"] <= radius, b["ids"], 0xFFFFFFFF).astype(np.uint32)       # the beam's results inside the ball
starts[:, 0] = np.where((starts != 0xFFFFFFFF).any(1), starts[:, 0], b["ids"][:, 0])     # (never an empty start list)
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 512
ix.range_search(starts[:64], radius, cap, queries=Q[:64])
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); r = ix.range_search(starts, radius, cap, queries=Q); best = min(best, time.perf_counter() - t0)
truth = (gd <= radius).sum(1)                       # exact count inside the ball, capped at 100 by the ground truth depth
found = np.minimum(r["counts"], 100)
print(f"range search {nq} queries, radius^2 {radius:.0f} (median rank-{rank} distance): {best * 1e3:.2f} ms host-inclusive = "
      f"{nq / best / 1e6:.2f} M queries/s; results/query {r['counts'].mean():.1f}, dist cmps/query {r['dist_cmps'].mean():.0f} "
      f"({r['dist_cmps'].sum() * 256 / best / 1e9:.0f} GB/s of candidate rows), truncated {int(r['truncated'].sum())}, "
      f"recall of the ball (first 100) {found.sum() / max(truth.sum(), 1):.4f}; the beam search before it: {tb * 1e3:.2f} ms")
