#!/usr/bin/env python3
"""wall time of the brute-force ground truth (pann_bruteforce_knn, host pointers in and out): nq x n, k = 100.
usage: gt_time.py [n=1000000] [nq=10000] [dtype=f16|bf16|u8|f32] [k=100]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parlayann_amd import DeviceIndex, bfloat16, datasets  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dt = {"f16": np.float16, "bf16": bfloat16, "u8": np.uint8, "f32": np.float32}[sys.argv[3] if len(sys.argv) > 3 else "f16"]
X = datasets.sift1m_like(n, 128, seed=1234, dtype=dt)
Q = datasets.sift1m_like(nq, 128, seed=4321, dtype=dt)
ix = DeviceIndex(X, max_degree=8)
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
if os.environ.get("GT_PIECES"):
    ix.set_option("gt_pieces", int(os.environ["GT_PIECES"]))
ix.bruteforce_knn(Q[:256], k)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); ids, d = ix.bruteforce_knn(Q, k); best = min(best, time.perf_counter() - t0)
print(f"bruteforce {nq} x {n} {sys.argv[3] if len(sys.argv) > 3 else 'f16'} k={k}: {best * 1e3:.1f} ms (host-inclusive, best of 3); checksum {int(ids.astype(np.uint64).sum())}")
