#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (pann_batch_search): queries and results in ordinary
numpy arrays, 10K queries per call, the bench workload (1M x 128 fp16, beam 64).  Never the bench `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parlayann_amd import DeviceIndex, datasets  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = datasets.sift1m_like(n, 128, seed=1234, dtype=np.float16)
Q = datasets.sift1m_like(10_000, 128, seed=4321, dtype=np.float16)
ix = DeviceIndex(X, max_degree=64)
ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1)
for what, kw in (("ids only", dict(want_dists=False)), ("ids + dists", dict())):
    best = 1e9
    for _ in range(12):
        t0 = time.perf_counter(); r = ix.batch_search(Q, k=10, beam=64, **kw); best = min(best, time.perf_counter() - t0)
    print(f"{what}: {len(Q) / best / 1e6:.2f} M queries/s host-inclusive ({best * 1e3:.2f} ms per 10K batch)")
