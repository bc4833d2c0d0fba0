#!/usr/bin/env python3
"""HCNNG build time on fp16 points (the dense MFMA kernel does the leaf kNN): 1M x 128, 30 trees, leaf 1000."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parlayann_amd import DeviceIndex, datasets
X = datasets.sift_like(1_000_000, 128, seed=1234, dtype=np.float16)
ix = DeviceIndex(X, max_degree=90)
ix.hcnng_build(2, 1000, 3, seed=5)
ix.close()
ix = DeviceIndex(X, max_degree=90)
t0 = time.time(); t = ix.hcnng_build(30, 1000, 3, seed=1); dt = time.time() - t0
G = ix.get_graph()
print(f"build {dt:.3f}s phases {t} avg_degree {G[:, 0].mean():.3f} checksum {int(G.astype(np.uint64).sum())}")
