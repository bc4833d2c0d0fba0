#!/usr/bin/env python3
"""insert-path counters of the ground-truth kernel (diagnostic build: make -C parlayann_amd/csrc alt ALTFLAGS=-DPANN_GT_COUNTERS;
run with PANN_LIBRARY=parlayann_amd/lib/libpann_alt.so).  usage: gt_counters.py [n] [nq] [k]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parlayann_amd import DeviceIndex, datasets, _capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
X = datasets.sift1m_like(n, 128, seed=1234, dtype=np.float16)
Q = datasets.sift1m_like(nq, 128, seed=4321, dtype=np.float16)
ix = DeviceIndex(X, max_degree=8)
lib = C.CDLL(_capi.LIB_PATH)
out = (C.c_ulonglong * 8)()
lib.pann_debug_gt_counters(out, 1)
ix.bruteforce_knn(Q, k)
lib.pann_debug_gt_counters(out, 1)
c = list(out)
wt = max(c[0], 1)
print(f"wave-tiles {c[0]}; with a survivor {c[1]} ({c[1] / wt:.3f}); (row set, column block) groups with an offer {c[5]} ({c[5] / wt:.3f}/wave-tile); "
      f"keys offered {c[3]} ({c[3] / wt:.3f}/wave-tile, {c[3] / nq:.0f}/query); flushes {c[4]} ({c[4] / wt:.4f}/wave-tile); "
      f"insert rounds {c[2]} ({c[2] / wt:.3f}/wave-tile, {c[3] / max(c[2], 1):.2f} keys/round)")
