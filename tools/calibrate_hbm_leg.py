#!/usr/bin/env python3
"""Calibration of bench.py's HBM-resident leg: the 12.5M x 128 fp16 table must reach recall@10 >= 0.95 at
beam 64 so that the leg is quoted at the metric's recall.  For each noise scale of datasets.sift_like_device: build, search,
recall.  usage: tools/calibrate_hbm_leg.py 12 11 10 9"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from parlayann_amd import DeviceIndex, datasets  # noqa: E402
from parlayann_amd.recall import recall_at_k  # noqa: E402

n = int(os.environ.get("HBM_N", 12_500_000))
dev = torch.device("cuda", 0)
for noise in [float(a) for a in sys.argv[1:]]:
    t0 = time.time()
    X = datasets.sift_like_device(n, 128, 1234, dev, np.float16, noise_scale=noise)
    Q = datasets.sift_like_device(10000, 128, 4321, dev, np.float16, noise_scale=noise)
    tg = time.time() - t0
    ix = DeviceIndex(X, max_degree=64)
    t0 = time.time(); ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1); tb = time.time() - t0
    r = ix.batch_search(Q, k=10, beam=64)
    gt, gd = ix.bruteforce_knn(Q, 100)
    print(json.dumps({"noise": noise, "n": n, "gen_s": tg, "build_s": tb, "recall_at_10": recall_at_k(r["ids"], gt, gd, 10),
                      "visited": float(r["visited_count"].mean()), "cmps": float(r["dist_cmps"].mean())}), flush=True)
    ix.close(); del X
