#!/usr/bin/env python3
"""C3-shaped build with the batch searches launched in locality order, for several pivot counts (round 3).
usage: locality_pivots_sweep.py [n=10000000] pivots..."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parlayann_amd import DeviceIndex, datasets
n = int(sys.argv[1]); piv = [a for a in sys.argv[2:]]      # "pivots" or "pivots:groups"; 0 = batch order
X = datasets.deep_like(n, 96, seed=1234)
ix = DeviceIndex(X, max_degree=64)
for spec in piv:
    p, g = (int(v) for v in (spec + ":32").split(":")[:2])
    ix.clear_graph()
    if p == 0:
        ix.set_option("locality_order", 0)
    else:
        ix.set_option("locality_order", 1); ix.set_option("locality_pivots", p); ix.set_option("locality_groups", g)
    t0 = time.time(); st = ix.vamana_build(64, 128, 1.05, num_passes=2, seed=1); tb = time.time() - t0
    print(json.dumps({"pivots": p, "groups": g, "build_s": tb, "search_s": st.t_search_s, "prune_s": st.t_prune_s, "cmps": st.search_dist_cmps}), flush=True)
