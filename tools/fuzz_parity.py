#!/usr/bin/env python3
"""Randomised parity soak: random element type / metric / dimension / degree / beam / k / cut / limit / starts, device
search and robustPrune against the oracle, bit for bit.  usage: fuzz_parity.py [rounds=30] [seed=0]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402
from parlayann_amd import DeviceIndex, datasets  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
o = oracle_api.load()
bad = 0
for it in range(rounds):
    dtype = [np.uint8, np.int8, np.float32, np.float16][rng.integers(4)]
    metric = ["l2", "mips"][rng.integers(2)]
    d = int(rng.choice([8, 16, 24, 32, 48, 64, 96, 100, 128, 160, 200, 256, 384]))
    n = int(rng.integers(800, 6000)); R = int(rng.choice([8, 16, 24, 32, 48, 64, 80])); nq = int(rng.choice([50, 300, 2300]))
    X = datasets.sift_like(n, d, seed=int(rng.integers(1 << 30)), dtype=np.float32)
    Q = datasets.sift_like(nq, d, seed=int(rng.integers(1 << 30)), dtype=np.float32)
    if dtype == np.int8:
        X, Q = (X - 128).clip(-127, 127), (Q - 128).clip(-127, 127)
    X, Q = X.astype(dtype), Q.astype(dtype)
    G, _ = o.vamana_build(X, R, min(2 * R, 100), 1.2 if metric == "l2" else 1.0, seed=int(rng.integers(1 << 20)), metric=metric)
    ix = DeviceIndex(X, G, metric=metric)
    beam = int(rng.choice([1, 5, 16, 33, 64, 65, 90, 100, 128, 129, 200]))
    k = int(rng.integers(0, min(beam, 20) + 1)); cut = float(rng.choice([0.0, 1.0, 1.35, 2.0]))
    limit = None if rng.random() < 0.6 else int(rng.integers(1, 3 * beam + 2)); dl = None if rng.random() < 0.6 else int(rng.integers(1, R + 1))
    ns = int(rng.choice([1, 1, 1, 2, 7])); starts = rng.choice(n, ns, replace=False).astype(np.uint32) if ns <= beam else np.array([0], np.uint32)
    a = o.batch_search(X, G, queries=Q, k=k, beam=beam, cut=cut, limit=limit, degree_limit=dl, starts=starts, metric=metric, out_k=min(beam, 12))
    b = ix.batch_search(Q, k=k, beam=beam, cut=cut, limit=limit, degree_limit=dl, starts=starts, out_k=min(beam, 12))
    ok = all(np.array_equal(a[f], b[f]) for f in ("ids", "frontier_size", "visited_count", "dist_cmps")) and np.array_equal(a["dists"].view(np.uint32), b["dists"].view(np.uint32))
    owners = rng.choice(n, 200, replace=False).astype(np.uint32)
    cands = [rng.choice(n, int(rng.integers(0, 300)), replace=True).astype(np.uint32) for _ in owners]
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.uint64); cid = np.concatenate(cands) if off[-1] else np.zeros(0, np.uint32)
    Rp = int(rng.integers(1, R + 1)); alpha = float(rng.choice([1.0, 1.05, 1.2, 1.5])); add = bool(rng.integers(2))
    ro, dco = o.robust_prune_batch(X, G, owners, cid, None, off, alpha, Rp, add=add, metric=metric)
    rg, dcg = ix.robust_prune_batch(owners, cid, off, alpha, Rp, add_out_nbrs=add)
    ok2 = np.array_equal(ro, rg) and np.array_equal(dco, dcg)
    ix.close()
    print(f"[{it}] {np.dtype(dtype).name} {metric} d={d} n={n} R={R} nq={nq} beam={beam} k={k} cut={cut} limit={limit} dl={dl} starts={ns} | search {'ok' if ok else 'MISMATCH'} | prune R={Rp} a={alpha} add={add} {'ok' if ok2 else 'MISMATCH'}", flush=True)
    bad += (not ok) + (not ok2)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
