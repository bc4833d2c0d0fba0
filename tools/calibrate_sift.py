import sys, time, json, numpy as np
sys.path.insert(0, ".")
from parlayann_amd import DeviceIndex, datasets
from parlayann_amd.recall import recall_at_k
n = 1_000_000
for (nc, rank, cs, bs, ns) in [(256,16,22.0,9.0,12.0), (256,32,22.0,9.0,12.0), (1024,16,22.0,9.0,12.0), (256,16,22.0,9.0,25.0), (256,48,18.0,9.0,15.0), (64,64,20.0,8.0,14.0)]:
    def gen(m, seed):
        x = datasets._mixture(m, 128, seed, nc, rank, center_scale=cs, basis_scale=bs, noise_scale=ns)
        return np.clip(np.rint(x + 100.0), 0, 255).astype(np.float16)
    X = gen(n, 1234); Q = gen(10000, 4321)
    ix = DeviceIndex(X, max_degree=64)
    t = time.time(); ix.vamana_build(64, 128, 1.15, num_passes=2, seed=1); tb = time.time() - t
    gt, gd = ix.bruteforce_knn(Q, 100)
    row = {"nc": nc, "rank": rank, "cs": cs, "bs": bs, "ns": ns, "build_s": round(tb, 2)}
    for beam in (16, 32, 64):
        r = ix.batch_search(Q, k=10, beam=beam)
        row[f"b{beam}"] = (round(recall_at_k(r["ids"], gt, gd, 10), 4), round(float(r["dist_cmps"].mean())), round(float(r["visited_count"].mean()), 1))
    G = ix.get_graph(); row["avgdeg"] = round(float(G[:, 0].mean()), 1)
    print(json.dumps(row), flush=True)
    ix.close()
