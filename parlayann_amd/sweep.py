"""search_and_parse (algorithms/utils/check_nn_recall.h:181-268) + parse_result
(parse_results.h:192-218): the reference's way of reporting QPS at recall -- a sweep over 43 beam
widths, 20 visit limits and one "best accuracy" point, then the best-QPS result per recall bucket.
Every point is one batched device search (host pointers in, so QPS here is PCIe-inclusive;
bench.py reports the device-resident number)."""
import time

import numpy as np

from .recall import recall_at_k

BEAMS = [10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 45, 50, 55, 60, 65, 70, 80, 90,
         100, 120, 140, 160, 180, 200, 225, 250, 275, 300, 375, 500, 750, 1000]                     # :217-219
LIMITS = [10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 30, 35]            # :243
BUCKETS = [.1, .2, .3, .4, .5, .6, .7, .75, .8, .85, .9, .93, .95, .97, .98, .99, .995, .999, .9995, .9999, .99995,
           .99999]                                                                                    # :259-261


def check_recall(index, queries, gt_ids, gt_dists, k, qp, verbose=False):
    """checkRecall (:17-125): time only the batched search; tie-aware recall; QPS = nq / time."""
    t0 = time.perf_counter()
    r = index.batch_search(queries, k=qp["k"], beam=qp["beam"], cut=qp["cut"], limit=qp["limit"],
                           degree_limit=qp["degree_limit"], out_k=k)
    dt = time.perf_counter() - t0
    res = {"recall": recall_at_k(r["ids"], gt_ids, gt_dists, k), "QPS": len(queries) / dt, "k": k, "beamQ": qp["beam"],
           "cut": qp["cut"], "limit": qp["limit"], "degree_limit": qp["degree_limit"],
           "avg_cmps": int(r["dist_cmps"].astype(np.uint64).sum() // len(queries)),
           "avg_visited": int(r["visited_count"].astype(np.uint64).sum() // len(queries))}
    if verbose:
        print(f"search: Q={qp['beam']}, k={qp['k']}, limit={qp['limit']}, recall={res['recall']:.6g}, "
              f"visited={res['avg_visited']}, comparisons={res['avg_cmps']}, QPS={res['QPS']:.6g}, "
              f"ctime={1 / (res['QPS'] * max(res['avg_cmps'], 1)) * 1e9:.6g}")
    return res


def parse_result(results, buckets=BUCKETS):
    """for each bucket b_i: among results with b_i <= recall <= b_{i+1} (last bucket: recall >= b)
    the one with the highest QPS."""
    out, out_b = [], []
    for i, b in enumerate(buckets):
        cand = [r for r in results if r["recall"] >= np.float32(b)]
        if i != len(buckets) - 1 and cand:
            cand = [r for r in cand if r["recall"] <= np.float32(buckets[i + 1])]
        if cand:
            out.append(max(cand, key=lambda r: r["QPS"])); out_b.append(b)
    return out, out_b


def search_and_parse(index, queries, gt_ids, gt_dists, k, fixed_beam_width=0, verbose=False, beams=BEAMS, limits=LIMITS):
    n, maxdeg = index.n, index.max_degree
    r = k if k else 10
    if fixed_beam_width:                                                            # the -Q path (:221-226)
        qp = dict(k=r, beam=fixed_beam_width, cut=1.35, limit=n, degree_limit=maxdeg)
        return [check_recall(index, queries, gt_ids, gt_dists, r, qp, verbose) for _ in range(5)], None
    results = []
    for Q in beams:                                                                 # :228-238
        if Q >= r:
            results.append(check_recall(index, queries, gt_ids, gt_dists, r,
                                        dict(k=r, beam=Q, cut=1.35, limit=n, degree_limit=maxdeg), verbose))
    for lim in limits:                                                              # "limited accuracy" :243-253
        results.append(check_recall(index, queries, gt_ids, gt_dists, r,
                                    dict(k=r, beam=max(lim, r), cut=1.35, limit=lim, degree_limit=min(maxdeg, 5 * lim)), verbose))
    if gt_ids.shape[1] >= 100 and 1000 in beams:                                    # "best accuracy" :255-256
        results.append(check_recall(index, queries, gt_ids, gt_dists, r,
                                    dict(k=100, beam=1000, cut=10.0, limit=n, degree_limit=maxdeg), verbose))
    return results, parse_result(results)
