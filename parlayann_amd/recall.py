"""Tie-aware recall, as checkRecall computes it (algorithms/utils/check_nn_recall.h:83-109):
the first k ground-truth ids plus every later ground-truth entry whose distance equals the k-th
distance form the accepted set; each accepted id found among the k reported ids counts once; the
sum is divided by k * nq."""
import numpy as np


def recall_at_k(result_ids, gt_ids, gt_dists, k):
    result_ids = np.asarray(result_ids)[:, :k]
    gt_ids = np.asarray(gt_ids)
    gt_dists = np.asarray(gt_dists)
    nq = len(result_ids)
    accept = np.zeros(gt_ids.shape, dtype=bool)
    accept[:, :k] = True
    accept[:, k:] = gt_dists[:, k:] == gt_dists[:, k - 1:k]
    hits = 0
    for i in range(nq):
        hits += int(np.isin(gt_ids[i][accept[i]], result_ids[i]).sum())
    return hits / float(k * nq)
