"""Batched evaluators: the reference's per-case / per-item / per-block Python loops as a handful of GPU calls.

  evaluate_protocol        src/ncf/ncf.ipynb cell 8 evaluate_metrics (115,056 forwards of 100 pairs)      
  autorec_ranking_metrics  src/autorec/utils/helper.py:232-349 get_ranking_metrics (3,706 forwards)       
  hybrid_ranking_metrics   hybridautorecncf.ipynb cell 13 evaluate_ranking_metrics (O(24 x 3706) forwards)

Scores and top-K come from the HIP kernels; HR / NDCG are computed on the host in float64 from the returned ids, with
the same formulas as the reference (helpers/ranking_metrics.py), so they are identical by construction.
"""
import numpy as np
import torch

from . import ops
from .helpers.ranking_metrics import hit_multi, hit_ndcg_batch, ndcg_multi


def evaluate_protocol(model, users, items, top_k=10):
    """Mean HR@k / NDCG@k over test cases; items[c, 0] is the positive, the rest its sampled negatives.

    One pair-list forward over all cases and one segmented top-k; `recommends` of the reference is
    torch.take(item, topk indices), `gt_item` is item[0] (ncf.ipynb cell 8).
    Returns (mean_hr, mean_ndcg, hr[c], ndcg[c]).
    """
    users = np.asarray(users); items = np.asarray(items)
    cases, width = items.shape
    u = torch.from_numpy(np.repeat(users.astype(np.int64), width))
    it = torch.from_numpy(items.astype(np.int64).reshape(-1))
    device = model._compute_device()
    scores = model(u.to(device), it.to(device))
    ids, _ = ops.topk_rows(scores.view(cases, width), min(top_k, width))
    top_items = np.take_along_axis(items, ids.cpu().numpy().astype(np.int64), axis=1)
    hr, nd = hit_ndcg_batch(top_items, items[:, 0])
    return float(hr.mean()), float(nd.mean()), hr, nd


def _multi_gt_metrics(top_ids, test_matrix):
    hrs, nds = [], []
    for u in range(test_matrix.shape[0]):
        gt = np.flatnonzero(test_matrix[u] > 0)
        if gt.size == 0:                                       # users without test items are skipped (helper.py:329-331)
            continue
        hrs.append(hit_multi(gt, top_ids[u])); nds.append(ndcg_multi(gt, top_ids[u]))
    if not hrs:
        return 0.0, 0.0
    return float(np.mean(hrs)), float(np.mean(nds))


def autorec_ranking_metrics(model, train_matrix, test_matrix, top_k=10):
    """HR@k / NDCG@k of an AutoRec model over the full catalog with seen items masked (helper.py:232-349)."""
    train = torch.as_tensor(np.asarray(train_matrix, np.float32))
    ids, _ = model.recommend_topk(train, top_k)
    return _multi_gt_metrics(ids.cpu().numpy(), np.asarray(test_matrix))


def hybrid_ranking_metrics(model, train_matrix, test_matrix, top_k=10):
    """Full-catalog HR@k / NDCG@k of the Hybrid model: encoders once, fused head over all pairs, seen items masked."""
    train = torch.as_tensor(np.asarray(train_matrix, np.float32))
    ids, _ = model.score_all(train, k=top_k, exclude=train > 0)
    return _multi_gt_metrics(ids.cpu().numpy(), np.asarray(test_matrix))
