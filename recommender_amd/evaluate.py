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
from .helpers.ranking_metrics import hit_ndcg_batch, multi_gt_metrics_batch


def evaluate_protocol(model, users, items, top_k=10):
    """Mean HR@k / NDCG@k over test cases; items[c, 0] is the positive, the rest its sampled negatives.

    One pair-list forward over all cases and one segmented top-k; `recommends` of the reference is
    torch.take(item, topk indices), `gt_item` is item[0] (ncf.ipynb cell 8).
    Returns (mean_hr, mean_ndcg, hr[c], ndcg[c]).
    """
    users = np.asarray(users); items = np.asarray(items)
    cases, width = items.shape
    device = model._compute_device()
    # Host work and PCIe, not the 2.7 ms of kernels, were the time of this function (32 ms at 115,056 x 100): the id lists go up
    # as they are (no np.repeat of the users, no int64 copy of an int64 matrix), users are repeated and the winners' item ids
    # gathered on the device, and one small copy brings back [positive | top-k items] per case.
    items_d = torch.from_numpy(np.ascontiguousarray(items)).to(device)
    if items_d.dtype != torch.int64:
        items_d = items_d.long()
    users_d = torch.from_numpy(np.ascontiguousarray(users)).to(device).long().repeat_interleave(width)
    scores = model(users_d, items_d.view(-1))
    ids, _ = ops.topk_rows(scores.view(cases, width), min(top_k, width))
    back = torch.cat([items_d[:, :1], torch.gather(items_d, 1, ids.long())], dim=1).cpu().numpy()
    hr, nd = hit_ndcg_batch(back[:, 1:], back[:, 0])
    return float(hr.mean()), float(nd.mean()), hr, nd


def _multi_gt_metrics(top_ids, test_matrix):
    hr, nd, has_gt = multi_gt_metrics_batch(top_ids, test_matrix)     # users without test items are skipped (helper.py:329-331)
    if not has_gt.any():
        return 0.0, 0.0
    return float(np.mean(hr[has_gt])), float(np.mean(nd[has_gt]))


def autorec_ranking_metrics(model, train_matrix, test_matrix, top_k=10):
    """HR@k / NDCG@k of an AutoRec model over the full catalog with seen items masked (helper.py:232-349)."""
    train = torch.as_tensor(np.asarray(train_matrix, np.float32))
    ids, _ = model.recommend_topk(train, top_k)
    return _multi_gt_metrics(ids.cpu().numpy(), np.asarray(test_matrix))


def hybrid_ranking_metrics(model, train_matrix, test_matrix, top_k=10):
    """Full-catalog HR@k / NDCG@k of the Hybrid model: encoders once, fused head over all pairs, seen items masked."""
    ops.require_gpu()
    p = next(model.parameters())
    device = p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())
    train = torch.as_tensor(np.asarray(train_matrix, np.float32)).to(device)      # one upload; the select reads the seen items off it
    ids, _ = model.score_all(train, k=top_k, mask_seen=True)
    return _multi_gt_metrics(ids.cpu().numpy(), np.asarray(test_matrix))
