"""HR / NDCG for one test case; host-side, float64, same results as src/helpers/ranking_metrics.py:7-54.

Batched forms used by the evaluators operate on whole [cases, K] top-K arrays at once.
"""
import numpy as np


def hit(gt_item, pred_items):
    """1 if gt_item is among pred_items else 0 (ranking_metrics.py:7-23)."""
    for p in pred_items:
        if p == gt_item:
            return 1
    return 0


def ndcg(gt_item, pred_items):
    """1/log2(rank+2) of the FIRST occurrence of gt_item, else 0.0 (ranking_metrics.py:26-54)."""
    for rank, p in enumerate(pred_items):
        if p == gt_item:
            return np.reciprocal(np.log2(rank + 2))
    return 0.0


def hit_ndcg_batch(top_items, gt_items):
    """Vectorised over cases: top_items [cases, K], gt_items [cases] -> (hr float64[cases], ndcg float64[cases])."""
    top_items = np.asarray(top_items)
    gt = np.asarray(gt_items).reshape(-1, 1)
    match = top_items == gt
    any_hit = match.any(axis=1)
    first = match.argmax(axis=1)
    nd = np.where(any_hit, np.reciprocal(np.log2(first + 2.0)), 0.0)
    return any_hit.astype(np.float64), nd


def hit_multi(gt_items, pred_items):
    """Multi-ground-truth HR (src/autorec/utils/helper.py:176-192)."""
    if len(gt_items) == 0:
        return 0.0
    gt = set(int(g) for g in gt_items)
    return 1.0 if any(int(p) in gt for p in pred_items) else 0.0


def ndcg_multi(gt_items, pred_items):
    """Multi-ground-truth NDCG normalised by the ideal DCG (src/autorec/utils/helper.py:195-229)."""
    if len(gt_items) == 0:
        return 0.0
    gt = set(int(g) for g in gt_items)
    dcg = 0.0
    for rank, p in enumerate(pred_items):
        if int(p) in gt:
            dcg += 1.0 / np.log2(rank + 2)
    ideal = sum(1.0 / np.log2(r + 2) for r in range(min(len(gt), len(pred_items))))
    return dcg / ideal if ideal else 0.0
