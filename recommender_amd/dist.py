"""User-sharded scoring across the GPUs of one node.

One process per GPU (torchrun / torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Tables and weights are replicated; every rank scores a contiguous user range against the whole catalog;
the only data-path collective is ONE all-gather of the per-user top-k lists (k x 8 bytes per user).
With user sharding the lists are disjoint, so the "merge" is a concatenation in rank order.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous split of range(total): the first total % world ranks get one extra element."""
    q, r = divmod(int(total), int(world))
    start = rank * q + min(rank, r)
    return start, q + (1 if rank < r else 0)


def all_gather_topk(ids, vals, counts=None, group=None):
    """Concatenate per-rank (ids [u_r,k], vals [u_r,k]) into ([U,k], [U,k]) on every rank.

    counts: list of per-rank row counts (needed when they differ: ranks pad to the maximum and the
    padding is dropped after the gather).  Equal counts use all_gather_into_tensor directly.
    """
    if not dist.is_initialized():
        return ids, vals
    world = dist.get_world_size(group)
    k = ids.shape[1]
    if counts is None:
        counts = [ids.shape[0]] * world
    mx = max(counts)
    if ids.shape[0] < mx:                                   # pad to a common shape
        pad = mx - ids.shape[0]
        ids = torch.cat([ids, ids.new_zeros((pad, k))]); vals = torch.cat([vals, vals.new_zeros((pad, k))])
    out_i = ids.new_empty((world * mx, k)); out_v = vals.new_empty((world * mx, k))
    dist.all_gather_into_tensor(out_i, ids.contiguous(), group=group)
    dist.all_gather_into_tensor(out_v, vals.contiguous(), group=group)
    if all(c == mx for c in counts):
        return out_i, out_v
    keep = torch.cat([torch.arange(r * mx, r * mx + c, device=ids.device) for r, c in enumerate(counts)])
    return out_i.index_select(0, keep), out_v.index_select(0, keep)


def sharded_score_topk(model, k, total_users=None, candidates=None, group=None, ws=None, out=None):
    """Every rank scores its user shard of `model` (full catalog) and all ranks receive the merged top-k.

    Returns (ids [U,k], scores [U,k]) for all U users in user order, identical on every rank.
    """
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    device = model._compute_device()
    tower = model._tower(device)
    U = tower.n_users if total_users is None else int(total_users)
    start, count = shard_range(U, rank, world)
    ids, vals = tower.score_topk((start, count), k, candidates, None, check_ids=False, ws=ws, out=out)
    if not dist.is_initialized():
        return ids, vals
    counts = [shard_range(U, r, world)[1] for r in range(world)]
    return all_gather_topk(ids, vals, counts, group)
