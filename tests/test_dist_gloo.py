"""N > 1 path on CPU: world_size-2 gloo ranks exercise the user sharding and the top-k all-gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from recommender_amd.dist import TopkGather, all_gather_topk, packed_topk_buffers, shard_range


def test_shard_range_covers_everything_contiguously():
    for total in (0, 1, 7, 6038, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _worker(rank, world, port, total, k, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)                       # every rank derives the same global answer ...
        ids = rng.integers(0, 3533, (total, k)).astype(np.int32)
        vals = rng.standard_normal((total, k)).astype(np.float32)
        start, count = shard_range(total, rank, world)       # ... but holds only its own user range
        counts = [shard_range(total, r, world)[1] for r in range(world)]
        gi, gv = all_gather_topk(torch.from_numpy(ids[start:start + count]), torch.from_numpy(vals[start:start + count]), counts)
        ok = np.array_equal(gi.numpy(), ids) and np.array_equal(gv.numpy(), vals)
        q.put((rank, bool(ok), tuple(gi.shape)))
    finally:
        dist.destroy_process_group()


class _StubTower:
    """Stands in for ops.Tower on the CPU: 'scores' a user range by writing a function of (user, step) into `out`."""
    def __init__(self, n_users, n_items):
        self.n_users, self.n_items, self.calls = n_users, n_items, 0

    def topk_workspace(self, U, N, k):
        return None

    def score_topk(self, users, k, cands, exclude, check_ids=True, ws=None, out=None):
        start, count = users
        u = torch.arange(start, start + count).view(-1, 1)
        out[0].copy_((u * 7 + torch.arange(k).view(1, -1) + self.calls).to(torch.int32))
        out[1].copy_((u * 0.5 - torch.arange(k).view(1, -1) - self.calls).to(torch.float32))
        self.calls += 1
        return out


class _StubModel:
    def __init__(self, tower):
        self.tower = tower

    def _compute_device(self):
        return torch.device("cpu")

    def _tower(self, device):
        return self.tower


def _pipe_worker(rank, world, port, total, k, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = TopkGather(_StubModel(_StubTower(total, 50)), k, total)
        got = [pipe.step() for _ in range(4)] + [pipe.flush()]
        ok = got[0] is None
        u = torch.arange(total).view(-1, 1)
        for step, (ids, vals) in enumerate(got[1:]):
            ok = ok and torch.equal(ids, (u * 7 + torch.arange(k).view(1, -1) + step).to(torch.int32))
            ok = ok and torch.equal(vals, (u * 0.5 - torch.arange(k).view(1, -1) - step).to(torch.float32))
        # three buffers in flight: step i hands out step i - 2, flush() the last one; the stub's step counter keeps running
        deep = TopkGather(_StubModel(_StubTower(total, 50)), k, total, depth=3)
        got3 = [deep.step() for _ in range(5)]
        ok = ok and got3[0] is None and got3[1] is None and deep.depth == 3
        for step, (ids, vals) in enumerate(got3[2:]):
            ok = ok and torch.equal(ids, (u * 7 + torch.arange(k).view(1, -1) + step).to(torch.int32))
        ids, vals = deep.flush()
        ok = ok and torch.equal(ids, (u * 7 + torch.arange(k).view(1, -1) + 4).to(torch.int32)) and deep.flush() is None
        try:
            TopkGather(_StubModel(_StubTower(total, 50)), k, total, depth=1)
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok), tuple(got[-1][0].shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(total, k, worker=_worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, total, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_all_gather_topk_equal_shards():
    for rank, ok, shape in _run(12, 10):
        assert ok and shape == (12, 10)


def test_all_gather_topk_ragged_shards():
    for rank, ok, shape in _run(13, 100):      # 7 + 6 users: padded gather, padding dropped
        assert ok and shape == (13, 100)


def test_pipelined_gather_two_buffers_in_flight_equal_and_ragged_shards():
    for total in (12, 13):
        for rank, ok, shape in _run(total, 10, _pipe_worker):
            assert ok and shape == (total, 10)


def test_packed_buffers_are_sent_in_place():
    from recommender_amd.dist import _is_packed, _pack
    buf, ids, vals = packed_topk_buffers(5, 3, "cpu")
    ids.copy_(torch.arange(15, dtype=torch.int32).view(5, 3)); vals.copy_(torch.arange(15, dtype=torch.float32).view(5, 3) * 0.25)
    assert _is_packed(ids, vals) and _pack(ids, vals, 5).data_ptr() == buf.data_ptr()
    assert torch.equal(_pack(ids, vals, 5), buf)
    assert not _is_packed(ids.clone(), vals) and _pack(ids[:4], vals[:4], 5).shape == (2, 5, 3)


class _ReservingTower(_StubTower):
    reserved_cus = 0

    def set_reserved_cus(self, cus):
        self.reserved_cus = int(cus)


def _reserve_worker(rank, world, port, total, k, q):
    """The CU reservation reaches the tower only when there are peers to gather from; results do not depend on it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RECO_RESERVE_CUS="6")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        by_arg, by_env = _ReservingTower(total, 50), _ReservingTower(total, 50)
        pipe = TopkGather(_StubModel(by_arg), k, total, reserve_cus=4)
        pipe_env = TopkGather(_StubModel(by_env), k, total)
        ok = by_arg.reserved_cus == 4 and pipe.reserve_cus == 4 and by_env.reserved_cus == 6 and pipe_env.reserve_cus == 6
        try:
            TopkGather(_StubModel(_ReservingTower(total, 50)), k, total, reserve_cus=500)
            ok = False
        except ValueError:
            pass
        got = [pipe.step() for _ in range(3)] + [pipe.flush()]
        u = torch.arange(total).view(-1, 1)
        for step, (ids, vals) in enumerate(got[1:]):
            ok = ok and torch.equal(ids, (u * 7 + torch.arange(k).view(1, -1) + step).to(torch.int32))
        q.put((rank, bool(ok), tuple(got[-1][0].shape)))
    finally:
        dist.destroy_process_group()


def _single_rank_worker(rank, world, port, total, k, q):
    """A process group of ONE rank (the driver's torchrun form at N = 1): no collective is issued, nothing is reserved."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RECO_RESERVE_CUS="6")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **kw: calls.append(1) or real(*a, **kw)
        tower = _ReservingTower(total, 50)
        pipe = TopkGather(_StubModel(tower), k, total)
        got = [pipe.step() for _ in range(3)] + [pipe.flush()]
        ok = got[0] is None and not pipe.on and tower.reserved_cus == 0 and tuple(got[-1][0].shape) == (total, k)
        ids = torch.arange(total * k, dtype=torch.int32).view(total, k)
        gi, gv = all_gather_topk(ids, ids.float())
        ok = ok and gi is ids and not calls
        # forced (the rehearsal switch of the RCCL leg): the same one-rank group now issues the collective, results unchanged
        gi, gv = all_gather_topk(ids, ids.float(), force_collective=True)
        ok = ok and len(calls) == 1 and gi is not ids and torch.equal(gi, ids) and torch.equal(gv, ids.float())
        forced = TopkGather(_StubModel(_ReservingTower(total, 50)), k, total, force_collective=True)
        got = [forced.step() for _ in range(3)] + [forced.flush()]
        ok = ok and forced.on and len(calls) == 4 and tuple(got[-1][0].shape) == (total, k)
        q.put((rank, bool(ok), tuple(got[-1][0].shape)))
    finally:
        dist.destroy_process_group()


def test_cu_reservation_argument_and_environment_reach_the_tower():
    for rank, ok, shape in _run(13, 10, _reserve_worker):
        assert ok and shape == (13, 10)


def test_single_rank_group_issues_no_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(0, 1, _free_port(), 9, 10, q))
    p.start()
    rank, ok, shape = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0 and ok and shape == (9, 10)
