"""N > 1 path on CPU: world_size-2 gloo ranks exercise the user sharding and the top-k all-gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from recommender_amd.dist import all_gather_topk, shard_range


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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(total, k):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, k, q)) for r in range(2)]
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
