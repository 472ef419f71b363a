"""N > 1 path with the REAL kernels: two gloo ranks share the box's one GPU, each scores its user shard with the HIP tower and the
lists are gathered (host-staged, recommender_amd/dist.py); every rank checks the merged result against the full single-process
answer it computes itself.  RCCL cannot be used with two ranks here (it refuses two ranks on one device); the collective is the only part that
differs from the driver's multi-GPU run, and tests/test_dist_gloo.py covers its packing / padding on the CPU.
The RCCL leg itself (device tensors, the collective enqueued on RCCL's stream behind the top-k kernel, work.wait() as a stream
dependency, two buffers in flight) runs in a ONE-rank "nccl" group with the collective forced (force_collective)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, users, items, k, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RECO_GATHER_TIMEOUT_S="30")
        import datetime
        import torch.distributed as dist
        from recommender_amd.dist import TopkGather, shard_range, sharded_score_topk
        from recommender_amd.helpers import NCF
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        try:
            torch.manual_seed(3)                                   # the same model on every rank
            model = NCF(users, items, 32, 3, 0.0, "NeuMF-end").eval()
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if "embed" in n:
                        p.mul_(25.0)
            model = model.cuda()
            tower = model._tower(torch.device("cuda", 0))
            want_ids, want_vals = tower.score_topk(None, k)        # all users, one process
            got_ids, got_vals = sharded_score_topk(model, k, users)
            ok = torch.equal(got_ids, want_ids) and torch.equal(got_vals.view(torch.int32), want_vals.view(torch.int32))
            for reserve in (0, 8):                                 # pipelined form, without and with CUs kept free for a collective
                pipe = TopkGather(model, k, users, reserve_cus=reserve)
                res = [pipe.step() for _ in range(5)] + [pipe.flush()]
                ok = ok and res[0] is None and pipe.reserve_cus == reserve
                for ids, vals in res[1:]:
                    ok = ok and torch.equal(ids, want_ids) and torch.equal(vals.view(torch.int32), want_vals.view(torch.int32))
            start, count = shard_range(users, rank, world)
            q.put((rank, bool(ok), tuple(got_ids.shape), count))
        finally:
            dist.destroy_process_group()
    except Exception as exc:                                       # the parent must not wait for a rank that died
        q.put((rank, False, repr(exc), -1))


@pytest.mark.parametrize("users,items,k", [(1001, 777, 10), (640, 3533, 100)])
def test_two_ranks_real_kernels_match_the_single_process_answer(users, items, k):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_rank, args=(r, 2, port, users, items, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=150) for _ in procs)
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert [r[0] for r in res] == [0, 1]
    for rank, ok, shape, count in res:
        assert ok, (rank, shape)
        assert shape == (users, k)
    assert sum(r[3] for r in res) == users and abs(res[0][3] - res[1][3]) <= 1


def _rccl_rank(port, users, items, k, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import datetime
        import torch.distributed as dist
        from recommender_amd.dist import TopkGather, all_gather_topk
        from recommender_amd.helpers import NCF
        device = torch.device("cuda", 0)
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=device)
        try:
            torch.manual_seed(5)
            model = NCF(users, items, 32, 3, 0.0, "NeuMF-end").eval().cuda()
            tower = model._tower(device)
            want_ids, want_vals = tower.score_topk(None, k)
            same = lambda ids, vals: (torch.equal(ids, want_ids) and
                                      torch.equal(vals.view(torch.int32), want_vals.view(torch.int32)))
            ids, vals = all_gather_topk(want_ids, want_vals, force_collective=True)      # blocking form, through RCCL
            ok = same(ids, vals) and ids.data_ptr() != want_ids.data_ptr()
            plain = TopkGather(model, k, users)                                          # one rank, not forced: no collective
            ok = ok and not plain.on
            for reserve in (0, 8):
                pipe = TopkGather(model, k, users, reserve_cus=reserve, force_collective=True)
                ok = ok and pipe.on and pipe.ordered and pipe.reserve_cus == reserve
                res = [pipe.step() for _ in range(6)] + [pipe.flush()]
                ok = ok and res[0] is None
                for ids, vals in res[1:]:
                    ok = ok and same(ids, vals)
            torch.cuda.synchronize()
            q.put((bool(ok), "ok"))
        finally:
            dist.destroy_process_group()
    except Exception as exc:
        q.put((False, repr(exc)))


def test_rccl_leg_in_a_one_rank_group_with_the_collective_forced():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = ctx.Process(target=_rccl_rank, args=(port, 1500, 900, 10, q))
    p.start()
    try:
        ok, msg = q.get(timeout=150)
    finally:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert ok, msg
