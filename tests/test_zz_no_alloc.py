"""reco_tower_score_topk_noalloc (include/reco_hip.h): the allocation-free form of reco_tower_score_topk, whose call
workspace is a static device array inside the library.  Called through the C ABI and compared with the existing entry
point bit for bit (same kernels, same arithmetic; only the user chunking may differ, and no result depends on it), on two
calls in a row, the second with a smaller input -- the shared array still holds the first call's bytes then."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tower(n_users, n_items, seed):
    from recommender_amd.helpers import NCF
    torch.manual_seed(seed)
    m = NCF(n_users, n_items, 32, 3, 0.0, "NeuMF-end").eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(25.0)
    m = m.to("cuda:0")
    return m, m._tower(torch.device("cuda", 0))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _both(tower, users, user_base, U, cands, N, k, exclude):
    """(ids, vals) from reco_tower_score_topk with a caller workspace and from the no-alloc form, raw C ABI."""
    from recommender_amd import _lib
    L = _lib.lib()
    dev = tower.device
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    res = []
    for noalloc in (False, True):
        ids = torch.full((U, k), -7, dtype=torch.int32, device=dev)
        vals = torch.full((U, k), float("nan"), dtype=torch.float32, device=dev)
        oob = torch.zeros(1, dtype=torch.int32, device=dev)
        ld_ex = N if exclude is not None else 0
        if noalloc:
            before = torch.cuda.memory_allocated(dev)
            _lib.check(L.reco_tower_score_topk_noalloc(tower._h, _p(users), user_base, U, _p(cands), N, k, _p(exclude), ld_ex,
                                                       _p(ids), _p(vals), _p(oob), st))
            assert torch.cuda.memory_allocated(dev) == before
        else:
            nbytes = L.reco_tower_topk_workspace_bytes(U, N, k)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            _lib.check(L.reco_tower_score_topk(tower._h, _p(users), user_base, U, _p(cands), N, k, _p(exclude), ld_ex,
                                               _p(ids), _p(vals), _p(oob), _p(ws), nbytes, st))
        torch.cuda.synchronize(dev)
        assert int(oob.item()) == 0
        res.append((ids.cpu().numpy(), vals.cpu().numpy()))
    return res


def test_noalloc_topk_equals_the_workspace_form_on_two_calls_in_a_row_second_smaller():
    from recommender_amd import _lib
    L = _lib.lib()
    assert L.reco_noalloc_scratch_bytes() == 32 << 20
    model, tower = _tower(5000, 3533, 0)
    dev = tower.device
    g = torch.Generator(device="cpu").manual_seed(1)

    # call 1: every user x the whole catalog, k = 100, with a seen-item mask: several user chunks inside 32 MiB
    U1, N1, k1 = 5000, 3533, 100
    ex1 = (torch.rand((U1, N1), generator=g) < 0.05).to(torch.uint8).to(dev)
    (ids_a, vals_a), (ids_b, vals_b) = _both(tower, None, 0, U1, None, N1, k1, ex1)
    assert np.array_equal(ids_a, ids_b) and np.array_equal(vals_a.view(np.uint32), vals_b.view(np.uint32))
    assert (ids_b >= 0).all() and (ids_b < N1).all()

    # call 2, right behind it and smaller in every extent: a user list, a candidate list, k = 10, no mask
    U2, N2, k2 = 37, 500, 10
    users = torch.randint(0, 5000, (U2,), generator=g).to(dev)
    cands = torch.randperm(3533, generator=g)[:N2].to(dev)
    (ids_c, vals_c), (ids_d, vals_d) = _both(tower, users, 0, U2, cands, N2, k2, None)
    assert np.array_equal(ids_c, ids_d) and np.array_equal(vals_c.view(np.uint32), vals_d.view(np.uint32))

    # and against the oracle, so that "equal" is not "equally wrong"
    from oracle import reco_oracle as O
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    full = O.ncf_score_all(sd, users.cpu().numpy())[:, cands.cpu().numpy()]
    ref_ids, ref_vals = O.topk_rows(full, k2)
    assert np.allclose(vals_d, ref_vals, rtol=1e-5, atol=1e-6)
    assert (ids_d == ref_ids).mean() > 0.99


def test_noalloc_topk_in_the_fused_form_when_the_score_rows_need_several_user_chunks():
    """6,100 users x 45,000 items = 1.1 GB of score rows: reco_tower_score_topk (and the no-alloc form with it) scores and selects
    in one kernel per user chunk; inside 32 MiB of scratch the chunks are a few hundred users, in the caller's workspace one."""
    _, tower = _tower(6100, 45000, 2)
    (ids_a, vals_a), (ids_b, vals_b) = _both(tower, None, 0, 6100, None, 45000, 20, None)
    assert np.array_equal(ids_a, ids_b) and np.array_equal(vals_a.view(np.uint32), vals_b.view(np.uint32))
    assert (ids_b >= 0).all() and (ids_b < 45000).all()
    # a second, smaller call right behind it (the scratch still holds the first call's lists and counters)
    users = torch.arange(100, 300, device=tower.device)
    (ids_c, vals_c), (ids_d, vals_d) = _both(tower, users, 0, 200, None, 45000, 5, None)
    assert np.array_equal(ids_c, ids_d) and np.array_equal(vals_c.view(np.uint32), vals_d.view(np.uint32))
    assert np.array_equal(ids_d, ids_b[100:300, :5])


def test_noalloc_topk_random_request_sweep():
    """Seeded sweep (RECO_FUZZ_MULT / RECO_FUZZ_SEED scale it, tools/fuzz_run.sh): random table sizes, user ranges and lists,
    candidate lists with repeats, masks, every k, back to back on the one shared scratch array -- always the workspace form's bits."""
    import os
    mult = max(1, int(os.environ.get("RECO_FUZZ_MULT", "1"))); seed = int(os.environ.get("RECO_FUZZ_SEED", "0"))
    rng = np.random.default_rng(4711 + seed)
    for case in range(6 * mult):
        n_users = int(rng.integers(1, 4000)); n_items = int(rng.choice([rng.integers(10, 400), rng.integers(400, 30000)]))
        _, tower = _tower(n_users, n_items, 100 + case)
        dev = tower.device
        for call in range(3):                                  # several calls per model, sizes going up and down
            k = int(rng.integers(1, min(n_items, 128) + 1))
            cands = torch.from_numpy(rng.integers(0, n_items, int(rng.integers(k, n_items + 50)))).to(dev) if call == 1 else None
            N = cands.numel() if cands is not None else n_items
            if call == 2:
                users = torch.from_numpy(rng.integers(0, n_users, int(rng.integers(1, 3000)))).to(dev); base = 0; U = users.numel()
            else:
                users = None; base = int(rng.integers(0, n_users)); U = int(rng.integers(1, n_users - base + 1))
            ex = (torch.from_numpy(rng.random((U, N)) < 0.1)).to(torch.uint8).to(dev) if (case + call) % 2 and U * N < 30_000_000 else None
            (ids_a, vals_a), (ids_b, vals_b) = _both(tower, users, base, U, cands, N, k, ex)
            assert np.array_equal(ids_a, ids_b) and np.array_equal(vals_a.view(np.uint32), vals_b.view(np.uint32)), (case, call, n_users, n_items, U, N, k)


def test_noalloc_topk_rejects_what_does_not_fit_and_launches_nothing():
    from recommender_amd import _lib
    L = _lib.lib()
    _, tower = _tower(64, 2000, 1)
    dev = tower.device
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    N = 200_000                                            # candidate list far beyond the static scratch (ids repeat)
    cands = (torch.arange(N, device=dev) % 2000).to(torch.int64)
    ids = torch.full((4, 10), -7, dtype=torch.int32, device=dev)
    vals = torch.zeros((4, 10), dtype=torch.float32, device=dev)
    assert L.reco_tower_topk_workspace_bytes(1, N, 10) > L.reco_noalloc_scratch_bytes()
    rc = L.reco_tower_score_topk_noalloc(tower._h, None, 0, 4, _p(cands), N, 10, None, 0, _p(ids), _p(vals), None, st)
    assert rc == _lib.EINVAL and b"static scratch" in L.reco_last_error()
    torch.cuda.synchronize(dev)
    assert (ids.cpu().numpy() == -7).all()
    # bad k / null outputs are refused like the workspace form's
    assert L.reco_tower_score_topk_noalloc(tower._h, None, 0, 4, None, 2000, 0, None, 0, _p(ids), _p(vals), None, st) == _lib.EINVAL
    assert L.reco_tower_score_topk_noalloc(tower._h, None, 0, 4, None, 2000, 10, None, 0, None, None, None, st) == _lib.EINVAL
    assert L.reco_tower_score_topk_noalloc(None, None, 0, 4, None, 2000, 10, None, 0, _p(ids), _p(vals), None, st) == _lib.EINVAL


def test_noalloc_topk_inside_a_captured_hip_graph():
    """The callers the entry point exists for: under stream capture nothing may allocate.  One call is captured into a HIP
    graph (torch.cuda.graph on a side stream), replayed twice on new user lists, and compared bit for bit with the plain
    workspace form on the same inputs."""
    from recommender_amd import _lib
    L = _lib.lib()
    _, tower = _tower(700, 2000, 3)
    dev = tower.device
    U, N, k = 300, 2000, 20
    g = torch.Generator(device="cpu").manual_seed(9)
    users = torch.randint(0, 700, (U,), generator=g).to(dev)
    ids = torch.full((U, k), -7, dtype=torch.int32, device=dev)
    vals = torch.zeros((U, k), dtype=torch.float32, device=dev)

    def call(stream):
        return L.reco_tower_score_topk_noalloc(tower._h, _p(users), 0, U, None, N, k, None, 0, _p(ids), _p(vals), None,
                                               ctypes.c_void_p(stream.cuda_stream))

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        _lib.check(call(side))                      # warm-up outside the capture: first launches set kernel attributes
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = call(torch.cuda.current_stream(dev))
    assert rc == 0, L.reco_last_error()
    for trial in range(2):
        users.copy_(torch.randint(0, 700, (U,), generator=g))
        ids.fill_(-7)
        torch.cuda.synchronize(dev)
        graph.replay()
        torch.cuda.synchronize(dev)
        want_ids, want_vals = tower.score_topk(users, k)
        assert np.array_equal(ids.cpu().numpy(), want_ids.cpu().numpy()), trial
        assert np.array_equal(vals.cpu().numpy().view(np.uint32), want_vals.cpu().numpy().view(np.uint32)), trial
