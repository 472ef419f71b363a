"""reco_tower_score_topk_fused: ONE kernel scores and selects (tower_tile_topk_kernel: the score rows never reach HBM).
Its results must be bit-identical to reco_tower_score_topk (score rows in the workspace + the stand-alone select) and to the
oracle's select over score_matrix.  Covers ragged user blocks and item tiles,
user / candidate lists, every k range, few users (many workgroups share a user block), many users (blocks outnumber the CUs),
chunked workspaces, the MLP-only tower and the Hybrid head over per-call row blocks."""
import numpy as np
import pytest
import torch

from oracle import reco_oracle as O

pytestmark = pytest.mark.gpu


def _model(n_users, n_items, seed, name="NeuMF-end", scale=25.0):
    from recommender_amd.helpers import NCF
    torch.manual_seed(seed)
    m = NCF(n_users, n_items, 32, 3, 0.0, name).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(scale)
    m = m.to("cuda:0")
    return m, m._tower(torch.device("cuda", 0))


def _same(a, b):
    return np.array_equal(a[0].cpu().numpy(), b[0].cpu().numpy()) and \
        np.array_equal(a[1].cpu().numpy().view(np.uint32), b[1].cpu().numpy().view(np.uint32))


def _both(tower, users, k, cands, U, N, ws=None):
    fused = tower.score_topk(users, k, cands, ws=ws, fused=True)
    plain = tower.score_topk(users, k, cands)
    return fused, plain


@pytest.mark.parametrize("name", ["NeuMF-end", "MLP"])
def test_fused_equals_two_kernel_form_and_the_oracle(name):
    m, t = _model(301, 517, 3, name)
    for k in (1, 10, 64, 65, 100, 128):
        fused, plain = _both(t, None, k, None, 301, 517)
        assert _same(fused, plain), (name, k)
    sm = m.score_matrix().cpu().numpy()
    oid, oval = O.topk_rows(sm, 100)
    ids, vals = t.score_topk(None, 100, fused=True)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)


def test_fused_ragged_shapes_lists_and_duplicates():
    m, t = _model(1000, 2000, 4)
    g = torch.Generator().manual_seed(7)
    for U, N, k in [(1, 64, 64), (1, 65, 3), (23, 63, 10), (24, 64, 10), (25, 129, 100), (49, 1999, 128), (7, 128, 128), (333, 100, 100)]:
        users = torch.randint(0, 1000, (U,), generator=g).cuda()            # repeats allowed
        cands = torch.randint(0, 2000, (N,), generator=g).cuda()            # repeated candidates: equal scores, positions decide
        fused, plain = _both(t, users, k, cands, U, N)
        assert _same(fused, plain), (U, N, k)
        ids = fused[0].cpu().numpy()
        assert (ids >= 0).all() and (ids < N).all()
        assert all(len(set(r)) == k for r in ids)                           # k distinct positions per user
    # user range form
    fused = t.score_topk((100, 77), 10, fused=True)
    plain = t.score_topk((100, 77), 10)
    assert _same(fused, plain)


def test_fused_few_users_long_catalog_many_workgroups_share_a_block():
    # 3 users x 40,000 items: ONE user block, 625 item tiles, 256 workgroups -> every workgroup joins that block with a partial
    # list of its own (256 are provided: one per CU), and the merge folds 256 lists per user, most of them empty.  16 users take the request
    # path (the handle's cached item projections); 40 users are two blocks.
    m, t = _model(64, 40000, 5, scale=10.0)
    for U in (3, 16, 17, 40):
        users = torch.arange(U).cuda()
        fused, plain = _both(t, users, 100, None, U, 40000)
        assert _same(fused, plain), U


def test_fused_more_user_blocks_than_workgroups_and_chunked_workspace():
    m, t = _model(9000, 300, 6)
    fused, plain = _both(t, None, 50, None, 9000, 300)          # 375 user blocks for 256 workgroups: the fresh-block counter
    assert _same(fused, plain)
    # a workspace that only admits a fraction of the users at a time
    from recommender_amd import _lib
    L = _lib.lib()
    full = L.reco_tower_topk_fused_workspace_bytes(t._h, 9000, 300, 50)
    assert 0 < full < L.reco_tower_topk_workspace_bytes(9000, 300, 50)
    small = torch.empty(full // 6, dtype=torch.uint8, device="cuda")
    chunked = t.score_topk(None, 50, ws=small, fused=True)
    assert _same(chunked, plain)
    # far too small: refused
    with pytest.raises(MemoryError):
        t.score_topk(None, 50, ws=torch.empty(4096, dtype=torch.uint8, device="cuda"), fused=True)


def test_fused_bounded_workspace_takes_the_largest_feasible_chunk():
    """The number of partial lists a chunk asks for is not monotone in its size (fewer than 128 user blocks -> one list per CU):
    a bounded workspace must still settle on the largest chunk that fits with the MINIMUM list count instead of halving past it
    (ADVICE r2).  Counted through the measurement hook: one tile launch per chunk."""
    import ctypes
    from recommender_amd import _lib
    L = _lib.lib()
    m, t = _model(6000, 2048, 11)
    k, U, N = 10, 6000, 2048
    plain = t.score_topk(None, k)
    ws_bytes = 6 << 20
    small = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    _lib.check(L.reco_profile_enable_kinds(256, 1 << _lib.PROF_TOWER))
    chunked = t.score_topk(None, k, ws=small, fused=True)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * 256)(); n = ctypes.c_int(0)
    _lib.check(L.reco_profile_collect(ms, 256, ctypes.byref(n)))
    _lib.check(L.reco_profile_enable(0))
    assert _same(chunked, plain)
    per_user = 128 * 4 + 4 * k * 8 + 1                       # one projection row + the minimum of four partial lists
    room = ws_bytes - N * 128 * 4 - 8192                     # minus the item projections and the queue / counter blocks
    needed = -(-U * per_user // room)
    assert needed <= n.value <= 2 * needed + 1, (n.value, needed)


def test_fused_out_of_range_ids_are_counted_once():
    m, t = _model(100, 200, 8)
    with pytest.raises(IndexError):
        t.score_topk(torch.tensor([0, 100]).cuda(), 5, fused=True)
    with pytest.raises(IndexError):
        t.score_topk(None, 5, torch.tensor([0, 1, 2, 3, 4, 200]).cuda(), fused=True)
    ids, _ = t.score_topk(torch.tensor([0, 99]).cuda(), 5, torch.arange(200).cuda(), fused=True)
    assert ids.shape == (2, 5)
    with pytest.raises(ValueError):
        t.score_topk(None, 5, exclude=torch.zeros((100, 200), dtype=torch.uint8).cuda(), fused=True)


def test_fused_is_refused_where_there_is_no_fused_kernel():
    from recommender_amd import _lib
    _, gmf = _model(50, 80, 9, "GMF")
    assert _lib.lib().reco_tower_topk_fused_workspace_bytes(gmf._h, 50, 80, 5) == 0
    with pytest.raises(ValueError):
        gmf.score_topk(None, 5, fused=True)


def test_fused_hybrid_head_shape_with_tables():
    """dg = dm = 64, two layers (the Hybrid head): 24 users per block, 768 threads, like the NeuMF shapes.  The latents stand in as tables."""
    import recommender_amd.hybridautorecncf.model as hm
    from recommender_amd import ops
    torch.manual_seed(11)
    for rating_output in (True, False):
        head = hm.NCF(64, (128, 64), 0.0, rating_output=rating_output).eval().cuda()
        lin = [m for m in head.mlp if isinstance(m, torch.nn.Linear)]
        for U, N, k in [(157, 1201, 20), (5, 3000, 100), (16, 64, 64)]:
            zu = torch.randn(U, 64, device="cuda"); zi = torch.randn(N, 64, device="cuda")
            t = ops.Tower(user_gmf=zu, item_gmf=zi, user_mlp=zu, item_mlp=zi,
                          layers=[(l.weight.detach(), l.bias.detach()) for l in lin], w_predict=head.output.weight.detach(),
                          b_predict=head.output.bias.detach(), act="rating" if rating_output else None, lo=1.0, hi=5.0)
            fused, plain = _both(t, None, k, None, U, N)
            assert _same(fused, plain), (rating_output, U, N, k)
            full = head.score_matrix(zu, zi)
            oid, oval = O.topk_rows(full.cpu().numpy(), k)
            assert np.array_equal(fused[0].cpu().numpy(), oid) and np.array_equal(fused[1].cpu().numpy(), oval)


def test_score_topk_takes_the_fused_form_when_the_score_rows_need_several_user_chunks():
    """300 users x 1M items = 1.2 GB of score rows: more than one 1 GiB user chunk, so reco_tower_score_topk itself scores and
    selects in one kernel.  An all-zero mask still takes the two-kernel form (chunks of 240 users): same bits."""
    m, t = _model(512, 1_000_000, 12, scale=10.0)
    users = (100, 300)
    got = t.score_topk(users, 100)
    zeros = torch.zeros((300, 1_000_000), dtype=torch.uint8, device="cuda")
    want = t.score_topk(users, 100, exclude=zeros)
    assert _same(got, want)
    ids = got[0].cpu().numpy()
    assert (ids >= 0).all() and (ids < 1_000_000).all() and all(len(set(r)) == 100 for r in ids[:16])
