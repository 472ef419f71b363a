"""BASELINE.json configurations at full size: HIP path vs the oracle on samples, plus size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import reco_oracle as O
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu


def _neumf(n_users, n_items, seed=0, name="NeuMF-end"):
    from recommender_amd.helpers import NCF
    torch.manual_seed(seed)
    m = NCF(n_users, n_items, 32, 3, 0.0, name).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(25.0)
            elif n.endswith("bias"):
                p.copy_(torch.empty_like(p).uniform_(-0.2, 0.2, generator=g))
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    return m.cuda(), sd


def test_cfg2_full_catalog_scoring_and_top10(tower_precision):
    m, sd = _neumf(6038, 3533)
    S = m.score_matrix()                                   # 21,332,254 pairs
    assert S.shape == (6038, 3533)
    rng = np.random.default_rng(2)
    users = np.sort(rng.choice(6038, 48, replace=False))
    ref32 = O.ncf_score_all(sd, users)
    ref64 = O.ncf_score_all(sd, users, np.float64)
    got = S[torch.from_numpy(users).cuda()].cpu().numpy()
    assert_close(got, ref32, what="cfg2 rows vs oracle fp32")
    assert_close(got, ref64, what="cfg2 rows vs oracle fp64")
    ids, vals = m.score_topk(None, 10)
    # the fused path equals top-k of the materialised matrix bit for bit, for every user
    oid, oval = O.topk_rows(S.cpu().numpy(), 10)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)
    # and the oracle's own top-10 on the sampled users, ids equal wherever the oracle's scores are not within tolerance
    rid, rval = O.topk_rows(ref32, 10)
    assert_close(vals.cpu().numpy()[users], rval, what="cfg2 top-10 scores")
    diff = ids.cpu().numpy()[users] != rid
    if diff.any():
        r, c = np.where(diff)
        gap = np.abs(np.take_along_axis(ref32, ids.cpu().numpy()[users].astype(np.int64), 1)[r, c] - rval[r, c])
        assert (gap <= 1e-5 * np.abs(rval[r, c]) + 1e-6).all()
    # determinism: a second run is bitwise identical
    ids2, vals2 = m.score_topk(None, 10)
    assert torch.equal(ids, ids2) and torch.equal(vals, vals2)
    # user ranges and explicit user lists agree with the all-users call
    a, b = m._tower(S.device).score_topk((1000, 77), 10)
    assert torch.equal(a, ids[1000:1077]) and torch.equal(b, vals[1000:1077])
    ul = torch.tensor([6037, 0, 3000, 3000], device="cuda")
    a, b = m.score_topk(ul, 10)
    assert torch.equal(a, ids[ul]) and torch.equal(b, vals[ul])


def test_cfg1_full_protocol_size_pair_list(tower_precision):
    """115,056 test cases x 100 candidates = 11,505,600 pairs through forward(user, item)."""
    from recommender_amd import ops
    from recommender_amd.helpers.ranking_metrics import hit_ndcg_batch
    m, sd = _neumf(6038, 3533, seed=3)
    rng = np.random.default_rng(7)
    cases = 115056
    users = rng.integers(0, 6038, cases)
    items = rng.integers(0, 3533, (cases, 100))
    u = torch.from_numpy(np.repeat(users, 100)).cuda(); it = torch.from_numpy(items.reshape(-1)).cuda()
    s = m(u, it)
    assert s.shape == (cases * 100,)
    # cross-check against the tile kernel (independent code path for gathers/GMF): same pairs looked up in the matrix
    S = m.score_matrix()
    look = S[u, it]
    assert_close(s.cpu().numpy(), look.cpu().numpy(), what="pair kernel vs tile kernel")
    idx = rng.choice(cases * 100, 8192, replace=False)
    ref = O.ncf_forward(sd, u.cpu().numpy()[idx], it.cpu().numpy()[idx])
    ref64 = O.ncf_forward(sd, u.cpu().numpy()[idx], it.cpu().numpy()[idx], np.float64)
    assert_close(s.cpu().numpy()[idx], ref, what="cfg1 pairs vs oracle fp32")
    assert_close(s.cpu().numpy()[idx], ref64, what="cfg1 pairs vs oracle fp64")
    ids, _ = ops.topk_rows(s.view(cases, 100), 10)
    top_items = np.take_along_axis(items, ids.cpu().numpy().astype(np.int64), 1)
    hr, nd = hit_ndcg_batch(top_items, items[:, 0])
    # oracle HR/NDCG on a slice of cases computed from oracle scores
    sl = slice(0, 4096)
    osc = O.ncf_forward(sd, np.repeat(users[sl], 100), items[sl].reshape(-1)).reshape(-1, 100)
    oid, _ = O.topk_rows(osc, 10)
    ohr, ond = O.protocol_metrics(np.take_along_axis(items[sl], oid.astype(np.int64), 1), items[sl, 0])
    assert abs(hr[sl].mean() - ohr.mean()) <= 1.0 / 4096 and abs(nd[sl].mean() - ond.mean()) <= 1e-3


def test_cfg3_autorec_full_reconstruction_and_masked_topk(tower_precision):
    from recommender_amd.autorec.utils.model import AutoRec
    torch.manual_seed(0)
    m = AutoRec(6040, 3706, 500, True).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_uniform_(p)
        m.decoder[0].bias.add_(3.0); m.decoder[0].weight.mul_(3.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(1)
    train = ((rng.random((6040, 3706)) < 0.045) * rng.integers(1, 6, (6040, 3706))).astype(np.float32)
    X = torch.from_numpy(np.ascontiguousarray(train.T))
    ref = T.autorec_forward(sd, X, clamp=False).numpy()            # same ATen ops as the reference, on the host
    m = m.cuda()
    got = m(X.cuda()).cpu().numpy()
    # conditioning: outputs are 500-term sums of O(1) terms; gate = 1e-5 relative + 4e-6 absolute (fp32 sgemm vs fp32 MFMA chain)
    assert_close(got, ref, atol=4e-6, what="cfg3 forward vs torch CPU")
    sd64 = {k: v.numpy() for k, v in sd.items()}
    rows = rng.choice(3706, 16, replace=False)
    ref64 = O.autorec_forward(sd64, train.T[rows], np.float64)
    assert_close(got[rows], ref64, atol=4e-6, what="cfg3 rows vs oracle fp64")
    ids, vals = m.recommend_topk(torch.from_numpy(train).cuda(), 10)
    # reconstruct_all reads R transposed and swaps the operand roles in the decoder (no transposed copies)
    pred = m.reconstruct_all(torch.from_numpy(train).cuda()).cpu().numpy()
    assert_close(pred, np.clip(got, 1.0, 5.0).T, atol=4e-6, what="cfg3 reconstruct_all vs forward")
    if tower_precision == "f32":            # exact mode: both are the same k-ordered fmaf chains, bit for bit
        assert np.array_equal(pred, np.clip(got, 1.0, 5.0).T)
    oid, oval = O.topk_rows(O.mask_seen(np.ascontiguousarray(pred), train > 0), 10)
    assert np.array_equal(vals.cpu().numpy(), oval)
    assert np.array_equal(ids.cpu().numpy(), oid)                  # clamp plateaus: ties resolved by ascending item id


def test_cfg4_hybrid_full_catalog(tower_precision):
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    torch.manual_seed(0)
    m = HybridAutoRecNCF(6040, 3706, 64, [128, 64], rating_output=False).eval()
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(4)
    R = (rng.random((6040, 3706)) < 0.045).astype(np.float32)
    m = m.cuda()
    S = m.score_all(torch.from_numpy(R).cuda())                    # 22,384,240 pairs
    assert S.shape == (6040, 3706)
    zu = O.hybrid_encode(sd, "user_autorec", R); zi = O.hybrid_encode(sd, "item_autorec", np.ascontiguousarray(R.T))
    users = rng.choice(6040, 24, replace=False)
    uu, ii = np.meshgrid(users, np.arange(3706), indexing="ij")
    ref = O.hybrid_head(sd, zu[uu.reshape(-1)], zi[ii.reshape(-1)], rating=False).reshape(24, 3706)
    ref64 = O.hybrid_head(sd, zu[uu.reshape(-1)].astype(np.float64), zi[ii.reshape(-1)].astype(np.float64), np.float64,
                          rating=False).reshape(24, 3706)
    got = S[torch.from_numpy(users).cuda()].cpu().numpy()
    assert_close(got, ref, atol=2e-6, what="cfg4 rows vs oracle fp32")
    assert_close(got, ref64, atol=2e-6, what="cfg4 rows vs oracle fp64 head on fp32 latents")
    ids, vals = m.score_all(torch.from_numpy(R).cuda(), k=10, exclude=torch.from_numpy(R > 0).cuda())
    oid, oval = O.topk_rows(O.mask_seen(S.cpu().numpy(), R > 0), 10)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)


def test_cfg5_slice_million_item_catalog_top100_user_chunking(tower_precision):
    """Synthetic-scale slice: 96 users x 1,000,000 items, top-100; exercises user chunking and multi-wave row splits."""
    m, sd = _neumf(4096, 1_000_000, seed=5)
    users = torch.arange(1000, 1096, device="cuda")
    tower = m._tower(users.device)
    ids, vals = tower.score_topk(users, 100)
    # a workspace small enough to force chunks of 32 users gives the same answer
    small = torch.empty(int(1_000_000 * 128 * 4 + 40 * (1_000_000 * 4 + 128 * 4 + 800) + 8192 * 800 + 4096), dtype=torch.uint8, device="cuda")
    ids2, vals2 = tower.score_topk(users, 100, ws=small)
    assert torch.equal(ids, ids2) and torch.equal(vals, vals2)
    for r in (0, 95):
        ref = O.ncf_score_all(sd, np.array([1000 + r]), chunk=1)
        oid, oval = O.topk_rows(ref, 100)
        assert_close(vals[r].cpu().numpy(), oval[0], what="cfg5 top-100 scores")
        same = (ids[r].cpu().numpy() == oid[0])
        if not same.all():
            gap = np.abs(ref[0][ids[r].cpu().numpy()[~same]] - oval[0][~same])
            assert (gap <= 1e-5 * np.abs(oval[0][~same]) + 1e-6).all()
    # property: every returned list is sorted descending and ids are unique
    v = vals.cpu().numpy(); i = ids.cpu().numpy()
    assert (np.diff(v, axis=1) <= 0).all() and all(len(set(row)) == 100 for row in i)


def test_split_mode_deviation_is_at_fp32_rounding_level():
    """RECO_PREC_SPLIT_BF16X3 against the fp64 oracle at cfg2 size: its error is of the size of the exact-fp32 kernel's own
    rounding error (both are fp32-accumulated dot products; the split drops only terms below 2^-24 relative)."""
    from recommender_amd import ops
    m, sd = _neumf(6038, 3533)
    rng = np.random.default_rng(5)
    users = np.sort(rng.choice(6038, 64, replace=False))
    ref64 = O.ncf_score_all(sd, users, np.float64)
    ut = torch.from_numpy(users).cuda()
    err = {}
    for mode in ("f32", "bf16x3"):
        old = ops.set_tower_precision(mode)
        try:
            got = m.score_matrix(ut).cpu().numpy().astype(np.float64)
        finally:
            ops.set_tower_precision(old)
        err[mode] = np.abs(got - ref64).max()
    scale = np.abs(ref64).max()
    print("max |score| %.3f; max abs deviation from fp64: exact fp32 MFMA %.3e, split bf16x3 %.3e" % (scale, err["f32"], err["bf16x3"]))
    assert err["f32"] <= 1e-6 * max(scale, 1.0)
    assert err["bf16x3"] <= 2.0 * err["f32"] + 1e-7 * max(scale, 1.0)
