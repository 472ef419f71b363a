"""HIP path vs golden fixtures (reference outputs) and vs the oracle, through the module interfaces and the C ABI."""
import numpy as np
import pytest
import torch

import os

from conftest import assert_close, golden_sd
from oracle import reco_oracle as O

# the seeded sweeps below run FUZZ_MULT times as many cases from FUZZ_SEED + their own seed (default: the committed cases);
# RECO_FUZZ_MULT=25 RECO_FUZZ_SEED=<n> is the long hunt of tools/fuzz_run.sh
FUZZ_MULT = max(1, int(os.environ.get("RECO_FUZZ_MULT", "1")))
FUZZ_SEED = int(os.environ.get("RECO_FUZZ_SEED", "0"))

pytestmark = pytest.mark.gpu


def _ncf_from_sd(sd, cls=None, **kw):
    from recommender_amd.helpers import NCF
    has_g = "embed_user_GMF.weight" in sd; has_m = "embed_user_MLP.weight" in sd
    name = "NeuMF-end" if has_g and has_m else ("GMF" if has_g else "MLP")
    ref = sd["embed_user_GMF.weight"] if has_g else sd["embed_user_MLP.weight"]
    refi = sd["embed_item_GMF.weight"] if has_g else sd["embed_item_MLP.weight"]
    cls = cls or NCF
    m = cls(ref.shape[0], refi.shape[0], 32, 3, 0.1, name, **kw)
    own = {k: torch.from_numpy(v) for k, v in sd.items() if not k.startswith(("GMF_model.", "MLP_model."))}
    m.load_state_dict(own)
    return m.eval().cuda()


def test_g1_real_gmf_protocol(g1):
    from recommender_amd.helpers.ranking_metrics import hit_ndcg_batch
    sd = {k.replace("__", "."): g1[k] for k in g1.files if k.endswith(("weight", "bias"))}
    m = _ncf_from_sd(sd)
    users = g1["users"].astype(np.int64)
    items = np.concatenate([g1["pos"].astype(np.int64)[:, None], g1["neg"].astype(np.int64)], axis=1)
    uu = np.repeat(users[:, None], 100, 1)
    s = m(torch.from_numpy(uu.reshape(-1)).cuda(), torch.from_numpy(items.reshape(-1)).cuda())
    sc = s.cpu().numpy().reshape(6040, 100)
    assert_close(sc[:256], g1["scores_head"], what="scores vs reference fp32")
    assert_close(sc[:256], g1["scores64_head"], what="scores vs reference fp64")
    assert_close(m(torch.zeros(5, dtype=torch.long).cuda(), torch.arange(5).cuda()).cpu().numpy(),
                 g1["known_answer"], what="known answer")
    from recommender_amd import ops
    ids, _ = ops.topk_rows(s.view(6040, 100), 10)
    top_items = np.take_along_axis(items, ids.cpu().numpy().astype(np.int64), axis=1)
    hr, nd = hit_ndcg_batch(top_items, items[:, 0])
    assert np.array_equal(hr.astype(np.uint8), g1["hr"])
    assert np.allclose(nd, g1["ndcg"], rtol=0, atol=1e-15)
    assert hr.mean() == float(g1["hr_mean"]) and abs(nd.mean() - float(g1["ndcg_mean"])) < 1e-15


@pytest.mark.parametrize("tag", ["end", "pre", "gmf", "mlp"])
def test_g2_small_models_pairs_matrix_topk(g2, tag, tower_precision):
    sd = golden_sd(g2, tag)
    m = _ncf_from_sd(sd)
    ref32 = g2[tag + "__scores"]; ref64 = g2[tag + "__scores64"]
    U, I = ref32.shape
    uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
    s = m(torch.from_numpy(uu.reshape(-1)).cuda(), torch.from_numpy(ii.reshape(-1)).cuda()).cpu().numpy().reshape(U, I)
    assert_close(s, ref32, what=tag + " pairs vs reference fp32")
    assert_close(s, ref64, what=tag + " pairs vs reference fp64")
    sm = m.score_matrix().cpu().numpy()
    assert_close(sm, ref32, what=tag + " matrix vs reference fp32")
    assert_close(sm, ref64, what=tag + " matrix vs reference fp64")
    ids, vals = m.score_topk(None, 10)
    oid, oval = O.topk_rows(sm, 10)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)
    ref_ids = g2[tag + "__top10_argsort"]
    rv = np.take_along_axis(ref32, ref_ids.astype(np.int64), 1)
    assert_close(vals.cpu().numpy(), rv, what=tag + " top-10 scores vs reference")
    # ids identical except where the reference's own neighbouring scores are within tolerance (ties)
    diff = ids.cpu().numpy() != ref_ids
    if diff.any():
        r, c = np.where(diff)
        gap = np.abs(np.take_along_axis(ref32, ids.cpu().numpy().astype(np.int64), 1)[r, c] - rv[r, c])
        assert (gap <= 1e-5 * np.abs(rv[r, c]) + 1e-6).all()


@pytest.mark.parametrize("tag", ["f16l2", "f64l3", "f8l4", "f128l1", "f64gmf"])
def test_g2b_other_tower_shapes_run_on_the_unfused_hip_path(g2b, tag, tower_precision):
    """factor_num / num_layers other than the shipped 32 x 3 (ncf-embedding.ipynb sweeps 8..128)."""
    from recommender_amd.helpers import NCF
    sd = golden_sd(g2b, tag)
    factor, layers = (int(v) for v in g2b[tag + "__cfg"])
    has_g = "embed_user_GMF.weight" in sd; has_m = "embed_user_MLP.weight" in sd
    name = "NeuMF-end" if has_g and has_m else ("GMF" if has_g else "MLP")
    ref = g2b[tag + "__scores"]
    U, I = ref.shape
    m = NCF(U, I, factor, layers, 0.1, name)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.eval().cuda()
    uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
    s = m(torch.from_numpy(uu.reshape(-1)).cuda(), torch.from_numpy(ii.reshape(-1)).cuda()).cpu().numpy().reshape(U, I)
    assert_close(s, ref, what=tag + " pairs"); assert_close(s, g2b[tag + "__scores64"], what=tag + " pairs vs fp64")
    sm = m.score_matrix().cpu().numpy()
    assert_close(sm, ref, what=tag + " matrix")
    ids, vals = m.score_topk(torch.arange(5, 25).cuda(), 7, torch.arange(3, 50).cuda())
    oid, oval = O.topk_rows(sm[5:25, 3:50], 7)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)
    with pytest.raises(IndexError):
        m(torch.tensor([U]).cuda(), torch.tensor([0]).cuda())


def test_g2_subset_users_and_candidates(g2, tower_precision):
    sd = golden_sd(g2, "end")
    m = _ncf_from_sd(sd)
    ref = g2["end__scores"]
    users = torch.tensor([5, 0, 256, 17, 17, 100])
    cands = torch.tensor([130, 3, 7, 64, 65, 66, 0, 99, 12, 13, 14, 15, 16])
    sm = m.score_matrix(users.cuda(), cands.cuda()).cpu().numpy()
    assert_close(sm, ref[users.numpy()][:, cands.numpy()], what="subset matrix")
    ids, vals = m.score_topk(users.cuda(), 5, cands.cuda())
    oid, oval = O.topk_rows(sm, 5)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)


def test_out_of_range_ids_raise_indexerror(g2):
    m = _ncf_from_sd(golden_sd(g2, "end"))
    with pytest.raises(IndexError):
        m(torch.tensor([0, 257]).cuda(), torch.tensor([0, 0]).cuda())
    with pytest.raises(IndexError):
        m(torch.tensor([0, 1]).cuda(), torch.tensor([-1, 0]).cuda())
    # and the handle is usable afterwards
    assert m(torch.tensor([0]).cuda(), torch.tensor([0]).cuda()).shape == (1,)
    assert m(torch.empty(0, dtype=torch.long).cuda(), torch.empty(0, dtype=torch.long).cuda()).shape == (0,)


def test_cpu_resident_model_and_inputs_are_served_from_gpu(g2):
    sd = golden_sd(g2, "end")
    m = _ncf_from_sd(sd).cpu()
    out = m(torch.tensor([1, 2, 3]), torch.tensor([4, 5, 6]))
    assert out.device.type == "cpu"
    assert_close(out.numpy(), g2["end__scores"][[1, 2, 3], [4, 5, 6]], what="cpu in/out")


def test_weight_update_invalidates_packed_cache(g2):
    m = _ncf_from_sd(golden_sd(g2, "end"))
    u = torch.tensor([3]).cuda(); i = torch.tensor([4]).cuda()
    a = m(u, i).item()
    with torch.no_grad():
        m.predict_layer.bias.add_(1.0)
    assert abs(m(u, i).item() - (a + 1.0)) < 1e-5


def test_g2r_rating_variants(g2r, tower_precision):
    from recommender_amd.ncf.rating_model import NCFRating
    for tag in ("neumf", "gmfreal"):
        sd = golden_sd(g2r, tag)
        has_m = "embed_user_MLP.weight" in sd
        ref = g2r[tag + "__scores"]
        m = NCFRating(ref.shape[0], ref.shape[1], 32, 3, 0.2, "NeuMF-end" if has_m else "GMF")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.eval().cuda()
        assert_close(m.score_matrix().cpu().numpy(), ref, what=tag + " rating matrix")
        assert_close(m.score_matrix().cpu().numpy(), g2r[tag + "__scores64"], what=tag + " rating matrix vs fp64")
        uu, ii = np.meshgrid(np.arange(ref.shape[0]), np.arange(ref.shape[1]), indexing="ij")
        s = m(torch.from_numpy(uu.reshape(-1)).cuda(), torch.from_numpy(ii.reshape(-1)).cuda()).cpu().numpy()
        assert_close(s.reshape(ref.shape), ref, what=tag + " rating pairs")


def test_g3_autorec(g3, tower_precision):
    from recommender_amd.autorec.utils.model import AutoRec
    for tag, (U, I, H) in (("a", (193, 89, 64)), ("b", (61, 37, 500))):
        sd = golden_sd(g3, tag)
        m = AutoRec(U, I, H, True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.eval().cuda()
        train = torch.from_numpy(g3[tag + "__train"].astype(np.float32)).cuda()
        X = train.t().contiguous()
        # 500-term dot products of O(1..10) terms that cancel: the reference's own fp32 result is 1.8e-6 / 4.6e-6 away
        # from its fp64 result on these fixtures, so the absolute gate is 2x that (stated, not tuned per element)
        atol = 2.0 * float(np.abs(g3[tag + "__raw"] - g3[tag + "__raw64"]).max())
        assert atol < 1e-5
        assert_close(m(X).cpu().numpy(), g3[tag + "__raw"], atol=atol, what=tag + " forward")
        assert_close(m(X).cpu().numpy(), g3[tag + "__raw64"], atol=atol, what=tag + " forward vs fp64")
        assert_close(m.reconstruct(X).cpu().numpy(), g3[tag + "__clamped"], atol=atol, what=tag + " clamped")
        assert_close(m.reconstruct_all(train).cpu().numpy(), g3[tag + "__clamped"].T, atol=atol, what=tag + " all")
        # single item vector, as the API engine calls it (inference_engine.py:230-233)
        one = m.reconstruct(train[:, 3].unsqueeze(0)).cpu().numpy()
        assert_close(one[0], g3[tag + "__clamped"][3], atol=atol, what=tag + " single")
        ids, vals = m.recommend_topk(train, 10)
        ref_ids = g3[tag + "__top10"].astype(np.int64)
        ref_vals = np.take_along_axis(g3[tag + "__masked"], ref_ids, 1)
        assert_close(vals.cpu().numpy(), ref_vals, atol=atol, what=tag + " top-10 scores")
        # serving path from the cached hidden layer
        m.cache_hidden(train)
        for u in (0, 7, U - 1):
            assert_close(m.predict_user(u).cpu().numpy(), g3[tag + "__clamped"][:, u], atol=atol, what="predict_user")


def test_g3_tie_storm(g3, tower_precision):
    from recommender_amd import ops
    ids, vals = ops.topk_rows(torch.from_numpy(g3["tie__scores"]).cuda(), 100)
    assert np.array_equal(ids.cpu().numpy(), g3["tie__api_top100"])


def test_g4_hybrid(g4, tower_precision):
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    for tag, rating, (U, I), hd in (("rating", True, (83, 47), (48, 40)), ("logit", False, (40, 26), (256, 128))):
        sd = golden_sd(g4, tag)
        m = HybridAutoRecNCF(U, I, 64, [128, 64], hidden_dims=hd, rating_output=rating)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.eval().cuda()
        R = torch.from_numpy(g4[tag + "__R"].astype(np.float32)).cuda()
        assert_close(m.user_autorec.encode(R).cpu().numpy(), g4[tag + "__zu"], atol=2e-6, what=tag + " zu")
        assert_close(m.item_autorec.encode(R.t().contiguous()).cpu().numpy(), g4[tag + "__zi"], atol=2e-6, what=tag + " zi")
        assert_close(m.score_all(R).cpu().numpy(), g4[tag + "__pred"], atol=2e-6, what=tag + " score_all")
        assert_close(m.score_all(R).cpu().numpy(), g4[tag + "__pred64"], atol=2e-6, what=tag + " score_all vs fp64")
        uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
        uid = torch.from_numpy(uu.reshape(-1)).cuda(); iid = torch.from_numpy(ii.reshape(-1)).cuda()
        pred, urec, irec = m(R[uid], R.t().contiguous()[iid], uid, iid)
        assert_close(pred.cpu().numpy().reshape(U, I), g4[tag + "__pred"], atol=2e-6, what=tag + " forward pred")
        assert_close(urec.cpu().numpy()[::I], g4[tag + "__user_recon"], atol=2e-6, what=tag + " user_recon")
        assert_close(irec.cpu().numpy()[:I], g4[tag + "__item_recon"], atol=2e-6, what=tag + " item_recon")
        ids, vals = m.score_all(R, k=5)
        oid, oval = O.topk_rows(m.score_all(R).cpu().numpy(), 5)
        assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)


@pytest.mark.parametrize("rating", [True, False])
def test_hybrid_forward_pair_rows_kernel_with_layer1_inside(g4, rating):
    """HybridAutoRecNCF.forward's head (evaluate_hybridautorecncf_rating.py:167-184) runs as ONE kernel when the GMF and MLP rows are
    the same latents (tower_rows.hip): every batch size class around the 32-pair group and the 8-wave workgroup, against the
    oracle's head in fp32 and fp64, against the two-GEMM + pair-kernel path (forced by handing the GMF rows in as copies), and
    with no workspace at all."""
    import ctypes
    from recommender_amd import _lib
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    tag = "rating" if rating else "logit"
    sd = golden_sd(g4, tag)
    U, I, hd = ((83, 47), (40, 26))[0 if rating else 1] + (((48, 40), (256, 128))[0 if rating else 1],)
    m = HybridAutoRecNCF(U, I, 64, [128, 64], hidden_dims=hd, rating_output=rating)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.eval().cuda()
    dev = torch.device("cuda", 0)
    t = m.ncf._tower(dev)
    rng = np.random.default_rng(3)
    for B in (1, 31, 32, 33, 255, 256, 257, 5000, 70001):
        zu = (rng.standard_normal((B, 64)) * 0.7).astype(np.float32); zi = (rng.standard_normal((B, 64)) * 0.7).astype(np.float32)
        zud, zid = torch.from_numpy(zu).cuda(), torch.from_numpy(zi).cuda()
        got = t.score_rows(zud, zid, zud, zid)
        assert t._L.reco_tower_rows_workspace_bytes_for(t._h, 1, B) == 0
        assert_close(got.cpu().numpy(), O.hybrid_head(sd, zu, zi, np.float32, rating=rating), atol=2e-6, what="rows kernel B=%d" % B)
        assert_close(got.cpu().numpy(), O.hybrid_head(sd, zu, zi, np.float64, rating=rating), atol=2e-6, what="rows kernel B=%d vs fp64" % B)
        unfused = t.score_rows(zud.clone(), zid.clone(), zud, zid)          # other pointers for the GMF rows: projections + pair kernel
        assert_close(got.cpu().numpy(), unfused.cpu().numpy(), atol=2e-6, what="rows kernel vs unfused B=%d" % B)
    # through the module: forward(user_z, item_z)
    pred = m.ncf(zud, zid)
    assert torch.equal(pred, got)
    # straight through the C ABI with a NULL workspace
    out = torch.empty(B, dtype=torch.float32, device=dev)
    rc = t._L.reco_tower_score_rows(t._h, ctypes.c_void_p(zud.data_ptr()), ctypes.c_void_p(zid.data_ptr()), ctypes.c_void_p(zud.data_ptr()),
                                    ctypes.c_void_p(zid.data_ptr()), B, ctypes.c_void_p(out.data_ptr()), None, 0,
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0 and torch.equal(out, got)


def test_topk_properties_long_rows_and_masks():
    from recommender_amd import ops
    rng = np.random.default_rng(11)
    for rows, N, k in ((3, 100000, 100), (130, 3533, 10), (2, 1000003, 100), (1, 64, 64), (5, 129, 128), (1, 1, 1)):
        sc = rng.standard_normal((rows, N)).astype(np.float32)
        sc[:, : min(N, 50)] = np.float32(0.5)                      # ties
        if N > 200:
            sc[0, 100] = np.nan; sc[0, 101] = np.inf; sc[0, 102] = -np.inf; sc[0, 103] = -0.0; sc[0, 104] = 0.0
        ex = rng.random((rows, N)) < 0.3
        t = torch.from_numpy(sc).cuda()
        ids, vals = ops.topk_rows(t, k)
        oid, oval = O.topk_rows(sc, k)
        assert np.array_equal(ids.cpu().numpy(), oid), (rows, N, k)
        assert np.array_equal(vals.cpu().numpy(), oval, equal_nan=True)
        ids, vals = ops.topk_rows(t, k, torch.from_numpy(ex).cuda())
        with np.errstate(invalid="ignore"):
            masked = O.mask_seen(sc, ex)
        oid, oval = O.topk_rows(masked, k)
        assert np.array_equal(ids.cpu().numpy(), oid), ("masked", rows, N, k)
        assert np.array_equal(vals.cpu().numpy(), oval, equal_nan=True)


def test_topk_seen_mask_from_the_rating_matrix_and_user_selection():
    """reco_topk_rows_seen: helper.py:272,313,335 with the mask read off the float rating matrix and users picked by id."""
    from recommender_amd import ops
    rng = np.random.default_rng(12)
    for U, N, k, ld_pad in ((97, 3706, 10, 6), (5, 70001, 100, 0), (33, 131, 128, 1), (2, 1, 1, 0)):
        sc = np.clip(rng.standard_normal((U, N)) * 2 + 3, 1.0, 5.0).astype(np.float32)        # clamp plateaus: ties
        R = ((rng.random((U, N)) < 0.2) * rng.integers(1, 6, (U, N))).astype(np.float32)
        R[rng.random((U, N)) < 0.01] = -1.0                          # not > 0: not seen
        buf = torch.zeros((U, N + ld_pad), dtype=torch.float32, device="cuda")
        buf[:, :N] = torch.from_numpy(sc).cuda()
        Rt = torch.from_numpy(R).cuda()
        masked = O.mask_seen(sc, R > 0)
        oid, oval = O.topk_rows(masked, k)
        ids, vals = ops.topk_rows_seen(buf[:, :N], k, Rt)
        assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval), (U, N, k)
        ids, vals = ops.topk_rows_seen(buf[:, :N], k, None)          # no mask: plain top-k
        pid, pval = O.topk_rows(sc, k)
        assert np.array_equal(ids.cpu().numpy(), pid) and np.array_equal(vals.cpu().numpy(), pval)
        users = rng.integers(0, U, 19)
        ids, vals = ops.topk_rows_seen(buf[:, :N], k, Rt, torch.from_numpy(users).cuda())
        assert np.array_equal(ids.cpu().numpy(), oid[users]) and np.array_equal(vals.cpu().numpy(), oval[users])
        ids, vals = ops.topk_rows_seen(buf[:, :N], k, Rt, torch.zeros(0, dtype=torch.int64))
        assert ids.shape == (0, k)
        with pytest.raises(IndexError):
            ops.topk_rows_seen(buf[:, :N], k, Rt, torch.tensor([0, U]))
        with pytest.raises(ValueError):
            ops.topk_rows_seen(buf[:, :N], k, Rt[:, : max(N - 1, 0)])


def test_sparse_linear_over_csr_input(tower_precision):
    """reco_csr_linear_f32: the encoder layer over the observed entries only equals the dense layer (zeros add nothing)."""
    from recommender_amd import ops
    rng = np.random.default_rng(21)
    for M, K, N, dens, act in ((67, 300, 500, 0.05, "sigmoid"), (5, 6040, 256, 0.5, "relu"), (40, 97, 37, 0.3, None),
                               (3, 1000, 1030, 0.9, "sigmoid"), (9, 11, 2, 0.0, None)):
        X = ((rng.random((M, K)) < dens) * rng.integers(1, 6, (M, K))).astype(np.float32)
        if dens > 0:
            X[0] = rng.integers(1, 6, K)                    # a fully observed row: several segments
            X[M - 1] = 0                                    # an empty row
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); b = rng.standard_normal(N).astype(np.float32)
        Xd = torch.from_numpy(X).cuda(); Wd = torch.from_numpy(W).cuda(); bd = torch.from_numpy(b).cuda()
        csr = ops.CsrMatrix.from_dense(Xd)
        assert csr.nnz == int((X != 0).sum()) and torch.equal(csr.to_dense(), Xd)
        flat = ops.CsrMatrix.from_dense(Xd, k_ranges=1)      # no k ranges: whole rows, cut only at the segment cap
        if dens > 0 and K > ops.CsrMatrix.SEG_LEN:
            assert flat.n_seg > int((X != 0).any(1).sum())   # the full row was cut into segments (an empty row owns none)
            assert flat.seg_len == ops.CsrMatrix.SEG_LEN_SKEWED      # a row longer than one segment: the short cap (balance)
        elif dens > 0:
            assert flat.seg_len == ops.CsrMatrix.SEG_LEN and flat.n_seg == int((X != 0).any(1).sum())     # whole rows
        if dens > 0:
            assert csr.order is not None and (csr.seg_len != flat.seg_len or csr.n_seg >= flat.n_seg)
            assert csr.n_seg == int(((csr.seg_end - csr.seg_begin) > 0).sum()) and int((csr.seg_end - csr.seg_begin).max()) <= csr.seg_len
            o = csr.order.cpu().numpy()
            assert sorted(o[o >= 0].tolist()) == list(range(csr.n_seg))       # every segment placed exactly once
        Wt = ops.transpose(Wd)
        assert torch.equal(Wt, Wd.t().contiguous())
        got = ops.csr_linear(csr, Wt, bd, act).cpu().numpy()
        got_flat = ops.csr_linear(flat, Wt, bd, act).cpu().numpy()
        pre64 = X.astype(np.float64) @ W.astype(np.float64).T + b
        ref64 = {"sigmoid": lambda v: 1 / (1 + np.exp(-v)), "relu": lambda v: np.maximum(v, 0), None: lambda v: v}[act](pre64)
        # gate: 1e-5 relative + 1e-6 absolute, plus what ANY fp32 summation of that many terms is entitled to: a few ulps of the
        # sum of the terms' magnitudes (the fully observed 6,040-entry row sums ~190 in magnitude to reach values of ~0.04; all
        # three activations are 1-Lipschitz, so the bound carries over)
        slack = 2e-7 * (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b))
        dense_out = ops.linear(Xd, Wd, bd, act, precision="f32").cpu().numpy()
        for mine in (got, got_flat):
            for name, other in (("fp64", ref64), ("the dense MFMA layer", dense_out)):
                err = np.abs(mine - other)
                tol = 1e-5 * np.maximum(np.abs(mine), np.abs(other)) + 1e-6 + slack
                assert (err <= tol).all(), "csr_linear vs %s (%d,%d,%d): worst err/tol %.2f" % (name, M, K, N, (err / tol).max())
        sparse_rows = (X != 0).sum(1) < 400                         # rows like the real matrix's: the plain gate, no slack
        if sparse_rows.any():
            assert_close(got[sparse_rows], ref64[sparse_rows], atol=2e-6, what="csr_linear, sparse rows vs fp64")
        # the transposed matrix from the same dense tensor, and from host triplets with a duplicate
        csr_t = ops.CsrMatrix.from_dense(Xd, transpose=True)
        assert torch.equal(csr_t.to_dense(), Xd.t())
        r, c = np.nonzero(X)
        trip = ops.CsrMatrix.from_triplets(np.concatenate([r[:1], r]), np.concatenate([c[:1], c]),
                                           np.concatenate([[9.0], X[r, c]]) if len(r) else X[r, c], (M, K), "cuda")
        assert torch.equal(trip.to_dense(), Xd)
        # determinism
        assert np.array_equal(ops.csr_linear(csr, Wt, bd, act).cpu().numpy(), got)
    with pytest.raises(ValueError):
        ops.csr_linear(csr, Wt[:, :1].contiguous().t().contiguous(), None)


def test_autorec_and_hybrid_with_the_rating_matrix_kept_sparse(g3, g4, tower_precision):
    """attach_ratings: same outputs as the dense path and as the reference-generated fixtures."""
    from recommender_amd.autorec.utils.model import AutoRec
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    for tag, (U, I, H) in (("a", (193, 89, 64)), ("b", (61, 37, 500))):
        m = AutoRec(U, I, H, True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g3, tag).items()})
        m = m.eval().cuda()
        train = torch.from_numpy(g3[tag + "__train"].astype(np.float32)).cuda()
        atol = 2.0 * float(np.abs(g3[tag + "__raw"] - g3[tag + "__raw64"]).max())
        dense_all = m.reconstruct_all(train).cpu().numpy()
        dense_ids, dense_vals = m.recommend_topk(train, 10)
        csr = m.attach_ratings(train)
        assert csr.shape == (I, U) and csr.nnz == int((g3[tag + "__train"] != 0).sum())
        sparse_all = m.reconstruct_all(train).cpu().numpy()
        assert_close(sparse_all, g3[tag + "__clamped"].T, atol=atol, what=tag + " reconstruct_all (sparse encoder) vs golden")
        assert_close(sparse_all, dense_all, atol=atol, what=tag + " sparse vs dense encoder")
        ids, vals = m.recommend_topk(train, 10)
        ref_ids = g3[tag + "__top10"].astype(np.int64)
        assert_close(vals.cpu().numpy(), np.take_along_axis(g3[tag + "__masked"], ref_ids, 1), atol=atol, what=tag + " top-10 (sparse)")
        m.cache_hidden(train)
        assert_close(m.predict_user(7).cpu().numpy(), g3[tag + "__clamped"][:, 7], atol=atol, what="predict_user (sparse H)")
        # a different tensor with the same values is NOT the attached one: dense path, same numbers
        assert_close(m.reconstruct_all(train.clone()).cpu().numpy(), dense_all, atol=0, rtol=0, what="clone takes the dense path")
        # an in-place edit bumps the version: the stale CSR is not used
        train[0, 0] += 1.0
        again = m.reconstruct_all(train).cpu().numpy()
        assert not np.array_equal(again, sparse_all)
    for tag, rating, (U, I), hd in (("rating", True, (83, 47), (48, 40)), ("logit", False, (40, 26), (256, 128))):
        m = HybridAutoRecNCF(U, I, 64, [128, 64], hidden_dims=hd, rating_output=rating)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g4, tag).items()})
        m = m.eval().cuda()
        R = torch.from_numpy(g4[tag + "__R"].astype(np.float32)).cuda()
        cu, ci = m.attach_ratings(R)
        assert cu.shape == (U, I) and ci.shape == (I, U)
        assert_close(m.user_autorec.encode(cu).cpu().numpy(), g4[tag + "__zu"], atol=2e-6, what=tag + " zu (sparse)")
        assert_close(m.item_autorec.encode(ci).cpu().numpy(), g4[tag + "__zi"], atol=2e-6, what=tag + " zi (sparse)")
        assert_close(m.score_all(R).cpu().numpy(), g4[tag + "__pred"], atol=2e-6, what=tag + " score_all (sparse encoders)")


def test_gather_kernels():
    from recommender_amd import ops
    rng = np.random.default_rng(5)
    eu_g = rng.standard_normal((1000, 32)).astype(np.float32); ei_g = rng.standard_normal((777, 32)).astype(np.float32)
    eu_m = rng.standard_normal((1000, 128)).astype(np.float32); ei_m = rng.standard_normal((777, 128)).astype(np.float32)
    u = rng.integers(0, 1000, 4099); i = rng.integers(0, 777, 4099)
    c = lambda a: torch.from_numpy(a).cuda()
    assert np.array_equal(ops.gather_rows(c(eu_m), c(u)).cpu().numpy(), eu_m[u])
    assert np.array_equal(ops.gather_rows(c(ei_g), c(i)).cpu().numpy(), ei_g[i])
    gp, mi = ops.neumf_gather(c(eu_g), c(ei_g), c(eu_m), c(ei_m), c(u), c(i))
    assert np.array_equal(gp.cpu().numpy(), eu_g[u] * ei_g[i])
    assert np.array_equal(mi.cpu().numpy(), np.concatenate([eu_m[u], ei_m[i]], 1))
    with pytest.raises(IndexError):
        ops.gather_rows(c(eu_m), c(np.array([0, 1000])))


def test_linear_shapes_and_epilogues(tower_precision):
    from recommender_amd import ops
    rng = np.random.default_rng(6)
    for M, N, K in ((1, 1, 1), (7, 5, 3), (130, 70, 33), (300, 500, 6040 // 8), (257, 129, 100), (64, 64, 32), (100, 6, 3706 // 2 + 1),
                    (300, 70, 2500), (128, 256, 4099),       # these two take the split-K path
                    # 16-byte aligned rows: LDS-direct staging (gemm.hip DMAT) -- single-buffered from 12 K tiles on, with a
                    # whole and with a ragged last tile, at its 384 threshold, and double-buffered beyond 768
                    (200, 300, 512), (130, 70, 500), (100, 100, 384), (100, 100, 388), (70, 200, 380), (129, 257, 1024), (260, 130, 1028)):
        x = rng.standard_normal((M, K)).astype(np.float32); w = rng.standard_normal((N, K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
        for act, f in ((None, lambda v: v), ("relu", lambda v: np.maximum(v, 0)), ("sigmoid", lambda v: 1 / (1 + np.exp(-v))),
                       ("clamp", lambda v: np.clip(v, -1, 2))):
            y = ops.linear(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), act, -1.0, 2.0)
            err = np.abs(y.cpu().numpy() - f(ref))
            assert (err <= 4e-7 * scale + 1e-6).all(), (M, N, K, act, err.max())
        # transposed-A operand and per-row bias give the same numbers as the plain form on transposed data
        xt = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
        y0 = ops.linear(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda())
        y1 = ops.linear(xt, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), x_transposed=True)
        assert torch.equal(y0, y1), (M, N, K, "x_transposed")
        bm = rng.standard_normal(M).astype(np.float32)
        y2 = ops.linear(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(bm).cuda(), bias_per_row=True)
        err = np.abs(y2.cpu().numpy() - (x.astype(np.float64) @ w.astype(np.float64).T + bm[:, None]))
        assert (err <= 4e-7 * (scale + 1) + 1e-6).all(), (M, N, K, "bias_per_row")


def test_cached_projections_equal_per_call_projections_bit_for_bit():
    """A user RANGE / the whole catalogue read the layer-1 projections the handle computed at creation; explicit id lists are
    projected inside the call (api.hip score_matrix_impl).  Same users, same items: same bits, in the matrix, the two-kernel
    top-k and the fused one."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(5)
    m = NCF(700, 900, 32, 3, 0.0, "NeuMF-end").eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(25.0)
    m = m.cuda()
    t = m._tower(torch.device("cuda", 0))
    users = torch.arange(100, 612, device="cuda")
    items = torch.arange(900, device="cuda")
    a = t.score_matrix((100, 512))
    b = t.score_matrix(users, items)
    c = t.score_matrix((100, 512), items)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(a.view(torch.int32), c.view(torch.int32))
    for fused in (False, True):
        i1, v1 = t.score_topk((100, 512), 10, fused=fused)
        i2, v2 = t.score_topk(users, 10, items, fused=fused)
        assert torch.equal(i1, i2) and torch.equal(v1.view(torch.int32), v2.view(torch.int32)), fused


def test_linear_tile_shape_never_changes_a_bit():
    """The launcher picks the tile shape from the problem's size (gemm.hip: grid fill, operand re-reads, short-K rule), so the SAME
    rows come out of differently tiled launches depending on how many are asked for.  Without split-K every output is one
    k-ordered chain whatever the tiling: the full launch (decoder-sized: 128 x 64 / 64 x 64 tiles by the short-K rule) must equal,
    bit for bit, the same rows computed 1,024 at a time (small grids: other shapes) -- both orientations, bias per column and per
    row, ragged edges, a padded leading dimension."""
    from recommender_amd import ops
    g = torch.Generator(device="cuda").manual_seed(17)
    for (M, N, K, per_row, pad) in ((6040, 3706, 500, False, 0), (3706, 6040, 500, True, 6), (5000, 4100, 256, False, 0)):
        A = torch.randn(M, K, device="cuda", generator=g)
        W = torch.randn(N, K, device="cuda", generator=g) * 0.05
        b = torch.randn(M if per_row else N, device="cuda", generator=g)
        assert -(-M // 128) * -(-N // 128) > 1024, "shape no longer triggers the large-output rule"
        buf = torch.empty((M, N + pad), device="cuda")
        full = ops.linear(A, W, b, act="clamp", lo=-1.0, hi=1.0, out=buf[:, :N], bias_per_row=per_row)
        for r0 in range(0, M, 1024):                                 # 8 x ceil(N / 128) <= 384 tiles of 128 x 128 per launch
            r1 = min(r0 + 1024, M)
            part = ops.linear(A[r0:r1], W, b[r0:r1] if per_row else b, act="clamp", lo=-1.0, hi=1.0, bias_per_row=per_row)
            assert torch.equal(full[r0:r1].view(torch.int32), part.view(torch.int32)), (M, N, K, r0)


def test_two_narrow_layers_in_one_launch_equal_two_linear_calls_bit_for_bit():
    """reco_mlp2_f32 (the Hybrid encoders' 256 -> 128 -> 64 and decoders' 64 -> 128 -> 256, evaluate_hybridautorecncf_rating.py:78-106)
    against two reco_linear_f32 calls: same bits, every row-count class around the 32-row tile, strided input rows, no biases,
    other activations; the oracle's fp64 layers within the fp32 gate; unsupported widths are refused."""
    from recommender_amd import ops
    rng = np.random.default_rng(17)
    for M, K1, N2, N3, act3 in ((1, 256, 128, 64, None), (31, 256, 128, 64, None), (32, 64, 128, 256, "relu"), (33, 256, 128, 64, None),
                                (6040, 256, 128, 64, None), (3706, 64, 128, 256, "relu"), (500, 8, 32, 32, "sigmoid"), (97, 512, 128, 1024, None)):
        x = torch.from_numpy(rng.standard_normal((M, K1)).astype(np.float32)).cuda()
        w2 = torch.from_numpy((rng.standard_normal((N2, K1)) / np.sqrt(K1)).astype(np.float32)).cuda()
        w3 = torch.from_numpy((rng.standard_normal((N3, N2)) / np.sqrt(N2)).astype(np.float32)).cuda()
        b2 = torch.from_numpy(rng.standard_normal(N2).astype(np.float32)).cuda(); b3 = torch.from_numpy(rng.standard_normal(N3).astype(np.float32)).cuda()
        assert ops.mlp2_supported(K1, N2, N3)
        got = ops.mlp2(x, w2, b2, w3, b3, "relu", act3)
        want = ops.linear(ops.linear(x, w2, b2, "relu", precision="f32"), w3, b3, act3, precision="f32")
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (M, K1, N2, N3)
        h = np.maximum(x.cpu().numpy().astype(np.float64) @ w2.cpu().numpy().astype(np.float64).T + b2.cpu().numpy(), 0)
        o = h @ w3.cpu().numpy().astype(np.float64).T + b3.cpu().numpy()
        o = np.maximum(o, 0) if act3 == "relu" else (1 / (1 + np.exp(-o)) if act3 == "sigmoid" else o)
        assert_close(got.cpu().numpy(), o, atol=2e-6 * max(1.0, float(np.abs(o).max())), what="mlp2 vs fp64 %s" % ((M, K1, N2, N3),))
    wide = torch.from_numpy(rng.standard_normal((70, 300)).astype(np.float32)).cuda()       # a column block of a wider buffer
    xs = wide[:, 16:272]
    w2 = torch.randn(128, 256, device="cuda") * 0.05; w3 = torch.randn(64, 128, device="cuda") * 0.05
    got = ops.mlp2(xs, w2, None, w3, None)
    want = ops.linear(ops.linear(xs.contiguous(), w2, None, "relu", precision="f32"), w3, None, None, precision="f32")
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert not ops.mlp2_supported(48, 40, 64) and not ops.mlp2_supported(3706, 256, 128)
    with pytest.raises(ValueError):
        ops.mlp2(torch.zeros(4, 48, device="cuda"), torch.zeros(40, 48, device="cuda"), None, torch.zeros(64, 40, device="cuda"), None)


def test_linear_random_shape_sweep(tower_precision):
    """Seeded sweep over ragged shapes and alignments (K odd -> 4-byte loads, K even -> 8-byte, strided A / C views)."""
    from recommender_amd import ops
    rng = np.random.default_rng(2026 + FUZZ_SEED)
    for case in range(36 * FUZZ_MULT):
        M = int(rng.integers(1, 400)); N = int(rng.integers(1, 300))
        K = int(rng.choice([rng.integers(1, 64), rng.integers(64, 700), rng.integers(1024, 3000)]))
        x = rng.standard_normal((M, K + 5)).astype(np.float32); w = rng.standard_normal((N, K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        xg = torch.from_numpy(x).cuda()[:, 3:3 + K] if case % 3 == 0 else torch.from_numpy(np.ascontiguousarray(x[:, 3:3 + K])).cuda()
        out = torch.full((M, N + 4), 7.0, device="cuda")[:, :N] if case % 4 == 1 else None
        y = ops.linear(xg, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), "relu", out=out)
        xs = x[:, 3:3 + K].astype(np.float64)
        ref = np.maximum(xs @ w.astype(np.float64).T + b, 0)
        scale = np.abs(xs) @ np.abs(w).astype(np.float64).T + np.abs(b)
        err = np.abs(y.cpu().numpy() - ref)
        # the parity gate (1e-5 relative) plus a few ulps of the sum of the terms' magnitudes.  The relative part matters for
        # outliers: the long run (seed 21, case 1289: M 61, N 201, K 432) has ONE output of 80.19 whose k-ordered fp32 chain --
        # numpy's float32 loop gives the same bits as the kernel -- is 1.27e-4 from fp64 against 1.24e-4 of scale slack alone
        # (tools/repro_linear_case.py)
        assert (err <= 1e-5 * np.abs(ref) + 4e-7 * scale + 1e-6).all(), (case, M, N, K, err.max())
        if out is not None:
            assert y.data_ptr() == out.data_ptr()


def test_topk_random_sweep():
    """Seeded sweep: row lengths around the vector-load switch (1024), k around the one/two-slot switch (64), row views with
    a leading dimension larger than N, masks, heavy ties."""
    from recommender_amd import ops
    rng = np.random.default_rng(77 + FUZZ_SEED)
    for case in range(40 * FUZZ_MULT):
        rows = int(rng.integers(1, 40))
        N = int(rng.choice([rng.integers(1, 130), rng.integers(1000, 1100), rng.integers(4000, 9000), 4096, 1024]))
        k = int(min(N, rng.choice([1, 5, 10, 63, 64, 65, 100, 128])))
        ld = N + int(rng.choice([0, 0, 1, 3, 4]))
        full = rng.standard_normal((rows, ld)).astype(np.float32)
        if case % 3 == 0:
            full = np.round(full * 2) / 2                       # many equal scores
        t = torch.from_numpy(full).cuda()[:, :N]
        sc = full[:, :N]
        ex = rng.random((rows, N)) < rng.choice([0.0, 0.3, 0.95])
        ids, vals = ops.topk_rows(t, k, torch.from_numpy(ex).cuda() if case % 2 else None)
        oid, oval = O.topk_rows(O.mask_seen(sc, ex) if case % 2 else sc, k)
        assert np.array_equal(ids.cpu().numpy(), oid), (case, rows, N, k, ld)
        assert np.array_equal(vals.cpu().numpy(), oval), (case, rows, N, k, ld)


@pytest.mark.parametrize("tag", ["end", "gmf"])
def test_empty_and_ragged_batches(g2, g3, tag, tower_precision):
    """Empty inputs return empty outputs (what the reference's ATen ops do); every batch length around the kernels' 32 / 64 /
    tile boundaries gives the same scores as the full matrix; k == N returns the whole row sorted."""
    from recommender_amd import ops
    m = _ncf_from_sd(golden_sd(g2, tag))
    ref = g2[tag + "__scores"]
    U, I = ref.shape
    e = torch.empty(0, dtype=torch.long).cuda()
    assert tuple(m(e, e).shape) == (0,)
    assert tuple(m.score_matrix(users=e).shape) == (0, I)
    ids, vals = m.score_topk(e, 5)
    assert tuple(ids.shape) == (0, 5) and tuple(vals.shape) == (0, 5)
    rng = np.random.default_rng(7)
    for B in (1, 2, 31, 32, 33, 63, 64, 65, 95, 127, 129, 255, 257, 1000):
        u = rng.integers(0, U, B); i = rng.integers(0, I, B)
        s = m(torch.from_numpy(u).cuda(), torch.from_numpy(i).cuda()).cpu().numpy()
        assert_close(s, ref[u, i], what="%s ragged batch B=%d" % (tag, B))
    for nu, ni in ((1, 1), (1, I), (U, 1), (23, 63), (25, 65), (24, 64), (49, 129)):
        nu, ni = min(nu, U), min(ni, I)
        uu = rng.choice(U, nu, replace=False); cc = rng.choice(I, ni, replace=False)
        sm = m.score_matrix(users=torch.from_numpy(uu).cuda(), candidates=torch.from_numpy(cc).cuda()).cpu().numpy()
        assert_close(sm, ref[np.ix_(uu, cc)], what="%s ragged tile %dx%d" % (tag, nu, ni))
    # k == N: the whole candidate list, sorted by descending score with the lower position first among equals
    cc = rng.choice(I, 17, replace=False)
    ids, vals = m.score_topk(torch.arange(9).cuda(), 17, candidates=torch.from_numpy(cc).cuda())
    oid, oval = O.topk_rows(ref[np.ix_(np.arange(9), cc)], 17)
    assert_close(vals.cpu().numpy(), oval, what="k == N scores")
    assert (np.sort(ids.cpu().numpy(), axis=1) == np.arange(17)).all()
    # the stand-alone ops
    assert tuple(ops.gather_rows(torch.randn(5, 8).cuda(), e).shape) == (0, 8)
    assert tuple(ops.linear(torch.empty(0, 12).cuda(), torch.randn(7, 12).cuda(), torch.randn(7).cuda(), "relu").shape) == (0, 7)
    i0, v0 = ops.topk_rows(torch.empty(0, 40).cuda(), 3)
    assert tuple(i0.shape) == (0, 3) and tuple(v0.shape) == (0, 3)
    from recommender_amd.autorec.utils.model import AutoRec
    a = AutoRec(193, 89, 64, True)
    a.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g3, "a").items()})
    a = a.eval().cuda()
    assert tuple(a(torch.empty(0, 193).cuda()).shape) == (0, 193)
    x1 = torch.from_numpy(g3["a__train"].astype(np.float32).T[:1].copy()).cuda()      # one item vector, as the engine sends it
    assert_close(a(x1).cpu().numpy(), g3["a__raw"][:1], atol=2.0 * float(np.abs(g3["a__raw"] - g3["a__raw64"]).max()), what="B = 1 AutoRec")
    torch.cuda.synchronize()


@pytest.mark.parametrize("tag", ["end", "mlp", "gmf"])
def test_tower_random_request_sweep(g2, tag, tower_precision):
    """Random requests against the reference's all-pairs matrix: user lists and candidate lists with repeats, every k up to the
    list length, exclusion masks; the select is checked against the oracle's rule on the kernel's own scores (bit-exact)."""
    m = _ncf_from_sd(golden_sd(g2, tag))
    ref = g2[tag + "__scores"]
    U, I = ref.shape
    rng = np.random.default_rng(11 + FUZZ_SEED)
    for trial in range(24 * FUZZ_MULT):
        nu = int(rng.integers(1, 70)); nc = int(rng.integers(1, 140))
        users = rng.integers(0, U, nu); cands = rng.integers(0, I, nc)            # repeats allowed on both sides
        ut = torch.from_numpy(users).cuda(); ct = torch.from_numpy(cands).cuda()
        sm = m.score_matrix(ut, ct).cpu().numpy()
        assert_close(sm, ref[np.ix_(users, cands)], what="%s trial %d matrix %dx%d" % (tag, trial, nu, nc))
        k = int(rng.integers(1, min(nc, 128) + 1))
        mask = rng.random((nu, nc)) < 0.3 if trial % 2 else None
        ids, vals = m.score_topk(ut, k, ct, exclude=None if mask is None else torch.from_numpy(mask).cuda())
        want = sm if mask is None else O.mask_seen(sm.copy(), mask)
        oid, oval = O.topk_rows(want, k)
        assert np.array_equal(ids.cpu().numpy(), oid), "%s trial %d ids (k=%d)" % (tag, trial, k)
        assert np.array_equal(vals.cpu().numpy(), oval), "%s trial %d scores" % (tag, trial)


def test_topk_many_short_rows_lane_kernel():
    """>= 32,768 short rows take the lane-per-row kernel (the 99-negative protocol shape): every row length class around the
    32-column staging slab, k on both register-list sizes, aligned and unaligned leading dimensions, ties, NaN / inf / -0.0.
    Rows of whole float4s up to 160 scores (N = 100 is the protocol's) take topk_lane_select_kernel: one to four 52-column
    segments, ragged last segment and last workgroup, padded leading dimension, k below and at the list size, an all-equal row
    (every element is marked), heavy ties."""
    from recommender_amd import ops
    rng = np.random.default_rng(5)
    for case, (rows, N, k, pad) in enumerate(((40000, 100, 10, 0), (33000, 37, 3, 0), (32768, 130, 16, 2), (35001, 31, 10, 1),
                                              (32769, 64, 11, 0), (33333, 512, 1, 0), (34000, 16, 16, 3),
                                              # 16-byte aligned rows of float4s: the select-then-order kernel, 1 to 4 segments
                                              (33001, 100, 10, 0), (32800, 100, 16, 0), (33000, 36, 7, 0), (32770, 124, 10, 0),
                                              (32773, 12, 10, 0), (32790, 20, 13, 0), (115056, 100, 10, 0), (32800, 160, 10, 0),
                                              (33000, 104, 10, 4), (33003, 52, 10, 0), (33000, 56, 5, 8), (32769, 156, 16, 0))):
        full = rng.standard_normal((rows, N + pad)).astype(np.float32)
        if case % 2:
            full = np.round(full * 2) / 2                        # heavy ties: the lower position must win
        full[7, :3] = [np.nan, np.inf, -np.inf]
        full[8, :N] = 0.0
        full[8, min(2, N - 1)] = -0.0                           # equal to +0.0: position decides
        t = torch.from_numpy(full).cuda()[:, :N]
        ids, vals = ops.topk_rows(t, k)
        oid, oval = O.topk_rows(full[:, :N], k)
        assert np.array_equal(ids.cpu().numpy(), oid), (rows, N, k, pad)
        got = vals.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(oval)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(oval)), (rows, N, k, pad)


def test_topk_lane_kernels_random_sweep():
    """Seeded sweep over the lane-per-row kernels' whole domain (>= 32,768 rows, N <= 512, k <= 16): random row lengths, k, padded
    leading dimensions (which decide between the select-then-order kernel and the register-list kernel), ties, a NaN row."""
    from recommender_amd import ops
    rng = np.random.default_rng(909 + FUZZ_SEED)
    for case in range(10 * FUZZ_MULT):
        rows = int(rng.integers(32768, 36000))
        N = int(rng.choice([rng.integers(1, 40), 4 * rng.integers(1, 41), rng.integers(40, 513), 100]))
        k = int(min(N, rng.integers(1, 17)))
        pad = int(rng.choice([0, 0, 4, 1, 2]))
        full = rng.standard_normal((rows, N + pad)).astype(np.float32)
        if case % 3 == 1:
            full = np.round(full * 2) / 2
        if case % 3 == 2:
            full = np.round(full)                                # very heavy ties
        full[rows - 1, : min(3, N)] = [np.nan, np.inf, -np.inf][: min(3, N)]
        ids, vals = ops.topk_rows(torch.from_numpy(full).cuda()[:, :N], k)
        oid, oval = O.topk_rows(full[:, :N], k)
        assert np.array_equal(ids.cpu().numpy(), oid), (case, rows, N, k, pad)
        got = vals.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(oval)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(oval)), (case, rows, N, k, pad)


def test_csr_linear_random_sweep():
    """Seeded sweep of the sparse first layer against the dense MFMA layer and fp64: random shapes on both column widths (4 / 8
    columns per lane, generic form for N % 4 != 0), densities from empty to full, skewed rows (segment cap 96), both tables."""
    from recommender_amd import ops
    rng = np.random.default_rng(4242 + FUZZ_SEED)
    for case in range(8 * FUZZ_MULT):
        M = int(rng.integers(1, 300)); K = int(rng.integers(1, 2500))
        N = int(rng.choice([4 * rng.integers(1, 65), 4 * rng.integers(65, 200), rng.integers(1, 600)]))
        dens = float(rng.choice([0.0, 0.01, 0.05, 0.3, 1.0]))
        X = ((rng.random((M, K)) < dens) * rng.integers(1, 6, (M, K))).astype(np.float32)
        if case % 2 and dens > 0:
            X[int(rng.integers(0, M))] = rng.integers(1, 6, K)          # one full row among sparse ones: the skewed cap
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); b = rng.standard_normal(N).astype(np.float32)
        act = [None, "relu", "sigmoid"][case % 3]
        Xd = torch.from_numpy(X).cuda(); Wd = torch.from_numpy(W).cuda(); bd = torch.from_numpy(b).cuda()
        csr = ops.CsrMatrix.from_dense(Xd, k_ranges=[8, 1, 3][case % 3])
        got = ops.csr_linear(csr, ops.transpose(Wd), bd if case % 4 else None, act).cpu().numpy()
        bb = b if case % 4 else np.zeros_like(b)
        pre64 = X.astype(np.float64) @ W.astype(np.float64).T + bb
        ref64 = {"sigmoid": lambda v: 1 / (1 + np.exp(-v)), "relu": lambda v: np.maximum(v, 0), None: lambda v: v}[act](pre64)
        slack = 2e-7 * (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(bb))
        err = np.abs(got - ref64)
        tol = 1e-5 * np.maximum(np.abs(got), np.abs(ref64)) + 1e-6 + slack
        assert (err <= tol).all(), (case, M, K, N, dens, float((err / tol).max()))


def test_neumf_random_table_size_sweep():
    """Seeded sweep over table sizes (1..3,000 users x 10..40,000 items) and all three model kinds: the full catalogue for a user
    range through the two-kernel form, the fused score + top-k kernel and a user LIST (per-call projections) must agree bit for
    bit; sampled rows against the oracle (fp32 gate) and the select against the oracle's rule on the kernel's own scores; a pair
    list of the same model equals the matrix entries bit for bit."""
    from recommender_amd.helpers import NCF
    rng = np.random.default_rng(808 + FUZZ_SEED)
    for case in range(6 * FUZZ_MULT):
        n_users = int(rng.integers(1, 3000)); n_items = int(rng.choice([rng.integers(10, 200), rng.integers(200, 5000), rng.integers(5000, 40000)]))
        name = ["NeuMF-end", "MLP", "GMF"][case % 3]
        torch.manual_seed(2000 + case)
        m = NCF(n_users, n_items, 32, 3, 0.0, name).eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "embed" in n:
                    p.mul_(float(rng.choice([1.0, 25.0])))
        sd = {k_: v.detach().numpy().copy() for k_, v in m.state_dict().items()}
        m = m.cuda()
        tower = m._tower(torch.device("cuda", 0))
        k = int(rng.integers(1, min(n_items, 128) + 1))
        lo = int(rng.integers(0, n_users)); cnt = int(rng.integers(1, n_users - lo + 1))
        ids, vals = tower.score_topk((lo, cnt), k)
        if name != "GMF":                                              # the fused form runs on the MFMA tower only
            fi, fv = tower.score_topk((lo, cnt), k, fused=True)
            assert torch.equal(fi, ids) and torch.equal(fv.view(torch.int32), vals.view(torch.int32)), ("fused", case, n_users, n_items, k)
        ul = torch.arange(lo, lo + cnt, device="cuda")
        li, lv = tower.score_topk(ul, k)
        assert torch.equal(li, ids) and torch.equal(lv.view(torch.int32), vals.view(torch.int32)), ("user list", case, n_users, n_items, k)
        pick = np.sort(rng.choice(cnt, min(cnt, 6), replace=False)) + lo
        S = tower.score_matrix(torch.from_numpy(pick).cuda()).cpu().numpy()
        assert_close(S, O.ncf_score_all(sd, pick), what="%s %dx%d rows vs oracle" % (name, n_users, n_items))
        oid, oval = O.topk_rows(S, k)
        assert np.array_equal(ids.cpu().numpy()[pick - lo], oid) and np.array_equal(vals.cpu().numpy()[pick - lo], oval), (case, name, n_users, n_items, k)
        B = int(rng.integers(1, 5000))
        pu = rng.choice(pick, B); pi = rng.integers(0, n_items, B)
        pairs = m(torch.from_numpy(pu).cuda(), torch.from_numpy(pi).cuda()).cpu().numpy()
        assert np.array_equal(pairs, S[np.searchsorted(pick, pu), pi]), ("pairs", case, name, B)


def test_topk_workspace_limited_chunking_sweep():
    """A workspace anywhere between what ONE user needs and what the whole request needs makes the call walk the users in chunks
    (two-kernel and fused form, masks, candidate lists, the allocation-free entry point's fixed scratch): the answer never depends
    on the chunking, bit for bit."""
    from recommender_amd.helpers import NCF
    rng = np.random.default_rng(1313 + FUZZ_SEED)
    for case in range(6 * FUZZ_MULT):
        n_users = int(rng.integers(40, 2500)); n_items = int(rng.choice([rng.integers(20, 300), rng.integers(300, 9000)]))
        name = ["NeuMF-end", "MLP"][case % 2]
        torch.manual_seed(3000 + case)
        m = NCF(n_users, n_items, 32, 3, 0.0, name).eval().cuda()
        tower = m._tower(torch.device("cuda", 0))
        L = tower._L
        k = int(rng.integers(1, min(n_items, 100) + 1))
        users = torch.from_numpy(rng.integers(0, n_users, int(rng.integers(1, 1500)))).cuda() if case % 3 else None
        U = users.numel() if users is not None else n_users
        cands = torch.from_numpy(rng.integers(0, n_items, int(rng.integers(k, 2 * n_items)))).cuda() if case % 4 == 1 else None
        N = cands.numel() if cands is not None else n_items
        mask = torch.from_numpy(rng.random((U, N)) < 0.2).cuda() if case % 2 else None
        want_i, want_v = tower.score_topk(users, k, cands, exclude=mask)
        for fused in ((False, True) if mask is None else (False,)):
            one = L.reco_tower_topk_fused_workspace_bytes(tower._h, 1, N, k) if fused else L.reco_tower_topk_workspace_bytes(1, N, k)
            full = L.reco_tower_topk_fused_workspace_bytes(tower._h, U, N, k) if fused else L.reco_tower_topk_workspace_bytes(U, N, k)
            for frac in (0.0, float(rng.random()) * 0.3, float(rng.random())):
                ws = torch.empty(int(one + frac * max(full - one, 0)) + 256, dtype=torch.uint8, device="cuda")
                ids, vals = tower.score_topk(users, k, cands, exclude=mask, ws=ws, fused=fused)
                assert torch.equal(ids, want_i) and torch.equal(vals.view(torch.int32), want_v.view(torch.int32)), (case, name, U, N, k, fused, frac)


def test_gather_mlp2_and_seen_select_random_sweep():
    """Seeded sweeps of the three small stand-alone ops: row gathers (any row width, repeated ids), the two-layer launch at every
    width combination it accepts, and the select that reads the seen mask off the rating matrix with users picked by id."""
    from recommender_amd import ops
    rng = np.random.default_rng(5150 + FUZZ_SEED)
    c = lambda a: torch.from_numpy(a).cuda()
    for case in range(8 * FUZZ_MULT):
        rows = int(rng.integers(1, 6000)); D = int(rng.choice([1, 3, 4, 8, 30, 32, 64, 100, 128, 256, 516]))
        table = rng.standard_normal((rows, D)).astype(np.float32)
        idx = rng.integers(0, rows, int(rng.integers(0, 9000)))
        assert np.array_equal(ops.gather_rows(c(table), c(idx)).cpu().numpy(), table[idx]), ("gather", case, rows, D)
        dg = int(rng.choice([8, 16, 32, 64, 10, 3])); dm = int(rng.choice([16, 32, 64, 128, 40, 5]))
        nu = int(rng.integers(1, 3000)); ni = int(rng.integers(1, 3000)); B = int(rng.integers(1, 7000))
        eu_g = rng.standard_normal((nu, dg)).astype(np.float32); ei_g = rng.standard_normal((ni, dg)).astype(np.float32)
        eu_m = rng.standard_normal((nu, dm)).astype(np.float32); ei_m = rng.standard_normal((ni, dm)).astype(np.float32)
        u = rng.integers(0, nu, B); i = rng.integers(0, ni, B)
        gp, mi = ops.neumf_gather(c(eu_g), c(ei_g), c(eu_m), c(ei_m), c(u), c(i))
        assert np.array_equal(gp.cpu().numpy(), eu_g[u] * ei_g[i]) and np.array_equal(mi.cpu().numpy(), np.concatenate([eu_m[u], ei_m[i]], 1)), ("neumf_gather", case)
    widths = [8, 16, 32, 64, 128, 256, 512, 1024]
    done = 0
    while done < 6 * FUZZ_MULT:
        K1, N2, N3 = (int(rng.choice(widths)) for _ in range(3))
        if not ops.mlp2_supported(K1, N2, N3):
            continue
        done += 1
        M = int(rng.integers(1, 5000))
        x = c(rng.standard_normal((M, K1)).astype(np.float32))
        w2 = c((rng.standard_normal((N2, K1)) / np.sqrt(K1)).astype(np.float32)); w3 = c((rng.standard_normal((N3, N2)) / np.sqrt(N2)).astype(np.float32))
        b2 = c(rng.standard_normal(N2).astype(np.float32)) if done % 2 else None
        b3 = c(rng.standard_normal(N3).astype(np.float32)) if done % 3 else None
        act3 = [None, "relu", "sigmoid"][done % 3]
        got = ops.mlp2(x, w2, b2, w3, b3, "relu", act3)
        want = ops.linear(ops.linear(x, w2, b2, "relu", precision="f32"), w3, b3, act3, precision="f32")
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("mlp2", M, K1, N2, N3, act3)
    for case in range(6 * FUZZ_MULT):
        U = int(rng.integers(1, 300)); N = int(rng.choice([rng.integers(1, 200), rng.integers(900, 1100), rng.integers(3000, 9000)]))
        k = int(min(N, rng.choice([1, 3, 10, 64, 100, 128]))); pad = int(rng.choice([0, 1, 4, 6]))
        sc = np.clip(rng.standard_normal((U, N)) * 2 + 3, 1.0, 5.0).astype(np.float32)
        R = ((rng.random((U, N)) < rng.choice([0.0, 0.05, 0.5, 0.98])) * rng.integers(1, 6, (U, N))).astype(np.float32)
        buf = torch.zeros((U, N + pad), dtype=torch.float32, device="cuda")
        buf[:, :N] = c(sc)
        users = rng.integers(0, U, int(rng.integers(1, 400)))
        ids, vals = ops.topk_rows_seen(buf[:, :N], k, c(R), c(users))
        oid, oval = O.topk_rows(O.mask_seen(sc, R > 0)[users], k)
        assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval), ("seen select", case, U, N, k, pad)


@pytest.mark.parametrize("factor,layers,name", [(10, 3, "NeuMF-end"), (6, 2, "MLP"), (7, 1, "GMF"), (10, 3, "NeuMF-pre")])
def test_factor_numbers_that_are_not_multiples_of_4(factor, layers, name):
    """nn.Embedding takes any width: an NCF with factor_num = 10 (GMF rows of 10 floats, MLP rows of 40) or 6 / 7 runs on the unfused
    HIP path through the one-float-per-thread gathers; pair lists, matrix and top-k against the oracle."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(factor)
    gmf = NCF(57, 83, factor, layers, 0.0, "GMF") if name == "NeuMF-pre" else None
    mlp = NCF(57, 83, factor, layers, 0.0, "MLP") if name == "NeuMF-pre" else None
    m = NCF(57, 83, factor, layers, 0.0, name, gmf, mlp).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in m.state_dict().items()}
    m = m.cuda()
    rng = np.random.default_rng(factor)
    u = rng.integers(0, 57, 700); i = rng.integers(0, 83, 700)
    got = m(torch.from_numpy(u).cuda(), torch.from_numpy(i).cuda()).cpu().numpy()
    assert_close(got, O.ncf_forward(sd, u, i), what="factor %d %s pairs" % (factor, name))
    assert_close(got, O.ncf_forward(sd, u, i, np.float64), what="factor %d %s pairs vs fp64" % (factor, name))
    S = m.score_matrix().cpu().numpy()
    assert_close(S, O.ncf_score_all(sd, np.arange(57)), what="factor %d %s matrix" % (factor, name))
    ids, vals = m.score_topk(None, 9)
    oid, oval = O.topk_rows(S, 9)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)


@pytest.mark.parametrize("latent,layers,hidden", [(32, (64, 32), (128, 64)), (16, (64,), (40,)), (64, (128, 64, 32), (256, 128)), (10, (20, 8), (12, 6))])
def test_hybrid_shapes_other_than_the_shipped_one(latent, layers, hidden):
    """HybridAutoRecNCF takes any latent width, MLP layer list and encoder widths (evaluate_hybridautorecncf_rating.py:58-208); the
    MFMA head covers latent 64 / [128, 64], everything else runs unfused, all-HIP: forward, score_all and the masked top-k
    against the oracle."""
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    torch.manual_seed(latent)
    U, I = 71, 53
    m = HybridAutoRecNCF(U, I, latent, list(layers), hidden_dims=hidden, rating_output=bool(latent % 3)).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in m.state_dict().items()}
    m = m.cuda()
    rng = np.random.default_rng(latent)
    R = ((rng.random((U, I)) < 0.2) * rng.integers(1, 6, (U, I))).astype(np.float32)
    Rd = torch.from_numpy(R).cuda()
    want = O.hybrid_score_all(sd, R, np.float32, rating=bool(latent % 3))
    want64 = O.hybrid_score_all(sd, R, np.float64, rating=bool(latent % 3))
    atol = 2.0 * float(np.abs(want - want64).max()) + 2e-6
    S = m.score_all(Rd).cpu().numpy()
    assert_close(S, want64, atol=atol, what="hybrid latent %d layers %s" % (latent, layers))
    uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
    uid = torch.from_numpy(uu.reshape(-1)).cuda(); iid = torch.from_numpy(ii.reshape(-1)).cuda()
    pred, urec, irec = m(Rd[uid], Rd.t().contiguous()[iid], uid, iid)
    assert_close(pred.cpu().numpy().reshape(U, I), want64, atol=atol, what="forward")
    ids, vals = m.score_all(Rd, k=7, mask_seen=True)
    oid, oval = O.topk_rows(O.mask_seen(S.copy(), R > 0), 7)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)
    ids, vals = m.score_all(Rd, k=7, exclude=torch.from_numpy(R > 0).cuda())
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval)


def test_autorec_and_hybrid_random_size_sweep():
    """Seeded sweep over matrix sizes the fixtures do not hold (users / items from a handful to ~700, densities from 1 % to 60 %,
    ratings 1..5 and 0/1): AutoRec reconstruct_all + masked top-k and Hybrid score_all + masked top-k, dense and with the matrix
    attached as CSR, against the oracle's fp32 restatement; the selects bit-exact on the kernels' own scores."""
    from recommender_amd.autorec.utils.model import AutoRec
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    rng = np.random.default_rng(606 + FUZZ_SEED)
    for case in range(6 * FUZZ_MULT):
        U = int(rng.integers(3, 700)); I = int(rng.integers(12, 700)); dens = float(rng.choice([0.01, 0.05, 0.2, 0.6]))
        R = (rng.random((U, I)) < dens).astype(np.float32)
        if case % 2 == 0:
            R *= rng.integers(1, 6, (U, I)).astype(np.float32)
        Rd = torch.from_numpy(R).cuda()
        k = int(rng.integers(1, min(I, 20) + 1))
        torch.manual_seed(1000 + case)
        # ---- AutoRec (item-based: rows of X are item vectors over the users) ----
        H = int(rng.choice([16, 64, 500, 37 * 4, 37]))
        a = AutoRec(U, I, H, True).eval()
        sd = {n: v.detach().numpy().copy() for n, v in a.state_dict().items()}
        a = a.cuda()
        ref = O.autorec_forward(sd, np.ascontiguousarray(R.T), np.float32, clamp=(1.0, 5.0)).T
        ref64 = O.autorec_forward(sd, np.ascontiguousarray(R.T), np.float64, clamp=(1.0, 5.0)).T
        atol = 2.0 * float(np.abs(ref - ref64).max()) + 2e-6
        for sparse in (False, True):
            if sparse:
                a.attach_ratings(Rd)
            pred = a.reconstruct_all(Rd).cpu().numpy()
            assert_close(pred, ref64, atol=atol, what="autorec %dx%d H=%d sparse=%s" % (U, I, H, sparse))
            ids, vals = a.recommend_topk(Rd, k)
            oid, oval = O.topk_rows(O.mask_seen(np.ascontiguousarray(pred), R > 0), k)
            assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval), ("autorec", case, U, I, H, k, sparse)
        # the serving path: one user's scores for a list of items from the cached hidden layer (any hidden width)
        a.cache_hidden(Rd)
        uid = int(rng.integers(0, U)); items = rng.integers(0, I, 9)
        assert_close(a.predict_user(uid, items).cpu().numpy(), ref64[uid, items], atol=atol, what="predict_user H=%d" % H)
        # ---- Hybrid ----
        rating = bool(case % 3 == 0)
        m = HybridAutoRecNCF(U, I, 64, [128, 64], rating_output=rating).eval()
        hsd = {n: v.detach().numpy().copy() for n, v in m.state_dict().items()}
        m = m.cuda()
        want = O.hybrid_score_all(hsd, R, np.float32, rating=rating)
        want64 = O.hybrid_score_all(hsd, R, np.float64, rating=rating)
        # the gate's absolute part follows the reference arithmetic's own fp32 error on this matrix, as for AutoRec above (a dense
        # 1..5 matrix drives the untrained encoders to latents whose head sums cancel: the oracle's fp32 and fp64 then differ by
        # 7.7e-6 on a 208 x 158 case, 2.5 times the plain gate)
        hatol = 2.0 * float(np.abs(want - want64).max()) + 2e-6
        for sparse in (False, True):
            if sparse:
                m.attach_ratings(Rd)
            S = m.score_all(Rd).cpu().numpy()
            assert_close(S, want64, atol=hatol, what="hybrid %dx%d sparse=%s" % (U, I, sparse))
            ids, vals = m.score_all(Rd, k=k, mask_seen=True)
            oid, oval = O.topk_rows(O.mask_seen(S.copy(), R > 0), k)
            assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy(), oval), ("hybrid", case, U, I, k, sparse)


@pytest.mark.parametrize("scale", [1e-6, 0.01, 1.0, 25.0, 3000.0])
def test_tile_kernels_equal_the_pair_kernel_bit_for_bit_at_any_embedding_scale(scale):
    """Tile kernels and pair-list kernel share the MFMA body, the clamped layer 1 (tower_shared.h layer1_scale) and the summation
    order of the GMF half (gmf_col / gmf_half_*: per float4 column, then in column order), so score_matrix / score_topk must
    reproduce score_pairs -- forward(u, i) -- to the bit whatever the size of the embeddings, with and without a GMF half."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(21)
    for name in ("NeuMF-end", "MLP"):
        m = NCF(211, 333, 32, 3, 0.0, name).eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "embed" in n:
                    p.mul_(100.0 * scale)
        m = m.cuda()
        t = m._tower(torch.device("cuda", 0))
        uu, ii = torch.meshgrid(torch.arange(211), torch.arange(333), indexing="ij")
        pairs = t.score_pairs(uu.reshape(-1).cuda(), ii.reshape(-1).cuda()).view(211, 333)
        full = t.score_matrix()
        assert torch.equal(pairs.view(torch.int32), full.view(torch.int32)), (name, scale)
        ids, vals = t.score_topk(None, 7)
        fids, fvals = t.score_topk(None, 7, fused=True)
        oid, oval = O.topk_rows(full.cpu().numpy(), 7)
        for a, b in ((ids, vals), (fids, fvals)):
            assert np.array_equal(a.cpu().numpy(), oid) and np.array_equal(b.cpu().numpy().view(np.uint32), oval.view(np.uint32))


@pytest.mark.parametrize("name", ["NeuMF-end", "MLP"])
def test_scores_do_not_depend_on_the_layer1_scale(name):
    """The exact kernels work on layer-1 activations scaled by a power of two that the handle derives from a bound over ALL of
    its table rows.  One extra user row with 4,096 times larger embeddings -- never scored -- moves that scale by 2^12; the
    scores of every other user must not change by a single bit (pair list, score matrix, both top-k forms)."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(33)
    a = NCF(150, 260, 32, 3, 0.0, name).eval()
    with torch.no_grad():
        for n, p in a.named_parameters():
            if "embed" in n:
                p.mul_(40.0)
    b = NCF(151, 260, 32, 3, 0.0, name).eval()
    sd = a.state_dict()
    with torch.no_grad():
        for k, v in b.state_dict().items():
            if k.startswith("embed_user"):
                v[:150] = sd[k]
                v[150] = sd[k][0] * 4096.0
            else:
                v.copy_(sd[k])
    ta, tb = a.cuda()._tower(torch.device("cuda", 0)), b.cuda()._tower(torch.device("cuda", 0))
    users = torch.arange(150).cuda()
    uu, ii = torch.meshgrid(torch.arange(150), torch.arange(260), indexing="ij")
    pa = ta.score_pairs(uu.reshape(-1).cuda(), ii.reshape(-1).cuda())
    pb = tb.score_pairs(uu.reshape(-1).cuda(), ii.reshape(-1).cuda())
    assert torch.equal(pa.view(torch.int32), pb.view(torch.int32))
    ma, mb = ta.score_matrix(users), tb.score_matrix(users)
    assert torch.equal(ma.view(torch.int32), mb.view(torch.int32))
    for fused in (False, True):
        ia, va = ta.score_topk(users, 9, fused=fused)
        ib, vb = tb.score_topk(users, 9, fused=fused)
        assert torch.equal(ia, ib) and torch.equal(va.view(torch.int32), vb.view(torch.int32))


def test_a_param_data_write_never_silently_saturates_layer1(monkeypatch):
    """A write through param.data moves no version counter, so the module's handle cache cannot see it (INTEGRATION section 1:
    unsupported).  What must never happen silently is the NEW failure mode of the clamped layer 1: embeddings that grew beyond
    the handle's scale s would be clipped at 1/s.  reco_tower_check_scale detects exactly that; with RECO_CHECK_STALE=1 the
    module rebuilds its handle and returns the oracle's scores for the new weights."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(8)
    m = NCF(60, 90, 32, 3, 0.0, "NeuMF-end").eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(30.0)
    m = m.cuda()
    dev = torch.device("cuda", 0)
    u = torch.arange(60).repeat_interleave(90).cuda(); i = torch.arange(90).repeat(60).cuda()
    before = m(u, i)
    t = m._tower(dev)
    assert not t.scale_is_stale()
    m.embed_user_MLP.weight.data.mul_(0.5)                             # smaller values: stale projections, but nothing can saturate
    assert not t.scale_is_stale()
    m.embed_user_MLP.weight.data.mul_(2000.0)                          # larger: the old scale no longer covers p + q
    assert m._tower(dev) is t and t.scale_is_stale()
    monkeypatch.setenv("RECO_CHECK_STALE", "1")
    after = m(u, i)
    assert m._tower(dev) is not t                                      # rebuilt
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = O.ncf_forward(sd, u.cpu().numpy(), i.cpu().numpy())
    assert_close(after.cpu().numpy(), want, atol=1e-6 * float(np.abs(want).max()), what="scores after the rebuild")
    assert not torch.equal(before, after) and not m._tower(dev).scale_is_stale()


@pytest.mark.parametrize("emb", [0.0, 1e-30, 1e-12, 1e12])
@pytest.mark.parametrize("name", ["NeuMF-end", "MLP"])
def test_degenerate_embedding_magnitudes_against_the_oracle(name, emb):
    """The layer-1 scale s is a power of two from a bound over the tables: all-zero tables (bound = |b1|, or 0), embeddings of
    1e-30 (s is capped at 2^24 so that s * b2 cannot overflow nor w_predict / s go denormal) and of 1e12 (s ~ 2^-50) must still
    give the oracle's scores: every form, fp32 gate relative to the largest score."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(5)
    m = NCF(40, 70, 32, 3, 0.0, name).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(100.0 * emb)                       # reference init: std 0.01
            elif n.endswith("bias"):
                p.copy_(torch.randn_like(p) * (0.0 if emb == 0.0 and "MLP_layers.0" in n else 0.7))
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    uu, ii = np.meshgrid(np.arange(40), np.arange(70), indexing="ij")
    want = O.ncf_forward(sd, uu.reshape(-1), ii.reshape(-1), dtype=np.float64).reshape(40, 70)
    assert np.isfinite(want).all()
    atol = 1e-6 * max(1.0, float(np.abs(want).max()))
    t = m.cuda()._tower(torch.device("cuda", 0))
    pairs = t.score_pairs(torch.from_numpy(uu.reshape(-1)).cuda(), torch.from_numpy(ii.reshape(-1)).cuda()).view(40, 70)
    full = t.score_matrix()
    assert_close(pairs.cpu().numpy(), want, rtol=1e-5, atol=atol, what="pairs %s %g" % (name, emb))
    assert_close(full.cpu().numpy(), want, rtol=1e-5, atol=atol, what="matrix %s %g" % (name, emb))
    for fused in (False, True):
        ids, vals = t.score_topk(None, 5, fused=fused)
        oid, oval = O.topk_rows(full.cpu().numpy(), 5)
        assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(vals.cpu().numpy().view(np.uint32), oval.view(np.uint32))


def test_an_absurd_or_non_finite_row_does_not_disturb_the_other_rows():
    """One user row of 1e36 and one holding inf / NaN.  The first one's layer-1 pre-activations (~2^122) are beyond the 2^80 the
    scale admits -- following it, s = 2^-123 would put the other rows' s * b2 into the denormals; the scale is floored at 2^-80
    instead.  Their own pairs get finite scores without meaning (header conventions); every other user keeps its scores bit for
    bit."""
    from recommender_amd.helpers import NCF
    torch.manual_seed(44)
    a = NCF(90, 130, 32, 3, 0.0, "NeuMF-end").eval()
    with torch.no_grad():
        for n, p in a.named_parameters():
            if "embed" in n:
                p.mul_(30.0)
    b = NCF(92, 130, 32, 3, 0.0, "NeuMF-end").eval()
    sd = a.state_dict()
    with torch.no_grad():
        for k, v in b.state_dict().items():
            if k.startswith("embed_user"):
                v[:90] = sd[k]
                v[90] = 1e36
                v[91] = sd[k][1]
                v[91, 0] = float("inf"); v[91, 1] = float("nan")
            else:
                v.copy_(sd[k])
    ta, tb = a.cuda()._tower(torch.device("cuda", 0)), b.cuda()._tower(torch.device("cuda", 0))
    users = torch.arange(90).cuda()
    assert torch.equal(ta.score_matrix(users).view(torch.int32), tb.score_matrix(users).view(torch.int32))
    uu, ii = torch.meshgrid(torch.arange(90), torch.arange(130), indexing="ij")
    pa = ta.score_pairs(uu.reshape(-1).cuda(), ii.reshape(-1).cuda())
    pb = tb.score_pairs(uu.reshape(-1).cuda(), ii.reshape(-1).cuda())
    assert torch.equal(pa.view(torch.int32), pb.view(torch.int32))
    # and against the oracle (the 2^-80 scale must not cost the ordinary rows any accuracy)
    sda = {k: v.cpu().numpy() for k, v in a.state_dict().items()}
    want = O.ncf_forward(sda, uu.reshape(-1).numpy(), ii.reshape(-1).numpy(), dtype=np.float64)
    assert_close(pb.cpu().numpy(), want, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(want).max())), what="ordinary rows beside an outlier")
    out = tb.score_matrix(torch.tensor([90]).cuda())
    assert out.shape == (1, 130)                     # the outlier's own scores: whatever they are, the call completes
