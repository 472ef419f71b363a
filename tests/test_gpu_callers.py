"""Callers either side of the path on the GPU: engines, batched evaluators."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden_sd
from oracle import reco_oracle as O

pytestmark = pytest.mark.gpu


def _ncf(g2, tag="end"):
    from recommender_amd.loaders import ncf_from_state_dict
    return ncf_from_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g2, tag).items()}).cuda()


def test_ncf_engine_matches_reference_engine_semantics(g2):
    from recommender_amd.api_core import NCFInferenceEngine, RequestError
    ref = g2["end__scores"]
    eng = NCFInferenceEngine(_ncf(g2))
    assert eng.config == {"user_num": 257, "item_num": 131}
    assert abs(eng.predict(5, 7) - ref[5, 7]) <= 1e-5 * abs(ref[5, 7]) + 1e-6
    items = [9, 3, 130, 3, 77, 0]
    got = eng.predict_batch(11, items)
    want = sorted(zip(items, ref[11, items].tolist()), key=lambda x: x[1], reverse=True)     # inference_engine.py:115-125
    assert [g[0] for g in got] == [w[0] for w in want]
    assert_close([g[1] for g in got], [w[1] for w in want], what="predict_batch scores")
    rec = eng.recommend(42, k=10)
    want_ids = np.argsort(ref[42])[::-1][:10]                                                # inference_engine.py:166
    assert [r[0] for r in rec] == want_ids.tolist()
    assert_close([r[1] for r in rec], ref[42, want_ids], what="recommend scores")
    cand = [5, 9, 5, 100, 64, 9, 2]
    rec = eng.recommend(42, k=50, candidate_item_ids=cand)                                  # k clipped to the de-duplicated list
    uniq = list(set(cand))
    want = [uniq[j] for j in np.argsort(ref[42, uniq])[::-1]]
    assert [r[0] for r in rec] == want
    assert eng.recommend(42, k=0) == []
    rec = eng.recommend(42, k=131)                                                          # past the wavefront select's k <= 128
    assert [r[0] for r in rec] == np.argsort(-ref[42], kind="stable").tolist()
    for bad in (lambda: eng.predict(257, 0), lambda: eng.predict(0, -1), lambda: eng.recommend(0, 5, [131]), lambda: eng.recommend(0, 5, [])):
        with pytest.raises(RequestError) as e:
            bad()
        assert e.value.status_code == 400
    with pytest.raises(RequestError) as e:
        NCFInferenceEngine(None, 10, 10).predict(0, 0)
    assert e.value.status_code == 503


def test_autorec_engine_serves_from_cached_hidden_layer(g3, tower_precision):
    from recommender_amd.api_core import AutoRecInferenceEngine, RequestError
    from recommender_amd.autorec.utils.model import AutoRec
    m = AutoRec(193, 89, 64, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g3, "a").items()})
    m = m.eval().cuda()
    train = torch.from_numpy(g3["a__train"].astype(np.float32)).cuda()
    eng = AutoRecInferenceEngine(m, train)
    clamped = g3["a__clamped"]                                   # [item, user] = reference forward of every item vector, clamped
    atol = 2.0 * float(np.abs(g3["a__raw"] - g3["a__raw64"]).max())
    assert abs(eng.predict(17, 33) - clamped[33, 17]) <= 1e-5 * 5 + atol
    items = list(range(0, 89, 3))
    got = eng.predict_batch(17, items)
    assert_close(sorted(g[1] for g in got), sorted(clamped[items, 17].tolist()), atol=atol, what="autorec predict_batch")
    assert [g[1] for g in got] == sorted((g[1] for g in got), reverse=True)
    # reference recommend = stable descending sort of all item scores, first k (inference_engine.py:294,333-341)
    rec = eng.recommend(17, k=10)
    mine = eng._user_scores(17).cpu().numpy()
    want = [i for i, _ in sorted(enumerate(mine.tolist()), key=lambda x: x[1], reverse=True)[:10]]
    assert [r[0] for r in rec] == want
    assert_close([r[1] for r in rec], np.sort(clamped[:, 17])[::-1][:10], atol=atol, what="autorec recommend scores")
    with pytest.raises(RequestError):
        eng.predict(193, 0)
    with pytest.raises(RequestError) as e:
        AutoRecInferenceEngine(None, None, 5, 5).recommend(0)
    assert e.value.status_code == 503


def test_evaluate_protocol_real_gmf(g1):
    import os
    from conftest import GOLDEN
    from recommender_amd.evaluate import evaluate_protocol
    from recommender_amd.loaders import load_ncf_model
    m = load_ncf_model(os.path.join(GOLDEN, "GMF.pth")).cuda()
    items = np.concatenate([g1["pos"].astype(np.int64)[:, None], g1["neg"].astype(np.int64)], axis=1)
    hr, nd, hr_c, nd_c = evaluate_protocol(m, g1["users"], items, 10)
    assert hr == float(g1["hr_mean"]) and abs(nd - float(g1["ndcg_mean"])) < 1e-15     # identical HR@10 / NDCG@10
    assert np.array_equal(hr_c.astype(np.uint8), g1["hr"])


def test_autorec_and_hybrid_ranking_evaluators(g3, g4, tower_precision):
    from recommender_amd.autorec.utils.model import AutoRec
    from recommender_amd.evaluate import autorec_ranking_metrics, hybrid_ranking_metrics
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    for tag, (U, I, H) in (("a", (193, 89, 64)), ("b", (61, 37, 500))):
        m = AutoRec(U, I, H, True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g3, tag).items()})
        m = m.eval().cuda()
        train = g3[tag + "__train"].astype(np.float32); test = g3[tag + "__test"]
        hr, nd = autorec_ranking_metrics(m, train, test, 10)
        # same rule applied to the reference's own clamped matrix
        oid, _ = O.topk_rows(g3[tag + "__masked"], 10)
        users = [u for u in range(U) if g3[tag + "__has_gt"][u]]
        ohr = np.mean([O.hit_multi(np.flatnonzero(test[u] > 0), oid[u]) for u in users])
        ond = np.mean([O.ndcg_multi(np.flatnonzero(test[u] > 0), oid[u]) for u in users])
        assert abs(hr - ohr) <= 1.0 / len(users) and abs(nd - ond) <= 0.01
        # the reference resolves clamp-plateau ties with an unstable argsort: its own figure differs only through those ties
        assert abs(hr - float(g3[tag + "__hr_mean"])) <= 0.1
    sd = golden_sd(g4, "logit")
    m = HybridAutoRecNCF(40, 26, 64, [128, 64], rating_output=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.eval().cuda()
    R = g4["logit__R"].astype(np.float32)
    rng = np.random.default_rng(0)
    test = ((rng.random(R.shape) < 0.1) & (R == 0)).astype(np.float32)
    hr, nd = hybrid_ranking_metrics(m, R, test, 5)
    oid, _ = O.topk_rows(O.mask_seen(g4["logit__pred"].copy(), R > 0), 5)
    users = [u for u in range(40) if test[u].any()]
    ohr = np.mean([O.hit_multi(np.flatnonzero(test[u] > 0), oid[u]) for u in users])
    assert abs(hr - ohr) <= 1.0 / len(users)


def test_c_abi_rejects_bad_arguments_with_error_codes_not_faults():
    """Straight through ctypes: every refused call returns a negative code and leaves a message, nothing is launched."""
    import ctypes
    from recommender_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(8, 16, device=dev); w = torch.randn(4, 16, device=dev); y = torch.empty(8, 4, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def refused(rc, code, fragment):
        assert rc == code, rc
        assert fragment in L.reco_last_error().decode(), L.reco_last_error()

    refused(L.reco_linear_f32(None, 16, None, 0, 8, p(w), 16, None, 4, 16, 0, 0.0, 0.0, p(y), 4, 0, None, 0, st), _lib.EINVAL, "null")
    refused(L.reco_linear_f32(p(x), 8, None, 0, 8, p(w), 16, None, 4, 16, 0, 0.0, 0.0, p(y), 4, 0, None, 0, st), _lib.EINVAL, "leading dimension")
    refused(L.reco_linear_f32(p(x), 16, None, 0, 8, p(w), 16, None, 4, 16, 99, 0.0, 0.0, p(y), 4, 0, None, 0, st), _lib.EINVAL, "activation")
    refused(L.reco_linear_f32(p(x), 16, None, 0, 8, p(w), 16, None, 4, 16, 0, 0.0, 0.0, p(y), 4, 64, None, 0, st), _lib.EINVAL, "flags")
    ids = torch.empty(8, 3, dtype=torch.int32, device=dev); vals = torch.empty(8, 3, device=dev)
    refused(L.reco_topk_rows(p(x), 8, 16, 16, 0, None, 0, p(ids), p(vals), None, 0, st), _lib.EINVAL, "k must be")
    refused(L.reco_topk_rows(p(x), 8, 16, 16, 17, None, 0, p(ids), p(vals), None, 0, st), _lib.EINVAL, "exceeds")
    refused(L.reco_topk_rows(p(x), 8, 16, 8, 3, None, 0, p(ids), p(vals), None, 0, st), _lib.EINVAL, "ld smaller")
    idx = torch.zeros(4, dtype=torch.int64, device=dev)
    refused(L.reco_gather_rows(p(x), 8, 6, p(idx), 4, p(y), None, st), _lib.EINVAL, "multiple of 4")

    d = _lib.TowerDesc()                                     # a tower with no branch at all
    h = ctypes.c_void_p()
    refused(L.reco_tower_create(ctypes.byref(d), st, ctypes.byref(h)), _lib.EINVAL, "neither")
    assert not h.value
    # a real handle: too small a workspace is ENOMEM, an unknown precision EINVAL
    from recommender_amd.helpers import NCF
    torch.manual_seed(0)
    t = NCF(40, 30, 32, 3, 0.0, "NeuMF-end").eval().cuda()._tower(dev)
    out = torch.empty(40, 30, device=dev)
    refused(L.reco_tower_score_matrix(t._h, None, 0, 40, None, 30, p(out), 30, None, None, 0, st), _lib.ENOMEM, "workspace")
    refused(L.reco_tower_score_matrix(t._h, None, 35, 10, None, 30, p(out), 30, None, None, 0, st), _lib.EINVAL, "user range")
    refused(L.reco_tower_score_pairs(None, None, None, 4, None, None, st), _lib.EINVAL, "null handle")
    with pytest.raises(ValueError):
        ops.Tower(w_predict=torch.zeros(64, device=dev), b_predict=torch.zeros(1, device=dev), precision="fp8")
    torch.cuda.synchronize()


def test_host_wrappers_refuse_tensors_whose_extents_the_kernels_would_overrun():
    """Caller-provided outputs / rows / weights go to the kernels as raw pointers: wrong extents raise before any launch."""
    from recommender_amd import ops
    from recommender_amd.helpers import NCF
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = NCF(40, 30, 32, 3, 0.0, "NeuMF-end").eval().cuda()
    t = m._tower(dev)
    ids = torch.empty(40, 5, dtype=torch.int32, device=dev); vals = torch.empty(40, 5, device=dev)
    got = t.score_topk(None, 5, out=(ids, vals))
    assert got[0] is ids and got[1] is vals
    for bad in ((ids[:39], vals), (ids, vals[:, :4]), (ids.long(), vals), (ids.cpu(), vals), (ids.t().contiguous().t(), vals)):
        with pytest.raises(ValueError):
            t.score_topk(None, 5, out=bad)
    with pytest.raises(ValueError):
        t.score_topk(None, 5, ws=torch.empty(1 << 20, dtype=torch.float32, device=dev))
    wide = torch.empty(40, 64, device=dev)
    assert t.score_matrix(out=wide[:, :30]).data_ptr() == wide.data_ptr()              # a column block is written in place
    for bad in (torch.empty(40, 29, device=dev), torch.empty(30, 40, device=dev).t(), torch.empty(40, 30, dtype=torch.float64, device=dev)):
        with pytest.raises(ValueError):
            t.score_matrix(out=bad)
    ug, ig = torch.randn(7, 32, device=dev), torch.randn(7, 32, device=dev)
    um, im = torch.randn(7, 128, device=dev), torch.randn(7, 128, device=dev)
    assert t.score_rows(ug, ig, um, im).shape == (7,)
    for bad in ((ug[:6], ig, um, im), (ug, ig, um[:, :64], im), (ug, ig, um, None)):
        with pytest.raises(ValueError):
            t.score_rows(*bad)
    sd = m.state_dict()
    layers = [(sd["MLP_layers.%d.weight" % i], sd["MLP_layers.%d.bias" % i]) for i in (1, 4, 7)]
    kw = dict(user_gmf=sd["embed_user_GMF.weight"], item_gmf=sd["embed_item_GMF.weight"], user_mlp=sd["embed_user_MLP.weight"],
              item_mlp=sd["embed_item_MLP.weight"], layers=layers, w_predict=sd["predict_layer.weight"], b_predict=sd["predict_layer.bias"])
    ops.Tower(**kw)
    for change in (dict(item_mlp=sd["embed_item_MLP.weight"][:29]), dict(user_gmf=sd["embed_user_GMF.weight"][:, :16]),
                   dict(layers=[layers[0], layers[2], layers[1]]), dict(w_predict=sd["predict_layer.weight"][:, :32]),
                   dict(b_predict=torch.zeros(2, device=dev))):
        with pytest.raises(ValueError):
            ops.Tower(**{**kw, **change})
    sc = torch.randn(6, 50, device=dev)
    with pytest.raises(ValueError):
        ops.topk_rows(sc, 3, exclude=torch.zeros(6, 49, dtype=torch.bool, device=dev))
    torch.cuda.synchronize()


def test_concurrent_callers_on_their_own_streams_share_one_model(g2):
    """SURVEY 8b threading: the library holds no per-call global state, so threads on separate streams may score through the
    same handle at once (every call carves its own workspace; the handle is read-only after creation)."""
    import threading
    m = _ncf(g2)
    ref_ids, ref_vals = m.score_topk(None, 10)
    ref_pairs = m(torch.arange(257).cuda() % 257, torch.arange(257).cuda() % 131)
    torch.cuda.synchronize()
    errors = []

    def worker(seed):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(20):
                    ids, vals = m.score_topk(None, 10)
                    pairs = m(torch.arange(257).cuda() % 257, torch.arange(257).cuda() % 131)
                st.synchronize()
                if not (torch.equal(ids, ref_ids) and torch.equal(vals, ref_vals) and torch.equal(pairs, ref_pairs)):
                    errors.append("thread %d: results differ from the serial call" % seed)
        except Exception as exc:  # noqa: BLE001
            errors.append("thread %d: %r" % (seed, exc))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_autorec_cached_hidden_layer_is_dropped_when_weights_change(g3):
    from recommender_amd.autorec.utils.model import AutoRec
    m = AutoRec(193, 89, 64, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g3, "a").items()})
    m = m.eval().cuda()
    train = torch.from_numpy(g3["a__train"].astype(np.float32)).cuda()
    m.cache_hidden(train)
    before = m.predict_user(5).clone()
    with torch.no_grad():
        m.decoder[0].bias.add_(0.25)
    with pytest.raises(RuntimeError):
        m.predict_user(5)                          # never served from the stale cache
    m.cache_hidden(train)
    after = m.predict_user(5, clamp=None)
    torch.testing.assert_close(after, m.reconstruct_all(train, clamp=None)[5], rtol=1e-5, atol=1e-5)
    assert not torch.equal(before, m.predict_user(5))
