#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports the reference package from /root/reference/src
(read-only) and, for classes that exist only inside scripts/notebooks that do
work at import time, extracts the class definitions with `ast` / the notebook
JSON and executes just those.  Nothing from the reference is written into the
repo except numeric inputs/outputs (fixtures are data, not source).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The fixtures pin `oracle/` (see oracle/reco_oracle.py header) and are the
fixed point every HIP parity test is compared against.

Reference symbols executed here:
  helpers.ncf_model.NCF                      src/helpers/ncf_model.py:16-249
  helpers.ranking_metrics.hit / ndcg         src/helpers/ranking_metrics.py:7-54
  autorec.utils.model.AutoRec                src/autorec/utils/model.py:5-33
  autorec.utils.helper.hit / ndcg (multi-GT) src/autorec/utils/helper.py:176-229
  NCFRating                                  src/ncf/evaluate_ncf_rating.py:66-146
  AutoEncoder / NCF / HybridAutoRecNCF       src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:58-208  (rating variant)
  AutoEncoder / NCF / HybridAutoRecNCF       src/hybridautorecncf/hybridautorecncf.ipynb cells 1-3          (logit variant)
"""
import ast
import collections
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))

from helpers.ncf_model import NCF  # noqa: E402
from helpers.ranking_metrics import hit, ndcg  # noqa: E402
from autorec.utils.model import AutoRec  # noqa: E402

torch.set_num_threads(4)


def classes_from_script(path, names):
    """exec only the named ClassDef nodes of a reference script."""
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "nn": nn, "np": np}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return [ns[n] for n in names]


def functions_from_script(path, names):
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "nn": nn, "np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return [ns[n] for n in names]


def classes_from_notebook(path, cells, names):
    nb = json.load(open(path))
    ns = {"torch": torch, "nn": nn, "np": np}
    for c in cells:
        exec("".join(nb["cells"][c]["source"]), ns)
    return [ns[n] for n in names]


def sd_np(model, prefix=""):
    return {prefix + k.replace(".", "__"): v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def f64(model):
    import copy
    return copy.deepcopy(model).double()


# ---------------------------------------------------------------------------
# G1  real GMF.pth on the 99-negative protocol (cfg1: first 6040 test rows)
# ---------------------------------------------------------------------------
def g1_gmf_real():
    import _codecs
    with torch.serialization.safe_globals([
            NCF, nn.Embedding, nn.Linear, collections.OrderedDict, set, np.dtype,
            np._core.multiarray.scalar, _codecs.encode, type(np.dtype("int32"))]):
        m = torch.load(os.path.join(REF, "models/GMF.pth"), weights_only=True)
    m.eval()
    assert m.model_name == "GMF"
    tr = np.loadtxt(os.path.join(REF, "data/ml-1m.test.rating"), dtype=np.int64, max_rows=6040)
    users, pos = tr[:, 0], tr[:, 1]
    rng = np.random.default_rng(20260220)
    neg = rng.integers(0, 3533, size=(6040, 99))
    items = np.concatenate([pos[:, None], neg], axis=1)          # positive is position 0 (load_all_data.py:49-54)
    uu = np.repeat(users[:, None], 100, axis=1)
    top10 = np.zeros((6040, 10), np.int32)
    hr = np.zeros(6040, np.uint8)
    nd = np.zeros(6040, np.float64)
    scores = np.zeros((6040, 100), np.float32)
    with torch.no_grad():
        for c in range(6040):                                    # ncf.ipynb#cell8 evaluate_metrics, batch of 100
            u = torch.from_numpy(uu[c]); it = torch.from_numpy(items[c])
            pred = m(u, it)
            _, idx = torch.topk(pred, 10)
            rec = torch.take(it, idx).tolist()
            gt = it[0].item()
            hr[c] = hit(gt, rec); nd[c] = ndcg(gt, rec)
            top10[c] = rec; scores[c] = pred.numpy()
        m64 = f64(m)
        s64 = m64(torch.from_numpy(uu.reshape(-1)), torch.from_numpy(items.reshape(-1))).numpy().reshape(6040, 100)
        ka = m(torch.zeros(5, dtype=torch.long), torch.arange(5)).numpy()
    np.savez_compressed(
        os.path.join(OUT, "g1_gmf_real.npz"),
        **sd_np(m), users=users.astype(np.int32), pos=pos.astype(np.int32), neg=neg.astype(np.int16),
        neg_sha=np.frombuffer(hashlib.sha256(neg.astype(np.int64).tobytes()).digest(), np.uint8),
        scores_head=scores[:256], scores64_head=s64[:256], score_sum64=np.float64(s64.sum()),
        score_abs_sum64=np.float64(np.abs(s64).sum()),
        max_abs_f32_vs_f64=np.float64(np.abs(scores - s64).max()),
        top10=top10, hr=hr, ndcg=nd, hr_mean=np.float64(hr.mean()), ndcg_mean=np.float64(nd.mean()),
        known_answer=ka)
    print("G1  HR@10 %.6f NDCG@10 %.6f  (ref f32 vs f64 max abs %.3g)" % (hr.mean(), nd.mean(), np.abs(scores - s64).max()))


# ---------------------------------------------------------------------------
# G2  small NeuMF-end / NeuMF-pre / MLP / GMF built by the reference ctor
# ---------------------------------------------------------------------------
def _liven(model, gen, emb_scale=25.0):
    """reference init is N(0,0.01) embeddings + zero biases: scale so scores are O(1) and biases are exercised."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "embed" in name:
                p.mul_(emb_scale)
            elif name.endswith("bias"):
                p.copy_(torch.empty_like(p).uniform_(-0.2, 0.2, generator=gen))


def g2_neumf_small():
    out = {}
    gen = torch.Generator().manual_seed(1234)
    U, I = 257, 131
    torch.manual_seed(0)
    end = NCF(U, I, 32, 3, 0.1, "NeuMF-end"); _liven(end, gen); end.eval()
    torch.manual_seed(1)
    Up, Ip = 97, 61
    gmf = NCF(Up, Ip, 32, 3, 0.1, "GMF"); _liven(gmf, gen)
    mlp = NCF(Up, Ip, 32, 3, 0.1, "MLP"); _liven(mlp, gen)
    pre = NCF(Up, Ip, 32, 3, 0.1, "NeuMF-pre", gmf, mlp); pre.eval(); gmf.eval(); mlp.eval()
    assert len(pre.state_dict()) == 26
    for tag, m, (nu, ni) in (("end", end, (U, I)), ("pre", pre, (Up, Ip)), ("gmf", gmf, (Up, Ip)), ("mlp", mlp, (Up, Ip))):
        uu, ii = np.meshgrid(np.arange(nu), np.arange(ni), indexing="ij")
        u = torch.from_numpy(uu.reshape(-1)); i = torch.from_numpy(ii.reshape(-1))
        with torch.no_grad():
            s32 = m(u, i).numpy().reshape(nu, ni)
            s64 = f64(m)(u, i).numpy().reshape(nu, ni)
        out.update(sd_np(m, tag + "__"))
        out[tag + "__scores"] = s32
        out[tag + "__scores64"] = s64
        # API recommend path (inference_engine.py:162-166): np.argsort(scores)[::-1][:k]
        top = np.stack([np.argsort(s32[r])[::-1][:10] for r in range(nu)]).astype(np.int32)
        out[tag + "__top10_argsort"] = top
        print("G2 %-4s U=%d I=%d keys=%d |score| mean %.3f  f32-f64 max abs %.3g" % (
            tag, nu, ni, len(m.state_dict()), np.abs(s64).mean(), np.abs(s32 - s64).max()))
    out["key_list_pre"] = np.array(list(pre.state_dict().keys()))
    out["key_list_end"] = np.array(list(end.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "g2_neumf_small.npz"), **out)


# ---------------------------------------------------------------------------
# G2b  other tower shapes (embedding sweep of ncf-embedding.ipynb: factor 8..128; other depths)
# ---------------------------------------------------------------------------
def g2b_other_shapes():
    out = {}
    gen = torch.Generator().manual_seed(4321)
    for tag, (factor, layers, name) in (("f16l2", (16, 2, "NeuMF-end")), ("f64l3", (64, 3, "NeuMF-end")),
                                         ("f8l4", (8, 4, "MLP")), ("f128l1", (128, 1, "NeuMF-end")), ("f64gmf", (64, 3, "GMF"))):
        torch.manual_seed(factor + layers)
        U, I = 71, 53
        m = NCF(U, I, factor, layers, 0.1, name); _liven(m, gen, 25.0 if factor <= 32 else 12.0); m.eval()
        uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
        u = torch.from_numpy(uu.reshape(-1)); i = torch.from_numpy(ii.reshape(-1))
        with torch.no_grad():
            out[tag + "__scores"] = m(u, i).numpy().reshape(U, I)
            out[tag + "__scores64"] = f64(m)(u, i).numpy().reshape(U, I)
        out.update(sd_np(m, tag + "__"))
        out[tag + "__cfg"] = np.array([factor, layers])
        print("G2b %-7s |score| mean %.3f f32-f64 %.3g" % (tag, np.abs(out[tag + "__scores64"]).mean(),
                                                           np.abs(out[tag + "__scores"] - out[tag + "__scores64"]).max()))
    np.savez_compressed(os.path.join(OUT, "g2b_other_shapes.npz"), **out)


# ---------------------------------------------------------------------------
# G2r  NCFRating (sigmoid*4+1 epilogue) small + slice of the real GMF-Rating.pth
# ---------------------------------------------------------------------------
def g2r_rating():
    (NCFRating,) = classes_from_script(os.path.join(REF, "src/ncf/evaluate_ncf_rating.py"), ["NCFRating"])
    out = {}
    gen = torch.Generator().manual_seed(77)
    torch.manual_seed(2)
    U, I = 83, 59
    m = NCFRating(U, I, 32, 3, 0.2, "NeuMF-end"); _liven(m, gen); m.eval()
    uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
    u = torch.from_numpy(uu.reshape(-1)); i = torch.from_numpy(ii.reshape(-1))
    with torch.no_grad():
        out["neumf__scores"] = m(u, i).numpy().reshape(U, I)
        out["neumf__scores64"] = f64(m)(u, i).numpy().reshape(U, I)
    out.update(sd_np(m, "neumf__"))
    # real GMF-Rating.pth (plain 4-key state_dict), sliced to 300 users x 200 items
    sd = torch.load(os.path.join(REF, "models/GMF-Rating.pth"), weights_only=True)
    Us, Is = 300, 200
    g = NCFRating(Us, Is, 32, 3, 0.2, "GMF")
    g.load_state_dict({"embed_user_GMF.weight": sd["embed_user_GMF.weight"][:Us].clone(),
                       "embed_item_GMF.weight": sd["embed_item_GMF.weight"][:Is].clone(),
                       "predict_layer.weight": sd["predict_layer.weight"], "predict_layer.bias": sd["predict_layer.bias"]})
    g.eval()
    uu, ii = np.meshgrid(np.arange(Us), np.arange(Is), indexing="ij")
    u = torch.from_numpy(uu.reshape(-1)); i = torch.from_numpy(ii.reshape(-1))
    with torch.no_grad():
        out["gmfreal__scores"] = g(u, i).numpy().reshape(Us, Is)
        out["gmfreal__scores64"] = f64(g)(u, i).numpy().reshape(Us, Is)
    out.update(sd_np(g, "gmfreal__"))
    np.savez_compressed(os.path.join(OUT, "g2r_rating.npz"), **out)
    print("G2r rating range [%.3f, %.3f]" % (out["gmfreal__scores"].min(), out["gmfreal__scores"].max()))


# ---------------------------------------------------------------------------
# G3  AutoRec: reconstruction, clamp, seen-mask top-K, tie storm
# ---------------------------------------------------------------------------
def _ratings_matrix(rng, users, items, density):
    m = (rng.random((users, items)) < density)
    return (m * rng.integers(1, 6, size=(users, items))).astype(np.float32)


def g3_autorec():
    hit_m, ndcg_m = functions_from_script(os.path.join(REF, "src/autorec/utils/helper.py"), ["hit", "ndcg"])
    out = {}
    rng = np.random.default_rng(3)
    for tag, (U, I, H) in (("a", (193, 89, 64)), ("b", (61, 37, 500))):
        torch.manual_seed(10 + H)
        m = AutoRec(U, I, H, True)
        with torch.no_grad():                                   # autorec-rating.ipynb#cell6 uses xavier_uniform_
            for p in m.parameters():
                if p.dim() == 2:
                    nn.init.xavier_uniform_(p)
            m.decoder[0].bias.add_(3.0)                         # centre outputs in the rating range so clamp has both sides
            m.decoder[0].weight.mul_(4.0)
        m.eval()
        train = _ratings_matrix(rng, U, I, 0.08)                # [users, items]
        test = _ratings_matrix(rng, U, I, 0.03) * (train == 0)
        X = torch.from_numpy(np.ascontiguousarray(train.T))    # item vectors [I, U]  (inference_engine.py:230)
        with torch.no_grad():
            raw = m(X).numpy()
            raw64 = f64(m)(X.double()).numpy()
            clamped = torch.clamp(m(X), min=1.0, max=5.0).numpy()
        # helper.py:275-340 semantics: predictions [U, I], seen-mask, argsort top-k, multi-GT metrics
        pred = clamped.T.copy()
        mask = train > 0
        masked = pred * (~mask).astype(np.float32) - mask.astype(np.float32) * 1e10
        top = np.zeros((U, 10), np.int32); hrs = []; nds = []; has_gt = np.zeros(U, np.uint8)
        for uid in range(U):
            gt = set(np.where(test[uid] > 0)[0])
            t = np.argsort(masked[uid])[-10:][::-1]
            top[uid] = t
            if len(gt):
                has_gt[uid] = 1
                hrs.append(hit_m(gt, t.tolist())); nds.append(ndcg_m(gt, t.tolist()))
        out.update(sd_np(m, tag + "__"))
        out[tag + "__train"] = train.astype(np.uint8); out[tag + "__test"] = test.astype(np.uint8)
        out[tag + "__raw"] = raw; out[tag + "__raw64"] = raw64; out[tag + "__clamped"] = clamped
        out[tag + "__masked"] = masked.astype(np.float32)
        out[tag + "__top10"] = top; out[tag + "__hr_mean"] = np.float64(np.mean(hrs)); out[tag + "__ndcg_mean"] = np.float64(np.mean(nds))
        out[tag + "__hr"] = np.array(hrs, np.float64); out[tag + "__ndcg"] = np.array(nds, np.float64); out[tag + "__has_gt"] = has_gt
        frac_clamped = float(((raw <= 1) | (raw >= 5)).mean())
        print("G3 %s U=%d I=%d H=%d clamped frac %.2f HR %.4f NDCG %.4f f32-f64 %.3g" % (
            tag, U, I, H, frac_clamped, np.mean(hrs), np.mean(nds), np.abs(raw - raw64).max()))
    # tie storm for the API path (inference_engine.py:294,341): Python stable sort, reverse=True, [:k]
    sc = rng.choice(np.array([1.0, 2.5, 5.0, 5.0, 5.0, 4.0], np.float32), size=(7, 400))
    api = np.stack([np.array([i for i, _ in sorted(enumerate(sc[r].tolist()), key=lambda x: x[1], reverse=True)[:100]])
                    for r in range(7)]).astype(np.int32)
    out["tie__scores"] = sc; out["tie__api_top100"] = api
    np.savez_compressed(os.path.join(OUT, "g3_autorec.npz"), **out)


# ---------------------------------------------------------------------------
# G4  Hybrid AutoRec->NCF: rating variant (script) and logit variant (notebook)
# ---------------------------------------------------------------------------
def g4_hybrid():
    out = {}
    rng = np.random.default_rng(4)
    AE, HN, Hy = classes_from_script(os.path.join(REF, "src/hybridautorecncf/evaluate_hybridautorecncf_rating.py"),
                                     ["AutoEncoder", "NCF", "HybridAutoRecNCF"])
    AEn, HNn, Hyn = classes_from_notebook(os.path.join(REF, "src/hybridautorecncf/hybridautorecncf.ipynb"), [1, 2, 3],
                                          ["AutoEncoder", "NCF", "HybridAutoRecNCF"])
    gen = torch.Generator().manual_seed(99)
    for tag, build, (U, I) in (("rating", lambda U, I: Hy(U, I, 64, [128, 64], hidden_dims=(48, 40)), (83, 47)),
                                ("logit", lambda U, I: Hyn(U, I, 64, [128, 64]), (40, 26))):
        torch.manual_seed(5)
        m = build(U, I)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if name.endswith("bias"):
                    p.copy_(torch.empty_like(p).uniform_(-0.1, 0.1, generator=gen))
        m.eval()
        R = _ratings_matrix(rng, U, I, 0.15)
        if tag == "logit":
            R = (R > 0).astype(np.float32)                      # implicit variant trains on binary interactions
        uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
        uid = uu.reshape(-1); iid = ii.reshape(-1)
        uv = torch.from_numpy(R[uid]); iv = torch.from_numpy(np.ascontiguousarray(R.T)[iid])
        with torch.no_grad():
            pred, urec, irec = m(uv, iv, torch.from_numpy(uid), torch.from_numpy(iid))
            p64, _, _ = f64(m)(uv.double(), iv.double(), None, None)
            zu = m.user_autorec.encode(torch.from_numpy(R)).numpy()
            zi = m.item_autorec.encode(torch.from_numpy(np.ascontiguousarray(R.T))).numpy()
            urec_all = m.user_autorec(torch.from_numpy(R))[0].numpy()
            irec_all = m.item_autorec(torch.from_numpy(np.ascontiguousarray(R.T)))[0].numpy()
        assert np.array_equal(urec.numpy()[::I], urec_all)
        out.update(sd_np(m, tag + "__"))
        out[tag + "__R"] = R.astype(np.uint8)
        out[tag + "__pred"] = pred.numpy().reshape(U, I); out[tag + "__pred64"] = p64.numpy().reshape(U, I)
        out[tag + "__zu"] = zu; out[tag + "__zi"] = zi
        out[tag + "__user_recon"] = urec_all; out[tag + "__item_recon"] = irec_all
        print("G4 %-6s U=%d I=%d params=%d pred range [%.3f, %.3f] f32-f64 %.3g" % (
            tag, U, I, sum(p.numel() for p in m.parameters()), pred.min(), pred.max(),
            np.abs(pred.numpy() - p64.numpy()).max()))
    np.savez_compressed(os.path.join(OUT, "g4_hybrid.npz"), **out)


# ---------------------------------------------------------------------------
# G5  hit / ndcg tables (single-GT and multi-GT)
# ---------------------------------------------------------------------------
def g5_metrics():
    hit_m, ndcg_m = functions_from_script(os.path.join(REF, "src/autorec/utils/helper.py"), ["hit", "ndcg"])
    rng = np.random.default_rng(5)
    preds = rng.integers(0, 30, size=(200, 10))                  # duplicates allowed (a negative may equal the positive)
    gts = rng.integers(0, 40, size=200)
    hr = np.array([hit(int(g), p.tolist()) for g, p in zip(gts, preds)], np.float64)
    nd = np.array([float(ndcg(int(g), p.tolist())) for g, p in zip(gts, preds)], np.float64)
    mg = [sorted(set(rng.integers(0, 40, size=rng.integers(0, 6)).tolist())) for _ in range(200)]
    mg_pad = np.full((200, 6), -1, np.int32)
    for r, g in enumerate(mg):
        mg_pad[r, :len(g)] = g
    mhr = np.array([hit_m(set(g), p.tolist()) for g, p in zip(mg, preds)], np.float64)
    mnd = np.array([ndcg_m(set(g), p.tolist()) for g, p in zip(mg, preds)], np.float64)
    np.savez_compressed(os.path.join(OUT, "g5_metrics.npz"), preds=preds.astype(np.int32), gts=gts.astype(np.int32),
                        hr=hr, ndcg=nd, multi_gt=mg_pad, multi_hr=mhr, multi_ndcg=mnd)
    print("G5 hit mean %.3f ndcg mean %.4f" % (hr.mean(), nd.mean()))


# ---------------------------------------------------------------------------
# G6  data files either side of the path: ratings.dat -> dense matrix, train.rating / test.negative parsers
# ---------------------------------------------------------------------------
def g6_files():
    import pandas as pd
    fdir = os.path.join(OUT, "files"); os.makedirs(fdir, exist_ok=True)
    rng = np.random.default_rng(6)
    # ratings.dat with gaps in both id spaces and one duplicated (user, item) pair (last occurrence wins in the reference loop)
    raw_users = np.sort(rng.choice(np.arange(1, 60), 23, replace=False)); raw_items = np.sort(rng.choice(np.arange(1, 200), 41, replace=False))
    rows = []
    for u in raw_users:
        for it in rng.choice(raw_items, rng.integers(3, 12), replace=False):
            rows.append((int(u), int(it), int(rng.integers(1, 6)), int(978300000 + rng.integers(0, 10 ** 6))))
    rows.append((rows[5][0], rows[5][1], 6 - rows[5][2], 978400000))
    with open(os.path.join(fdir, "mini_ratings.dat"), "w", encoding="latin-1") as f:
        for r in rows:
            f.write("%d::%d::%d::%d\n" % r)
    cfg = {"user_num": 25, "item_num": 44}            # a little larger than the id spaces, as AUTOREC_CONFIG is fixed in the reference
    tree = ast.parse(open(os.path.join(REF, "src/api/core/data_loader.py")).read())
    ns = {"np": np, "pd": pd, "torch": torch, "os": os, "Optional": __import__("typing").Optional, "AUTOREC_CONFIG": cfg}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("_create_matrix_from_ratings", "_load_training_file"):
            exec(compile(ast.Module([node], []), "data_loader.py", "exec"), ns)
    df = pd.read_csv(os.path.join(fdir, "mini_ratings.dat"), sep="::", header=None, names=["user_id", "item_id", "rating", "timestamp"],
                     engine="python", encoding="latin-1",
                     dtype={"user_id": np.int32, "item_id": np.int32, "rating": np.float32, "timestamp": np.int32})
    mat = ns["_create_matrix_from_ratings"](df)
    # NCF-format files
    U, I = 19, 37
    train = sorted({(int(rng.integers(0, U)), int(rng.integers(0, I))) for _ in range(160)})
    train.append((U - 1, I - 1))
    with open(os.path.join(fdir, "mini.train.rating"), "w") as f:
        for u, it in train:
            f.write("%d\t%d\n" % (u, it))
    with open(os.path.join(fdir, "mini.test.negative"), "w") as f:
        for c in range(31):
            u, it = int(rng.integers(0, U)), int(rng.integers(0, I))
            f.write("(%d, %d)\t" % (u, it) + "\t".join(str(int(x)) for x in rng.integers(0, I, 9)) + "\n")
    sys.path.insert(0, os.path.join(REF, "src", "ncf"))
    from utils.load_all_data import load_all_data
    train_data, test_data, user_num, item_num, train_mat = load_all_data(os.path.join(fdir, "mini.train.rating"),
                                                                         os.path.join(fdir, "mini.test.negative"))
    binmat = ns["_load_training_file"](os.path.join(fdir, "mini.train.rating"))
    np.savez_compressed(os.path.join(OUT, "g6_files.npz"), ratings_matrix=mat, cfg=np.array([cfg["user_num"], cfg["item_num"]]),
                        train_data=np.array(train_data, np.int32), test_data=np.array(test_data, np.int32),
                        user_num=np.int64(user_num), item_num=np.int64(item_num), train_mat=np.asarray(train_mat.todense(), np.float32),
                        binary_matrix=binmat)
    print("G6 ratings matrix nnz %d, test_data %s, user_num %d item_num %d" % (np.count_nonzero(mat), np.array(test_data).shape, user_num, item_num))


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # regenerate only the named fixtures, e.g. `make_golden.py g2b_other_shapes`
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    g1_gmf_real()
    g2_neumf_small()
    g2b_other_shapes()
    g2r_rating()
    g3_autorec()
    g4_hybrid()
    g5_metrics()
    g6_files()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("%-24s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))
