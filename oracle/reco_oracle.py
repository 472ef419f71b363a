"""CPU oracle for the recommender scoring path -- TEST INFRASTRUCTURE ONLY.

This is a numpy restatement of the reference's forward/scoring/top-K algorithm.
It is the checker for the HIP kernels; it is never the product path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned: every function below is checked in tests/test_oracle_golden.py against
the fixtures in tests/golden/*.npz, which were produced by running the
reference itself in the build container (tests/golden/make_golden.py).

All functions take a plain dict of numpy arrays keyed by the reference's
state_dict names, and a `dtype` (np.float32 restates the reference's fp32 math
op-for-op in numpy; np.float64 is the high-precision arbiter).
"""
import numpy as np


# --------------------------------------------------------------------------
# NCF / NeuMF            reference: src/helpers/ncf_model.py:188-249
# --------------------------------------------------------------------------
def ncf_model_name(sd):
    """GMF / MLP / NeuMF from the key set (ncf_model.py:60-70,106-113)."""
    has_g = "embed_user_GMF.weight" in sd
    has_m = "embed_user_MLP.weight" in sd
    if has_g and has_m:
        return "NeuMF"
    return "GMF" if has_g else "MLP"


def _mlp_layers(sd):
    """Linear layers of MLP_layers in order: indices 1,4,7,... (ncf_model.py:84-99)."""
    idx = sorted(int(k.split(".")[1]) for k in sd if k.startswith("MLP_layers.") and k.endswith(".weight"))
    return [(sd["MLP_layers.%d.weight" % i], sd["MLP_layers.%d.bias" % i]) for i in idx]


def ncf_forward(sd, user, item, dtype=np.float32, rating=False, rating_min=1.0, rating_max=5.0):
    """score[b] for pairs (user[b], item[b]).

    ncf_model.py:205-249: GMF product (:211), concat user-half first (:224),
    Linear+ReLU x num_layers with ReLU after the last layer too (:227, :84-99),
    concat GMF-half first (:240), predict layer (:246), NO sigmoid.
    rating=True adds the NCFRating epilogue sigmoid(logit)*range+min
    (src/ncf/evaluate_ncf_rating.py:144-146).
    Out-of-range ids raise IndexError like nn.Embedding on CPU.
    """
    user = np.asarray(user, np.int64)
    item = np.asarray(item, np.int64)
    name = ncf_model_name(sd)
    parts = []
    if name != "MLP":
        eu = sd["embed_user_GMF.weight"].astype(dtype)
        ei = sd["embed_item_GMF.weight"].astype(dtype)
        _check_ids(user, eu.shape[0]); _check_ids(item, ei.shape[0])
        parts.append(eu[user] * ei[item])
    if name != "GMF":
        eu = sd["embed_user_MLP.weight"].astype(dtype)
        ei = sd["embed_item_MLP.weight"].astype(dtype)
        _check_ids(user, eu.shape[0]); _check_ids(item, ei.shape[0])
        x = np.concatenate([eu[user], ei[item]], axis=-1)
        for w, b in _mlp_layers(sd):
            x = np.maximum(x @ w.astype(dtype).T + b.astype(dtype), 0)
        parts.append(x)
    concat = np.concatenate(parts, axis=-1)
    logit = concat @ sd["predict_layer.weight"].astype(dtype).reshape(-1) + sd["predict_layer.bias"].astype(dtype)[0]
    if rating:
        return _sigmoid(logit) * dtype(rating_max - rating_min) + dtype(rating_min)
    return logit


def ncf_score_all(sd, users, dtype=np.float32, chunk=256, **kw):
    """scores[len(users), n_items]: full-catalog scoring (inference_engine.py:149-164 with all items)."""
    key = "embed_item_GMF.weight" if "embed_item_GMF.weight" in sd else "embed_item_MLP.weight"
    n_items = sd[key].shape[0]
    users = np.asarray(users, np.int64)
    out = np.empty((len(users), n_items), dtype)
    items = np.arange(n_items, dtype=np.int64)
    for s in range(0, len(users), chunk):
        u = users[s:s + chunk]
        out[s:s + chunk] = ncf_forward(sd, np.repeat(u, n_items), np.tile(items, len(u)), dtype, **kw).reshape(len(u), n_items)
    return out


def _check_ids(ids, n):
    if ids.size and (ids.min() < 0 or ids.max() >= n):
        raise IndexError("index out of range in self")


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x)) if x.dtype == np.float64 else (np.float32(1.0) / (np.float32(1.0) + np.exp(-x))).astype(np.float32)


# --------------------------------------------------------------------------
# AutoRec                reference: src/autorec/utils/model.py:25-33
# --------------------------------------------------------------------------
def autorec_forward(sd, x, dtype=np.float32, clamp=None):
    """decoder(sigmoid(encoder(x))) (model.py:31-33); clamp=(1,5) adds inference_engine.py:232."""
    x = np.asarray(x).astype(dtype)
    h = _sigmoid(x @ sd["encoder.0.weight"].astype(dtype).T + sd["encoder.0.bias"].astype(dtype))
    r = h @ sd["decoder.0.weight"].astype(dtype).T + sd["decoder.0.bias"].astype(dtype)
    if clamp is not None:
        r = np.clip(r, dtype(clamp[0]), dtype(clamp[1]))
    return r


def mask_seen(pred, seen):
    """src/autorec/utils/helper.py:313: pred*(~seen) - seen*1e10 (fp32)."""
    seen = np.asarray(seen, bool)
    return (pred * (~seen).astype(pred.dtype) - seen.astype(pred.dtype) * pred.dtype.type(1e10)).astype(pred.dtype)


# --------------------------------------------------------------------------
# Hybrid AutoRec -> NCF  reference: src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:58-208
# --------------------------------------------------------------------------
def _seq_linear_relu(sd, prefix, x, dtype, last_act=None):
    """nn.Sequential of Linear/ReLU/Dropout triples with a final bare Linear (:78-89, :94-106)."""
    idx = sorted(int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight"))
    for n, i in enumerate(idx):
        x = x @ sd["%s%d.weight" % (prefix, i)].astype(dtype).T + sd["%s%d.bias" % (prefix, i)].astype(dtype)
        if n + 1 < len(idx):
            x = np.maximum(x, 0)
    if last_act == "sigmoid":
        x = _sigmoid(x)
    return x


def hybrid_encode(sd, which, x, dtype=np.float32):
    """AutoEncoder.encode (:116-117): no activation on the bottleneck."""
    return _seq_linear_relu(sd, which + ".encoder.", np.asarray(x).astype(dtype), dtype)


def hybrid_decode(sd, which, z, dtype=np.float32, sigmoid_out=False):
    """AutoEncoder.decode (:119-120); the notebook (logit) variant ends in Sigmoid (hybridautorecncf.ipynb#cell1)."""
    return _seq_linear_relu(sd, which + ".decoder.", z, dtype, "sigmoid" if sigmoid_out else None)


def hybrid_head(sd, zu, zi, dtype=np.float32, rating=True, rating_min=1.0, rating_max=5.0):
    """NCF head (:167-184): g = zu*zi; m = mlp([zu|zi]) with ReLU after EVERY mlp Linear; logit = output([g|m]).

    rating=True: sigmoid(logit)*range+min (:181-182); rating=False: logit (hybridautorecncf.ipynb#cell2).
    """
    zu = zu.astype(dtype); zi = zi.astype(dtype)
    g = zu * zi
    x = np.concatenate([zu, zi], axis=1)
    idx = sorted(int(k.split(".")[2]) for k in sd if k.startswith("ncf.mlp.") and k.endswith(".weight"))
    for i in idx:
        x = np.maximum(x @ sd["ncf.mlp.%d.weight" % i].astype(dtype).T + sd["ncf.mlp.%d.bias" % i].astype(dtype), 0)
    logit = np.concatenate([g, x], axis=1) @ sd["ncf.output.weight"].astype(dtype).reshape(-1) + sd["ncf.output.bias"].astype(dtype)[0]
    if rating:
        return _sigmoid(logit) * dtype(rating_max - rating_min) + dtype(rating_min)
    return logit


def hybrid_score_all(sd, R, dtype=np.float32, rating=True):
    """pred[U, I] for every (user, item): encoders once, head over all pairs (cfg4)."""
    R = np.asarray(R).astype(dtype)
    zu = hybrid_encode(sd, "user_autorec", R, dtype)
    zi = hybrid_encode(sd, "item_autorec", np.ascontiguousarray(R.T), dtype)
    U, I = R.shape
    uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
    return hybrid_head(sd, zu[uu.reshape(-1)], zi[ii.reshape(-1)], dtype, rating).reshape(U, I)


# --------------------------------------------------------------------------
# top-K and ranking metrics
# --------------------------------------------------------------------------
def topk_rows(scores, k):
    """Build rule: descending score, ascending candidate position among equals.

    Exactly reproduces the API AutoRec path (Python stable sort with reverse=True then [:k],
    inference_engine.py:294,341) and is a valid answer for np.argsort(...)[::-1][:k] (:166),
    torch.topk (ncf.ipynb#cell8) and np.argsort(p)[-k:][::-1] (helper.py:335), whose tie order is
    unspecified.  NaN sorts first (numpy puts NaN last ascending; torch.topk treats it as largest).
    Returns (ids int32 [rows,k], vals [rows,k]).
    """
    scores = np.asarray(scores)
    rows, n = scores.shape
    k = min(k, n)
    key = np.where(np.isnan(scores), np.inf, scores)
    nan_first = np.isnan(scores)
    order = np.lexsort((np.broadcast_to(np.arange(n), scores.shape), -key, ~nan_first), axis=1)[:, :k]
    return order.astype(np.int32), np.take_along_axis(scores, order, axis=1)


def hit(gt_item, pred_items):
    """src/helpers/ranking_metrics.py:7-23."""
    return 1 if gt_item in list(pred_items) else 0


def ndcg(gt_item, pred_items):
    """src/helpers/ranking_metrics.py:26-54: first occurrence, 1/log2(pos+2) in float64."""
    pred_items = list(pred_items)
    if gt_item in pred_items:
        return float(np.reciprocal(np.log2(pred_items.index(gt_item) + 2)))
    return 0.0


def hit_multi(gt_items, pred_items):
    """src/autorec/utils/helper.py:176-192."""
    if len(gt_items) == 0:
        return 0.0
    pred_items = list(pred_items)
    return 1.0 if any(i in pred_items for i in gt_items) else 0.0


def ndcg_multi(gt_items, pred_items):
    """src/autorec/utils/helper.py:195-229."""
    if len(gt_items) == 0:
        return 0.0
    gt = set(int(g) for g in gt_items)
    dcg = 0.0
    for idx, it in enumerate(pred_items):
        if int(it) in gt:
            dcg += 1.0 / np.log2(idx + 2)
    idcg = sum(1.0 / np.log2(i + 2) for i in range(min(len(gt), len(pred_items))))
    return 0.0 if idcg == 0 else dcg / idcg


def protocol_metrics(top_items, gt_items):
    """mean HR@K / NDCG@K over test cases (ncf.ipynb#cell8): top_items [cases,K] item ids, gt_items [cases]."""
    hr = np.array([hit(int(g), t.tolist()) for g, t in zip(gt_items, top_items)], np.float64)
    nd = np.array([ndcg(int(g), t.tolist()) for g, t in zip(gt_items, top_items)], np.float64)
    return hr, nd
