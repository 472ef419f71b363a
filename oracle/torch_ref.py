"""Stock-PyTorch CPU restatement of the reference forward -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

The reference's arithmetic lives in ATen CPU kernels (embedding, cat, addmm,
relu, sigmoid).  This module drives the SAME ATen ops, in the same order, from
a plain state_dict, so that timing it on the GPU box's host cores is the
"reference PyTorch CPU forward" of BASELINE.md section 3 without the reference
source having to travel.  Checked against tests/golden in
tests/test_oracle_golden.py before its timing is trusted.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import torch
import torch.nn.functional as F


def _t(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in sd.items()}


@torch.no_grad()
def ncf_forward(sd, user, item, rating=False):
    """src/helpers/ncf_model.py:205-249 as functional ATen calls (eval mode: Dropout is identity)."""
    parts = []
    if "embed_user_GMF.weight" in sd:
        parts.append(F.embedding(user, sd["embed_user_GMF.weight"]) * F.embedding(item, sd["embed_item_GMF.weight"]))
    if "embed_user_MLP.weight" in sd:
        x = torch.cat((F.embedding(user, sd["embed_user_MLP.weight"]), F.embedding(item, sd["embed_item_MLP.weight"])), -1)
        idx = sorted(int(k.split(".")[1]) for k in sd if k.startswith("MLP_layers.") and k.endswith(".weight"))
        for i in idx:
            x = F.relu(F.linear(x, sd["MLP_layers.%d.weight" % i], sd["MLP_layers.%d.bias" % i]))
        parts.append(x)
    concat = parts[0] if len(parts) == 1 else torch.cat(parts, -1)
    out = F.linear(concat, sd["predict_layer.weight"], sd["predict_layer.bias"]).view(-1)
    if rating:                                   # src/ncf/evaluate_ncf_rating.py:144-146
        out = torch.sigmoid(out) * 4.0 + 1.0
    return out


@torch.no_grad()
def ncf_score_topk(sd, users, k, chunk_users=512):
    """Full-catalog scoring + top-k, chunked at `chunk_users` x catalog (un-chunked OOMs).

    Per user this is inference_engine.py:162-166 (all items, then arg-top-k); torch.topk stands in for
    np.argsort()[::-1][:k] (same result up to tie order).
    """
    sd = _t(sd)
    key = "embed_item_GMF.weight" if "embed_item_GMF.weight" in sd else "embed_item_MLP.weight"
    n_items = sd[key].shape[0]
    items = torch.arange(n_items)
    ids, vals = [], []
    for s in range(0, len(users), chunk_users):
        u = users[s:s + chunk_users]
        sc = ncf_forward(sd, u.repeat_interleave(n_items), items.repeat(len(u))).view(len(u), n_items)
        v, i = torch.topk(sc, k, dim=1)
        ids.append(i); vals.append(v)
    return torch.cat(ids), torch.cat(vals)


@torch.no_grad()
def autorec_forward(sd, x, clamp=True):
    """src/autorec/utils/model.py:31-33 (+ clamp of inference_engine.py:232)."""
    h = torch.sigmoid(F.linear(x, sd["encoder.0.weight"], sd["encoder.0.bias"]))
    r = F.linear(h, sd["decoder.0.weight"], sd["decoder.0.bias"])
    return torch.clamp(r, min=1.0, max=5.0) if clamp else r
