"""NCF / NeuMF module whose forward runs on the MI355X HIP kernels.

Drop-in for the reference class `helpers.ncf_model.NCF` (src/helpers/ncf_model.py:16-249):
same constructor signature, same attribute and state_dict names (so `load_state_dict` of a
reference checkpoint and unpickling of a reference whole-module `.pth` both work), same
`forward(user: int64[B], item: int64[B]) -> float32[B]`.

What differs is where the arithmetic happens: `forward` hands raw device pointers to
libreco_hip.so (fused gather + factorised layer 1 + MFMA layers 2/3 + GMF + predict, one kernel).
Parameters are ordinary nn.Parameters, only used as storage.

Unpickling note: nn.Module.__setstate__ restores __dict__ without calling __init__, so
nothing here may rely on attributes that __init__ alone creates; the packed-weight cache lives in
a module-level WeakKeyDictionary and is rebuilt whenever a parameter changes.
"""
import weakref

import os

import torch
import torch.nn as nn

from .. import ops

_VARIANTS = ("MLP", "GMF", "NeuMF-end", "NeuMF-pre")
_cache = weakref.WeakKeyDictionary()      # module -> (signature, ops.Tower, device tensors kept alive)


class NCF(nn.Module):
    def __init__(self, user_num, item_num, factor_num, num_layers, dropout, model_name, GMF_model=None, MLP_model=None):
        super().__init__()
        if model_name not in _VARIANTS:
            raise ValueError("model_name must be one of %s, got %r" % (_VARIANTS, model_name))
        self.dropout = dropout
        self.model_name = model_name
        # registered as sub-modules exactly like the reference does (ncf_model.py:50-51), which is why a
        # NeuMF-pre state_dict carries 14 extra GMF_model.* / MLP_model.* entries
        self.GMF_model = GMF_model
        self.MLP_model = MLP_model

        use_gmf = model_name != "MLP"
        use_mlp = model_name != "GMF"
        if use_gmf:
            self.embed_user_GMF = nn.Embedding(user_num, factor_num)
            self.embed_item_GMF = nn.Embedding(item_num, factor_num)
        if use_mlp:
            width = factor_num << (num_layers - 1)            # per-side MLP embedding (ncf_model.py:68)
            self.embed_user_MLP = nn.Embedding(user_num, width)
            self.embed_item_MLP = nn.Embedding(item_num, width)
            stack = []
            fan_in = 2 * width
            for _ in range(num_layers):                       # Dropout, Linear, ReLU triples -> keys MLP_layers.{1,4,7}
                stack += [nn.Dropout(p=dropout), nn.Linear(fan_in, fan_in // 2), nn.ReLU()]
                fan_in //= 2
            self.MLP_layers = nn.Sequential(*stack)
        self.predict_layer = nn.Linear(factor_num * (2 if use_gmf and use_mlp else 1), 1)
        self._init_weight_()

    # -- initialisation (ncf_model.py:118-186) ---------------------------------------------------------------
    def _init_weight_(self):
        if self.model_name == "NeuMF-pre":
            self._init_from_pretrained()
            return
        for name in ("embed_user_GMF", "embed_item_GMF", "embed_user_MLP", "embed_item_MLP"):
            if hasattr(self, name):
                nn.init.normal_(getattr(self, name).weight, std=0.01)
        if hasattr(self, "MLP_layers"):
            for layer in self.MLP_layers:
                if isinstance(layer, nn.Linear):
                    nn.init.xavier_uniform_(layer.weight)
        nn.init.kaiming_uniform_(self.predict_layer.weight, a=1, nonlinearity="sigmoid")
        for layer in self.modules():
            if isinstance(layer, nn.Linear) and layer.bias is not None:
                layer.bias.data.zero_()

    @torch.no_grad()
    def _init_from_pretrained(self):
        g, m = self.GMF_model, self.MLP_model
        if g is None or m is None:
            raise ValueError("NeuMF-pre needs both GMF_model and MLP_model")
        self.embed_user_GMF.weight.copy_(g.embed_user_GMF.weight)
        self.embed_item_GMF.weight.copy_(g.embed_item_GMF.weight)
        self.embed_user_MLP.weight.copy_(m.embed_user_MLP.weight)
        self.embed_item_MLP.weight.copy_(m.embed_item_MLP.weight)
        for mine, theirs in zip(self.MLP_layers, m.MLP_layers):
            if isinstance(mine, nn.Linear):
                mine.weight.copy_(theirs.weight)
                mine.bias.copy_(theirs.bias)
        # predict = 0.5 * [gmf | mlp], bias = mean of the two (ncf_model.py:179-186)
        self.predict_layer.weight.copy_(0.5 * torch.cat([g.predict_layer.weight, m.predict_layer.weight], dim=1))
        self.predict_layer.bias.copy_((g.predict_layer.bias + m.predict_layer.bias) / 2)

    # -- packed weights -------------------------------------------------------------------------------------
    def _own_tensors(self):
        """The 4-12 top-level tensors that feed the kernels; nested GMF_model.* / MLP_model.* are ignored."""
        t = {}
        mods = self._modules
        for name in ("embed_user_GMF", "embed_item_GMF", "embed_user_MLP", "embed_item_MLP"):
            if mods.get(name) is not None:
                t[name] = mods[name].weight
        if mods.get("MLP_layers") is not None:
            t["layers"] = [(l.weight, l.bias) for l in mods["MLP_layers"] if isinstance(l, nn.Linear)]
        t["wp"] = mods["predict_layer"].weight
        t["bp"] = mods["predict_layer"].bias
        return t

    def _rating_epilogue(self):
        """None for the logit model (ncf_model.py:246-249); (lo, hi) for the NCFRating variant."""
        return None

    def _tower(self, device):
        own = self._own_tensors()
        flat = [own[k] for k in own if k != "layers"] + [x for pair in own.get("layers", []) for x in pair]
        sig = (str(device), self._rating_epilogue(), ops.tower_precision()) + tuple((p.data_ptr(), p._version, str(p.device)) for p in flat)
        hit = _cache.get(self)
        if hit is not None and hit[0] == sig:
            # RECO_CHECK_STALE=1 (debug): a write through param.data moves no version counter, so `sig` cannot see it; the handle
            # would go on with stale projections and -- if the embeddings grew -- with layer-1 activations clamped at the old
            # scale.  The check costs a pass over the MLP tables and a sync per call: the handle is rebuilt when it fires.
            if not (os.environ.get("RECO_CHECK_STALE") == "1" and isinstance(hit[1], ops.Tower) and hit[1].scale_is_stale()):
                return hit[1]

        def dev(p):
            return p.detach().to(device=device, dtype=torch.float32)

        rating = self._rating_epilogue()
        layers = own.get("layers", [])
        widths = [w.shape[0] for w, _ in layers]
        # GMF-only: the float4 GMF kernels (any width that is a multiple of 4); NeuMF / MLP factor 32 x 3 layers: the MFMA tower;
        # everything else (other factors and depths, widths that are not multiples of 4): the unfused all-HIP path
        fused = (not layers and own["embed_user_GMF"].shape[1] % 4 == 0) or \
                (widths == [128, 64, 32] and ("embed_user_GMF" not in own or own["embed_user_GMF"].shape[1] == 32))
        tower = (ops.Tower if fused else ops.GenericTower)(
            user_gmf=dev(own["embed_user_GMF"]) if "embed_user_GMF" in own else None,
            item_gmf=dev(own["embed_item_GMF"]) if "embed_item_GMF" in own else None,
            user_mlp=dev(own["embed_user_MLP"]) if "embed_user_MLP" in own else None,
            item_mlp=dev(own["embed_item_MLP"]) if "embed_item_MLP" in own else None,
            layers=[(dev(w), dev(b)) for w, b in own.get("layers", [])],
            w_predict=dev(own["wp"]), b_predict=dev(own["bp"]),
            act="rating" if rating else None, lo=rating[0] if rating else 1.0, hi=rating[1] if rating else 5.0)
        _cache[self] = (sig, tower)
        return tower

    def _compute_device(self, *tensors):
        ops.require_gpu()
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                return t.device
        p = self._modules["predict_layer"].weight
        return p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())

    def _check_mode(self):
        if self.training and float(self.dropout) > 0.0:
            raise RuntimeError("recommender_amd.NCF is an inference engine: call .eval() first "
                               "(training-time Dropout is not implemented on the HIP path)")

    # -- reference interface --------------------------------------------------------------------------------
    def forward(self, user, item):
        """score[b] for the pairs (user[b], item[b]); float32 on the inputs' device (ncf_model.py:188-249)."""
        self._check_mode()
        src_device = user.device if isinstance(user, torch.Tensor) else torch.device("cpu")
        device = self._compute_device(user, item)
        out = self._tower(device).score_pairs(user, item)
        return out if out.device == src_device else out.to(src_device)

    # -- extended entry points -------------------------------------------------------------------
    def score_pairs(self, user, item):
        return self.forward(user, item)

    def score_matrix(self, users=None, candidates=None):
        """scores[u, n] for users x candidates (None = everyone / the full catalog), on the GPU."""
        self._check_mode()
        device = self._compute_device(users, candidates)
        return self._tower(device).score_matrix(users, candidates)

    def score_topk(self, users=None, k=10, candidates=None, exclude=None, check_ids=True, out=None):
        """Per-user top-k over the candidates: (ids int32 [U,k], scores float32 [U,k]) on the GPU.

        ids index into `candidates` (they are item ids when candidates is None).  Replaces the
        model(...) + np.argsort(...)[::-1][:k] pair of inference_engine.py:162-166.
        check_ids=False skips the out-of-range report (one device-to-host sync) for callers that validated the ids.
        """
        self._check_mode()
        device = self._compute_device(users, candidates)
        return self._tower(device).score_topk(users, k, candidates, exclude, check_ids=check_ids, out=out)
