"""Hybrid AutoRec -> NCF model on the MI355X kernels.

Mirrors the classes of src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:58-208 (rating variant:
raw decoder, sigmoid-scaled prediction) and of hybridautorecncf.ipynb cells 1-3 (logit variant: Sigmoid
decoder, logit prediction).  The two share every state_dict key (user_autorec.encoder.{0,3,6}, ...
ncf.mlp.{0,3}, ncf.output), so the variant is chosen by constructor flags, not by the checkpoint.

forward(user_vecs, item_vecs, user_ids, item_ids) -> (pred[B], user_recon, item_recon) keeps the
reference semantics (encoders, decoders and head for every pair of the batch).  `score_all` is the
full-catalog path (cfg4): encode every user and item ONCE, then the fused head over all pairs --
instead of the reference's O(blocks x items) full-model loop (hybridautorecncf.ipynb cell 13).
"""
import os
import weakref

import torch
import torch.nn as nn

from .. import ops

_tower_cache = weakref.WeakKeyDictionary()
_weight_cache = weakref.WeakKeyDictionary()      # AutoEncoder -> {id(Sequential): (signature, [(weight, bias) on device])}
_csr_cache = weakref.WeakKeyDictionary()         # HybridAutoRecNCF -> (attached tensor, its version, (CSR of R, CSR of R^T))


_side_streams = {}                               # device index -> the stream the item encoder runs on beside the user encoder


def _side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        st = _side_streams[idx] = torch.cuda.Stream(device=idx)
    return st


def _f32_on(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class AutoEncoder(nn.Module):
    def __init__(self, input_dim: int, latent_dim: int, hidden_dims=(256, 128), dropout_rate: float = 0.1,
                 sigmoid_output: bool = False):
        super().__init__()
        self.sigmoid_output = sigmoid_output
        enc, width = [], input_dim
        for h in hidden_dims:                                   # Linear, ReLU, Dropout triples -> keys encoder.{0,3}
            enc += [nn.Linear(width, h), nn.ReLU(), nn.Dropout(dropout_rate)]
            width = h
        enc.append(nn.Linear(width, latent_dim))                # bottleneck, no activation -> key encoder.6
        self.encoder = nn.Sequential(*enc)
        dec, width = [], latent_dim
        for h in reversed(hidden_dims):
            dec += [nn.Linear(width, h), nn.ReLU(), nn.Dropout(dropout_rate)]
            width = h
        dec.append(nn.Linear(width, input_dim))
        if sigmoid_output:
            dec.append(nn.Sigmoid())                            # notebook (implicit) variant
        self.decoder = nn.Sequential(*dec)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def _device_weights(self, seq, device):
        """[(weight, bias)] of seq's Linear layers as fp32 tensors on `device`; copies (CPU-resident or non-fp32 parameters)
        are made once and kept until a parameter's version counter changes."""
        linears = [m for m in seq if isinstance(m, nn.Linear)]
        ps = [p for lin in linears for p in (lin.weight, lin.bias)]
        sig = (str(device),) + tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        cache = _weight_cache.setdefault(self, {})
        hit = cache.get(id(seq))
        if hit is None or hit[0] != sig:
            hit = (sig, [(_f32_on(lin.weight, device), _f32_on(lin.bias, device)) for lin in linears])
            cache[id(seq)] = hit
        return hit[1]

    def _run(self, seq, x, final_act, transposed=False):
        """x: dense [batch, in] tensor, or an ops.CsrMatrix of the batch (first layer over the observed entries only)."""
        device = x.device
        weights = self._device_weights(seq, device)
        n = 0
        while n < len(weights):
            w, b = weights[n]
            last = n + 1 == len(weights)
            act = final_act if last else "relu"
            if n == 0 and isinstance(x, ops.CsrMatrix):
                cache = _weight_cache.setdefault(self, {})
                wt = cache.get(("T", id(seq)))
                if wt is None or wt[0] is not w:                       # one transposed copy per weight version
                    wt = (w, ops.transpose(w))
                    cache[("T", id(seq))] = wt
                x = ops.csr_linear(x, wt[1], b, act=act)
                n += 1
            elif (not last and not (transposed and n == 0) and w.shape[1] <= 512
                  and ops.mlp2_supported(w.shape[1], w.shape[0], weights[n + 1][0].shape[0])):
                # two narrow layers in one launch, the intermediate stays in LDS (reco_mlp2_f32; same bits as two linear calls)
                w3, b3 = weights[n + 1]
                x = ops.mlp2(x, w, b, w3, b3, "relu", final_act if n + 2 == len(weights) else "relu")
                n += 2
            else:
                x = ops.linear(x, w, b, act=act, x_transposed=(transposed and n == 0))
                n += 1
        return x

    def _prep(self, x):
        ops.require_gpu()
        if any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.modules()) and self.training:
            raise RuntimeError("recommender_amd AutoEncoder is inference-only: call .eval() first")
        device = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        return x.to(device=device, dtype=torch.float32)

    def encode(self, x, transposed=False):
        """transposed=True: x is [input_dim, batch] (e.g. the rating matrix itself for the item encoder) and is read
        transposed by the first layer's GEMM instead of being copied.  x may also be an ops.CsrMatrix of the batch."""
        if isinstance(x, ops.CsrMatrix):
            if any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.modules()) and self.training:
                raise RuntimeError("recommender_amd AutoEncoder is inference-only: call .eval() first")
            return self._run(self.encoder, x, None)
        src = x.device
        z = self._run(self.encoder, self._prep(x), None, transposed)
        return z if z.device == src else z.to(src)

    def decode(self, z):
        src = z.device
        r = self._run(self.decoder, self._prep(z), "sigmoid" if self.sigmoid_output else None)
        return r if r.device == src else r.to(src)

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z), z


class NCF(nn.Module):
    """The hybrid's scoring head: GMF product + MLP over the two latents, one output unit."""

    def __init__(self, latent_dim, mlp_layers=(128, 64), dropout_rate=0.1, rating_min=1.0, rating_max=5.0,
                 rating_output: bool = True):
        super().__init__()
        self.rating_min = rating_min
        self.rating_max = rating_max
        self.rating_output = rating_output
        stack, width = [], 2 * latent_dim
        for h in mlp_layers:                                    # keys mlp.{0,3}
            stack += [nn.Linear(width, h), nn.ReLU(), nn.Dropout(dropout_rate)]
            width = h
        self.mlp = nn.Sequential(*stack)
        self.output = nn.Linear(latent_dim + (mlp_layers[-1] if len(mlp_layers) else 2 * latent_dim), 1)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _fused_shape(self):
        """The MFMA head covers the shipped shape: latent 64, MLP 128 -> 128 -> 64 (hybridautorecncf notebooks); any other latent
        width / layer list runs on the unfused all-HIP path (ops.GenericTower over this call's latents)."""
        linears = [m for m in self.mlp if isinstance(m, nn.Linear)]
        return [l.out_features for l in linears] == [128, 64] and self.output.in_features == 64 + 64

    def _generic(self, zu, zi):
        linears = [m for m in self.mlp if isinstance(m, nn.Linear)]
        if not linears:
            raise ValueError("recommender_amd hybrid NCF: an empty mlp_layers list is not supported")
        dev = zu.device
        return ops.GenericTower(user_gmf=zu, item_gmf=zi, user_mlp=zu, item_mlp=zi,
                                layers=[(_f32_on(l.weight, dev), _f32_on(l.bias, dev)) for l in linears],
                                w_predict=_f32_on(self.output.weight, dev), b_predict=_f32_on(self.output.bias, dev),
                                act="rating" if self.rating_output else None, lo=float(self.rating_min), hi=float(self.rating_max))

    def _tower(self, device):
        linears = [m for m in self.mlp if isinstance(m, nn.Linear)]
        ps = [p for lin in linears for p in (lin.weight, lin.bias)] + [self.output.weight, self.output.bias]
        sig = (str(device), bool(self.rating_output), float(self.rating_min), float(self.rating_max), ops.tower_precision()) + tuple(
            (p.data_ptr(), p._version, str(p.device)) for p in ps)
        hit = _tower_cache.get(self)
        if hit is not None and hit[0] == sig:
            return hit[1]
        latent = self.output.in_features - linears[-1].out_features
        tower = ops.Tower(layers=[(_f32_on(l.weight, device), _f32_on(l.bias, device)) for l in linears],
                          w_predict=_f32_on(self.output.weight, device), b_predict=_f32_on(self.output.bias, device),
                          dg=latent, dm=latent, act="rating" if self.rating_output else None,
                          lo=float(self.rating_min), hi=float(self.rating_max))
        _tower_cache[self] = (sig, tower)
        return tower

    def forward(self, user_z, item_z):
        ops.require_gpu()
        src = user_z.device
        device = src if user_z.is_cuda else torch.device("cuda", torch.cuda.current_device())
        zu = _f32_on(user_z, device); zi = _f32_on(item_z, device)
        if zu.shape != zi.shape:
            raise ValueError("user_z and item_z must have the same shape")
        if self._fused_shape():
            out = self._tower(device).score_rows(zu, zi, zu, zi)
        else:
            rows = torch.arange(zu.shape[0], device=device)
            out = self._generic(zu, zi).score_pairs(rows, rows, check_ids=False)
        return out if out.device == src else out.to(src)

    def score_matrix(self, zu, zi, k=None, exclude=None, seen=None):
        """pred[u, i] for every (row of zu, row of zi) -- the full-catalog form of forward -- or, with k, the per-user top-k
        (ids = row numbers of zi).  Runs on the cached head (weights packed once, keyed on parameter versions); the latents are
        handed to the kernels per call, so nothing is allocated besides the outputs and nothing synchronises."""
        ops.require_gpu()
        device = zu.device if zu.is_cuda else torch.device("cuda", torch.cuda.current_device())
        zu = _f32_on(zu, device); zi = _f32_on(zi, device)
        if not self._fused_shape():
            tower = self._generic(zu, zi)
            if k is None:
                return tower.score_matrix(check_ids=False)
            if seen is None:
                return tower.score_topk(None, k, None, exclude, check_ids=False)
            if exclude is not None:
                raise ValueError("give the seen items as a mask or as the rating matrix, not both")
            return ops.topk_rows_seen(tower.score_matrix(check_ids=False), k, seen)
        tower = self._tower(device)
        if k is None:
            return tower.score_matrix_rows(zu, zi, zu, zi)
        return tower.score_topk_rows(zu, zi, zu, zi, k, exclude, seen=seen)


class HybridAutoRecNCF(nn.Module):
    def __init__(self, num_users, num_items, latent_dim, mlp_layers, hidden_dims=(256, 128), dropout_rate=0.1,
                 rating_output: bool = True):
        """rating_output=True: the `-Rating` variant (raw decoders, sigmoid*4+1 prediction);
        rating_output=False: HybridAutoRecNCF.pth / `-Ranking` (Sigmoid decoders, logit prediction)."""
        super().__init__()
        self.user_autorec = AutoEncoder(num_items, latent_dim, hidden_dims, dropout_rate, sigmoid_output=not rating_output)
        self.item_autorec = AutoEncoder(num_users, latent_dim, hidden_dims, dropout_rate, sigmoid_output=not rating_output)
        self.ncf = NCF(latent_dim, mlp_layers, dropout_rate, rating_output=rating_output)

    def forward(self, user_vecs, item_vecs, user_ids=None, item_ids=None):
        user_recon, user_z = self.user_autorec(user_vecs)
        item_recon, item_z = self.item_autorec(item_vecs)
        return self.ncf(user_z, item_z), user_recon, item_recon

    def attach_ratings(self, rating_matrix):
        """Keeps this [users, items] rating matrix sparse on the device (CSR of the matrix for the user encoder, of its
        transpose for the item encoder).  score_all() handed the same tensor afterwards runs both first encoder
        layers over the observed entries only."""
        ops.require_gpu()
        device = rating_matrix.device if rating_matrix.is_cuda else torch.device("cuda", torch.cuda.current_device())
        R = rating_matrix.to(device=device, dtype=torch.float32).contiguous()
        wide_u = next(m for m in self.user_autorec.encoder if isinstance(m, nn.Linear)).out_features
        wide_i = next(m for m in self.item_autorec.encoder if isinstance(m, nn.Linear)).out_features
        pair = (ops.CsrMatrix.from_dense(R, k_ranges=ops.csr_k_ranges(R.shape[1], wide_u)),
                ops.CsrMatrix.from_dense(R, transpose=True, k_ranges=ops.csr_k_ranges(R.shape[0], wide_i)))
        # the entry holds the caller's tensor and is matched by identity + version counter: an address alone can be recycled by the
        # caching allocator for the next same-shaped matrix (whose version counter starts at 0 again)
        _csr_cache[self] = (rating_matrix, rating_matrix._version, pair)
        return pair

    def score_all(self, rating_matrix, k=None, exclude=None, mask_seen=False):
        """Full-catalog joint scoring (cfg4): pred[users, items], or per-user top-k when k is given.
        mask_seen=True (with k): items the user rated (rating_matrix > 0) are pushed to -1e10 before the select, read straight off
        the rating matrix inside the kernel (helper.py:272,313) -- no mask tensor is formed."""
        ops.require_gpu()
        device = rating_matrix.device if rating_matrix.is_cuda else torch.device("cuda", torch.cuda.current_device())
        R = rating_matrix.to(device=device, dtype=torch.float32).contiguous()
        hit = _csr_cache.get(self)
        sparse = hit is not None and hit[0] is rating_matrix and hit[1] == rating_matrix._version
        xu = hit[2][0] if sparse else R
        xi = hit[2][1] if sparse else R
        # The two encoders are independent chains of short launches (a CSR or split-K first layer, then the narrow layers): the
        # item side runs on a second stream beside the user side and joins before the head (fork / join by events, so the pattern
        # also captures into a HIP graph).  RECO_HYBRID_STREAMS=0 keeps both on the caller's stream.
        if os.environ.get("RECO_HYBRID_STREAMS", "1") != "0":
            cur = torch.cuda.current_stream(device)
            side = _side_stream(device)
            fork = torch.cuda.Event()
            fork.record(cur)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                zi = self.item_autorec.encode(xi) if sparse else self.item_autorec.encode(xi, transposed=True)
                join = torch.cuda.Event()
                join.record(side)
            zu = self.user_autorec.encode(xu)
            cur.wait_event(join)
            zi.record_stream(cur)              # allocated on the side stream's pool, consumed on the caller's
        else:
            zu = self.user_autorec.encode(xu)
            zi = self.item_autorec.encode(xi) if sparse else self.item_autorec.encode(xi, transposed=True)
        return self.ncf.score_matrix(zu, zi, k, exclude, seen=R if (mask_seen and k is not None) else None)
