"""Item-based / user-based AutoRec whose forward runs on the MI355X fp32-MFMA GEMM kernel.

Drop-in for `autorec.utils.model.AutoRec` (src/autorec/utils/model.py:5-33): same constructor, same
state_dict keys (encoder.0.{weight,bias}, decoder.0.{weight,bias}), same
forward(x: float32[B, n_in]) -> float32[B, n_in].  Sigmoid and clamp are GEMM epilogues.

Extended entry points cover what the reference's engines build on top of forward with Python loops
(SURVEY 3.2 / 3.4 / 8f-N1): whole-matrix reconstruction, masked per-user top-K, and a cached-hidden
serving path that turns a /autorec/recommend request into one 500-wide GEMV row.
"""
import weakref

import torch
import torch.nn as nn

from ... import ops

_dev_cache = weakref.WeakKeyDictionary()
_csr_cache = weakref.WeakKeyDictionary()         # model -> (key of the attached rating matrix, ops.CsrMatrix of the encoder's batch)


class _Attached:
    """A rating matrix whose CSR form is kept: the entry HOLDS the tensor that was handed in and is matched by identity + version
    counter.  (Keyed on data_ptr alone, a freed matrix's address could be handed by the caching allocator to the next same-shaped
    matrix, whose fresh version counter 0 would match too: the encoder would silently run on the old matrix's CSR.)"""

    def __init__(self, source, csr):
        self.source, self.version, self.csr = source, source._version, csr

    def matches(self, t):
        return t is self.source and t._version == self.version


class AutoRec(nn.Module):
    def __init__(self, num_users: int, num_items: int, num_hidden_units: int, item_based: bool = True) -> None:
        super().__init__()
        self.num_users = num_users
        self.num_items = num_items
        self.num_hidden_units = num_hidden_units
        self.item_based = item_based
        width = num_users if item_based else num_items        # an item vector holds one rating per user
        self.encoder = nn.Sequential(nn.Linear(width, num_hidden_units), nn.Sigmoid())
        self.decoder = nn.Sequential(nn.Linear(num_hidden_units, width))

    # -- weights on the compute device ------------------------------------------------------------------------
    def _weights(self, device):
        ps = (self.encoder[0].weight, self.encoder[0].bias, self.decoder[0].weight, self.decoder[0].bias)
        sig = (str(device),) + tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        hit = _dev_cache.get(self)
        if hit is None or hit[0] != sig:
            hit = (sig, tuple(p.detach().to(device=device, dtype=torch.float32).contiguous() for p in ps), {})
            _dev_cache[self] = hit
        return hit

    @staticmethod
    def _device_of(x):
        ops.require_gpu()
        return x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())

    # -- N2: the rating matrix kept sparse on the device ------------------------------------------------------
    def attach_ratings(self, train_matrix):
        """Builds (once) the CSR form of the encoder's batch for this [users, items] rating matrix -- for the item-based
        model the transposed matrix, row i = item i's ratings -- and keeps it on the device.  Every later call that is
        handed the same tensor (reconstruct_all, recommend_topk, cache_hidden) then runs the encoder over the observed
        entries only (reco_csr_linear_f32); zeros contribute nothing to an fp32 sum, so the result is the dense layer's with
        its zero terms left out.  Replaces the dense float matrix that src/api/core/data_loader.py:96-103,149-161 builds and
        ships to the device as the encoder's only input form.  Returns the CsrMatrix."""
        device = self._device_of(train_matrix)
        m = train_matrix.to(device=device, dtype=torch.float32).contiguous()
        k_dim = m.shape[0] if self.item_based else m.shape[1]
        csr = ops.CsrMatrix.from_dense(m, transpose=self.item_based, k_ranges=ops.csr_k_ranges(k_dim, self.num_hidden_units))
        # matched later against the tensor the CALLER holds (a CPU or non-fp32 matrix was copied above: the copy dies here)
        _csr_cache[self] = _Attached(train_matrix, csr)
        return csr

    def _attached(self, given):
        """The CSR form of `given` if that very tensor (unchanged since) was attached, else None."""
        hit = _csr_cache.get(self)
        return hit.csr if hit is not None and hit.matches(given) else None

    def _encoder_t(self, device):
        """encoder weight as [n_in, hidden] (the sparse kernel reads one contiguous row per observed entry), per weight version"""
        _, (we, be, wd, bd), extra = self._weights(device)
        if "WeT" not in extra:
            extra["WeT"] = ops.transpose(we)
        return extra["WeT"]

    def _hidden_of(self, m, device, given=None):
        """H = sigmoid(encoder(batch)) [items, hidden] for the item-based model and the [users, items] matrix m (`given`: the
        caller's tensor m was made from, for the attach_ratings match)."""
        we, be, wd, bd = self._weights(device)[1]
        csr = self._attached(m if given is None else given)
        if csr is not None:
            return ops.csr_linear(csr, self._encoder_t(device), be, act="sigmoid")
        return ops.linear(m, we, be, act="sigmoid", x_transposed=True)

    # -- reference interface ----------------------------------------------------------------------------------
    def forward(self, input_data: torch.Tensor) -> torch.Tensor:
        return self.reconstruct(input_data, clamp=None)

    # -- extended ---------------------------------------------------------------------------------------------
    def reconstruct(self, input_data, clamp=(1.0, 5.0)):
        """decoder(sigmoid(encoder(x))) with the engines' clamp fused (inference_engine.py:231-232)."""
        device = self._device_of(input_data)
        x = input_data.to(device=device, dtype=torch.float32)
        squeeze = x.dim() == 1
        if squeeze:
            x = x.unsqueeze(0)
        we, be, wd, bd = self._weights(device)[1]
        out = ops.autorec_forward(x, we, be, wd, bd, clamp)
        if squeeze:
            out = out.squeeze(0)
        return out if out.device == input_data.device else out.to(input_data.device)

    def reconstruct_all(self, train_matrix, clamp=(1.0, 5.0)):
        """R_hat[users, items] for a [users, items] rating matrix, in one batched forward.

        Replaces the 3,706-iteration loop of helper.py:281-292 / inference_engine.py:268-272.
        Item-based: the batch is the transposed matrix (row i = item i's ratings from all users) and the
        result is transposed back, so R_hat[u, i] = forward(train[:, i])[u].  The returned tensor is a [users, items] view
        whose row stride is items rounded up to 32 (call .contiguous() if a dense buffer is needed).
        """
        device = self._device_of(train_matrix)
        m = train_matrix.to(device=device, dtype=torch.float32).contiguous()
        if not self.item_based:
            csr = self._attached(train_matrix)
            if csr is None:
                return self.reconstruct(m, clamp)
            we, be, wd, bd = self._weights(device)[1]
            h = ops.csr_linear(csr, self._encoder_t(device), be, act="sigmoid")
            act = (None, 0.0, 0.0) if clamp is None else ("clamp", float(clamp[0]), float(clamp[1]))
            return ops.linear(h, wd, bd, act=act[0], lo=act[1], hi=act[2])
        # no transposed copies: the encoder reads R as the [K = users, M = items] operand (or, with attach_ratings, only its
        # observed entries), and the decoder is run with its operands swapped, R_hat^T[u, i] = Wd[u, :] . H[i, :] + bd[u],
        # which lands directly in [users, items]
        we, be, wd, bd = self._weights(device)[1]
        h = self._hidden_of(m, device, train_matrix)
        act = (None, 0.0, 0.0) if clamp is None else ("clamp", float(clamp[0]), float(clamp[1]))
        # rows padded to whole 128-byte lines (3706 -> 3712 floats): aligned stores in the GEMM epilogue, float4 reads in the
        # top-k scan; the result is the [users, items] view of that buffer
        n_users, n_items = wd.shape[0], h.shape[0]
        buf = torch.empty((n_users, (n_items + 31) // 32 * 32), dtype=torch.float32, device=device)
        return ops.linear(wd, h, bd, act=act[0], lo=act[1], hi=act[2], bias_per_row=True, out=buf[:, :n_items])

    def recommend_topk(self, train_matrix, k=10, users=None, mask_seen=True, clamp=(1.0, 5.0)):
        """Per-user top-k of the reconstruction, seen items pushed to -1e10 as helper.py:313.

        Returns (item ids int32 [U,k], scores [U,k]); order: descending score, ascending item id on ties.
        """
        device = self._device_of(train_matrix)
        m = train_matrix.to(device=device, dtype=torch.float32).contiguous()
        pred = self.reconstruct_all(train_matrix, clamp)
        # the seen mask (train > 0, helper.py:272) is read off the rating matrix inside the select and the user subset is an
        # index list handed to the kernel: no mask tensor, no gathered copies
        return ops.topk_rows_seen(pred, k, m if mask_seen else None, users)

    # -- N1: cached hidden layer for serving ------------------------------------------------------------------
    def cache_hidden(self, train_matrix):
        """H = sigmoid(encoder(X^T)) [items, hidden] for the item-based model; cached until weights change."""
        if not self.item_based:
            raise ValueError("cache_hidden is for the item-based model")
        device = self._device_of(train_matrix)
        sig, (we, be, wd, bd), extra = self._weights(device)
        x = train_matrix.to(device=device, dtype=torch.float32).contiguous()
        extra["H"] = self._hidden_of(x, device, train_matrix)
        return extra["H"]

    def predict_user(self, user_id, item_ids=None, clamp=(1.0, 5.0)):
        """Scores of one user for all (or the given) items from the cached H: H . Wd[u] + bd[u].

        Same numbers as inference_engine.py:266-272 (one full forward per item, keep column user_id).
        """
        hit = _dev_cache.get(self)
        if hit is not None:
            hit = self._weights(hit[1][0].device)           # a weight update since cache_hidden drops the cached H with the old copy
        if hit is None or "H" not in hit[2]:
            raise RuntimeError("call cache_hidden(train_matrix) first (and again after the weights change)")
        _, (we, be, wd, bd), extra = hit
        if not 0 <= int(user_id) < wd.shape[0]:
            raise IndexError("user id %d out of range [0, %d)" % (int(user_id), wd.shape[0]))
        H = extra["H"]
        if item_ids is not None:
            idx = torch.as_tensor(item_ids, dtype=torch.int64, device=H.device)
            H = ops.gather_rows(H, idx)                       # any hidden width (the one-float-per-thread form when it is not a multiple of 4)
        u = int(user_id)
        act = (None, 0.0, 0.0) if clamp is None else ("clamp", float(clamp[0]), float(clamp[1]))
        return ops.linear(H, wd[u:u + 1], bd[u:u + 1], act=act[0], lo=act[1], hi=act[2]).reshape(-1)
