"""Inference engines with the reference's method surface, built on the fused GPU paths.

Mirror of src/api/core/inference_engine.py:41-344 minus the presentation layer (movie metadata / pydantic items are the
routes' business): `predict`, `predict_batch`, `recommend` take the same arguments, apply the same id validation and
return plain `(item_id, score)` pairs in the reference's order:

  NCF.predict_batch       stable sort by score, descending (list.sort(reverse=True), :124)
  NCF.recommend           np.argsort(scores)[::-1][:k] over de-duplicated candidates (:149-166)  -> fused score + top-k
  AutoRec.predict_batch   one forward per item in the reference (:266-279)                        -> one GEMV row from the cached H
  AutoRec.recommend       predict_batch(...)[:k] (:333-341)                                       -> same + wavefront top-k

Errors carry the reference's HTTP status (400 bad id / empty candidates, 503 model not loaded); when FastAPI is
importable they ARE fastapi.HTTPException, so the unchanged routes keep working.
"""
import numpy as np
import torch

from .. import ops
from ..dist import packed_topk_buffers

try:
    from fastapi import HTTPException as _Base
except Exception:  # pragma: no cover - fastapi is optional
    class _Base(Exception):
        def __init__(self, status_code, detail=None):
            super().__init__(detail)
            self.status_code, self.detail = status_code, detail


class RequestError(_Base):
    pass


def _check_id(value, limit, what, model_name):
    if not (0 <= value < limit):                                   # inference_engine.py:17-31
        raise RequestError(status_code=400, detail="%s ID must be in [0, %d) for %s, got %s" % (what, limit, model_name, value))


class NCFInferenceEngine:
    def __init__(self, model, user_num=None, item_num=None):
        self.model = model
        self.model_name = "NCF"
        if model is not None:
            t = model._own_tensors()
            ref_u = t.get("embed_user_GMF", t.get("embed_user_MLP")); ref_i = t.get("embed_item_GMF", t.get("embed_item_MLP"))
            user_num = user_num or ref_u.shape[0]; item_num = item_num or ref_i.shape[0]
        self.config = {"user_num": user_num, "item_num": item_num}

    def _ready(self):
        if self.model is None:
            raise RequestError(status_code=503, detail="NCF model not loaded")

    def predict(self, user_id, item_id):
        self._ready()
        _check_id(user_id, self.config["user_num"], "User", self.model_name)
        _check_id(item_id, self.config["item_num"], "Item", self.model_name)
        self.model._check_mode()
        tower = self.model._tower(self.model._compute_device())              # ids validated above: no range report needed
        return float(tower.score_pairs(torch.tensor([user_id]), torch.tensor([item_id]), check_ids=False).cpu().item())

    def predict_batch(self, user_id, item_ids):
        self._ready()
        _check_id(user_id, self.config["user_num"], "User", self.model_name)
        for i in item_ids:
            _check_id(i, self.config["item_num"], "Item", self.model_name)
        if not len(item_ids):
            return []
        cands = torch.as_tensor(list(item_ids), dtype=torch.int64)
        if len(item_ids) <= 128:           # one fused call: the stable descending sort IS the top-n select's order
            device = self.model._compute_device()
            buf, ids, vals = packed_topk_buffers(1, len(item_ids), device)
            self.model.score_topk((int(user_id), 1), len(item_ids), cands, check_ids=False, out=(ids, vals))
            host = buf.cpu()
            order = host[0, 0].numpy(); vals = host[1, 0].view(torch.float32).numpy()
        else:                              # long lists: tile path + stable descending sort on the host
            sc = self.model.score_matrix((int(user_id), 1), cands)[0].cpu().numpy()
            order = np.argsort(-sc, kind="stable"); vals = sc[order]
        return [(int(item_ids[j]), float(v)) for j, v in zip(order, vals)]

    def recommend(self, user_id, k=10, candidate_item_ids=None):
        self._ready()
        _check_id(user_id, self.config["user_num"], "User", self.model_name)
        if candidate_item_ids is None:
            cands = None
            n = self.config["item_num"]
        else:
            for i in candidate_item_ids:
                _check_id(i, self.config["item_num"], "Item", self.model_name)
            uniq = list(set(candidate_item_ids))                               # the reference de-duplicates with set() (:154)
            if not uniq:
                raise RequestError(status_code=400, detail="No valid candidate items")
            cands = torch.as_tensor(uniq, dtype=torch.int64)
            n = len(uniq)
        k = min(k, n)
        if k <= 0:
            return []                                                          # scores[...][:0]
        if k > 128:                        # beyond the schema's k <= 100 (schemas.py:61) and the wavefront select: host sort
            sc = self.model.score_matrix((int(user_id), 1), cands)[0].cpu().numpy()
            order = np.argsort(-sc, kind="stable")[:k]
            ids = order if cands is None else cands.numpy()[order]
            return [(int(i), float(v)) for i, v in zip(ids, sc[order])]
        # ids were validated above, so the kernel-side range report (a sync) is skipped; ids and scores come back in ONE copy
        device = self.model._compute_device()
        buf, ids, vals = packed_topk_buffers(1, k, device)
        self.model.score_topk((int(user_id), 1), k, cands, check_ids=False, out=(ids, vals))
        host = buf.cpu()
        ids = host[0, 0].numpy(); vals = host[1, 0].view(torch.float32).numpy()
        if cands is not None:
            ids = cands.numpy()[ids]
        return [(int(i), float(v)) for i, v in zip(ids, vals)]


class AutoRecInferenceEngine:
    def __init__(self, model, training_tensor, user_num=None, item_num=None):
        self.model = model
        self.training_tensor = training_tensor
        self.model_name = "AutoRec"
        self.item_based = True if model is None else bool(model.item_based)
        if model is not None and training_tensor is not None:
            user_num = user_num or training_tensor.shape[0]; item_num = item_num or training_tensor.shape[1]
            if self.item_based:
                if training_tensor.is_cuda:
                    model.attach_ratings(training_tensor)                       # N2: the matrix stays sparse (CSR) on the device
                model.cache_hidden(training_tensor)                             # H = sigmoid(enc(X^T)), once at start-up
        self.config = {"user_num": user_num, "item_num": item_num}

    def _ready(self):
        if self.model is None or self.training_tensor is None:
            raise RequestError(status_code=503, detail="AutoRec model or training matrix not loaded")

    def _user_scores(self, user_id, item_ids=None):
        if self.item_based:
            return self.model.predict_user(user_id, item_ids)
        row = self.model.reconstruct(self.training_tensor[user_id].unsqueeze(0))[0]
        return row if item_ids is None else row[torch.as_tensor(list(item_ids), dtype=torch.int64, device=row.device)]

    def predict(self, user_id, item_id):
        self._ready()
        _check_id(user_id, self.config["user_num"], "User", self.model_name)
        _check_id(item_id, self.config["item_num"], "Item", self.model_name)
        return float(self._user_scores(user_id, [item_id]).cpu().item())

    def predict_batch(self, user_id, item_ids):
        self._ready()
        _check_id(user_id, self.config["user_num"], "User", self.model_name)
        for i in item_ids:
            _check_id(i, self.config["item_num"], "Item", self.model_name)
        if not len(item_ids):
            return []
        sc = self._user_scores(user_id, list(item_ids)).cpu().numpy()
        order = np.argsort(-sc, kind="stable")                                 # list.sort(reverse=True) keeps first-seen order on ties
        return [(int(item_ids[j]), float(sc[j])) for j in order]

    def recommend(self, user_id, k=10, candidate_item_ids=None):
        self._ready()
        _check_id(user_id, self.config["user_num"], "User", self.model_name)
        if candidate_item_ids is None:
            cand = None
            n = self.config["item_num"]
        else:
            for i in candidate_item_ids:
                _check_id(i, self.config["item_num"], "Item", self.model_name)
            cand = list(set(candidate_item_ids))
            if not cand:
                raise RequestError(status_code=400, detail="No valid candidate items")
            n = len(cand)
        k = min(k, n)
        if k <= 0:
            return []
        sc = self._user_scores(user_id, cand)
        if k <= 128:
            ids, vals = ops.topk_rows(sc.view(1, -1), k)                        # ties: lowest candidate position first, as the stable sort
            ids = ids[0].cpu().numpy(); vals = vals[0].cpu().numpy()
        else:
            h = sc.cpu().numpy(); ids = np.argsort(-h, kind="stable")[:k]; vals = h[ids]
        if cand is not None:
            ids = np.asarray(cand)[ids]
        return [(int(i), float(v)) for i, v in zip(ids, vals)]
