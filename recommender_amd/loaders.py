"""Checkpoint loaders for the three on-disk shapes the reference produces.

  (1) whole-module pickle   torch.save(model, path)                      src/ncf/ncf.ipynb cells 10-17 -> NeuMF.pth, GMF.pth, MLP.pth
  (2) plain state_dict      torch.save(model.state_dict(), path)         *-Rating.pth, *-Ranking.pth
  (3) dict checkpoint       {'epoch', 'model_state_dict', ...}           autorec-rating.ipynb cell 9 -> AutoRec-best.pth

Mirrors src/api/core/model_loader.py:11-77 (load_ncf_model / load_autorec_model: return None when the file is missing
or unreadable, model in eval mode) but never unpickles arbitrary code: every file is read with weights_only=True and an
allow-list in which the reference's class path `helpers.ncf_model.NCF` is mapped to the HIP-backed class.
"""
import collections
import os

import torch
import torch.nn as nn

from .autorec.utils.model import AutoRec
from .helpers.ncf_model import NCF
from .ncf.rating_model import NCFRating


def _safe_globals():
    import _codecs
    import numpy as np
    allow = [(NCF, "helpers.ncf_model.NCF"), NCF, nn.Embedding, nn.Linear, nn.Sequential, nn.Dropout, nn.ReLU,
             collections.OrderedDict, set, np.dtype, _codecs.encode]
    try:
        allow += [np._core.multiarray.scalar, type(np.dtype("int32")), type(np.dtype("int64")), type(np.dtype("float64"))]
    except AttributeError:
        pass
    return allow


def load_any(path, map_location="cpu"):
    """torch.load with weights_only=True and the allow-list above."""
    with torch.serialization.safe_globals(_safe_globals()):
        return torch.load(path, map_location=map_location, weights_only=True)


def ncf_from_state_dict(sd, dropout=0.0, rating=False, rating_min=1.0, rating_max=5.0):
    """Build an NCF / NCFRating whose shape is read off the state_dict (nested GMF_model.* / MLP_model.* ignored, F3)."""
    own = collections.OrderedDict((k, v) for k, v in sd.items() if not k.startswith(("GMF_model.", "MLP_model.")))
    has_g = "embed_user_GMF.weight" in own
    has_m = "embed_user_MLP.weight" in own
    if not (has_g or has_m):
        raise ValueError("not an NCF state_dict: no embed_user_GMF / embed_user_MLP")
    name = "NeuMF-end" if has_g and has_m else ("GMF" if has_g else "MLP")
    ref_u = own["embed_user_GMF.weight" if has_g else "embed_user_MLP.weight"]
    ref_i = own["embed_item_GMF.weight" if has_g else "embed_item_MLP.weight"]
    n_layers = sum(1 for k in own if k.startswith("MLP_layers.") and k.endswith(".weight")) or 3
    factor = ref_u.shape[1] if has_g else own["embed_user_MLP.weight"].shape[1] >> (n_layers - 1)
    if rating:
        model = NCFRating(ref_u.shape[0], ref_i.shape[0], factor, n_layers, dropout, name, rating_min, rating_max)
    else:
        model = NCF(ref_u.shape[0], ref_i.shape[0], factor, n_layers, dropout, name)
    model.load_state_dict(own)
    return model.eval()


def load_ncf_model(model_path, device=None, rating=None):
    """src/api/core/model_loader.py:11-34.  Accepts shape (1) and shape (2); `rating` defaults to '-Rating' in the name."""
    if not os.path.exists(model_path):
        print("Warning: NCF model not found at %s" % model_path)
        return None
    try:
        obj = load_any(model_path)
        if isinstance(obj, nn.Module):
            model = obj.eval()
        else:
            if rating is None:
                rating = "-Rating" in os.path.basename(model_path)
            model = ncf_from_state_dict(obj.get("model_state_dict", obj), rating=rating)
        return model.to(device) if device is not None else model
    except Exception as e:  # the reference swallows loader errors and serves HTTP 503 (model_loader.py:32-34)
        print("Error loading NCF model: %s" % e)
        return None


def load_autorec_model(model_path, num_users=6040, num_items=3706, hidden_units=500, item_based=True, device=None):
    """src/api/core/model_loader.py:37-77.  Accepts shape (3) and shape (2)."""
    if not os.path.exists(model_path):
        print("Warning: AutoRec model not found at %s" % model_path)
        return None
    try:
        ckpt = load_any(model_path)
        sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
        model = AutoRec(num_users, num_items, hidden_units, item_based)
        model.load_state_dict(sd)
        model.eval()
        return model.to(device) if device is not None else model
    except Exception as e:
        print("Error loading AutoRec model: %s" % e)
        return None
