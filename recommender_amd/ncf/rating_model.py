"""NCFRating: the rating-prediction variant of NCF, forward = sigmoid(logit) * (max - min) + min.

Mirror of the class defined in src/ncf/evaluate_ncf_rating.py:66-146 (and ncf-rating.ipynb); the
`*-Rating.pth` files are plain state_dicts with the same 4/10/12 keys as the logit models, so the
epilogue is selected by the class, not by the checkpoint (SURVEY 8a A6').
The sigmoid-scale epilogue is fused into the HIP tower kernel (RECO_ACT_RATING).
"""
import torch.nn as nn

from ..helpers.ncf_model import NCF


class NCFRating(NCF):
    def __init__(self, num_users, num_items, factor_num, num_layers, dropout, model_name="NeuMF-end",
                 rating_min=1.0, rating_max=5.0, GMF_model=None, MLP_model=None):
        super().__init__(num_users, num_items, factor_num, num_layers, dropout, model_name, GMF_model, MLP_model)
        self.rating_min = rating_min
        self.rating_max = rating_max

    def _init_weight_(self):
        """evaluate_ncf_rating.py:110-119: embeddings N(0, 0.01), other weights Xavier, biases 0; NeuMF-pre untouched."""
        if self.model_name == "NeuMF-pre":
            return
        for name, p in self.named_parameters():
            if "embed" in name:
                nn.init.normal_(p, std=0.01)
            elif "weight" in name:
                nn.init.xavier_uniform_(p)
            elif "bias" in name:
                nn.init.zeros_(p)

    def _rating_epilogue(self):
        return (float(self.rating_min), float(self.rating_max))
