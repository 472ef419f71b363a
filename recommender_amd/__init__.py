"""recommender_amd: MI355X (gfx950) scoring engine behind the reference recommender's model interfaces.

Package layout mirrors the reference's import paths for the hot path only:
  recommender_amd.helpers.ncf_model.NCF          <- src/helpers/ncf_model.py
  recommender_amd.helpers.ranking_metrics        <- src/helpers/ranking_metrics.py
  recommender_amd.autorec.utils.model.AutoRec    <- src/autorec/utils/model.py
  recommender_amd.hybridautorecncf.model         <- src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:58-208
The arithmetic is in csrc/*.hip (libreco_hip.so, C ABI in include/reco_hip.h).
"""
__version__ = "0.1.0"

from .ops import set_tower_precision, tower_precision  # noqa: E402,F401
