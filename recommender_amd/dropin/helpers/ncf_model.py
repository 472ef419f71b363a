"""`helpers.ncf_model` as the reference names it; the class lives in recommender_amd.helpers.ncf_model."""
from recommender_amd.helpers.ncf_model import NCF  # noqa: F401
