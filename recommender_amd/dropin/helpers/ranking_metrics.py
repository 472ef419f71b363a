"""`helpers.ranking_metrics` as the reference names it."""
from recommender_amd.helpers.ranking_metrics import hit, ndcg  # noqa: F401
