"""Mirror of the reference's `helpers` package for the scoring path (src/helpers/__init__.py:1-4)."""
from .ncf_model import NCF
from .ranking_metrics import hit, ndcg

__all__ = ["NCF", "hit", "ndcg"]
