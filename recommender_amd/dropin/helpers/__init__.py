"""Drop-in `helpers` package: NCF / hit / ndcg from recommender_amd, the rest from the reference tree."""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)

from recommender_amd.helpers.ncf_model import NCF  # noqa: E402
from recommender_amd.helpers.ranking_metrics import hit, ndcg  # noqa: E402

try:  # src/helpers/__init__.py:1-4 also re-exports the dataset downloader (not on the scoring path)
    from .data_downloader import download_ml1m_dataset  # noqa: E402,F401
except ImportError:  # reference tree not on sys.path
    pass

__all__ = ["NCF", "hit", "ndcg"]
