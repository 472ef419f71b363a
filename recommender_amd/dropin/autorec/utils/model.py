"""`autorec.utils.model` as the reference names it."""
from recommender_amd.autorec.utils.model import AutoRec  # noqa: F401
