"""Engines that sit where src/api/core/inference_engine.py sits: id validation, batched scoring, sorting."""
from .inference_engine import AutoRecInferenceEngine, NCFInferenceEngine, RequestError

__all__ = ["NCFInferenceEngine", "AutoRecInferenceEngine", "RequestError"]
