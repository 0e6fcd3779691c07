from .causal_inference import CausalInferencePipeline
from .causvid_inference import CausVidInferencePipeline

__all__ = ["CausalInferencePipeline", "CausVidInferencePipeline"]
