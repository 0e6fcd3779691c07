from .causal_inference import CausalInferencePipeline
from .causvid_inference import CausVidInferencePipeline
from .self_forcing import SelfForcingPipeline

__all__ = ["CausalInferencePipeline", "CausVidInferencePipeline", "SelfForcingPipeline"]
