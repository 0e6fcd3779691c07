from .base_pipeline import AbstractInferencePipeline
from .causal_inference import CausalInferencePipeline
from .causvid import CausVidPipeline
from .causvid_inference import CausVidInferencePipeline
from .self_forcing import SelfForcingPipeline

__all__ = ["AbstractInferencePipeline", "CausalInferencePipeline", "CausVidInferencePipeline", "CausVidPipeline",
           "SelfForcingPipeline"]
