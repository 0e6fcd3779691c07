from .causal_inference import CausalInferencePipeline

__all__ = ["CausalInferencePipeline"]
