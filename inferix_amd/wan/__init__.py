from .causal_model import HipCausalWanModel, ParallelConfig
from .wrapper import HipWanDiffusionWrapper

__all__ = ["HipCausalWanModel", "HipWanDiffusionWrapper", "ParallelConfig"]
