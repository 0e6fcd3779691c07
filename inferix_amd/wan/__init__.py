from .causal_model import HipCausalWanModel, ParallelConfig
from .wrapper import HipCausVidDiffusionWrapper, HipWanDiffusionWrapper

__all__ = ["HipCausalWanModel", "HipCausVidDiffusionWrapper", "HipWanDiffusionWrapper", "ParallelConfig"]
