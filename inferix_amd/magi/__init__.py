"""MAGI context-parallel attention path: Ulysses exchange + range attention over an in-place cache."""
from .types import InferenceParams, ModelMetaArgs, PackedCoreAttnParams, PackedCrossAttnParams  # noqa: F401
