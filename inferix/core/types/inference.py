"""inferix/core/types/inference.py:11-101"""
from inferix_amd.core.types import DecodeMode, MemoryMode, StreamingMode  # noqa: F401
from inferix_amd.magi.types import InferenceParams, ModelMetaArgs, PackedCoreAttnParams, PackedCrossAttnParams  # noqa: F401
