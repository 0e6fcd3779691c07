"""Enums of the pipeline API surface (inferix/core/types/inference.py:11-48)."""
from enum import Enum


class DecodeMode(Enum):
    AFTER_ALL = "after_all"
    PER_BLOCK = "per_block"
    NO_DECODE = "no_decode"


class StreamingMode(Enum):
    TRUE_STREAMING = "true_streaming"
    DEFERRED_DECODE = "deferred_decode"
    AUTO = "auto"


class MemoryMode(Enum):
    AGGRESSIVE = "aggressive"
    BALANCED = "balanced"
    RELAXED = "relaxed"
