"""inferix/core/types/__init__.py: the enums and records of the pipeline API"""
from inferix_amd.core.interactive import (CheckpointResult, ControlCommand, GenerationStatus, InputApplyPolicy, InputState,  # noqa: F401
                                          QueuedInput, SegmentBoundary, SessionState, calculate_total_frames,
                                          validate_overlap_config)
from inferix_amd.core.types import DecodeMode, MemoryMode, StreamingMode  # noqa: F401
