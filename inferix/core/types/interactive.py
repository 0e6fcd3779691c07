"""inferix/core/types/interactive.py"""
from inferix_amd.core.interactive import (CheckpointResult, ControlCommand, GenerationStatus, InputApplyPolicy, InputState,  # noqa: F401
                                          QueuedInput, SegmentBoundary, SessionState, calculate_total_frames,
                                          validate_overlap_config)
