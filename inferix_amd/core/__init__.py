from .interactive import (CheckpointResult, ControlCommand, GenerationStatus, InputApplyPolicy, InputState,
                          InteractiveSession, QueuedInput, SegmentBoundary, SessionState, calculate_total_frames,
                          validate_overlap_config)
from .types import DecodeMode, MemoryMode, StreamingMode

__all__ = ["DecodeMode", "MemoryMode", "StreamingMode", "InteractiveSession", "InputApplyPolicy", "InputState", "SessionState",
           "ControlCommand", "QueuedInput", "GenerationStatus", "CheckpointResult", "SegmentBoundary",
           "calculate_total_frames", "validate_overlap_config"]
