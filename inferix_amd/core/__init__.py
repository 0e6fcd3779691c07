from .types import DecodeMode, MemoryMode, StreamingMode

__all__ = ["DecodeMode", "MemoryMode", "StreamingMode"]
