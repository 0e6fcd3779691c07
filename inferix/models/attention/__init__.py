"""inferix/models/attention/__init__.py: `attention` / `flash_attention` (flash_attention.py:42-200)"""
from inferix_amd.attention import attention, flash_attention  # noqa: F401
