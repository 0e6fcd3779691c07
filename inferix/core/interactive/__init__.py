"""inferix/core/interactive/session.py:38 -> inferix_amd.core.interactive"""
from inferix_amd.core.interactive import InteractiveSession  # noqa: F401
