"""inferix/models/wan_base/utils/parallel_config.py:3-30"""
from inferix_amd.wan import ParallelConfig  # noqa: F401
