"""inferix/distributed/parallelism/__init__.py"""
from inferix_amd.magi.context_parallel import UlyssesScheduler, cp_post_process, cp_pre_process  # noqa: F401
