"""inferix/distributed/parallelism/context_parallel.py:30-598 (cp_ulysses)"""
from inferix_amd.magi.context_parallel import *  # noqa: F401,F403
from inferix_amd.magi.context_parallel import (UlyssesScheduler, all_to_all_input_split, all_to_all_output_split,  # noqa: F401
                                               cp_post_process, cp_pre_process, cp_update_cross_attn_qkv_range,
                                               fused_qkv_communication, gather_from_context_parallel_region,
                                               scatter_to_context_parallel_region)
