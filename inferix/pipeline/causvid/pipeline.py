"""inferix/pipeline/causvid/pipeline.py:16 -> inferix_amd.pipeline.causvid"""
from inferix_amd.pipeline.causvid import CausVidPipeline, get_prompt as get_prompt_from_shell  # noqa: F401
