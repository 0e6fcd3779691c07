"""inferix/pipeline/causvid/CausalInferencePipeline.py:21 -> inferix_amd.pipeline.causvid_inference"""
from inferix_amd.pipeline.causvid_inference import CausVidInferencePipeline as CausalInferencePipeline  # noqa: F401
