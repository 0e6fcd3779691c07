"""inferix/pipeline/self_forcing/CausalInferencePipeline.py:58 -> inferix_amd.pipeline.causal_inference"""
from inferix_amd.pipeline.causal_inference import CausalInferencePipeline  # noqa: F401
