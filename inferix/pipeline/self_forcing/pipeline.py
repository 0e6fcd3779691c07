"""inferix/pipeline/self_forcing/pipeline.py:26 -> inferix_amd.pipeline.self_forcing"""
from inferix_amd.pipeline.self_forcing import SelfForcingPipeline  # noqa: F401
