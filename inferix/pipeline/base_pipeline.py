"""inferix/pipeline/base_pipeline.py:16 -> inferix_amd.pipeline.base_pipeline"""
from inferix_amd.pipeline.base_pipeline import AbstractInferencePipeline  # noqa: F401
