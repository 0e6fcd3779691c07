"""inferix/models/self_forcing/wrapper.py: WanDiffusionWrapper :171-385, WanTextEncoder :15-59, WanVAEWrapper :62-168"""
from inferix_amd.t5 import HipWanTextEncoder as WanTextEncoder  # noqa: F401
from inferix_amd.vae import HipWanVAEWrapper as WanVAEWrapper  # noqa: F401
from inferix_amd.wan import HipWanDiffusionWrapper as WanDiffusionWrapper  # noqa: F401
