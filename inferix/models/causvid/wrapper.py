"""inferix/models/causvid/wrapper.py:269-304"""
from inferix_amd.wan import HipCausVidDiffusionWrapper as WanDiffusionWrapper  # noqa: F401
