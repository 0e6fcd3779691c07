"""inferix/models/magi/dit/dit_model.py: VideoDiTModel :44-596 (forward_pre_process / forward / forward_post_process / forward_3cfg /
forward_dispatcher)"""
from inferix_amd.magi.model import HipVideoDiTModel as VideoDiTModel  # noqa: F401
