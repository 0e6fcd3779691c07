"""inferix/models/schedulers/flow_match.py:8-176"""
from inferix_amd.schedulers import FlowMatchScheduler  # noqa: F401
