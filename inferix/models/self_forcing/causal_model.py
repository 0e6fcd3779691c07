"""inferix/models/self_forcing/causal_model.py:493-1243 (CausalWanModel, inference branch)"""
from inferix_amd.wan import HipCausalWanModel as CausalWanModel  # noqa: F401
