"""inferix/models/attention/backends.py:36-166: the registry; the one backend here is "HipPagedFA"."""
from inferix_amd.attention import collect_supported_attn, hip_paged_fa_forward  # noqa: F401
