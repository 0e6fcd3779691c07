"""inferix/kvcache_manager/model/magi_kv_cache_manager.py:76-187"""
from inferix_amd.magi.attention import MagiKVCacheManager  # noqa: F401
