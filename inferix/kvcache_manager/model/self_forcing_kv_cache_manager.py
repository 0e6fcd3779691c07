"""inferix/kvcache_manager/model/self_forcing_kv_cache_manager.py:8-217"""
from inferix_amd.kvcache_manager.model.self_forcing_kv_cache_manager import (SelfForcingKVCacheManager,  # noqa: F401
                                                                              SelfForcingKVCacheManagerFactory)
