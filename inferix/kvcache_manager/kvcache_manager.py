"""inferix/kvcache_manager/kvcache_manager.py:56-244"""
from inferix_amd.kvcache_manager.kvcache_manager import *  # noqa: F401,F403
from inferix_amd.kvcache_manager import (KVCacheManager, KVCacheRequest, KVCacheRequestSpec, KVCacheSpec, KVCaches,  # noqa: F401
                                         KVCacheTensorSpec)
