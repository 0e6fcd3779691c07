"""inferix/kvcache_manager/__init__.py"""
from inferix_amd.kvcache_manager import (KVCacheManager, KVCacheRequest, KVCacheRequestSpec, KVCacheSpec, KVCaches,  # noqa: F401
                                         KVCacheTensorSpec)
