from .kvcache_manager import (KVCacheManager, KVCacheRequest, KVCacheRequestSpec, KVCacheSpec,
                              KVCacheTensorSpec, KVCaches, PageTable)

__all__ = ["KVCacheManager", "KVCacheRequest", "KVCacheRequestSpec", "KVCacheSpec", "KVCacheTensorSpec",
           "KVCaches", "PageTable"]
