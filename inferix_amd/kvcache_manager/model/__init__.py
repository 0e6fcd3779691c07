from .self_forcing_kv_cache_manager import SelfForcingKVCacheManager, SelfForcingKVCacheManagerFactory

__all__ = ["SelfForcingKVCacheManager", "SelfForcingKVCacheManagerFactory"]
