"""Stand-in for cachetools (memoising decorator + dict-backed LRUCache)."""
from . import keys  # noqa: F401


class LRUCache(dict):
    def __init__(self, maxsize=128):
        super().__init__()
        self.maxsize = maxsize


def cached(cache, key=keys.hashkey):
    def deco(func):
        def wrapper(*args, **kwargs):
            k = key(*args, **kwargs)
            if k in cache:
                return cache[k]
            v = func(*args, **kwargs)
            cache[k] = v
            return v
        wrapper.__wrapped__ = func
        return wrapper
    return deco
