"""liquid_cache_amd — MI355X-native decode + predicate-pushdown path of LiquidCache.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of include/liquid_cache_amd.h) and the
host-side mirror of the reference's cache interface (`cache.py`).  See DESIGN.md.
"""
from ._native import LiquidCacheError, LIB_PATH, EXPORTED_SYMBOLS  # noqa: F401
from .cache import (EntryID, ParquetArrayID, CacheExpression, LiquidExpr, LiquidCacheBuilder, LiquidCache, Scan,  # noqa: F401
                    Date32Field, ExtractDate32, boolean_buffer_and_then, Column, Cast, ToTimestampSeconds)
