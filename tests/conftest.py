import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _deterministic_index_builds():
    """The tests of the evaluation paths pin LC_OPT_LIKE_INDEX_ASYNC = 0 (the first LIKE of a scan waits for its scan-level
    index), so that which kernel answers an evaluation is a property of the test, not of a background build's progress.  The
    asynchronous default is what tests/test_gpu_round6.py exercises (before / during / after the build); LC_TEST_INDEX_ASYNC=1
    runs the whole suite with it."""
    from liquid_cache_amd import _native as N
    from liquid_cache_amd.cache import LiquidCacheBuilder
    if os.environ.get("LC_TEST_INDEX_ASYNC") != "1":
        LiquidCacheBuilder.default_options[N.OPT_LIKE_INDEX_ASYNC] = 0
    yield
    LiquidCacheBuilder.default_options.pop(N.OPT_LIKE_INDEX_ASYNC, None)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/liboracle.so on demand."""
    from oracle import liquid_oracle as lo
    lo.build()
    return lo


@pytest.fixture(scope="session")
def product_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from liquid_cache_amd import _native
    return _native.load()


@pytest.fixture()
def gpu_cache(product_lib):
    import liquid_cache_amd as lc
    cache = lc.LiquidCacheBuilder.new().build()  # raises loudly when no HIP device is present
    yield cache
    cache.close()
