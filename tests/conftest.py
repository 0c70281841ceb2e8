import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


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
