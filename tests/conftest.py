import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def load_pkg():
    """Import the hyphen-named package directory as module `sgrt_amd`."""
    if "sgrt_amd" in sys.modules:
        return sys.modules["sgrt_amd"]
    d = os.path.join(ROOT, "simd-gaussian-ray-tracing_amd")
    spec = importlib.util.spec_from_file_location("sgrt_amd", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sgrt_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def _shared_renderer(pkg):
    r = pkg.Renderer(0)   # raises if the HIP library or the GPU is missing -- no fallback
    yield r
    r.close()


@pytest.fixture
def renderer(pkg, _shared_renderer):
    """One context for the whole session; every test gets it with the library's default table settings (tests that switch
    to the exact kernels or another step must not leak that into the next test)."""
    _shared_renderer.set_table_step(pkg.TABLE_STEP_DEFAULT)
    _shared_renderer.set_table_budget(2.5e-5)
    _shared_renderer.set_cull_prune(6.0)
    return _shared_renderer


GOLDEN = os.path.join(ROOT, "tests", "golden")
