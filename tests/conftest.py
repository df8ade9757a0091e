import os
import sys

import pytest

# One process holds one HIP runtime: torch bundles its own libamdhip64.so.7 and libmi355rt.so asks for the
# same soname, so whichever is loaded first serves both.  torch only finds the GPU through its own copy,
# so tests that use torch tensors beside the library need torch imported first (bench.py does the same).
try:
    import torch  # noqa: F401
except ImportError:   # the library itself does not need torch
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def rt():
    import ray_tracer_amd
    return ray_tracer_amd


@pytest.fixture
def switch(rt):
    """Sets one of the library's tuning switches for the rest of the test (mi355rt_debug_switch: the library reads the
    MI355RT_* environment once, so tests cannot flip them through os.environ); value None = the default."""
    yield rt.debug_switch
    rt.debug_switch(None, None)
