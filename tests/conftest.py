import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def art():
    """The product binding.  Builds the native libraries in-tree if they are missing."""
    import accelerated_ray_tracer_amd as art
    if not (os.path.exists(art.RT_LIB_PATH) and os.path.exists(art.HOST_LIB_PATH)):
        art.build()
    return art


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu(art):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this environment")
    art.init(0)
    return art


@pytest.fixture(autouse=True)
def _shipped_options():
    """The scheduling knobs (rt_set_option) are process-wide: every test starts from, and leaves behind, the shipped defaults."""
    import accelerated_ray_tracer_amd as art
    if art._rt is not None:
        art.reset_options()
    yield
    if art._rt is not None:
        art.reset_options()


@pytest.fixture(scope="session")
def earth(art):
    return art.default_texture()
