import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ANCSH_REDZONE=1: EVERY gpu test runs with every buffer the host layer allocates (torch.empty / empty_like) carved out of a
# sentinel-guarded allocation (tests/redzone.py) and the guards checked when the test ends -- the dedicated file tests/test_redzone_gpu.py
# covers the ABI in the default run; this switch re-runs the whole suite that way (profiles/r05_redzone_full_suite.txt).
@pytest.fixture(autouse=True)
def _redzone_everything(request):
    if os.environ.get("ANCSH_REDZONE", "0") != "1" or request.node.get_closest_marker("gpu") is None:
        yield
        return
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from redzone import guarded
    with guarded():
        yield
