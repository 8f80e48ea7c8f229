import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tsukuba():
    """Frames 2 (reference) and 3 (matching) of tsukuba/param_in.txt, as BGRA uint8."""
    from stereomatch_amd import pnm
    d = os.path.join(GOLDEN, "tsukuba")
    return (pnm.read_image(os.path.join(d, "scene1.row3.col3.ppm")),
            pnm.read_image(os.path.join(d, "scene1.row3.col4.ppm")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    from stereomatch_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
