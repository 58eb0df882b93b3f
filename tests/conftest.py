import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # the step-wise Python statements of the provers are TEST infrastructure (tests/stepwise): importing the package registers them with
    # gemini_amd.snark / gemini_amd.psnark, which is what `native=False` (and the Python-level sharded keys) reach
    import tests.stepwise  # noqa: F401


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def pyref():
    from oracle import pyref

    return pyref
