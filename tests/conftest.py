import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pls_oracle as po
    return po.OracleLib()


@pytest.fixture(scope="session")
def po():
    from oracle import pls_oracle
    return pls_oracle


@pytest.fixture(scope="session")
def handle():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pls_amd
    torch.cuda.set_device(0)
    h = pls_amd.Handle()
    yield h
    h.close()
