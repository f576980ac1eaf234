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


import contextlib


@contextlib.contextmanager
def handle_with_env(**env):
    """A fresh pls_amd.Handle created under the given environment switches: the library reads its switches ONCE, when a
    handle is created (pls_hip_create), so a test that compares two settings makes a handle for the other one."""
    import pls_amd
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        h = pls_amd.Handle()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        yield h
    finally:
        h.close()
