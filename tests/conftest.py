import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def st():
    """The product package; on a GPU box the HIP library must be present and a device visible."""
    import computervisionimagestich2_amd as pkg
    pkg.capi.lib()
    return pkg


@pytest.fixture(scope="session")
def gpu(st):
    import torch
    if not torch.cuda.is_available() or st.device_count() < 1:
        pytest.fail("gpu-marked test started without a visible HIP device")
    torch.cuda.set_device(0)
    return torch.device("cuda:0")
