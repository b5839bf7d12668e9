import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import pyoracle
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref/libspla_ref.so not built (needs /root/reference)")
    return pyoracle.Reference()
