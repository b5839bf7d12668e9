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


def pytest_terminal_summary(terminalreporter):
    """the full-size fp32 tests state their parity the way the contract does: the measured fraction of rows within plain 1e-5 of the reference's fold"""
    mod = sys.modules.get("test_configs_gpu") or sys.modules.get("tests.test_configs_gpu")
    rows = getattr(mod, "PARITY_REPORT", []) if mod else []
    if rows:
        terminalreporter.write_sep("-", "fp32 parity against the reference's sequential fold (tolerance 1e-5 relative)")
        for what, n, frac, worst in rows:
            terminalreporter.write_line(f"{what}: {n} rows, {100.0 * frac:.4f} % within plain 1e-5, worst relative difference {worst:.3e}")
