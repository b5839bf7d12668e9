"""Short randomised soaks on the GPU (the tools under tools/exp/ with a fixed seed and a small time budget): random shapes, skews,
long rows, semirings, select ops, init values and vector densities through every kernel form, against the oracle."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,seed", [("fuzz_mxv.py", 11), ("fuzz_ops.py", 12)])
def test_soak(tool, seed):
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "exp" / tool), "12", str(seed)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "random cases passed" in res.stdout, res.stdout[-500:]
