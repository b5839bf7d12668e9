"""The reference's OWN unit tests of the hot path — tests/test_mxv.cpp, test_vxm.cpp, test_mxm.cpp, test_mxmT.cpp of
SparseLinearAlgebra/spla, compiled unmodified (with the reference's vendored googletest) against this repo's include/spla.hpp
and linked to libspla_amd.so (oracle/Makefile target `reftests`; the binaries live in oracle/_ref/, which is git-ignored but
travels to the GPU box).  They call spla::exec_* exactly as the reference's users do and check the reference's known answers:
the drop-in claim, tested with the reference's own test code."""
import os
import subprocess
from pathlib import Path

import pytest

REF_DIR = Path(__file__).resolve().parents[1] / "oracle" / "_ref"
SUITES = ["mxv", "vxm", "mxm", "mxmT", "op", "library"]  # test_op.cpp / test_library.cpp: the op registry and library info through the same API


@pytest.mark.gpu
@pytest.mark.parametrize("suite", SUITES)
def test_reference_unit_tests_pass_on_the_hip_backend(suite):
    exe = REF_DIR / f"reftest_{suite}"
    if not exe.exists():
        pytest.skip(f"{exe} not built (oracle/Makefile `reftests` needs /root/reference)")
    env = dict(os.environ)
    # the known-answer cases; the *.perf* cases build 10^8-entry matrices element by element on the host and assert nothing
    # about the values: they are run by tools/gpu_reftests_perf.sh, not in the parity suite
    out = subprocess.run([str(exe), "--gtest_filter=-*perf*", "--gtest_color=no"], env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((out.stdout + out.stderr).splitlines()[-25:])
    assert out.returncode == 0, tail
    assert "[  PASSED  ]" in out.stdout and "[  FAILED  ]" not in out.stdout, tail


def test_reference_suite_binaries_link_the_product_library():
    """CPU: where the binaries exist they must resolve libspla_amd.so from the tree (rpath), not some other copy."""
    built = [REF_DIR / f"reftest_{s}" for s in SUITES if (REF_DIR / f"reftest_{s}").exists()]
    if not built:
        pytest.skip("reference test binaries not built")
    for exe in built:
        ldd = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
        line = next((l for l in ldd.splitlines() if "libspla_amd.so" in l), "")
        assert "spla_amd/lib/libspla_amd.so" in line, ldd
