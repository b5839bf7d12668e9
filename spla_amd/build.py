"""Build libspla_amd.so (HIP kernels + kernel-level C ABI + host layer) for gfx950 with hipcc.

In-tree build: objects under spla_amd/build/, the shared library at spla_amd/lib/libspla_amd.so (git-ignored,
but it travels to the GPU box with the repo snapshot).  hipcc cross-compiles without a GPU.

    python -m spla_amd.build            # incremental
    python -m spla_amd.build --force    # rebuild everything
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
REPO = ROOT.parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIB = ROOT / "lib" / "libspla_amd.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-std=c++17", "-O3", "-fPIC", f"-I{REPO / 'include'}", f"-I{CSRC}", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("SPLA_EXTRA_FLAGS", "").split()  # experiments only (e.g. -DSPLAHIP_GATHER_MODE=2)
if os.environ.get("SPLA_BUILD_TAG"):  # experiments only: side-by-side builds
    OBJ = ROOT / ("build_" + os.environ["SPLA_BUILD_TAG"])
    LIB = ROOT / "lib" / f"libspla_amd_{os.environ['SPLA_BUILD_TAG']}.so"


def _sources():
    hip = sorted((CSRC / "kernels").glob("*.hip"))
    cpp = sorted((CSRC / "host").glob("*.cpp")) if (CSRC / "host").is_dir() else []
    return hip, cpp


def _headers_mtime() -> float:
    newest = 0.0
    for pat in ("csrc/**/*.hpp", "csrc/**/*.h"):
        for p in ROOT.glob(pat):
            newest = max(newest, p.stat().st_mtime)
    for p in (REPO / "include").rglob("*.h*"):
        newest = max(newest, p.stat().st_mtime)
    return newest


def _compile(src: Path, force: bool, hdr_mtime: float) -> Path:
    obj = OBJ / (src.stem + ("_hip" if src.suffix == ".hip" else "_cpp") + ".o")
    if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime):
        return obj
    cmd = [HIPCC, *COMMON]
    if src.suffix == ".hip":
        cmd += [f"--offload-arch={ARCH}"]
    else:
        cmd += ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    cmd += ["-c", str(src), "-o", str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"compile failed: {' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    return obj


def build_all(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    LIB.parent.mkdir(parents=True, exist_ok=True)
    hip, cpp = _sources()
    hdr = _headers_mtime()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, hdr), hip + cpp))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest_obj:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed: {' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"[spla_amd.build] {LIB} ({LIB.stat().st_size // 1024} KiB, {len(objs)} objects)")
    return LIB


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
