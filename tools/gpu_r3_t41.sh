#!/bin/bash
# round 3: last sanity of the shipped library: shard + mxv tests, smoke
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{ timeout 1500 python -m pytest tests/test_shard_gpu.py tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120; } 2>&1 | tee gpurun_out/r3_t41.txt
