#!/bin/bash
# round 3: the new test of the non-default forms of the blocks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{ timeout 1500 python -m pytest tests/test_mxv_gpu.py -m gpu -x -q -k "knobs or reused" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15; } 2>&1 | tee gpurun_out/r3_t27.txt
