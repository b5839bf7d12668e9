#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python3 tools/exp/exp_r6_fused_compact.py 2>&1 | grep density | tee gpurun_out/r6_fused_micro.txt
bash tools/gpu_r6_fused.sh
