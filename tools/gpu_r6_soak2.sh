#!/bin/bash
# Round 6, after the host-round-trip changes (polled mailbox, small uploads as kernel arguments, v_count_mf by compaction): the randomised soaks against the oracle,
# the algorithm fuzz, two ranks on one GPU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 700 python3 tools/exp/fuzz_mxv.py ${SOAK:-400} 60611 > gpurun_out/r6_fuzz2_mxv.txt 2>&1; tail -2 gpurun_out/r6_fuzz2_mxv.txt | cut -c1-300
timeout 900 python3 tools/exp/fuzz_ops.py ${SOAK:-600} 60612 > gpurun_out/r6_fuzz2_ops.txt 2>&1; tail -2 gpurun_out/r6_fuzz2_ops.txt | cut -c1-300
timeout 500 python3 tools/exp/fuzz_algos.py ${SOAK_ALGOS:-300} 60613 > gpurun_out/r6_fuzz2_algos.txt 2>&1; tail -3 gpurun_out/r6_fuzz2_algos.txt | cut -c1-300
SCALE=19 bash tools/gpu_n2_debug_r06.sh > gpurun_out/r6_n2_debug2.txt 2>&1; tail -14 gpurun_out/r6_n2_debug2.txt | cut -c1-220
