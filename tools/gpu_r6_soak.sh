#!/bin/bash
# Round 6: randomised soaks against the oracle on this round's build (every form incl. the reference-order mode and the forced selected-rows kernels), two ranks on one GPU,
# then the counter passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 700 python3 tools/exp/fuzz_mxv.py ${SOAK:-500} 60601 > gpurun_out/r6_fuzz_mxv.txt 2>&1; tail -2 gpurun_out/r6_fuzz_mxv.txt | cut -c1-300
timeout 700 python3 tools/exp/fuzz_ops.py ${SOAK:-500} 60602 > gpurun_out/r6_fuzz_ops.txt 2>&1; tail -2 gpurun_out/r6_fuzz_ops.txt | cut -c1-300
SCALE=19 bash tools/gpu_n2_debug_r06.sh > gpurun_out/r6_n2_debug.txt 2>&1; tail -14 gpurun_out/r6_n2_debug.txt | cut -c1-220
ONLY="spmv_fp32_scale23_wavetile mxm mxm_scale19 pagerank_vxm_orkut_like sssp_step bfs_step tc_mxmT" bash tools/gpu_pmc_r06_ops.sh 2>&1 | grep -v "exit 0" | tail -12
bash tools/gpu_pmc_r06.sh > gpurun_out/r6_pmc_headline.log 2>&1; grep "traffic_bytes" gpurun_out/r6_pmc_headline.log | cut -c1-300; tail -30 gpurun_out/r6_pmc_headline.log | cut -c1-200
