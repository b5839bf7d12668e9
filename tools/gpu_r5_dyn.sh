#!/bin/bash
# round 5: the run-time schedule of the blocks kernel (mxv_cob_dyn.hpp) against the static pieces, then the mxv / vxm / shard GPU tests on it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for s in ${SCALES:-22 23}; do timeout ${EXP_TIMEOUT:-150} python3 tools/exp/exp_r5_dyn.py $s "$@" 2>&1 | tail -6; done | tee gpurun_out/r5_dyn_ab.txt
if [ "${TESTS:-1}" = "1" ]; then
timeout 1500 python3 -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q -m gpu --timeout 600 > gpurun_out/r5_dyn_tests.txt 2>&1; tail -5 gpurun_out/r5_dyn_tests.txt
fi
