#!/bin/bash
# round 5: run-time schedule A/B at two steal thresholds + timeline + the mxv tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for s in ${SCALES:-22 23}; do timeout 100 python3 tools/exp/exp_r5_dyn.py $s 2>&1 | tail -4; done | tee gpurun_out/r5_dyn_ab.txt
for s in ${SCALES:-22 23}; do timeout 100 python3 tools/exp/exp_r5_dyn.py $s SPLAHIP_COB_DYN_MIN_HUB=1200 SPLAHIP_COB_DYN_MIN_NAT=4000 2>&1 | grep "dyn=1" | tail -2; done | tee gpurun_out/r5_dyn_ab_lowthr.txt
for s in ${SCALES:-22 23}; do SPLA_LIB_TAG=cobtime SPLAHIP_COB_TRACE=1 timeout 100 python3 tools/exp/exp_r5_dyn_timing.py $s 2>&1 | grep -v "attach phase\|^\[cob\]" > gpurun_out/r5_dyn_timeline$s.txt; head -22 gpurun_out/r5_dyn_timeline$s.txt; done
if [ "${TESTS:-1}" = "1" ]; then
timeout 900 python3 -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -x -q -m gpu --timeout 300 > gpurun_out/r5_dyn_tests.txt 2>&1; tail -5 gpurun_out/r5_dyn_tests.txt
fi
