#!/bin/bash
# round 4: piece claiming inside cob_mxv_kernel (SPLAHIP_COB_PERSIST=1, the default) against one workgroup per piece (=0) and against the build before it (tag base)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do
  for sc in 23 22; do
    echo "== scale $sc rep $rep"
    SPLA_LIB_TAG=base timeout 300 python3 tools/exp/exp_r4_ab.py $sc 2>&1 | tail -1
    SPLAHIP_COB_PERSIST=0 timeout 300 python3 tools/exp/exp_r4_ab.py $sc 2>&1 | tail -1 | sed 's/^/persist=0 /'
    SPLAHIP_COB_PERSIST=1 timeout 300 python3 tools/exp/exp_r4_ab.py $sc 2>&1 | tail -1 | sed 's/^/persist=1 /'
  done
done
timeout 1500 python3 -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q -m gpu --timeout 600 > gpurun_out/r4_persist_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r4_persist_tests.txt | tail -3
