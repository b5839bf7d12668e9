#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; out=gpurun_out/r04_t15.txt; : > $out
timeout 600 python -m pytest tests/test_mxv_gpu.py -x -q --timeout 200 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 >> $out
for pm in 1 0; do
  SPLAHIP_COB_TRACE=1 timeout 300 python tools/exp/exp_r4_cob.py 23 $pm 2>&1 | grep -v "amdgpu.ids\|schedule" | cut -c1-260 >> $out
  SPLAHIP_COB_TRACE=1 timeout 300 python tools/exp/exp_r4_cob.py 23 $pm SPLAHIP_COB_RELABEL=0 2>&1 | grep -v "amdgpu.ids\|schedule\|columns ref" | cut -c1-200 >> $out
done

timeout 600 python bench.py --workload pagerank-vxm --steps 10 --warmup 2 --no-cpu-baseline 2>> $out | cut -c1-400 >> $out
SPLAHIP_COB_RELABEL=0 timeout 600 python bench.py --workload pagerank-vxm --steps 10 --warmup 2 --no-cpu-baseline 2> /dev/null | cut -c1-400 >> $out
cat $out
