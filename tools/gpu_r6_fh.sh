#!/bin/bash
# Round 6: the first-hit kernel with two rows per lane (SPLAHIP_FH_ROWS=2) against one: tests, the BFS step of configs[1], the traversal end to end
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_fh.txt
: > $out
timeout 900 python3 -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_api_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -2 | tee -a $out
for nr in 2 1 2 1; do
  echo -n "SPLAHIP_FH_ROWS=$nr: " | tee -a $out
  SPLAHIP_FH_ROWS=$nr timeout 300 python3 bench.py --workload bfs-step --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'GTEPS', d['value'], 'kernel', d['roofline'].get('kernel_avg_us'), 'verified', d.get('verified'))" | tee -a $out
done
for nr in 2 1; do
  echo -n "algorithms SPLAHIP_FH_ROWS=$nr: " | tee -a $out
  SPLAHIP_FH_ROWS=$nr timeout 600 python3 bench.py --workload algorithms --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v.get('ms') for k,v in d['algorithms'].items()}, d.get('verified'))" | tee -a $out
done
