#!/bin/bash
# Round 6: polled mailbox words against stream synchronisation, alternating in one process; then the algorithms workload with v_count_mf by (limited) compaction on / off
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_mailbox3.txt
: > $out
timeout 1500 python3 -m pytest tests/test_api_gpu.py tests/test_vxm_gpu.py tests/test_reference_suite_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -3 | tee -a $out
timeout 600 python3 tools/exp/exp_r6_mailbox_ab.py 2>&1 | grep round | tee -a $out
for cfg in 1 0 1 0; do
  echo "== SPLA_HIP_COUNT_BY_COMPACTION=$cfg" | tee -a $out
  SPLA_HIP_COUNT_BY_COMPACTION=$cfg timeout 600 python3 bench.py --workload algorithms --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v.get('ms') for k,v in d['algorithms'].items()}, d.get('verified'))" | tee -a $out
done
