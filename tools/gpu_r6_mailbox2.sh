#!/bin/bash
# Round 6: the traversal algorithms end to end after (a) polled mailbox words, (b) uploads of <= 16 words as kernel arguments, (c) v_count_mf by compaction
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_mailbox2.txt
: > $out
timeout 1500 python3 -m pytest tests/test_api_gpu.py tests/test_vxm_gpu.py tests/test_reference_suite_gpu.py tests/test_threads_gpu.py tests/test_user_ops_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -3 | tee -a $out
for cfg in "1 100" "0 100" "1 100" "1 0"; do
  set -- $cfg
  echo "== SPLA_HIP_COUNT_BY_COMPACTION=$1 SPLAHIP_MAILBOX_POLL_US=$2" | tee -a $out
  SPLA_HIP_COUNT_BY_COMPACTION=$1 SPLAHIP_MAILBOX_POLL_US=$2 timeout 600 python3 bench.py --workload algorithms --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v.get('ms') for k,v in d['algorithms'].items()}, d.get('verified'))" | tee -a $out
done
