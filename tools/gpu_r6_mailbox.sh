#!/bin/bash
# Round 6: host decisions read by POLLING the mailbox words (SPLAHIP_MAILBOX_POLL_US, default 100) against synchronising the stream (0): the traversal algorithms end to end
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_mailbox.txt
: > $out
timeout 900 python3 -m pytest tests/test_api_gpu.py tests/test_vxm_gpu.py tests/test_reference_suite_gpu.py tests/test_threads_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -3 | tee -a $out
for us in 0 100 0 100 400; do
  echo "== SPLAHIP_MAILBOX_POLL_US=$us" | tee -a $out
  SPLAHIP_MAILBOX_POLL_US=$us timeout 600 python3 bench.py --workload algorithms --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v.get('ms') for k,v in d['algorithms'].items()}, d.get('verified'))" | tee -a $out
done
