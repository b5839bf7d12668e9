#!/bin/bash
# Round 6: the combine pass as the first workgroups of the short rows' launch, behind the tasks (SPLAHIP_ROWS_COMBINE_FUSED=1) against a launch of its own (0)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows10.txt
: > $out
timeout 900 python3 -m pytest tests/test_mxv_gpu.py tests/test_configs_gpu.py tests/test_shard_gpu.py -x -q -m gpu --timeout 600 2>&1 | grep -a "passed\|failed" | tail -1 | tee -a $out
timeout 300 python3 tools/exp/fuzz_mxv.py 150 60631 2>&1 | tail -1 | tee -a $out
for f in 1 0 1 0; do
  echo "== SPLAHIP_ROWS_COMBINE_FUSED=$f" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_COMBINE_FUSED=$f 2>&1 | grep "random 10\|no hub" | sed 's/selected nnz *[0-9.]* M ([0-9.]*) *//' | cut -c1-70 | tee -a $out
done
