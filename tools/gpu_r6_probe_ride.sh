#!/bin/bash
# Round 6: a masked call served by the blocks: its mask probe as the first workgroups of the hub ranges' finishing launch (SPLAHIP_MASK_PROBE_FUSED=1) against a launch of its own (0)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_probe_ride.txt
: > $out
timeout 900 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 600 -k "masked or mask or selected" 2>&1 | tail -2 | tee -a $out
for f in 1 0 1 0; do
  echo "== SPLAHIP_MASK_PROBE_FUSED=$f" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_MASK_PROBE_FUSED=$f 2>&1 | grep "all rows (\|random" | cut -c1-100 | tee -a $out
done
