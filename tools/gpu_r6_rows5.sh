#!/bin/bash
# Round 6: the compaction pass's last flush behind one 8-byte atomic (short | medium counts): tests, a short forced-rows soak, timings
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows5.txt
: > $out
timeout 900 python3 -m pytest tests/test_mxv_gpu.py tests/test_configs_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -2 | tee -a $out
timeout 300 python3 tools/exp/fuzz_mxv.py 150 60621 2>&1 | tail -1 | tee -a $out
for cfg in "1 1024" "1 1024"; do
  set -- $cfg
  echo "== SPLAHIP_ROWS_SMALL_LDS=$1 SPLAHIP_ROWS_COMPACT_BLOCKS=$2" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_SMALL_LDS=$1 SPLAHIP_ROWS_COMPACT_BLOCKS=$2 2>&1 | grep "random 10\|no hub" | cut -c1-100 | tee -a $out
done
MASKS="random10" X10=1 bash tools/gpu_r6_mask2.sh 2>&1 | tee -a $out
