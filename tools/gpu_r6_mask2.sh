#!/bin/bash
# Round 6: kernel trace of one masked call shape at a time, the selected-rows kernels forced
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${MASKS:-random10 stride10 random50}; do
  cd /tmp; rm -rf $R/gpurun_out/prof_r6_mask
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6_mask --output-format csv -- python3 $R/tools/exp/exp_r6_mask.py 23 $m SPLAHIP_MASK_ROWS_X10=${X10:-10} SPLAHIP_MASK_LONG_X10=${X10:-10} 2>&1 | grep "R-MAT"
  f=$(ls $R/gpurun_out/prof_r6_mask/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "splahip" in r["Name"] or "cob_" in r["Name"]:
        if int(r["Calls"]) > 20: print(f"   {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} us  {r['Name'][:100]}")
PY
done
rm -rf $R/gpurun_out/prof_r6_mask
