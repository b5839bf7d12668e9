#!/bin/bash
# Round 6: masked calls — random 10 % / 50 % of all rows on the blocks (default verdict) against the selected-rows kernels forced (probe weights lowered), + a kernel trace of the forced run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
echo "== default weights"; timeout 300 python3 tools/exp/exp_r4_mask.py 23 2>&1 | grep -v "^\[cob\]"
echo "== rows forced (SPLAHIP_MASK_ROWS_X10=10 SPLAHIP_MASK_LONG_X10=10)"; timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_MASK_ROWS_X10=10 SPLAHIP_MASK_LONG_X10=10 2>&1 | grep -v "^\[cob\]"
cd /tmp
rm -rf $R/gpurun_out/prof_r6_mask
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6_mask --output-format csv -- python3 $R/tools/exp/exp_r4_mask.py 23 SPLAHIP_MASK_ROWS_X10=10 SPLAHIP_MASK_LONG_X10=10 > /dev/null 2>&1
f=$(ls $R/gpurun_out/prof_r6_mask/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'calls':>6} {'avg us':>10} {'total ms':>10} {'%':>6}  name")
for r in rows[:25]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}  {r['Name'][:110]}")
PY
find $R/gpurun_out/prof_r6_mask -name "*.csv" -size +1M -delete
