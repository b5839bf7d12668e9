#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/pmc_sq"
SPLAHIP_NO_HOT=${NOHOT:-1} timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_sq" -- python3 "$R/tools/exp/exp_gather.py" 22 ${HOTFLAG} > "$R/gpurun_out/pmc_sq.log" 2>&1; echo "exit $?"
cd "$R"; python - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
rows = defaultdict(dict)
for row in csv.DictReader(open(f)):
    if "spmv_" in row["Kernel_Name"] and "combine" not in row["Kernel_Name"]:
        rows[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
ids = sorted(rows)
# first 23 dispatches = "real", next 23 = "k%64"
for name, sl in (("real", ids[3:23]), ("coalesced k%64", ids[26:46])):
    if not sl: continue
    avg = {c: sum(rows[i].get(c, 0) for i in sl) / len(sl) for c in rows[sl[0]]}
    print(name, {k: round(v) for k, v in avg.items()})
    w = avg.get("SQ_WAVES", 1)
    print("   per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f  wave_cycles(quad) %.0f" % (avg["SQ_INSTS_VALU"]/w, avg["SQ_INSTS_SALU"]/w, avg["SQ_INSTS_LDS"]/w, avg["SQ_INSTS_VMEM_RD"]/w, avg["SQ_INSTS_VMEM_WR"]/w, avg["SQ_WAVE_CYCLES"]/w))
PY
find gpurun_out/pmc_sq -name "*.csv" -size +2M -delete
