#!/bin/bash
# FETCH_SIZE per kernel of tools/exp/sector_calib.hip (what does a 4-byte gather's miss bring: a 128-B line or a 64-B sector?)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/sector_calib "$R/tools/exp/sector_calib.hip" || exit 1
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum; do
  rm -rf /tmp/sc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/sc_$c -- /tmp/sector_calib > /tmp/sc_$c.out 2>&1; echo "$c exit $?"
done
python3 - <<'PY' | tee "$R/gpurun_out/r03_sector_calib.txt"
import csv, glob, collections
for c in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"/tmp/sc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "gather" in k or "stream_read" in k:
                per[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in sorted(per.items()):
        print(f"{c:24s} {k:28s} per dispatch: {sorted(d.values())}")
print("table: 8192 lines = 1024 KB; gather_random: 16777216 loads of one word in a random line of a 512 MB table")
PY
