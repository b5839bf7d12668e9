#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/pmc_exp"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_exp" -- python3 "$R/tools/exp/exp_gather.py" 22 > "$R/gpurun_out/pmc_exp.log" 2>&1; echo "exit $?"
cd "$R"; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_exp/**/*counter_collection.csv", recursive=True)[0]
vals=[]
for row in csv.DictReader(open(f)):
    if "spmv_wavetile_kernel" in row["Kernel_Name"]:
        vals.append(float(row["Counter_Value"]))
# 23 launches per variant (3 warmup + 20), in order
names=["real","k%64","k%n","rand1K","rand8K","rand64K","rand256K","rand1M","rand4M","relabel-random","relabel-indeg","relabel-indeg2"]
for i,n in enumerate(names):
    chunk=vals[i*23:(i+1)*23]
    if chunk: print(f"{n:16s} FETCH_SIZE KB median {sorted(chunk)[len(chunk)//2]:12.0f}  -> {sorted(chunk)[len(chunk)//2]*1024/1e6:8.1f} MB")
PY
grep -v amdgpu "$R/gpurun_out/pmc_exp.log" | tail -14
find gpurun_out/pmc_exp -name "*.csv" -size +2M -delete
