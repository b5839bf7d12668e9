#!/bin/bash
# round 6: counter passes (separate rocprofv3 --pmc runs) and the kernel trace of the contract-line command, north-star SpMV at scale 23 and 22
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
for sc in 23 22; do
  name=spmv_fp32_scale$sc
  rm -rf "$R/gpurun_out/pmc_r06/$name"
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r06/$name/$c" -- python3 "$R/bench.py" --scale $sc --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-variants > /dev/null 2> "$R/gpurun_out/pmc_r06_${name}_$c.err"; echo "$name $c exit $?"
  done
  python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r06/$name" "cob_mxv_kernel<float" "cob_hub_finish_kernel<float" > "$R/gpurun_out/r06_pmc_$name.raw.json"
  python3 - "$R/gpurun_out/r06_pmc_$name.raw.json" "$R/gpurun_out/r06_pmc_$name.json" $sc <<'PY'
import json, sys
raw = json.load(open(sys.argv[1])); k = raw.get("cob_mxv_kernel<float", {})
nnz = {23: 131162881, 22: 65244781}[int(sys.argv[3])]
out = dict(raw)
h = raw.get("cob_hub_finish_kernel<float", {})
if "traffic_bytes_median" in k:
    out["_per_call"] = {"traffic_bytes": k["traffic_bytes_median"] + h.get("traffic_bytes_median", 0), "tcp_tcc_read_req": k.get("TCP_TCC_READ_REQ", {}).get("median"), "nnz": nnz,
                        "fetch_size_kb": k["FETCH_SIZE"]["median"] + h.get("FETCH_SIZE", {}).get("median", 0), "write_size_kb": k["WRITE_SIZE"]["median"] + h.get("WRITE_SIZE", {}).get("median", 0),
                        "hub_finish_kernel_bytes": h.get("traffic_bytes_median", 0),
                        "method": "(2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the two kernels a call launches (cob_mxv_kernel + cob_hub_finish_kernel), median over the dispatches of separate rocprofv3 --pmc passes"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out.get("_per_call")))
PY
  find "$R/gpurun_out/pmc_r06/$name" -name "*.csv" -size +1M -delete
done
# kernel trace + stats of the contract-line command (program directly after --)
rm -rf "$R/gpurun_out/prof_r06"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r06" --output-format csv -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify > "$R/gpurun_out/r06_bench_under_rocprof.json" 2> "$R/gpurun_out/r06_bench_under_rocprof.err"
f=$(ls "$R"/gpurun_out/prof_r06/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== rocprof kernel stats ($f)"
python3 - "$f" <<'PY' | tee "$R/gpurun_out/r06_rocprof_kernel_stats.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify   (kernel names cut to 150 characters)")
print(f"{'calls':>6} {'avg us':>10} {'total ms':>10} {'%':>6}  name")
for r in rows[:40]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}  {r['Name'][:150]}")
PY
t=$(ls "$R"/gpurun_out/prof_r06/*/*kernel_trace.csv 2>/dev/null | tail -1); python3 "$R/tools/cob_trace_groups.py" "$t" | tee -a "$R/gpurun_out/r06_rocprof_kernel_stats.txt"
