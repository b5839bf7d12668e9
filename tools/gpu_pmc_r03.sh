#!/bin/bash
# round 3: counter passes (separate rocprofv3 --pmc runs) and the kernel trace of the contract-line command, north-star SpMV at scale 23 and 22
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
for sc in 23 22; do
  name=spmv_fp32_scale$sc
  rm -rf "$R/gpurun_out/pmc_r03/$name"
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r03/$name/$c" -- python3 "$R/bench.py" --scale $sc --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-variants > /dev/null 2> "$R/gpurun_out/pmc_r03_${name}_$c.err"; echo "$name $c exit $?"
  done
  python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r03/$name" "cob_mxv_kernel<float" > "$R/gpurun_out/r03_pmc_$name.raw.json"
  python3 - "$R/gpurun_out/r03_pmc_$name.raw.json" "$R/gpurun_out/r03_pmc_$name.json" $sc <<'PY'
import json, sys
raw = json.load(open(sys.argv[1])); k = raw.get("cob_mxv_kernel<float", {})
nnz = {23: 131162881, 22: 65244781}[int(sys.argv[3])]
out = dict(raw)
if "traffic_bytes_median" in k:
    out["_per_call"] = {"traffic_bytes": k["traffic_bytes_median"], "tcp_tcc_read_req": k.get("TCP_TCC_READ_REQ", {}).get("median"), "nnz": nnz,
                        "fetch_size_kb": k["FETCH_SIZE"]["median"], "write_size_kb": k["WRITE_SIZE"]["median"],
                        "method": "(2*FETCH_SIZE + WRITE_SIZE)*1024 of the one kernel a call launches, median over the dispatches of separate rocprofv3 --pmc passes"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out.get("_per_call")))
PY
  find "$R/gpurun_out/pmc_r03/$name" -name "*.csv" -size +1M -delete
done
# kernel trace + stats of the contract-line command (program directly after --)
rm -rf "$R/gpurun_out/prof_r03"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r03" --output-format csv -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify > "$R/gpurun_out/r03_bench_under_rocprof.json" 2> "$R/gpurun_out/r03_bench_under_rocprof.err"
f=$(ls "$R"/gpurun_out/prof_r03/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== rocprof kernel stats ($f)"; head -25 "$f" | cut -c1-260 | tee "$R/gpurun_out/r03_rocprof_kernel_stats.txt"
