#!/bin/bash
# round 3: counter passes (separate rocprofv3 --pmc runs) of the SSSP step of configs[3] and of the PageRank pull step on the round-3 layout of the blocks
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
run() { # name, reference kernel, other kernels (|-separated), command...
  name=$1; ref=$2; others=$3; shift 3
  rm -rf "$R/gpurun_out/pmc_r03/$name"
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r03/$name/$c" -- "$@" > /dev/null 2> "$R/gpurun_out/pmc_r03_${name}_$c.err"; echo "$name $c exit $?"
  done
  IFS='|' read -ra N <<< "$ref|$others"
  python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r03/$name" "${N[@]}" > "$R/gpurun_out/r03_pmc_$name.raw.json"
  python3 - "$R/gpurun_out/r03_pmc_$name.raw.json" "$R/gpurun_out/r03_pmc_$name.json" "$ref" <<'PY'
import json, sys
raw = json.load(open(sys.argv[1])); ref = sys.argv[3]
calls = raw[ref]["FETCH_SIZE"]["dispatches"]
tot = 0.0
for k, cs in raw.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        tot += (2 * cs["FETCH_SIZE"]["median"] + cs["WRITE_SIZE"]["median"]) * 1024 * cs["FETCH_SIZE"]["dispatches"] / calls
req = sum(cs["TCP_TCC_READ_REQ"]["median"] * cs["TCP_TCC_READ_REQ"]["dispatches"] / calls for cs in raw.values() if "TCP_TCC_READ_REQ" in cs)
out = dict(raw)
out["_per_call"] = {"calls_in_pass": calls, "reference_kernel": ref, "traffic_bytes": round(tot), "tcp_tcc_read_req": round(req),
                    "how": "sum over the listed kernels of (median per dispatch) x (dispatches / calls); (2*FETCH_SIZE + WRITE_SIZE) KB * 1024"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["_per_call"]))
PY
  find "$R/gpurun_out/pmc_r03/$name" -name "*.csv" -size +1M -delete
}
run sssp_step "cob_mxv_kernel<int" "cob_hub_finish_kernel<int|cob_vbits_kernel<int" python3 "$R/bench.py" --workload sssp-step --steps 5 --warmup 2 --no-cpu-baseline --no-verify
run pagerank_vxm "cob_mxv_kernel<float" "cob_hub_finish_kernel<float|flag_write_kernel" python3 "$R/bench.py" --workload pagerank-vxm --steps 5 --warmup 2 --no-cpu-baseline --no-verify
