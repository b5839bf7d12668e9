#!/bin/bash
# round 6: counter passes (separate rocprofv3 --pmc runs) of the SSSP step of configs[3] and of the PageRank pull step on the round-5 layout of the blocks
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }   # ONLY="mxm tc_mxmT" tools/gpu_pmc_r06_ops.sh: just those passes
run() { # name, reference kernel, other kernels (|-separated), command...
  name=$1; ref=$2; others=$3; shift 3
  rm -rf "$R/gpurun_out/pmc_r06/$name"
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r06/$name/$c" -- "$@" > /dev/null 2> "$R/gpurun_out/pmc_r06_${name}_$c.err"; echo "$name $c exit $?"
  done
  IFS='|' read -ra N <<< "$ref|$others"
  python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r06/$name" "${N[@]}" > "$R/gpurun_out/r06_pmc_$name.raw.json"
  python3 - "$R/gpurun_out/r06_pmc_$name.raw.json" "$R/gpurun_out/r06_pmc_$name.json" "$ref" <<'PY'
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
  find "$R/gpurun_out/pmc_r06/$name" -name "*.csv" -size +1M -delete
}
want sssp_step && run sssp_step "cob_mxv_kernel<int" "cob_hub_finish_kernel<int|cob_vbits_kernel<int" python3 "$R/bench.py" --workload sssp-step --steps 5 --warmup 2 --no-cpu-baseline --no-verify
want pagerank_vxm_orkut_like && run pagerank_vxm_orkut_like "cob_mxv_kernel<float" "cob_hub_finish_kernel<float|flag_write_kernel" python3 "$R/bench.py" --workload pagerank-vxm --steps 5 --warmup 2 --no-cpu-baseline --no-verify
# the BFS step of configs[1] (profiles/pmc_traffic.json was round 2's pass): the three kernels of the sliced-ELL bit form
# (round 6: the step runs on the first-hit kernel; the sliced-ELL form of the same step with SPLAHIP_FIRST_HIT=0)
if want bfs_step; then
run bfs_step "mxv_first_hit_kernel<int" "vstat_kernel<int" python3 "$R/bench.py" --workload bfs-step --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-variants
export SPLAHIP_FIRST_HIT=0
run bfs_step_sell "sell_mxv_dual_kernel<int" "sell_vprep_kernel<int|sell_post_kernel<int" python3 "$R/bench.py" --workload bfs-step --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants
unset SPLAHIP_FIRST_HIT
python3 - "$R/gpurun_out/r06_pmc_bfs_step.json" "$R/gpurun_out/r06_pmc_traffic.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); pc = d["_per_call"]
n, nnz, nnz_sel = 4194304, 65244781, 28570647
alg = 4 * nnz_sel + 4 * (n + 1) + 4 * n + 4 * n + 4 * n
out = {"kernel": "vstat_kernel<int>|mxv_first_hit_kernel<int>", "traffic_bytes": pc["traffic_bytes"], "algorithmic_bytes": alg,
       "traffic_over_algorithmic": round(pc["traffic_bytes"] / alg, 3), "tcp_tcc_read_req": pc["tcp_tcc_read_req"], "config": {"n": n, "nnz": nnz, "nnz_selected": nnz_sel},
       "method": "round 6: tools/gpu_pmc_r06_ops.sh - rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ in separate passes (with --kernel-trace only) of `python3 bench.py --workload bfs-step --steps 10 "
                 "--warmup 2 --no-cpu-baseline --no-verify --no-variants`; per-kernel medians by tools/pmc_kernels.py (profiles/r06_pmc_bfs_step.json; the sliced-ELL form of the same step: r06_pmc_bfs_step_sell.json); traffic = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024",
       "per_kernel": {k: v for k, v in d.items() if not k.startswith("_")}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("traffic_bytes", "algorithmic_bytes", "traffic_over_algorithmic")}))
PY
fi
# round 6: the two SpGEMM workloads on this round's forms (owner form of mxmT; mxm with the light rows in LDS)
want tc_mxmT && run tc_mxmT "mxmT_owner_kernel<int" "mxmT_entry_kernel<int|mxmT_compact_kernel|mxmT_heavy_kernel<int|mxmT_merge_kernel<int|onesweep_iteration_kernel|onesweep_histograms_kernel" python3 "$R/bench.py" --workload tc-mxmT --steps 5 --warmup 2 --no-cpu-baseline --no-verify
want mxm && run mxm "mxm_spa_kernel<float, splahip::StaBin<2>, 0, true" "mxm_spa_kernel<float, splahip::StaBin<2>, 0, false|mxm_spa_ranked_kernel<float|light_batch_kernel<float|place_light_kernel|rowprod_kernel" python3 "$R/bench.py" --workload mxm --steps 3 --warmup 1 --no-cpu-baseline --no-verify
# round 6: mxm at the largest representable size, and the CSR wave-tile kernel on the north-star matrix (the blocks switched off: SPLAHIP_COB=0)
want mxm_scale19 && run mxm_scale19 "mxm_spa_kernel<float, splahip::StaBin<2>, 0, true" "mxm_spa_kernel<float, splahip::StaBin<2>, 0, false|mxm_spa_ranked_kernel<float|light_batch_kernel<float|place_light_kernel|rowprod_kernel" python3 "$R/bench.py" --workload mxm-19 --steps 2 --warmup 1 --no-cpu-baseline --no-verify
if want spmv_fp32_scale23_wavetile; then
export SPLAHIP_COB=0
run spmv_fp32_scale23_wavetile "spmv_wavetile_kernel<splahip::MxvCfg, float" "spmv_combine_kernel<splahip::MxvCfg, float" python3 "$R/bench.py" --scale 23 --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-variants
unset SPLAHIP_COB
fi
