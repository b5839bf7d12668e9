#!/bin/bash
# Round-2 counter evidence for every reported line: FETCH_SIZE, WRITE_SIZE, TCP_TCC_READ_REQ in SEPARATE rocprofv3 --pmc passes
# (with --kernel-trace only), per workload; per-kernel medians by tools/pmc_kernels.py into gpurun_out/r02_pmc_<workload>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
run() { # name, kernel substrings (|-separated), command...
  name=$1; needles=$2; shift 2
  rm -rf "$R/gpurun_out/pmc_r02/$name"
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r02/$name/$c" -- "$@" > /dev/null 2> "$R/gpurun_out/pmc_r02_${name}_$c.err"; echo "$name $c exit $?"
  done
  IFS='|' read -ra N <<< "$needles"
  python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r02/$name" "${N[@]}" > "$R/gpurun_out/r02_pmc_$name.json"; head -c 1500 "$R/gpurun_out/r02_pmc_$name.json"; echo
  find "$R/gpurun_out/pmc_r02/$name" -name "*.csv" -size +1M -delete
}
export EXP_SELL=1 EXP_INT=1
run spmv_fp32_scale22 "cob_mxv_kernel<float|cob_combine_kernel<float|cob_mxv_kernel<int|cob_combine_kernel<int|spmv_wavetile_kernel<splahip::MxvCfg, float|sell_mxv_kernel<float" python3 "$R/tools/exp/exp_cob.py" 22 14:0
run bfs_step "sell_vprep_kernel<int|sell_mxv_dual_kernel<int|sell_post_kernel<int" python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants
run sssp_step "sell_vprep_kernel<int|sell_mxv_dual_kernel<int|sell_post_kernel<int|cob_mxv_kernel<int" python3 "$R/bench.py" --workload sssp-step --steps 5 --warmup 2 --no-cpu-baseline --no-verify
run pagerank_vxm "cob_mxv_kernel<float|cob_combine_kernel<float|sell_mxv_kernel<float|flag_write_kernel" python3 "$R/bench.py" --workload pagerank-vxm --steps 5 --warmup 2 --no-cpu-baseline --no-verify
run mxm "mxm_spa_kernel|mxm_spa_ranked_kernel|onesweep|mxm_expand_kernel" python3 "$R/bench.py" --workload mxm --steps 2 --no-cpu-baseline --no-verify
run tc_mxmT "mxmT_heavy_kernel|mxmT_kernel|mxmT" python3 "$R/bench.py" --workload tc-mxmT --steps 3 --warmup 1 --no-cpu-baseline --no-verify
