#!/bin/bash
# Round-2 counter evidence, second pass: the workloads whose kernels changed after tools/gpu_pmc_r02.sh ran (BFS step: frontier walk; mxmT: galloping)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
run() { # name, kernel substrings (|-separated), command...
  name=$1; needles=$2; shift 2
  rm -rf "$R/gpurun_out/pmc_r02/$name"
  for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
    timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r02/$name/$c" -- "$@" > /dev/null 2> "$R/gpurun_out/pmc_r02_${name}_$c.err"; echo "$name $c exit $?"
  done
  IFS='|' read -ra N <<< "$needles"
  python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r02/$name" "${N[@]}" > "$R/gpurun_out/r02_pmc_$name.json"; head -c 600 "$R/gpurun_out/r02_pmc_$name.json"; echo
  find "$R/gpurun_out/pmc_r02/$name" -name "*.csv" -size +1M -delete
}
run bfs_step "sell_vprep_kernel<int|sell_mxv_dual_kernel<int|sell_post_kernel<int" python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants --no-scale23
run tc_mxmT "mxmT_heavy_kernel|mxmT_entry_kernel|mxmT_compact" python3 "$R/bench.py" --workload tc-mxmT --steps 3 --warmup 1 --no-cpu-baseline --no-verify
# kernel trace of the default bench line (the judged command)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r02_bench" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-variants --no-scale23 > "$R/gpurun_out/r02_bench_under_rocprof.json" 2> /dev/null; echo "trace exit $?"
f=$(find "$R/gpurun_out/prof_r02_bench" -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
