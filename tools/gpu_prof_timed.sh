#!/bin/bash
# only the "timed configuration" rocprof pass of tools/gpu_round.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"; rm -rf "$R/gpurun_out/prof_timed"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_timed" -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-verify --no-variants > "$R/gpurun_out/prof_timed_bench.json" 2> /dev/null; echo "rocprof timed exit $?"
python3 "$R/tools/timed_kernel_from_trace.py" "$R/gpurun_out/prof_timed" 'sell_vprep_kernel<int>|sell_mxv_dual_kernel<int, splahip::StaBin<13>, splahip::StaBin<12> >|sell_post_kernel<int' 30 "$R/gpurun_out/prof_timed_bench.json" | tee "$R/gpurun_out/prof_timed_summary.txt"
cd "$R"; python tools/summarize_prof.py gpurun_out/prof_timed | head -8 | tee -a gpurun_out/prof_timed_summary.txt
find gpurun_out/prof_timed -name "*kernel_trace.csv" -size +3M -delete
