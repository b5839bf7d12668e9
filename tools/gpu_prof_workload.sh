#!/bin/bash
# rocprofv3 kernel stats of one bench workload: tools/gpu_prof_workload.sh <workload> [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-mxm}; S=${2:-2}
cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_$W"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$W" -- python3 "$R/bench.py" --workload $W --steps $S --warmup 1 --no-cpu-baseline --no-verify > "$R/gpurun_out/prof_$W.json" 2> "$R/gpurun_out/prof_$W.err"; echo "rocprof exit $?"
cd "$R"; python tools/summarize_prof.py gpurun_out/prof_$W 2>&1 | head -40
find gpurun_out/prof_$W -name "*kernel_trace.csv" -size +3M -delete
