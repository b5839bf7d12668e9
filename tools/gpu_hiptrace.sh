#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; S=${1:-tools/exp/exp_bfs_e2e.py}
cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/hiptrace"
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d "$R/gpurun_out/hiptrace" -- python3 "$R/$S" > "$R/gpurun_out/hiptrace.log" 2>&1; echo "exit $?"; tail -2 "$R/gpurun_out/hiptrace.log"
cd "$R"; f=$(ls gpurun_out/hiptrace/*/*hip_api_stats.csv 2>/dev/null | head -1); echo "$f"; head -25 "$f" | cut -c1-160
f=$(ls gpurun_out/hiptrace/*/*kernel_stats.csv | head -1); head -22 "$f" | cut -c1-200
find gpurun_out/hiptrace -name "*trace.csv" -size +2M -delete
