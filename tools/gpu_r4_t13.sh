#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_mask" -- python3 "$R/tools/exp/exp_r4_mask.py" 23 > "$R/gpurun_out/r04_t13.txt" 2>&1
cd "$R"; python3 tools/summarize_prof.py gpurun_out/prof_mask | cut -c1-200 > gpurun_out/r04_t13_stats.txt; head -40 gpurun_out/r04_t13_stats.txt
