#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_sc"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_sc" -- python3 "$R/bench.py" --scale ${1:-20} --steps 30 --warmup 5 --no-cpu-baseline --no-verify --no-variants > /dev/null 2>&1; echo "exit $?"
cd "$R"; python tools/summarize_prof.py gpurun_out/prof_sc | grep "sell_vprep\|sell_mxv_dual\|sell_post" | cut -c1-120
rm -rf gpurun_out/prof_sc
