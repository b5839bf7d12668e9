#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for m in 1 2 3 0; do echo "SPLAHIP_MXMT_ROWS=$m"; SPLAHIP_MXMT_ROWS=$m timeout 900 python bench.py --workload tc-mxmT --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-150; done
