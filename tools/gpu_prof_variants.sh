#!/bin/bash
# kernel trace of the default bench run with its variants (per-kernel averages of every form)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"; rm -rf "$R/gpurun_out/prof_var"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_var" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify > "$R/gpurun_out/prof_var_bench.json" 2> /dev/null; echo "rocprof exit $?"
cd "$R"; python tools/summarize_prof.py gpurun_out/prof_var | grep -v "plan_\|sell_[a-z]*_kernel(\|sell_rowinfo\|sell_emit\|sell_permute\|sell_width\|sell_col\|sell_rank\|sell_fill\|sell_tail" | head -30 | tee gpurun_out/prof_var_summary.txt
find gpurun_out/prof_var -name "*kernel_trace.csv" -size +3M -delete
