#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_bits"
SPLA_EXP_ONLY=A timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bits" -- python3 "$R/tools/exp/exp_bits.py" prof > /dev/null 2>&1; echo "rocprof exit $?"
cd "$R"; python tools/summarize_prof.py gpurun_out/prof_bits | grep -v "plan_\|sell_rowinfo\|sell_emit\|sell_permute\|sell_width\|sell_col\|sell_rank\|sell_fill\|sell_tail\|sell_cold_flag\|sell_cold_emit" | head -20
find gpurun_out/prof_bits -name "*kernel_trace.csv" -size +3M -delete
