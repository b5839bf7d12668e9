#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/pmc_fetch" "$R/gpurun_out/pmc_write"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants > /dev/null 2> "$R/gpurun_out/pmc_fetch.err"; echo "pmc fetch exit $?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_write" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants > /dev/null 2> "$R/gpurun_out/pmc_write.err"; echo "pmc write exit $?"
cd "$R"; python tools/summarize_prof.py gpurun_out/pmc_fetch > gpurun_out/pmc_fetch_summary.txt 2>&1; python tools/summarize_prof.py gpurun_out/pmc_write > gpurun_out/pmc_write_summary.txt 2>&1
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -size +3M -delete
