#!/bin/bash
# One GPU-box session: parity tests, smoke, bench line, rocprofv3 kernel trace.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; tail -12 gpurun_out/bench.err; cat gpurun_out/bench.json
echo "== rocprof"; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof_bench.err"; echo "rocprof exit $?"
find "$R/gpurun_out/prof" -name "*kernel_stats*" | head -3
f=$(find "$R/gpurun_out/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
