#!/bin/bash
# One GPU-box session: parity tests, smoke, bench line, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; grep -v amdgpu.ids gpurun_out/bench.err | tail -6; cat gpurun_out/bench.json
cd /tmp; export TMPDIR=/tmp
echo "== rocprof kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-verify > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof_bench.err"; echo "rocprof exit $?"
echo "== rocprof kernel trace of the timed configuration only"
rm -rf "$R/gpurun_out/prof_timed"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_timed" -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-verify --no-variants > "$R/gpurun_out/prof_timed_bench.json" 2> /dev/null; echo "rocprof timed exit $?"
python3 "$R/tools/timed_kernel_from_trace.py" "$R/gpurun_out/prof_timed" 'sell_vprep_kernel<int>|sell_mxv_dual_kernel<int, splahip::StaBin<13>, splahip::StaBin<12> >|sell_post_kernel<int' 30 "$R/gpurun_out/prof_timed_bench.json" | tee "$R/gpurun_out/prof_timed_summary.txt"
echo "== rocprof pmc FETCH_SIZE"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants > /dev/null 2> "$R/gpurun_out/pmc_fetch.err"; echo "pmc fetch exit $?"
echo "== rocprof pmc WRITE_SIZE"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_write" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-variants > /dev/null 2> "$R/gpurun_out/pmc_write.err"; echo "pmc write exit $?"
cd "$R"
python tools/summarize_prof.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1
python tools/summarize_prof.py gpurun_out/pmc_fetch > gpurun_out/pmc_fetch_summary.txt 2>&1
python tools/summarize_prof.py gpurun_out/pmc_write > gpurun_out/pmc_write_summary.txt 2>&1
cat gpurun_out/prof_summary.txt; grep -E "FETCH|WRITE" gpurun_out/pmc_fetch_summary.txt gpurun_out/pmc_write_summary.txt | head -20
# keep the merged output small
find gpurun_out/prof gpurun_out/prof_timed gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection.csv" -size +8M -delete
