#!/bin/bash
# round 3: last check of the shipped build: whole GPU suite, smoke, the default bench command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
  echo "== pytest -m gpu"; timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error" gpurun_out/r03/pytest_gpu.txt | tail -2
  echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.txt 2>&1; echo "smoke exit $?"
  echo "== bench"; /usr/bin/time -v timeout 900 python3 bench.py > gpurun_out/r03/bench_last.json 2> gpurun_out/r03/bench_last.err; echo "bench exit $?"; grep "Elapsed" gpurun_out/r03/bench_last.err; cut -c1-400 gpurun_out/r03/bench_last.json
} 2>&1 | tee gpurun_out/r3_t34.txt
