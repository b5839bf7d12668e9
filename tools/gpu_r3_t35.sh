#!/bin/bash
# round 3: the default bench command on the shipped build, with its wall time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{ t0=$(date +%s); timeout 900 python3 bench.py > gpurun_out/r03/bench_last.json 2> gpurun_out/r03/bench_last.err; echo "bench exit $? wall $(( $(date +%s) - t0 )) s"; cut -c1-330 gpurun_out/r03/bench_last.json; } 2>&1 | tee gpurun_out/r3_t35.txt
