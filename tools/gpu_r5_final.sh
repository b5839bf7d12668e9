#!/bin/bash
# Round 5 closing session on the final build: the GPU test suite, the default bench line, then the two randomised soaks against the oracle (new seeds).
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -x -q -m gpu --timeout 900 > gpurun_out/r5_pytest_gpu.txt 2>&1
tail -3 gpurun_out/r5_pytest_gpu.txt
timeout 900 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
cat gpurun_out/r5_bench.json
timeout 400 python3 tools/exp/fuzz_mxv.py 240 50501 > gpurun_out/r5_fuzz_mxv.txt 2>&1; tail -3 gpurun_out/r5_fuzz_mxv.txt
timeout 400 python3 tools/exp/fuzz_ops.py 240 50502 > gpurun_out/r5_fuzz_ops.txt 2>&1; tail -3 gpurun_out/r5_fuzz_ops.txt
