#!/bin/bash
# Round 5 closing session on the final build: the GPU test suite, the default bench line, two ranks on one GPU (control flow of the N > 1 paths), the counter passes
# of the two SpGEMM workloads and of the headline call + the kernel statistics of the contract-line command, then the two randomised soaks against the oracle.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu --timeout 900 > gpurun_out/r5_pytest_gpu.txt 2>&1
grep "passed\|failed" gpurun_out/r5_pytest_gpu.txt
timeout 900 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r5_bench.json'))
print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['verified'])
for v in d['variants']:
    if 'value' in v: print('  ', v.get('name','')[:60], v['value'], v['unit'], v['ms_per_step'], v.get('verified'))
"
SCALE=19 bash tools/gpu_n2_debug.sh > gpurun_out/r5_n2_debug.txt 2>&1; tail -12 gpurun_out/r5_n2_debug.txt | cut -c1-200
ONLY="mxm tc_mxmT" bash tools/gpu_pmc_r05_ops.sh 2>&1 | tail -4
bash tools/gpu_pmc_r05.sh > gpurun_out/r5_pmc_headline.log 2>&1; grep "traffic_bytes" gpurun_out/r5_pmc_headline.log | cut -c1-300
cd $R
timeout 400 python3 tools/exp/fuzz_mxv.py 200 50503 > gpurun_out/r5_fuzz_mxv.txt 2>&1; tail -1 gpurun_out/r5_fuzz_mxv.txt
timeout 400 python3 tools/exp/fuzz_ops.py 200 50504 > gpurun_out/r5_fuzz_ops.txt 2>&1; tail -1 gpurun_out/r5_fuzz_ops.txt
