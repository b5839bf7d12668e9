#!/bin/bash
# Round 6 closing session on the final build: the GPU test suite, the default bench line, the operator workloads, the soaks
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 2700 python3 -m pytest tests -x -q -m gpu --timeout 900 > gpurun_out/r6_pytest_gpu.txt 2>&1
grep -a "passed\|failed" gpurun_out/r6_pytest_gpu.txt | tail -2
timeout 1200 python3 bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r6_bench.json'))
print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_measured_read_stream'), d.get('verified'))
for v in d['variants']:
    print('  ', v.get('name','')[:80], v.get('value'), v.get('ms_per_step', v.get('kernel_avg_us')), v.get('roofline',{}).get('frac', v.get('roofline_frac')), v.get('ratio_to_all_rows'), v.get('verified', v.get('bit_identical_to_oracle')), v.get('error', v.get('skipped','')))
"
for w in algorithms mxm-dense vxm-push transpose; do timeout 900 python3 bench.py --workload $w > gpurun_out/r6_ops_$w.json 2> gpurun_out/r6_ops_$w.err; echo "$w exit $?"; cut -c1-500 gpurun_out/r6_ops_$w.json; echo; done
timeout 700 python3 tools/exp/fuzz_mxv.py ${SOAK:-400} 60621 > gpurun_out/r6_fuzz_mxv.txt 2>&1; tail -1 gpurun_out/r6_fuzz_mxv.txt | cut -c1-300
timeout 700 python3 tools/exp/fuzz_ops.py ${SOAK:-400} 60622 > gpurun_out/r6_fuzz_ops.txt 2>&1; tail -1 gpurun_out/r6_fuzz_ops.txt | cut -c1-300
