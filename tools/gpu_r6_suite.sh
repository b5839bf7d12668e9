#!/bin/bash
# Round 6: the whole -m gpu suite and the default bench line
export TMPDIR=/tmp
timeout 2700 python3 -m pytest tests -x -q -m gpu --timeout 900 > gpurun_out/r6_pytest_gpu.txt 2>&1
tail -25 gpurun_out/r6_pytest_gpu.txt | cut -c1-250
timeout 1200 python3 bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r6_bench.json'))
print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_measured_read_stream'), d.get('verified'))
for v in d['variants']:
    print('  ', v.get('name','')[:90], v.get('value'), v.get('unit'), v.get('ms_per_step', v.get('kernel_avg_us')), v.get('roofline',{}).get('frac', v.get('roofline_frac')), v.get('ratio_to_all_rows'), v.get('verified', v.get('bit_identical_to_oracle')), v.get('error', v.get('skipped','')))
"
tail -3 gpurun_out/r6_bench.err
