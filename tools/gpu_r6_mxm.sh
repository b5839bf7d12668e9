#!/bin/bash
# Round 6: mxm at R-MAT 18 and 19 in the -m gpu suite, the mxm-19 and mxm-dense bench workloads, the reference-order calls' times
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu --timeout 900 -k "cfg5_full_size_mxm_rmat or reference_order" -s > gpurun_out/r6_mxm_tests.txt 2>&1
grep -a "reference order\|passed\|failed\|Error" gpurun_out/r6_mxm_tests.txt | tail
timeout 600 python3 bench.py --workload mxm-19 > gpurun_out/r6_ops_mxm_scale19.json 2> gpurun_out/r6_ops_mxm_scale19.err; tail -3 gpurun_out/r6_ops_mxm_scale19.err; cut -c1-400 gpurun_out/r6_ops_mxm_scale19.json
timeout 900 python3 bench.py --workload mxm-dense > gpurun_out/r6_ops_mxm_dense.json 2> gpurun_out/r6_ops_mxm_dense.err; grep "mxm-dense" gpurun_out/r6_ops_mxm_dense.err; tail -2 gpurun_out/r6_ops_mxm_dense.err
