#!/bin/bash
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_threads_gpu.py tests/test_mxm_gpu.py tests/test_vxm_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -15
timeout 900 python3 bench.py --workload mxm-dense > gpurun_out/r6_ops_mxm_dense.json 2> gpurun_out/r6_ops_mxm_dense.err; grep "mxm-dense parity" gpurun_out/r6_ops_mxm_dense.err | cut -c1-600; tail -2 gpurun_out/r6_ops_mxm_dense.err | cut -c1-300
