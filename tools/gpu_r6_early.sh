#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_mxv_gpu.py tests/test_user_ops_gpu.py tests/test_golden.py tests/test_algorithms.py -x -q -m gpu --timeout 900 2>&1 | tail -8
timeout 300 python3 tools/exp/fuzz_mxv.py 150 60611 2>&1 | tail -2 | cut -c1-300
