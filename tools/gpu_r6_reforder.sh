#!/bin/bash
# Round 6, step 1: the reference-order mode (splahip_set_reference_order) — unit tests (every kernel form against the oracle directly), the operator API test, the two
# full-size bit-exact tests.
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_mxv_gpu.py tests/test_api_gpu.py tests/test_vxm_gpu.py tests/test_golden.py -x -q -m gpu --timeout 900 > gpurun_out/r6_reforder_unit.txt 2>&1
tail -5 gpurun_out/r6_reforder_unit.txt
timeout 1500 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu --timeout 900 -k "reference_order" -s > gpurun_out/r6_reforder_full.txt 2>&1
grep -a "reference order\|passed\|failed\|Error\|error" gpurun_out/r6_reforder_full.txt | tail -20
