#!/bin/bash
# round 4: A/B of two builds of the library on the north-star call and the min-plus calls (tools/exp/exp_r4_ab.py), then the mxv / shard / config GPU tests.
# usage: gpu_r4_ab.sh <tag-of-the-baseline-build> [pytest args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
TAG=${1:-base}; shift
for rep in 1 2; do
  for t in "$TAG" ""; do
    echo "== lib tag '${t}' (rep $rep)"
    SPLA_LIB_TAG=$t timeout 300 python3 tools/exp/exp_r4_ab.py 23 2>&1 | tail -3
  done
done
timeout 1500 python3 -m pytest tests/test_mxv_gpu.py tests/test_shard_gpu.py tests/test_configs_gpu.py -x -q -m gpu --timeout 600 "$@" > gpurun_out/r4_ab_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r4_ab_tests.txt | tail -3
