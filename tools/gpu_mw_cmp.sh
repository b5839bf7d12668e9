#!/bin/bash
# int min-plus cob kernel: compiler's register budget (default build) vs a forced 8 waves / SIMD (mw8 build); scale 22 and 24
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
export EXP_SELL=0 EXP_INT=1
for tag in "" mw8; do
  echo "== lib tag '$tag'"
  SPLA_LIB_TAG=$tag timeout 200 python3 tools/exp/exp_cob.py 22 14:0 15:0 2>&1 | grep -v "^csr"
  SPLA_LIB_TAG=$tag timeout 200 python3 tools/exp/exp_cob.py 24 14:0 15:0 2>&1 | grep -v "^csr"
done
