#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for tag in "" nog nof nogf; do
  echo "=== lib tag '$tag' (sliced-ELL copy)"; SPLA_LIB_TAG=$tag timeout 600 python tools/exp/exp_gather.py 22 --sell 2>&1 | grep -v amdgpu.ids | head -3
done
