#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for tag in "" nt; do
  echo "=== lib tag '$tag' (non-persistent kernel)"; SPLAHIP_NO_HOT=1 SPLA_LIB_TAG=$tag timeout 600 python tools/exp/exp_gather.py 22 2>&1 | grep -v amdgpu.ids
done
