#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do for tag in r3 ""; do SPLA_LIB_TAG=$tag timeout 300 python3 tools/exp/exp_r4_ab.py 23 2>&1 | grep -v amdgpu.ids; done; done
