#!/bin/bash
# the *.perf* cases of the reference's own unit tests on the HIP backend (lap times in ms, as spla::Timer::print writes them)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for t in mxv vxm mxm mxmT; do
  echo "== reftest_$t --gtest_filter=*perf*"; SECONDS=0
  timeout ${T:-500} oracle/_ref/reftest_$t --gtest_filter='*perf*' --gtest_color=no 2>&1 | grep -v "^\[ *RUN\|^Note\|^\[---\|^\[===\|^$" | cut -c1-400 | tail -8; echo "wall ${SECONDS}s"
done 2>&1 | tee gpurun_out/reftests_perf.txt
