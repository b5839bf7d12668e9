#!/bin/bash
# the other hot-path operators: one JSON line each into gpurun_out/ops_<name>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for w in ${WORKLOADS:-pagerank-vxm sssp-step tc-mxmT mxm}; do
  echo "== $w"; timeout 1500 python bench.py --workload $w --steps ${STEPS:-20} --warmup 3 > gpurun_out/ops_$w.json 2> gpurun_out/ops_$w.err; echo "exit $?"
  grep "^\[bench\]" gpurun_out/ops_$w.err | tail -4; tail -2 gpurun_out/ops_$w.err | grep -v "^\[bench\]" | cut -c1-300; cut -c1-900 gpurun_out/ops_$w.json
done
