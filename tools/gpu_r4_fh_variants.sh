#!/bin/bash
# round 4: compile-time variants of mxv_first_hit_kernel (SPLA_LIB_TAG builds: walk unroll, entries a lane walks before the wave takes over) on the BFS-step line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do
for tag in "" ${TAGS:-}; do
  SPLA_LIB_TAG=$tag timeout 300 python3 bench.py --workload bfs-step --steps 30 --warmup 5 --no-cpu-baseline --no-verify --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tag=%-8s' % '$tag', d['ms_per_step'], d['roofline']['kernel_avg_us'], d['roofline']['kernel'][:30])"
done
done
