#!/bin/bash
# the bench line at other problem sizes (same workload definition), one JSON line each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for sc in 20 21 23 24; do
  timeout 1200 python bench.py --scale $sc --steps 30 --warmup 5 --no-variants > gpurun_out/bench_scale$sc.json 2> gpurun_out/bench_scale$sc.err; echo "scale $sc exit $?"
  grep "parity\|attached" gpurun_out/bench_scale$sc.err | cut -c1-160
  python - gpurun_out/bench_scale$sc.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "verified")}, d["config"]["nnz"], d["config"]["nnz_selected"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
done
