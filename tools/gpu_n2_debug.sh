#!/bin/bash
# control-flow check of the N=2 bench path on a 1-GPU box: two ranks share GPU 0, exchange staged through gloo; both exchange forms
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
for mode in dense sparse; do
  SPLA_EXCHANGE=$mode SPLA_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
     bench.py --gpus 2 --steps 5 --warmup 2 --scale 18 > gpurun_out/bench_n2_debug_$mode.json 2> gpurun_out/bench_n2_debug_$mode.err
  echo "$mode exit $?"; grep -E "^\[bench\]|Error|error" gpurun_out/bench_n2_debug_$mode.err | tail -5
  python - gpurun_out/bench_n2_debug_$mode.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "verified")}, d["exchange"])
PY
done
