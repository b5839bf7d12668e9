#!/bin/bash
# rounds 4 - 6: two ranks on ONE GPU (control flow of the N > 1 paths; gloo staged through the host between the C-ABI phases — RCCL cannot host two ranks on one device):
#   1. the DEFAULT command at reduced scales: the weak-split headline step + its variants (SSSP step of configs[3], the raw-label step, mxmT by rows of the mask)
#   2. --workload tc-mxmT and --workload mxm across the two ranks (rows of the mask / of A, B whole on every rank)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06n2
export SPLA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --scale ${SCALE:-19} --steps 10 --warmup 2 \
    > gpurun_out/r06n2/bench_n2_default_gloo_debug.json 2> gpurun_out/r06n2/bench_n2_default_gloo_debug.log
echo "default exit $?"; tail -3 gpurun_out/r06n2/bench_n2_default_gloo_debug.log | cut -c1-300
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r06n2/bench_n2_default_gloo_debug.json').read().strip().splitlines()[-1])  # (gloo prints a banner line on stdout)
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'verified') if k in d})
for v in d.get('variants', []):
    print('  -', v.get('name', '?')[:110], '|', v.get('value'), v.get('unit'), '|', v.get('ms_per_step'), '| verified', v.get('verified'), '|', (v.get('config') or {}).get('triangles', ''))
PY
for w in tc-mxmT mxm; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29632 bench.py --gpus 2 --workload $w --scale ${SPG_SCALE:-16} --steps 5 --warmup 1 \
      > gpurun_out/r06n2/bench_n2_${w}_gloo_debug.json 2> gpurun_out/r06n2/bench_n2_${w}_gloo_debug.log
  echo "$w exit $?"; tail -2 gpurun_out/r06n2/bench_n2_${w}_gloo_debug.log | cut -c1-300; head -c 1500 gpurun_out/r06n2/bench_n2_${w}_gloo_debug.json; echo
done
