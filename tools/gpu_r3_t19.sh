#!/bin/bash
# round 3: one workgroup per CU for the 4-byte slices too: the other int paths (sparse-vector forms, SSSP step at R-MAT 24, BFS step variants), parity
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 2400 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
  for pc in 1 2; do
    echo "== sssp-step R-MAT 24, workgroups per CU $pc"; SPLAHIP_COB_PER_CU=$pc timeout 900 python bench.py --workload sssp-step --steps 20 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))"
    echo "== pagerank vxm, workgroups per CU $pc"; SPLAHIP_COB_PER_CU=$pc timeout 900 python bench.py --workload pagerank-vxm --steps 20 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))"
    echo "== bfs variants, workgroups per CU $pc"; SPLAHIP_COB_PER_CU=$pc timeout 900 python bench.py --workload bfs-step --steps 30 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for v in d.get('variants', []):
    if 'column-ordered' in v.get('name','') or 'min-plus' in v.get('name','') or 'fp32' in v.get('name',''): print(v['name'][:110], v['kernel_avg_us'])"
  done
} 2>&1 | tee gpurun_out/r3_t19.txt
