#!/bin/bash
# round 3: the two-rank run at the driver's N = 2 size with the graph's own labels (--permute 0) against owner-sorted labels (default)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SPLA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
{
  for pm in 0 2; do
    echo "== permute $pm"; timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2964$pm bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --permute $pm > gpurun_out/r03_n2_permute$pm.json 2> gpurun_out/r03_n2_permute$pm.log; echo "exit $?"
    python3 - $pm <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r03_n2_permute{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(d["verified"], d["config"]["nnz_rank0"], d["config"]["rows_rank0"], d["roofline"]["kernel_avg_us"], d["exchange"].get("local_part_avg_us"), d["exchange"].get("remote_part_avg_us"), d["shard"])
PY
    grep "rank\|rows\|nnz" gpurun_out/r03_n2_permute$pm.log | tail -3 | cut -c1-250
  done
} 2>&1 | tee gpurun_out/r3_t39.txt
