#!/bin/bash
# round 3: what layout / schedule the attach picks for the blocks of the two-rank run (trace on)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SPLA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp SPLAHIP_COB_TRACE=1 SPLAHIP_DIST_FORM=whole
{ timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29651 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/r03_n2_trace.json 2> gpurun_out/r03_n2_trace.log; echo "exit $?"; grep "\[cob\] schedule 1" gpurun_out/r03_n2_trace.log | cut -c1-330 | sort | uniq -c | head -12; } 2>&1 | tee gpurun_out/r3_t40.txt
