#!/bin/bash
# round 3: the two-rank control-flow run at the REAL per-rank size of the driver's N = 2 run (R-MAT 24 split in two; gloo between the ranks, both on one GPU)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SPLA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
{ timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r03_bench_n2_fullsize_gloo_debug.json 2> gpurun_out/r03_bench_n2_fullsize_gloo_debug.log; echo "exit $?"; tail -5 gpurun_out/r03_bench_n2_fullsize_gloo_debug.log | cut -c1-300; head -c 3000 gpurun_out/r03_bench_n2_fullsize_gloo_debug.json; echo; } 2>&1 | tee gpurun_out/r3_t36.txt
