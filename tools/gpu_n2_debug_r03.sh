#!/bin/bash
# round 3: two ranks on ONE GPU (control flow of the N > 1 paths; the transport is gloo staged through the host — RCCL cannot host two ranks on one device —
# while every phase of the products and of the sparse exchange runs through the C ABI, spla_amd/csrc/kernels/dist.hip):
#   1. the DEFAULT command (spmv-fp32, weak split, + the strong-split SSSP step of configs[3] as its variant) at reduced scales
#   2. the BFS step with its default sparse exchange (splahip_dist_sx_pack / _scatter)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SPLA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --scale ${SCALE:-20} --steps 20 --warmup 3 \
    > gpurun_out/r03_bench_n2_default_gloo_debug.json 2> gpurun_out/r03_bench_n2_default_gloo_debug.log
tail -4 gpurun_out/r03_bench_n2_default_gloo_debug.log; head -c 2500 gpurun_out/r03_bench_n2_default_gloo_debug.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29622 bench.py --gpus 2 --workload bfs-step --scale ${SCALE:-20} --steps 20 --warmup 3 --no-variants \
    > gpurun_out/r03_bench_n2_bfs_gloo_debug.json 2> gpurun_out/r03_bench_n2_bfs_gloo_debug.log
tail -4 gpurun_out/r03_bench_n2_bfs_gloo_debug.log; head -c 2500 gpurun_out/r03_bench_n2_bfs_gloo_debug.json; echo
