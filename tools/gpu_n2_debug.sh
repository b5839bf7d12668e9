#!/bin/bash
# two ranks on ONE GPU (control flow + overlap of the exchange with the local-column part): the exchange is staged through the host with gloo
# (RCCL cannot host two ranks on one device), the phases of the product run through the C ABI (spla_amd/csrc/kernels/dist.hip)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export SPLA_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload sssp-step --scale ${SCALE:-21} --steps 20 --warmup 3 \
    > gpurun_out/bench_n2_sssp_gloo_debug.json 2> gpurun_out/bench_n2_sssp_gloo_debug.log
tail -3 gpurun_out/bench_n2_sssp_gloo_debug.log; cat gpurun_out/bench_n2_sssp_gloo_debug.json
SPLA_EXCHANGE=abi python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --scale ${SCALE:-21} --steps 20 --warmup 3 --no-variants \
    > gpurun_out/bench_n2_bfs_abi_gloo_debug.json 2> gpurun_out/bench_n2_bfs_abi_gloo_debug.log
tail -3 gpurun_out/bench_n2_bfs_abi_gloo_debug.log; cat gpurun_out/bench_n2_bfs_abi_gloo_debug.json
