#!/bin/bash
# control-flow check of the N=2 bench path on a 1-GPU box: two ranks share GPU 0, exchange staged through gloo
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
SPLA_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus 2 --steps 5 --warmup 2 --scale 18 > gpurun_out/bench_n2_debug.json 2> gpurun_out/bench_n2_debug.err
echo "exit $?"; grep -v amdgpu.ids gpurun_out/bench_n2_debug.err | tail -15; cat gpurun_out/bench_n2_debug.json | cut -c1-1500
