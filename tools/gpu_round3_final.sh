#!/bin/bash
# Round-3 final session on a GPU box: the whole GPU suite, smoke, the bench line (as the driver runs it), the R-MAT 22 line, the other workloads, the
# two-rank control-flow runs, then the counter passes and the kernel trace of the contract line.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r03; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error" gpurun_out/r03/pytest_gpu.txt | tail -2
echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r03/smoke.txt | cut -c1-200
echo "== bench"; timeout 900 python3 bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; echo "bench exit $?"; cut -c1-1500 gpurun_out/r03/bench.json
echo "== bench --scale 22"; timeout 900 python3 bench.py --scale 22 --no-cpu-baseline > gpurun_out/r03/bench_scale22.json 2> /dev/null; echo "exit $?"; cut -c1-1200 gpurun_out/r03/bench_scale22.json
echo "== bench --workload bfs-step"; timeout 900 python3 bench.py --workload bfs-step > gpurun_out/r03/bench_bfs_step.json 2> /dev/null; echo "exit $?"; cut -c1-700 gpurun_out/r03/bench_bfs_step.json
for w in pagerank-vxm sssp-step tc-mxmT mxm algorithms; do
  echo "== $w"; timeout 1500 python3 bench.py --workload $w --steps ${STEPS:-10} --warmup 2 > gpurun_out/r03/ops_$w.json 2> gpurun_out/r03/ops_$w.err; echo "exit $?"; cut -c1-500 gpurun_out/r03/ops_$w.json
done
echo "== n2 debug"; timeout 1200 bash tools/gpu_n2_debug_r03.sh 2>&1 | tail -6 | cut -c1-400
echo "== pmc + trace"; timeout 2400 bash tools/gpu_pmc_r03.sh 2>&1 | tail -45 | cut -c1-300
echo "== pmc of the SSSP / PageRank steps"; timeout 2400 bash tools/gpu_pmc_r03_ops.sh 2>&1 | tail -12 | cut -c1-300
for sc in 22 23 24; do echo "== layout chosen at scale $sc"; SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python3 tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1\|min-plus" | cut -c1-420; done
