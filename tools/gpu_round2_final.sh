#!/bin/bash
# Round-2 final session on a GPU box: the whole GPU suite, smoke, the bench line (as the driver runs it), the other workloads, the scale sweep.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r02
echo "== pytest -m gpu"; timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error" gpurun_out/r02/pytest_gpu.txt | tail -2
echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02/smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r02/smoke.txt | cut -c1-200
echo "== bench"; timeout 900 python3 bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; echo "bench exit $?"; cut -c1-700 gpurun_out/r02/bench.json
for w in pagerank-vxm sssp-step tc-mxmT mxm transpose algorithms; do
  echo "== $w"; timeout 1500 python3 bench.py --workload $w --steps ${STEPS:-10} --warmup 2 > gpurun_out/r02/ops_$w.json 2> gpurun_out/r02/ops_$w.err; echo "exit $?"; cut -c1-500 gpurun_out/r02/ops_$w.json
done
for sc in 20 21 23 24; do
  echo "== scale $sc"; timeout 900 python3 bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-variants --no-scale23 > gpurun_out/r02/bench_scale$sc.json 2> /dev/null; echo "exit $?"; cut -c1-260 gpurun_out/r02/bench_scale$sc.json
done
