#!/bin/bash
# Round-4 final session on a GPU box: the whole GPU suite, smoke, the bench line as the driver runs it, the other workloads, then the counter passes and the
# kernel trace of the contract line.  Outputs under gpurun_out/r04/ and gpurun_out/r04_*; the keepers are copied to profiles/r04_* afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r04; export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu"; timeout 3000 python3 -m pytest tests -m gpu -x -q --timeout 900 > gpurun_out/r04/pytest_gpu.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error|within plain" gpurun_out/r04/pytest_gpu.txt | tail -8
echo "== smoke"; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r04/smoke.txt | cut -c1-200
fi
echo "== bench"; t0=$(date +%s); timeout 1500 python3 bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err; echo "bench exit $? in $(( $(date +%s) - t0 )) s"; cut -c1-900 gpurun_out/r04/bench.json
if [ -z "$SKIP_OPS" ]; then
for w in bfs-step algorithms transpose vxm-push; do
  echo "== $w"; timeout 1500 python3 bench.py --workload $w --steps ${STEPS:-10} --warmup 2 > gpurun_out/r04/ops_$w.json 2> gpurun_out/r04/ops_$w.err; echo "exit $?"; cut -c1-500 gpurun_out/r04/ops_$w.json
done
fi
echo "== pmc + trace"; timeout 2400 bash tools/gpu_pmc_r04.sh 2>&1 | tail -45 | cut -c1-300
echo "== pmc of the SSSP / PageRank / BFS steps"; timeout 2400 bash tools/gpu_pmc_r04_ops.sh 2>&1 | tail -14 | cut -c1-300
