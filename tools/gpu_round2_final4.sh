#!/bin/bash
# last pass of round 2: full GPU suite, the lines of the workloads touched at the end (mxmT block merge in count mode, 512-word tile), mxmT counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r02
echo "== pytest -m gpu"; timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error" gpurun_out/r02/pytest_gpu.txt | tail -2
for w in tc-mxmT algorithms; do
  echo "== $w"; timeout 1500 python3 bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/r02/ops_$w.json 2> gpurun_out/r02/ops_$w.err; echo "exit $?"; cut -c1-260 gpurun_out/r02/ops_$w.json
done
cd /tmp; export TMPDIR=/tmp
name=tc_mxmT
rm -rf "$R/gpurun_out/pmc_r02/$name"
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
  timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r02/$name/$c" -- python3 "$R/bench.py" --workload tc-mxmT --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> "$R/gpurun_out/pmc_r02_${name}_$c.err"; echo "$name $c exit $?"
done
python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r02/$name" "mxmT_merge_kernel" "mxmT_heavy_kernel" "mxmT_entry_kernel" "mxmT_compact" "value_range_kernel" > "$R/gpurun_out/r02_pmc_$name.json"
find "$R/gpurun_out/pmc_r02/$name" -name "*.csv" -size +1M -delete
