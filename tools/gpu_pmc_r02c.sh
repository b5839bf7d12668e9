#!/bin/bash
# counter passes of the SSSP step after the switch to 2^15-row ranges (tools/gpu_pmc_r02.sh took it with 2^14)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
name=sssp_step
rm -rf "$R/gpurun_out/pmc_r02/$name"
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
  timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r02/$name/$c" -- python3 "$R/bench.py" --workload sssp-step --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2> "$R/gpurun_out/pmc_r02_${name}_$c.err"; echo "$name $c exit $?"
done
python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r02/$name" "cob_mxv_kernel<int" "cob_combine_kernel<int" > "$R/gpurun_out/r02_pmc_$name.json"; head -c 900 "$R/gpurun_out/r02_pmc_$name.json"; echo
find "$R/gpurun_out/pmc_r02/$name" -name "*.csv" -size +1M -delete
