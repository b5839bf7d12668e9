#!/bin/bash
# round-2 check on the GPU box: the whole GPU suite, then the counter passes of the fp32 / int SpMV experiment (scale 22)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r02_pytest_gpu.txt
cd /tmp; export TMPDIR=/tmp
export EXP_SELL=1 EXP_INT=1
name=spmv_fp32_scale22
rm -rf "$R/gpurun_out/pmc_r02/$name"
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ; do
  timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r02/$name/$c" -- python3 "$R/tools/exp/exp_cob.py" 22 14:0 > /dev/null 2> "$R/gpurun_out/pmc_r02_${name}_$c.err"; echo "$name $c exit $?"
done
python3 "$R/tools/pmc_kernels.py" "$R/gpurun_out/pmc_r02/$name" "cob_mxv_kernel<float" "cob_combine_kernel<float" "cob_mxv_kernel<int" "cob_combine_kernel<int" "spmv_wavetile_kernel<splahip::MxvCfg, float" "sell_mxv_kernel<float" > "$R/gpurun_out/r02_pmc_$name.json"; head -c 2500 "$R/gpurun_out/r02_pmc_$name.json"; echo
find "$R/gpurun_out/pmc_r02/$name" -name "*.csv" -size +1M -delete
