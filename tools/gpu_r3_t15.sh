#!/bin/bash
# round 3: sell attach phases after the LDS-privatised column count; vprep with preloaded quads; int32 on 2^15-slot ranges (how much do larger natural ranges give?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
  echo "== attach"; SPLAHIP_SELL_TRACE=1 timeout 600 python tools/exp/exp_sell_attach.py 22 2>&1 | grep -v "amdgpu.ids" | tail -40
  for sc in 22 23; do
    echo "== scale $sc int32 R 20224 vs 32768"; SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 15:0 2>&1 | grep "^cob\|schedule 0\|min-plus"
  done
  echo "== bfs variants"; timeout 900 python bench.py --workload bfs-step --steps 30 > gpurun_out/r03_bench_bfs.json 2> gpurun_out/r03_bench_bfs.log; tail -3 gpurun_out/r03_bench_bfs.log
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_bfs.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_avg_us"))
for v in d.get("variants", []):
    if "bfs-step" in v.get("name", ""):
        print(v["name"][:90], v["kernel_avg_us"])
PY
} 2>&1 | tee gpurun_out/r3_t15.txt
