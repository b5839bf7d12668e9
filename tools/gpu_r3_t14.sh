#!/bin/bash
# round 3: hub slots in row order within a range; sliced-ELL window descriptors; FETCH_SIZE calibration for gathers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
  for sc in 22 23; do
    for ro in 1 0; do echo "== scale $sc hub_row_order $ro"; SPLAHIP_COB_HUB_ROW_ORDER=$ro SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1\|min-plus" | sort -u; done
  done
  echo "== sector calibration"; timeout 900 bash tools/gpu_sector_calib.sh 2>&1 | tail -30
  echo "== bfs variants"; timeout 900 python bench.py --workload bfs-step --steps 30 > gpurun_out/r03_bench_bfs.json 2> gpurun_out/r03_bench_bfs.log; tail -3 gpurun_out/r03_bench_bfs.log
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_bfs.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_avg_us"))
for v in d.get("variants", []):
    if "bfs-step" in v.get("name", ""):
        print(v["name"][:90], v["kernel_avg_us"])
PY
} 2>&1 | tee gpurun_out/r3_t14.txt
