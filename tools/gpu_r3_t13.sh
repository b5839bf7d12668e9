#!/bin/bash
# round 3: early exit over a bitmap of v; 4 / 5 taper levels; counters + kernel trace of the contract line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 1500 python -m pytest tests/test_mxv_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
  for sc in 22 23; do
    for lv in 3 4 5; do echo "== scale $sc levels $lv"; SPLAHIP_COB_LEVELS=$lv SPLAHIP_COB_TRACE=1 EXP_INT=0 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1" | sort -u; done
  done
  echo "== bfs variants"; timeout 900 python bench.py --workload bfs-step --steps 30 > gpurun_out/r03_bench_bfs.json 2> gpurun_out/r03_bench_bfs.log; tail -3 gpurun_out/r03_bench_bfs.log
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_bfs.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for v in d.get("variants", []):
    if "early" in v.get("name", ""):
        print({k: v[k] for k in v if k != "note"})
PY
  echo "== pmc"; timeout 2400 bash tools/gpu_pmc_r03.sh 2>&1 | tail -40
} 2>&1 | tee gpurun_out/r3_t13.txt
