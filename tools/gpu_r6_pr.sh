#!/bin/bash
# Round 6: PageRank step (configs[2] stand-in): natural-range geometry sweep
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 600 python3 bench.py --workload pagerank-vxm --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/r6_pr.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], r['frac'], r.get('kernel_avg_us'), d.get('verified'), d['config'].get('layout'))"; grep -a "\[cob\] schedule" gpurun_out/r6_pr.err | tail -2 | cut -c1-400; }
run A=1
run SPLAHIP_COB_NAT_MAX_DEGREE=160 SPLAHIP_COB_NAT_SLOTS=32704
run SPLAHIP_COB_NAT_MAX_DEGREE=160 SPLAHIP_COB_NAT_SLOTS=40832
run SPLAHIP_COB_NAT_MAX_DEGREE=320 SPLAHIP_COB_NAT_SLOTS=40832 SPLAHIP_COB_HUB_MIN_DEGREE=304
run SPLAHIP_COB_HUB_RANGES=16
