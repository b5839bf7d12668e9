#!/bin/bash
# round 3: A/B of the schedule / geometry knobs on one box, twice: gather-line weight x hub ranges x natural slots
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  C22="0:0:0:25:16:20224 0:0:0:40:16:20224 0:0:0:60:16:20224 0:0:0:40:16:32704 0:0:0:25:16:20224 0:0:0:40:16:20224 0:0:0:60:16:20224 0:0:0:40:16:32704"
  C23="0:0:0:25:16:20224 0:0:0:40:16:20224 0:0:0:60:16:20224 0:0:0:40:24:20224 0:0:0:40:24:32704 0:0:0:25:24:32704 0:0:0:25:16:20224 0:0:0:40:16:20224 0:0:0:60:16:20224 0:0:0:40:24:20224 0:0:0:40:24:32704 0:0:0:25:24:32704"
  echo "== scale 22"; EXP_INT=1 EXP_SELL=0 timeout 1500 python tools/exp/exp_cob.py 22 $C22 2>&1 | grep "^cob\|min-plus" | cut -c1-64
  echo "== scale 23"; EXP_INT=1 EXP_SELL=0 timeout 2400 python tools/exp/exp_cob.py 23 $C23 2>&1 | grep "^cob\|min-plus" | cut -c1-64
  echo "== scale 24"; EXP_INT=1 EXP_SELL=0 timeout 2400 python tools/exp/exp_cob.py 24 0:0:0:25:16:20224 0:0:0:40:16:20224 0:0:0:40:32:32704 0:0:0:25:16:20224 0:0:0:40:16:20224 0:0:0:40:32:32704 2>&1 | grep "^cob\|min-plus" | cut -c1-64
} 2>&1 | tee gpurun_out/r3_t22.txt
