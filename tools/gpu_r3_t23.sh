#!/bin/bash
# round 3: natural slot counts between 20224 and 32704 (does an intermediate size pay where 32704 has to be cut?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== scale 22"; EXP_SELL=0 SPLAHIP_COB_TRACE=1 timeout 1500 python tools/exp/exp_cob.py 22 0:0:0:40:32:20224 0:0:0:40:32:22528 0:0:0:40:32:24576 0:0:0:40:32:26624 0:0:0:40:32:28672 0:0:0:40:32:20224 0:0:0:40:32:22528 0:0:0:40:32:24576 0:0:0:40:32:26624 0:0:0:40:32:28672 2>&1 | grep "^cob\|schedule 1" | cut -c1-330
  echo "== scale 23"; EXP_SELL=0 timeout 1500 python tools/exp/exp_cob.py 23 0:0:0:40:32:32704 0:0:0:40:32:28672 0:0:0:40:32:24576 0:0:0:40:32:32704 0:0:0:40:32:28672 0:0:0:40:32:24576 2>&1 | grep "^cob" | cut -c1-100
} 2>&1 | tee gpurun_out/r3_t23.txt
