#!/bin/bash
# round 3: natural pieces first, hub pieces as the tail (SPLAHIP_COB_ORDER=2) against dearest-first (0); fp32 LDS atomics with return value / inline asm
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23 24; do
    for om in 0 2 0 2; do echo "== scale $sc order $om"; SPLAHIP_COB_ORDER=$om SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|min-plus\|schedule 1" | cut -c1-75 | grep -v "^\[cob\]" ; SPLAHIP_COB_ORDER=$om SPLAHIP_COB_TRACE=1 true; done
  done
  echo "== lds atomics"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/lds_bench2 tools/exp/lds_atomic_bench2.hip 2>/dev/null && timeout 300 /tmp/lds_bench2 | grep "pattern\|f32\|f64\|CAS"
} 2>&1 | tee gpurun_out/r3_t25.txt
