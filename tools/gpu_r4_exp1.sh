#!/bin/bash
# round 4, session 1: where the time of the fp32 plus-times call over the column-ordered blocks goes (diagnostic builds), R-MAT 23
out=gpurun_out/r04_exp1.txt
: > $out
for tag in "" exp18 exp2 exp3; do
  SPLA_LIB_TAG=$tag EXP_CHECK=$([ -z "$tag" ] && echo 1 || echo 0) python tools/exp/exp_r4_cob.py 23 0 >> $out 2>&1
done
python tools/exp/exp_r4_cob.py 23 1 >> $out 2>&1
SPLA_LIB_TAG=exp18 EXP_CHECK=0 python tools/exp/exp_r4_cob.py 23 1 >> $out 2>&1
python tools/exp/exp_r4_cob.py 22 0 >> $out 2>&1
python tools/exp/exp_r4_cob.py 22 1 >> $out 2>&1
cat $out
