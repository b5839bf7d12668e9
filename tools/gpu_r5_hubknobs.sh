#!/bin/bash
# round 5: more rows in hub ranges? (a natural chunk walks at ~220 ns, a hub chunk at ~90)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for s in 23 22; do
for kv in "X=1" "SPLAHIP_COB_HUB_MIN_DEGREE=24" "SPLAHIP_COB_HUB_MIN_DEGREE=16" "SPLAHIP_COB_HUB_MIN_DEGREE=12" "SPLAHIP_COB_HUB_MIN_DEGREE=8"; do
  EXP_CHECK=0 timeout 120 python3 tools/exp/exp_r4_cob.py $s 0 $kv 2>&1 | tail -1 | cut -c1-330
done; done
