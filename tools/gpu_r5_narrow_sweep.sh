#!/bin/bash
# round 5: the packed streams (7 bytes per slot): gather-line weight of the piece cost, and the min-plus calls' forms
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for lp in 40 55 70 90; do echo "line_pct $lp:"; SPLAHIP_COB_LINE_PCT=$lp SPLA_ENV_AUTOSYNC=1 timeout 150 python3 tools/exp/exp_r4_ab.py 23 2>&1 | tail -1; done
echo "bits form off:"; SPLAHIP_COB_BITS=0 timeout 150 python3 tools/exp/exp_r4_ab.py 23 2>&1 | tail -1
echo "scale 22 line_pct 70:"; SPLAHIP_COB_LINE_PCT=70 timeout 150 python3 tools/exp/exp_r4_ab.py 22 2>&1 | tail -1
echo "scale 24 line_pct 70:"; SPLAHIP_COB_LINE_PCT=70 timeout 150 python3 tools/exp/exp_r4_ab.py 24 2>&1 | tail -1
