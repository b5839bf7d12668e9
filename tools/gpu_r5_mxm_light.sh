cd $GRAFT_REPO_ROOT
for L in 4096 1024 256 16384; do SPLAHIP_MXM_LIGHT=$L timeout 300 python3 bench.py --workload mxm --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('light <= $L:', d['ms_per_step'], 'ms', d['value'], d['config'].get('row_classes'))"; done
