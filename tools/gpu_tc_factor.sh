#!/bin/bash
# mxmT triangle form: block-merge threshold sweep + kernel split per setting
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for f in 32 128 100000; do
  if [ $f = 0 ]; then export SPLAHIP_MXMT_MERGE=0; else export SPLAHIP_MXMT_MERGE=1 SPLAHIP_MXMT_MERGE_FACTOR=$f; fi
  rm -rf $R/gpurun_out/prof_tcf
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tcf --output-format csv -- python3 $R/bench.py --workload tc-mxmT --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /tmp/line.json 2>/dev/null
  python3 - $f <<'PY'
import csv,glob,sys,json,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
fs=glob.glob(R+'/gpurun_out/prof_tcf/*/*kernel_stats.csv')
ks={}
for r in csv.DictReader(open(fs[0])):
    n=r['Name']
    for k in ('mxmT_merge','mxmT_entry','mxmT_heavy','mxmT_compact'):
        if k in n: ks[k]=round(float(r['AverageNs'])/1e6,2)
try: ms=json.load(open('/tmp/line.json'))['ms_per_step']
except Exception as e: ms=None
print('factor',sys.argv[1],'ms/step (under rocprof)',ms,ks)
PY
done
