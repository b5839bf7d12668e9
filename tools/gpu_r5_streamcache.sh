#!/bin/bash
# Round 5: the per-thread cache of released stream-ordered blocks (DevMem): workload lines with the cache on / off, then the whole GPU suite.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_streamcache.txt
: > $out
for c in 1 0 1 0; do
  for w in vxm-push pagerank-vxm algorithms transpose; do
    SPLAHIP_STREAM_CACHE=$c timeout 900 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
extra = {k: v for k, v in d.items() if k in ('bfs_ms','sssp_ms','pr_ms','tc_ms','lines')}
print('cache=$c', '$w', d.get('value'), d.get('unit'), d.get('ms_per_step'), 'ms', json.dumps(extra)[:600])" >> $out
  done
done
timeout 2400 python3 -m pytest tests -x -q -m gpu --timeout 900 2>&1 | tail -3 >> $out
