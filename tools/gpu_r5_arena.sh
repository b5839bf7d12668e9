#!/bin/bash
# Round 5: the scratch arena of the blocking calls: mxm / mxmT / algorithm parity tests, the two workload lines with the arena on and off.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_arena.txt
: > $out
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "mxm or tc or algorithm or shard" --timeout 900 2>&1 | tail -3 >> $out
for a in 1 0 1 0; do
  for w in tc-mxmT mxm; do
    SPLAHIP_SCRATCH_ARENA=$a timeout 600 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena=$a', '$w', d['value'], d['unit'], d['ms_per_step'], 'ms')" >> $out
  done
done
