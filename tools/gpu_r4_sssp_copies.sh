#!/bin/bash
# round 4: memory copies of spla_Algorithm_sssp end to end (rocprofv3 --memory-copy-trace), by direction and size
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r04b"
rm -rf "$R/gpurun_out/r04b/prof_sssp_mc"
timeout 600 rocprofv3 --memory-copy-trace --kernel-trace -d "$R/gpurun_out/r04b/prof_sssp_mc" --output-format csv -- python3 "$R/tools/exp/exp_sssp_e2e.py" 2>&1 | grep "sssp e2e"
f=$(ls -t "$R"/gpurun_out/r04b/prof_sssp_mc/*/*memory_copy_trace.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), 'copies; columns:', list(rows[0].keys()) if rows else None)
agg = collections.Counter(); dur = collections.Counter()
for r in rows:
    size = int(r.get('Bytes', r.get('Size', 0)) or 0) if ('Bytes' in r or 'Size' in r) else 0
    key = (r.get('Direction', r.get('Kind', '?')), 'large' if size >= 1 << 20 else 'small')
    agg[key] += 1
    dur[key] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k in agg: print(k, agg[k], 'copies', round(dur[k] / agg[k], 1), 'us avg')
PY
find "$R/gpurun_out/r04b/prof_sssp_mc" -name "*.csv" -size +1M -delete
