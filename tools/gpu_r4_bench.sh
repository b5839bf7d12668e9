#!/bin/bash
# round 4: the default bench line as the driver runs it (timed), into gpurun_out/r04_bench.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
t0=$(date +%s)
timeout 1500 python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "bench exit $? in $(( $(date +%s) - t0 )) s"
tail -25 gpurun_out/r04_bench.err | cut -c1-300
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench.json'))
print({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'verified') if k in d}, d['roofline']['frac'], d.get('cpu_baseline', {}).get('value'))
for v in d.get('variants', []):
    r = v.get('roofline', {})
    print('  -', v.get('name', '?')[:110], '|', v.get('value', v.get('gteps', v.get('gteps_selected', ''))), v.get('unit', ''), '| us', v.get('kernel_avg_us', r.get('kernel_avg_us')), '| frac', r.get('frac', v.get('roofline_frac', v.get('roofline_frac_selected'))), '|', v.get('served_by', ''), v.get('skipped', ''), v.get('error', ''), v.get('variant_wall_s', ''))
PY
