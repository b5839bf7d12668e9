#!/bin/bash
# round 4: the first-hit form of BFS-like steps — tests of everything the dispatch change touches, then the BFS-step line and the default line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r04b
timeout 1500 python3 -m pytest tests/test_mxv_gpu.py tests/test_api_gpu.py tests/test_algorithms.py tests/test_configs_gpu.py tests/test_golden.py -x -q -m gpu --timeout 600 > gpurun_out/r04b/tests.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error|Error" gpurun_out/r04b/tests.txt | tail -5
timeout 600 python3 bench.py --workload bfs-step --steps 20 --warmup 3 > gpurun_out/r04b/bfs_step.json 2> gpurun_out/r04b/bfs_step.err; echo "bfs-step exit $?"
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r04b/bfs_step.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'verified') if k in d}, d['roofline'].get('kernel'), d['roofline'].get('kernel_avg_us'), d['roofline'].get('frac'))
for v in d.get('variants', []):
    print('  -', v.get('name', '?')[:120], '|', v.get('kernel_avg_us'), '|', v.get('kernel', '')[:70])
PY
timeout 600 python3 bench.py --workload algorithms --steps 10 --warmup 2 > gpurun_out/r04b/algorithms.json 2> /dev/null; cut -c1-300 gpurun_out/r04b/algorithms.json; python3 -c "
import json; d=json.load(open('gpurun_out/r04b/algorithms.json')); print(d['algorithms'])"
