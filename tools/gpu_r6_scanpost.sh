#!/bin/bash
# Round 6: the compactions' count posted to the host by the scan pass (SPLAHIP_SCAN_POST=1) against a mailbox launch behind the write pass (0)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_scanpost.txt
: > $out
timeout 900 python3 -m pytest tests/test_algorithms.py tests/test_api_gpu.py tests/test_vxm_gpu.py tests/test_reference_suite_gpu.py tests/test_user_ops_gpu.py tests/test_golden_next.py -x -q -m gpu --timeout 600 2>&1 | grep -a "passed\|failed" | tail -1 | tee -a $out
timeout 200 python3 tools/exp/fuzz_ops.py 100 60641 2>&1 | tail -1 | cut -c1-200 | tee -a $out
timeout 200 python3 tools/exp/fuzz_algos.py 60 60642 2>&1 | tail -1 | cut -c1-200 | tee -a $out
sed -e 's/SPLAHIP_MAILBOX_POLL_US/SPLAHIP_SCAN_POST/; s/for us in ("0", "100", "400"):/for us in ("0", "1"):/; s/poll_us/scan_post/' tools/exp/exp_r6_mailbox_ab.py > /tmp/ab_scanpost.py
sed -i "s#sys.path.insert(0, str(Path(__file__).resolve().parents\[2\]))#sys.path.insert(0, '$GRAFT_REPO_ROOT')#" /tmp/ab_scanpost.py
timeout 300 python3 /tmp/ab_scanpost.py 2>&1 | grep round | tee -a $out
