#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for tag in "" exp21 exp22; do
  rm -rf "$R/gpurun_out/prof_vp"
  SPLA_LIB_TAG=$tag SPLA_EXP_ONLY=A timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_vp" -- python3 "$R/tools/exp/exp_bits.py" x > /dev/null 2>&1
  echo "== ${tag:-base}"; python3 "$R/tools/summarize_prof.py" "$R/gpurun_out/prof_vp" | grep "vprep\|sell_mxv_kernel<int" | cut -c1-110
done
rm -rf "$R/gpurun_out/prof_vp"
