#!/usr/bin/env python3
"""Per-(kernel, grid size) averages of the column-ordered-blocks kernels in a rocprofv3 kernel trace: the default bench command runs the fp32 plus-times
kernel on two matrices (R-MAT 23 for the contract line, R-MAT 22 inside variants[0]), so the per-kernel line of --stats mixes the two.
    python tools/cob_trace_groups.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "cob_mxv_kernel<float, splahip::StaBin<2>" in n or "cob_hub_finish_kernel<float, 0" in n:
        g[(n.split("(")[0].replace("void splahip::", "")[:70], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("fp32 MULT / PLUS kernels of the column-ordered blocks by grid size (threads = 1024 per piece; the larger grid is the R-MAT 23 matrix of the contract line, the smaller the R-MAT 22 matrix inside variants[0]), from the kernel trace of the same command:")
print(f"{'calls':>6} {'avg us':>9} {'min':>9} {'max':>9}  kernel, grid")
for k, v in sorted(g.items()):
    print(f"{len(v):6d} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f}  {k[0]}, {k[1]}")
