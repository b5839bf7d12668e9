#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels in a built object:  tools/kernel_regs.py spla_amd/build/cob_hip.o [substring ...]
(the gfx950 code object is cut out of the object's .hip_fatbin section; numbers from its AMDGPU metadata note)"""
import re, subprocess, sys, tempfile, os
obj = sys.argv[1]
pats = sys.argv[2:]
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, d + "/fat.bin"], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + d + "/fat.bin", "--output=" + d + "/k.co", "--unbundle"], check=True)
    t = subprocess.run([LLVM + "llvm-readelf", "--notes", d + "/k.co"], capture_output=True, text=True).stdout
rows = []
for b in t.split("- .agpr_count")[1:]:
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
    rows.append((re.search(r"\.name:\s+(\S+)", b).group(1), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("vgpr spill sgpr  lds scratch  kernel")
for r, n in zip(rows, names):
    n = n.replace("splahip::", "").replace("(anonymous namespace)::", "")
    if not pats or all(p in n for p in pats):
        print(f"{r[1]:4d} {r[2]:5d} {r[3]:4d} {r[4]:5d} {r[5]:6d}  {n[:200]}")
