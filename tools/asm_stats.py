#!/usr/bin/env python3
"""Static instruction mix of kernels in a hipcc -save-temps .s file:  tools/asm_stats.py file.s prefix [prefix...]"""
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
for prefix in sys.argv[2:]:
    start = next((i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l), None)
    if start is None:
        print("not found:", prefix); continue
    ins, labels = [], 0
    for l in lines[start + 1:]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")): continue
        if t.split(";")[0].strip().endswith(":"):
            labels += 1; continue
        ins.append(t)
        if t.startswith("s_endpgm"): break
    c = Counter(l.split()[0] for l in ins)
    g = Counter()
    for k, v in c.items():
        g["valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_")) else "other"] += v
    print(f"{prefix[:70]}: {len(ins)} instructions, {labels} labels, {dict(g)}")
    print("   ", c.most_common(18))
