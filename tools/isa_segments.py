#!/usr/bin/env python3
"""Static scratch profile of a kernel's ISA: instructions / scratch loads / scratch stores between consecutive calls
(s_swappc), with the labels (loop headers) each segment carries.  Shows WHERE a kernel's spills sit — in the window loop's
straight-line code, or in rare exceptional-case blocks.
usage: hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -I <csrc> <csrc>/k_mnt4_753.hip -o k.s
       tools/isa_segments.py k.s lazy_a_kernel G2Cfg"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for f in re.split(r"\n(?=_Z[\w]+:\s)", txt):
    name = f.split(":")[0]
    if not name.startswith("_Z") or not all(p in name for p in pats):
        continue
    lines = [l for l in f.splitlines() if (l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")) or re.match(r"^\.LBB", l)]
    seg, cur = [], {"ld": 0, "st": 0, "n": 0, "lbl": []}
    for l in lines:
        if l.startswith(".LBB"):
            cur["lbl"].append(l.split(";")[-1].strip() if ";" in l else l.strip(":"))
            continue
        cur["n"] += 1
        cur["ld"] += "scratch_load" in l
        cur["st"] += "scratch_store" in l
        if "s_swappc" in l:
            seg.append(cur)
            cur = {"ld": 0, "st": 0, "n": 0, "lbl": []}
    seg.append(cur)
    print(name)
    print("segment  instrs  scratch_ld  scratch_st  labels")
    for i, s in enumerate(seg):
        print("%7d %7d %11d %11d  %s" % (i, s["n"], s["ld"], s["st"], "; ".join(s["lbl"][:3])))
    print("total: %d instructions, %d scratch loads, %d scratch stores, %d calls" % (sum(s["n"] for s in seg), sum(s["ld"] for s in seg), sum(s["st"] for s in seg), len(seg) - 1))
