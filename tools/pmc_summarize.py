#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one directory per pass) into a per-kernel, per-launch
summary: profiles/rNN_pmc_summary.json.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (128-B streaming requests are tallied at 64 B).

usage: pmc_summarize.py <dir-with-pass-subdirs> <out.json> [command string]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").strip()
    base = n.split("<")[0].split("::")[-1]
    grp = ""
    for tag in ("BlsG1", "BlsG2", "Bw6G1", "Bw6G2", "Mnt4G1", "Mnt4G2", "Mnt6G1", "Mnt6G2"):
        if tag + "Cfg" in name:
            grp = "<" + tag + ">"
    return base + grp


def main():
    root, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(set))
    regs = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            launches[k][c].add((f, row.get("Dispatch_Id", row.get("Correlation_Id", ""))))
            # NB rocprofv3's VGPR_Count column is NOT the allocation on gfx950's unified VGPR/AGPR file (it showed 128 for a
            # kernel whose code object allocates 256): registers, spills and occupancy come from the code objects themselves —
            # tools/kernel_resources.sh -> profiles/rNN_kernel_resources*.txt.  Scratch and LDS below are per-dispatch facts.
            regs[k] = {"scratch_bytes_per_lane": int(row.get("Scratch_Size", 0) or 0), "lds_bytes_per_block": int(row.get("LDS_Block_Size", 0) or 0),
                       "grid": int(row.get("Grid_Size", 0) or 0), "workgroup": int(row.get("Workgroup_Size", 0) or 0)}
    kernels = {}
    for k, cs in acc.items():
        d = dict(regs.get(k, {}))
        for c, v in cs.items():
            n = max(1, len(launches[k][c]))
            d[c + "_per_launch"] = v / n
            d[c + "_launches"] = n
        if "FETCH_SIZE_per_launch" in d and "WRITE_SIZE_per_launch" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE_per_launch"] + d["WRITE_SIZE_per_launch"]) * 1024.0
            d["hbm_bytes_per_launch_raw_fetch"] = (d["FETCH_SIZE_per_launch"] + d["WRITE_SIZE_per_launch"]) * 1024.0
        if "SQ_WAIT_ANY_per_launch" in d and d.get("SQ_WAVE_CYCLES_per_launch"):
            d["wait_any_frac_of_wave_cycles"] = d["SQ_WAIT_ANY_per_launch"] / d["SQ_WAVE_CYCLES_per_launch"]
        kernels[k] = d
    json.dump({"source": "rocprofv3 --pmc <counter> (separate passes) -- " + cmd,
               "unit_note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half "
                            "of a streaming read); Infinity-Cache hits are counted, so window-table re-reads appear here",
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, "kernels:", ", ".join(sorted(kernels)))


if __name__ == "__main__":
    main()
