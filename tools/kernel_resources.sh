#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy as the compiler allocates them (the code objects' own numbers — rocprofv3's
# VGPR_Count column is not the allocation on gfx950's unified register file).  Recompiles the four curve TUs with
# -Rpass-analysis=kernel-resource-usage (CPU box, ~2 min on 8 cores) and prints one line per kernel:
#   tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
set -e
CS=$(cd "$(dirname "$0")/../snark-setup-operator_amd/csrc" && pwd)
T=$(mktemp -d)
for f in k_bls12_377 k_bw6_761 k_mnt4_753 k_mnt6_753; do
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c -I "$CS" "$CS/$f.hip" -o "$T/$f.o" \
      -Rpass-analysis=kernel-resource-usage 2> "$T/$f.log" || true ) &
done
wait
python3 - "$T" <<'PY'
import re, subprocess, sys, glob, os
rows = []
for log in sorted(glob.glob(os.path.join(sys.argv[1], "*.log"))):
    cur = None
    for line in open(log):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill|Dynamic Stack):\s*(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%-6s %-6s %-8s %-9s %-8s %-6s %s" % ("VGPRs", "AGPRs", "scratch", "vgprSpill", "LDS", "occ", "kernel"))
for r, n in zip(rows, names):
    n = re.sub(r"ptau::", "", n)
    n = re.sub(r"\(.*", "", n)
    print("%-6s %-6s %-8s %-9s %-8s %-6s %s" % (r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                               r.get("VGPRs Spill", "?"), r.get("LDS Size [bytes/block]", "?"),
                                               r.get("Occupancy [waves/SIMD]", "?"), n))
PY
rm -rf "$T"
