#!/usr/bin/env python3
"""Per-kernel timeline of ONE chunk verification (BLS12-377 2^20, transform_pok_and_correctness with ratio check).

  run  : rocprofv3 --kernel-trace -d gpurun_out/vt -o vt --output-format csv -- python3 tools/verify_timeline.py run
  read : python3 tools/verify_timeline.py read gpurun_out/vt > profiles/rNN_verify_timeline.txt

`run` does setup + one warm verification, sleeps 1.5 s, then the verification that `read` isolates (the kernels after the last
idle gap of more than one second).  Measurement tooling only."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(lg):
    import shutil
    import tempfile
    from bench import load_pkg
    pkg = load_pkg()
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, lg, n)
    d = tempfile.mkdtemp(prefix="ptau_vt_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    f = lambda s: os.path.join(d, s)
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                         prm, b"bench verify seed".ljust(64, b"."), device=0)
        for k in range(2):
            if os.path.exists(f("c1")):
                os.remove(f("c1"))
            if k == 1:
                time.sleep(1.5)
            pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.hash"),
                                              pkg.CheckForCorrectness.OnlyNonZero, f("c1"), f("c1.hash"), pkg.SubgroupCheckMode.Auto,
                                              True, prm, device=0)
            print(k, pkg.phase1_cli.last_timing(0), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_lanes(lg, lanes):
    """One warm verification, a pause, then `lanes` concurrent verifications of the same chunk (in-process lanes on one GPU):
    `read` then shows how the device is shared."""
    import shutil
    import tempfile
    import threading
    from bench import load_pkg
    pkg = load_pkg()
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, lg, n)
    d = tempfile.mkdtemp(prefix="ptau_vl_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    f = lambda s: os.path.join(d, s)
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                         prm, b"bench verify seed".ljust(64, b"."), device=0)

        def lane(k):
            pkg.transform_pok_and_correctness(f("c0"), f("L%d.c.hash" % k), pkg.CheckForCorrectness.No, f("r0"), f("L%d.r.hash" % k),
                                              pkg.CheckForCorrectness.OnlyNonZero, f("L%d.c1" % k), f("L%d.c1.hash" % k), pkg.SubgroupCheckMode.Auto,
                                              True, prm, device=0)
        lane(99)
        time.sleep(1.5)
        th = [threading.Thread(target=lane, args=(k,)) for k in range(lanes)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        print("lanes", lanes, "chunks/s", lanes / (time.perf_counter() - t0), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_contribute(lg, asserted):
    """Two contributions of chunk 0 (the second one is what `read` isolates): default flags, or membership asserted."""
    import shutil
    import tempfile
    from bench import load_pkg
    pkg = load_pkg()
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, lg, n)
    d = tempfile.mkdtemp(prefix="ptau_ct_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    f = lambda s: os.path.join(d, s)
    mode = int(pkg.BatchExpMode.Auto) | (pkg.phase1_cli.INPUT_IN_SUBGROUP if asserted else 0)
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        for k in range(2):
            if k == 1:
                time.sleep(1.5)
            pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h%d" % k), f("r%d" % k), f("r%d.hash" % k), pkg.CheckForCorrectness.No, mode,
                                             prm, b"bench verify seed".ljust(64, b"."), device=0)
            print(k, pkg.phase1_cli.last_timing(0), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def read(path):
    rows = []
    for fn in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[max(0, i - 64):i]) > 1_000_000_000:
            cut = i
    rows = rows[cut:]
    t0 = rows[0][0]
    span = (max(r[1] for r in rows) - t0) / 1e6

    def union(rs):
        tot, cur_s, cur_e = 0, None, None
        for s, e, *_ in sorted(rs):
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        return (tot + (cur_e - cur_s if cur_e is not None else 0)) / 1e6
    print("one verification: %d kernel launches, first start -> last end %.1f ms, device busy (union of all kernels) %.1f ms" % (len(rows), span, union(rows)))
    streams = sorted(set(r[3] for r in rows))
    for s in streams:
        rs = [r for r in rows if r[3] == s]
        print("  stream/queue %s: %d launches, busy %.1f ms" % (s, len(rs), union(rs)))
    agg = {}
    for s, e, name, st in rows:
        k = name.split("(")[0].replace("void ptau::", "").replace("ptau::", "")[:110]
        a = agg.setdefault(k, [0, 0.0, t0 and (s - t0) / 1e6])
        a[0] += 1
        a[1] += (e - s) / 1e6
    print("%-112s %6s %10s %10s" % ("kernel", "calls", "total ms", "first at"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-112s %6d %10.2f %10.1f" % (k, a[0], a[1], a[2]))
    if "--launches" in sys.argv:
        for s, e, name, st in rows:
            print("%9.2f %9.2f %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, st, name.split("(")[0][:100]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif sys.argv[1] == "lanes":
        run_lanes(20, int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    elif sys.argv[1] in ("contribute", "contribute_asserted"):
        run_contribute(int(sys.argv[2]) if len(sys.argv) > 2 else 20, sys.argv[1] == "contribute_asserted")
    else:
        read(sys.argv[2])
