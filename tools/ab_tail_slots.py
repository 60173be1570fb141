"""Whole-chunk verification (file level, ratio check + PoK) of the three bench shapes, three runs each: wall and device time.
Run twice for the A/B of DESIGN §4.4a(5d):   python3 tools/ab_tail_slots.py ;  PTAU_TAIL_SLOTS=1 python3 tools/ab_tail_slots.py"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_pkg, initial_challenge_file
pkg = load_pkg()
for curve, power, lg in (("bls12_377", 20, 20), ("bw6", 20, 18), ("mnt4_753", 18, 16)):
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk(curve, 0, 0, n, 0, power, n)
    d = tempfile.mkdtemp(prefix="ptau_ab_", dir="/dev/shm")
    f = lambda s: os.path.join(d, s)
    try:
        initial_challenge_file(pkg, curve, prm, f("c0"), f("c0.hash"))
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm,
                                         b"ab".ljust(64, b"."), device=0)
        res = []
        for k in range(4):
            if os.path.exists(f("c1")):
                os.remove(f("c1"))
            t0 = time.perf_counter()
            pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.hash"),
                                              pkg.CheckForCorrectness.OnlyNonZero, f("c1"), f("c1.hash"), pkg.SubgroupCheckMode.Auto, True, prm, device=0)
            dt = time.perf_counter() - t0
            tm = pkg.phase1_cli.last_timing(0)
            if k:
                res.append((dt, tm["kernel_ms"], tm["decompress_ms"], tm["msm_ms"], tm["msm_tail_ms"]))
        best = min(res)
        print("slots=%s %-10s 2^%d: wall %.3f s  device %.1f ms (decompress %.1f, msm %.1f; tails' latency %.1f)  [runs: %s]" % (
            os.environ.get("PTAU_TAIL_SLOTS", "2"), curve, lg, best[0], best[1], best[2], best[3], best[4],
            ", ".join("%.1f" % r[1] for r in res)), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
