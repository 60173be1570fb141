#!/usr/bin/env python3
"""Device time of ONE whole-chunk contribution + verification through the file-level entry points, per curve:
    python3 tools/chunk_probe.py mnt4_753 16 [mnt6_753 16 ...]
prints the ptau_timing record of each call (kernel_ms, decompress_ms, subgroup_ms, msm_ms, msm_tail_ms) — the quick A/B for
changes to the verify-side kernels (env switches: PTAU_LAZY_A=0 keeps the round-2 kernels of the a != 0 groups)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.is_available()
    pkg = bench.load_pkg()
    args = sys.argv[1:]
    for k in range(0, len(args), 2):
        curve, lg = args[k], int(args[k + 1])
        n = 1 << lg
        prm = pkg.Phase1Parameters.new_chunk(curve, 0, 0, n, 0, lg, n)
        d = tempfile.mkdtemp(prefix="ptau_probe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        f = lambda s: os.path.join(d, s)
        bench.initial_challenge_file(pkg, curve, prm, f("c0"), f("c0.hash"))
        C = pkg.CheckForCorrectness
        for rep in range(2):
            for p in ("r0", "c1"):
                if os.path.exists(f(p)):
                    os.remove(f(p))
            t0 = time.perf_counter()
            pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), C.No, pkg.BatchExpMode.Auto, prm, b"probe".ljust(64, b"."))
            t1 = time.perf_counter()
            tc = pkg.phase1_cli.last_timing(0)
            pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), C.No, f("r0"), f("v.r.hash"), C.OnlyNonZero, f("c1"), f("c1.hash"),
                                              pkg.SubgroupCheckMode.Auto, True, prm)
            t2 = time.perf_counter()
            tv = pkg.phase1_cli.last_timing(0)
        r = lambda t: {k: round(v, 2) for k, v in t.items() if isinstance(v, float) and v}
        print(curve, "2^%d" % lg, "contribute %.3f s" % (t1 - t0), r(tc), flush=True)
        print(curve, "2^%d" % lg, "verify     %.3f s" % (t2 - t1), r(tv), "ratio_checks", tv["ratio_checks"], flush=True)
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
