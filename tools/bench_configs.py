#!/usr/bin/env python3
"""One chunk of BASELINE configs 3 and 4 on ONE GPU through the file-level entry points (phase1_cli::contribute /
transform_pok_and_correctness): BW6-761 power 20 chunk 2^18, MNT4-753 and MNT6-753 power 18 chunk 2^16 — chunk 0 (all five
vectors) and a tauG1-only chunk.  Prints wall and device times; the 8-GPU runs of those configs farm such chunks one per rank.
Usage (GPU box):  python tools/bench_configs.py [curve ...]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (helpers only)

CONFIGS = {"bw6": (20, 18), "mnt4_753": (18, 16), "mnt6_753": (18, 16), "bls12_377": (20, 20)}


def main():
    pkg = bench.load_pkg()
    pkg.load_library()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    for curve in (sys.argv[1:] or ["bw6", "mnt4_753", "mnt6_753"]):
        power, lcs = CONFIGS[curve]
        cs = 1 << lcs
        n_chunks = -(-((2 << power) - 1) // cs)
        keys = [(1 << 250) // 3 + 7 * i for i in range(3)]      # below every group order
        d = tempfile.mkdtemp(prefix="ptau_cfg_", dir=base)
        f = lambda s: os.path.join(d, s)
        try:
            tau_g2_check = None
            for idx in (0, n_chunks - 1):
                prm = pkg.Phase1Parameters.new_chunk(curve, 0, idx, cs, 0, power, cs)
                tag = "c%d" % idx
                pkg.new_challenge(f(tag), f(tag + ".h"), prm)
                key = pkg.PrivateKey(*keys, bytes(prm.public_key_size))
                t0 = time.perf_counter()
                pkg.contribute(f(tag), f(tag + ".h2"), f(tag + ".r"), f(tag + ".r.h"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, key)
                t1 = time.perf_counter()
                tc = pkg.phase1_cli.last_timing(0)
                kw = {"tau_g2_check": tau_g2_check} if prm.other_chunk_size == 0 else {}
                t2 = time.perf_counter()
                pkg.transform_pok_and_correctness(f(tag), f(tag + ".vh"), pkg.CheckForCorrectness.No, f(tag + ".r"), f(tag + ".r.vh"),
                                                  pkg.CheckForCorrectness.OnlyNonZero, f(tag + ".n"), f(tag + ".n.h"),
                                                  pkg.SubgroupCheckMode.Auto, True, prm, **kw)
                t3 = time.perf_counter()
                tv = pkg.phase1_cli.last_timing(0)
                if idx == 0:
                    L = pkg.load_library()
                    u1, u2 = L.ptau_point_size(prm.curve, 1, 0), L.ptau_point_size(prm.curve, 2, 0)
                    nc = open(f(tag + ".n"), "rb").read()
                    off = 64 + prm.g1_chunk_size * u1
                    tau_g2_check = nc[off:off + 2 * u2]
                print("%-9s power %d chunk 2^%d idx %d/%d (g1 %d, other %d; challenge %d B): contribute %.3f s (device %.0f ms)   "
                      "verify+ratio %.3f s (device %.0f ms: sqrt %.0f, subgroup %.0f, msm %.0f)" %
                      (curve, power, lcs, idx, n_chunks, prm.g1_chunk_size, prm.other_chunk_size, prm.accumulator_size, t1 - t0,
                       tc["kernel_ms"], t3 - t2, tv["kernel_ms"], tv["decompress_ms"], tv["subgroup_ms"], tv["msm_ms"]), flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
