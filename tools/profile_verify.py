#!/usr/bin/env python3
"""Whole-chunk verify (and contribute) of a BLS12-377 power-N chunk 0 for rocprofv3 --kernel-trace --stats runs:
   rocprofv3 --kernel-trace --stats -d out -o v --output-format csv -- python3 tools/profile_verify.py 18
Prints the library's per-phase HIP-event timings."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (helpers only)


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    curve = sys.argv[2] if len(sys.argv) > 2 else "bls12_377"
    pkg = bench.load_pkg()
    pkg.load_library()
    n = 1 << log2n
    prm = pkg.Phase1Parameters.new_chunk(curve, 0, 0, n, 0, log2n, n)
    assert curve == "bls12_377"
    r = bench.BLS_R
    k0 = [bench.det_scalar("bench/k0/%d" % i) % r for i in range(3)]
    ch0 = pkg.phase1_cli.initialization(prm)
    t0 = time.perf_counter()
    resp0 = bytearray(pkg.phase1_cli.computation(ch0, prm, pkg.PrivateKey(*k0, bytes(prm.public_key_size)), device=0))
    t1 = time.perf_counter()
    print("contribute %.3f s" % (t1 - t0), pkg.phase1_cli.last_timing(0))
    resp0[:64] = pkg.calculate_hash(ch0)
    resp0 = bytes(resp0)
    for ratio in (True, False):
        t0 = time.perf_counter()
        pkg.phase1_cli.verification_and_decompress(ch0, resp0, prm, ratio_check=ratio, device=0)
        t1 = time.perf_counter()
        print("verify ratio_check=%s %.3f s" % (ratio, t1 - t0), pkg.phase1_cli.last_timing(0))


if __name__ == "__main__":
    main()
