"""Proactive stress of the asynchronous machinery of the device sections (queued copies with host callbacks, reused output buffers
and MSM-tail slots, input prefetch, the speculative endomorphism launch): ROUNDS rounds of LANES in-process lanes on one GPU,
verify and contribute alternating, at two chunk sizes; every file must be byte-identical to the single-lane reference.
   python3 tools/stress_lanes.py [rounds] [lanes]"""
import hashlib, os, shutil, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_pkg
pkg = load_pkg()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
digest = lambda p: hashlib.blake2b(open(p, "rb").read()).digest()
t00 = time.time()
for lg in (14, 18):
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, lg, n)
    d = tempfile.mkdtemp(prefix="ptau_sl_", dir="/dev/shm")
    f = lambda s: os.path.join(d, s)
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, b"a".ljust(64, b"."), device=0)

        def verify(tag):
            pkg.transform_pok_and_correctness(f("c0"), f(tag + ".c.hash"), pkg.CheckForCorrectness.No, f("r0"), f(tag + ".r.hash"),
                                              pkg.CheckForCorrectness.OnlyNonZero, f(tag + ".c1"), f(tag + ".c1.hash"), pkg.SubgroupCheckMode.Auto, True, prm, device=0)

        def contribute(tag):
            pkg.phase1_cli.contribute_seeded(f("ref.c1"), f(tag + ".h"), f(tag + ".r1"), f(tag + ".r1.hash"), pkg.CheckForCorrectness.No,
                                             pkg.BatchExpMode.Auto, prm, b"b".ljust(64, b"."), device=0)
        verify("ref"); contribute("refc")
        want_v = [digest(f("ref" + s)) for s in (".c1", ".c1.hash", ".c.hash", ".r.hash")]
        want_c = [digest(f("refc" + s)) for s in (".r1", ".r1.hash", ".h")]
        bad = []
        for rnd in range(rounds):
            def lane(k):
                tag = "L%d" % k
                try:
                    if (k + rnd) % 2 == 0:
                        verify(tag)
                        if [digest(f(tag + s)) for s in (".c1", ".c1.hash", ".c.hash", ".r.hash")] != want_v: bad.append((lg, rnd, k, "verify"))
                        for s in (".c1", ".c1.hash", ".c.hash", ".r.hash"): os.remove(f(tag + s))
                    else:
                        contribute(tag)
                        if [digest(f(tag + s)) for s in (".r1", ".r1.hash", ".h")] != want_c: bad.append((lg, rnd, k, "contribute"))
                        for s in (".r1", ".r1.hash", ".h"): os.remove(f(tag + s))
                except Exception as e:
                    bad.append((lg, rnd, k, repr(e)))
            th = [threading.Thread(target=lane, args=(k,)) for k in range(lanes)]
            [t.start() for t in th]; [t.join() for t in th]
        print("2^%d chunk: %d rounds x %d lanes, files differing from the single-lane reference: %d %s" % (lg, rounds, lanes, len(bad), bad[:5]), flush=True)
        if bad: sys.exit(1)
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("lanes stress ok, %.0f s" % (time.time() - t00))
