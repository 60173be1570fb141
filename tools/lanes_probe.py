import os, sys, time, threading, tempfile, shutil
sys.path.insert(0, os.getcwd())
from bench import load_pkg
pkg = load_pkg()
n = 1 << 20
prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, 20, n)
d = tempfile.mkdtemp(prefix="ptau_l_", dir="/dev/shm")
f = lambda s: os.path.join(d, s)
pkg.new_challenge(f("c0"), f("c0.hash"), prm)
pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, b"s".ljust(64, b"."), device=0)
Lb = pkg.load_library()
MODE = sys.argv[1] if len(sys.argv) > 1 else "verify"
VERIFIED = int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP
def lane(k):
    if MODE == "contribute":
        pkg.phase1_cli.contribute_seeded(f("c0"), f("L%d.h" % k), f("L%d.c1" % k), f("L%d.r1.hash" % k), pkg.CheckForCorrectness.No,
                                         VERIFIED, prm, b"t".ljust(64, b"."), device=0)
        return
    pkg.transform_pok_and_correctness(f("c0"), f("L%d.c.hash" % k), pkg.CheckForCorrectness.No, f("r0"), f("L%d.r.hash" % k),
                                      pkg.CheckForCorrectness.OnlyNonZero, f("L%d.c1" % k), f("L%d.c1.hash" % k), pkg.SubgroupCheckMode.Auto, True, prm, device=0)
lane(99); os.remove(f("L99.c1"))
for lanes in (1, 4, 8, 12):
    for rep in range(2):
        th = [threading.Thread(target=lane, args=(k,)) for k in range(lanes)]
        b0 = Lb.ptau_device_busy_ms(0); t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print(MODE + " register=%s lanes %d: %.2f chunks/s busy %.2f" % (os.environ.get("PTAU_HOST_REGISTER", "1"), lanes, lanes / dt, (Lb.ptau_device_busy_ms(0) - b0) / 1e3 / dt), flush=True)
        tm = pkg.phase1_cli.last_timing(0)
        print("   last call under load: kernel %.0f (decompress %.0f msm %.0f) h2d %.1f d2h %.1f ms" % (tm["kernel_ms"], tm["decompress_ms"], tm["msm_ms"], tm["h2d_ms"], tm["d2h_ms"]), flush=True)
        for k in range(lanes): os.remove(f("L%d.c1" % k))
shutil.rmtree(d, ignore_errors=True)
