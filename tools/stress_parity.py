"""One-off randomized parity stress of the window paths against the C oracle on more points than the test suite uses:
subgroup points, curve points outside the subgroup and cofactor-only points mixed, Auto (exact / probed) and asserted-
membership modes.  python3 tools/stress_parity.py [n]   (GPU box; ~1-2 minutes for n = 20000)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.cuda.is_available()
from conftest import load_package
from oracle import cref
from oracle.pyref import serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group

pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(2026)
n_arg = n
# the MNT groups (round 3: carry-free lane-per-coefficient kernels, GLS on the twists) at sizes the CPU oracle finishes in seconds
for name, group, div in (("bls12_377", 1, 1), ("bls12_377", 2, 1)) if os.environ.get("STRESS_ONLY_BLS") else (("bls12_377", 1, 1), ("bls12_377", 2, 1), ("bw6", 1, 1), ("bw6", 2, 1), ("mnt4_753", 1, 8), ("mnt6_753", 1, 8),
                         ("mnt4_753", 2, 16), ("mnt6_753", 2, 32)):
    n = max(64, n_arg // div)
    c = CURVES[name]; G = Group(c, group); F = G.F
    def curve_point():
        while True:
            x = tuple(rng.randrange(c.q) for _ in range(F.deg))
            y = F.sqrt(F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b))
            if y is not None:
                return (x, y)
    base = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(24)]
    off = [curve_point() for _ in range(12)]
    cof = [G.mul(P, c.r) for P in off[:4]]
    # subgroup-only vector (all modes) and mixed vector (exact modes), drawn from pools of serialized points
    ser = lambda P: serde.write_point(G, P, False)
    pool_good = [ser(P) for P in base]
    pool_mixed = pool_good + [ser(P) for P in off + cof + [None]]
    good = b"".join(pool_good[rng.randrange(len(pool_good))] for _ in range(n))
    mixed = b"".join(pool_mixed[rng.randrange(len(pool_mixed))] for _ in range(n))
    for tag, data, modes in (("subgroup", good, (pkg.BatchExpMode.Auto, int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP)),
                             ("mixed", mixed, (pkg.BatchExpMode.Auto, pkg.BatchExpMode.Direct))):
        tau, coeff, start = rng.randrange(c.r), rng.randrange(1, c.r), rng.randrange(1 << 30)
        t = time.time()
        exp = cref.apply_powers(name, group, data, n, start, tau, coeff, False, True, nthreads=16)
        t_cpu = time.time() - t
        for mode in modes:
            got = pkg.phase1_cli.apply_powers(name, group, data, n, start, tau, coeff, False, True, batch_exp_mode=mode)
            bad = 0 if got == exp else sum(1 for i in range(n) if got[i * G.comp_size:(i + 1) * G.comp_size] != exp[i * G.comp_size:(i + 1) * G.comp_size])
            print(name, group, tag, "mode", int(mode), "mismatches", bad, "of", n, "(oracle %.1f s)" % t_cpu, flush=True)
            if bad:   # say WHICH side is off: both are recomputed once (a CPU-side or a GPU-side fluke shows as a result that changes)
                idx = [i for i in range(n) if got[i * G.comp_size:(i + 1) * G.comp_size] != exp[i * G.comp_size:(i + 1) * G.comp_size]]
                print("   first mismatches", idx[:10], "last", idx[-5:], "flags of first bad input", data[(idx[0] + 1) * G.uncomp_size - 1], flush=True)
                exp2 = cref.apply_powers(name, group, data, n, start, tau, coeff, False, True, nthreads=16)
                got2 = pkg.phase1_cli.apply_powers(name, group, data, n, start, tau, coeff, False, True, batch_exp_mode=mode)
                exp1 = cref.apply_powers(name, group, data, n, start, tau, coeff, False, True, nthreads=1) if n <= 65536 else None
                print("   recomputed: oracle same as before:", exp2 == exp, "| GPU same as before:", got2 == got, "| GPU(2) == oracle(2):", got2 == exp2,
                      "| oracle 1 thread == 16 threads:", None if exp1 is None else exp1 == exp2, flush=True)
            assert bad == 0
print("stress parity ok")
