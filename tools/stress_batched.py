"""Randomized stress of the batched subgroup check that rides on the ratio check's MSM: a genuine 2^lg BLS12-377
chunk whose response gets, trial after trial, elements outside the prime-order subgroup planted at random places — single
points, cofactor-only points, pairs P / -P and triples Q1, Q2, -(Q1 + Q2) (their images in the cofactor group cancel only if
they share a bucket in EVERY window) — must be rejected every time with PTAU_E_NOT_IN_SUBGROUP, under Auto and under Batched,
and the untouched response must keep passing.   python3 tools/stress_batched.py [lg] [trials]   (GPU box)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
from oracle.pyref import phase1 as P, serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group

pkg = load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 200
name = "bls12_377"
c = CURVES[name]
rng = random.Random(4242)
n = 1 << lg
prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, n, 0, lg, n)
p = P.Params(c, P.CHUNKED, 0, n, lg, n)
ch0 = pkg.phase1_cli.initialization(prm)
key = pkg.PrivateKey(rng.randrange(1, c.r), rng.randrange(1, c.r), rng.randrange(1, c.r), bytes(prm.public_key_size))
resp = bytearray(pkg.phase1_cli.computation(ch0, prm, key))
resp[:64] = pkg.calculate_hash(ch0)
offs = {nm: (g, o, cnt) for (nm, g, o, cnt) in p.layout(True)[0]}
groups = {1: Group(c, 1), 2: Group(c, 2)}

def curve_point(G):
    F = G.F
    while True:
        x = tuple(rng.randrange(c.q) for _ in range(F.deg))
        y = F.sqrt(F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b))
        if y is not None and not G.in_subgroup((x, y)):
            return (x, y)

pool = {g: [curve_point(G) for _ in range(6)] for g, G in groups.items()}
for g, G in groups.items():
    pool[g] += [G.mul(pool[g][0], c.r), G.mul(pool[g][1], c.r)]      # cofactor-only points
Mode = pkg.SubgroupCheckMode
t0 = time.time()
ok = pkg.phase1_cli.verification_and_decompress(ch0, bytes(resp), prm, ratio_check=True)
assert ok[:64] == pkg.calculate_hash(bytes(resp))
rejected = 0
kinds = {"single": 0, "pair": 0, "triple": 0}
for t in range(trials):
    vec = rng.choice(["tau_g1", "tau_g2", "alpha_g1", "beta_g1"])
    g, o, cnt = offs[vec]
    G = groups[g]
    kind = rng.choice(["single", "single", "pair", "triple"])
    if kind == "single":
        pts = [rng.choice(pool[g])]
    elif kind == "pair":
        q = rng.choice(pool[g][:6])
        pts = [q, G.neg(q)]
    else:
        q1, q2 = rng.sample(pool[g][:6], 2)
        pts = [q1, q2, G.neg(G.add(q1, q2))]
    pos = rng.sample(range(cnt), len(pts)) if rng.random() < 0.7 else [(s + i) % cnt for s in [rng.randrange(cnt)] for i in range(len(pts))]
    r2 = bytearray(resp)
    for q, at in zip(pts, pos):
        r2[o + at * G.comp_size:o + (at + 1) * G.comp_size] = serde.write_point(G, q, True)
    mode = Mode.Auto if t % 2 == 0 else Mode.Batched
    try:
        pkg.phase1_cli.verification_and_decompress(ch0, bytes(r2), prm, subgroup_check_mode=mode, ratio_check=True)
        print("ACCEPTED a corrupted response: trial", t, vec, kind, pos, flush=True)
        sys.exit(1)
    except pkg.PtauError as e:
        assert e.code == -5, (t, vec, kind, e)
        rejected += 1
        kinds[kind] += 1
ok = pkg.phase1_cli.verification_and_decompress(ch0, bytes(resp), prm, ratio_check=True)
assert ok[:64] == pkg.calculate_hash(bytes(resp))
print("batched subgroup check, %s 2^%d chunk: %d / %d corrupted responses rejected with PTAU_E_NOT_IN_SUBGROUP (%s), genuine response "
      "accepted before and after; %.1f s" % (name, lg, rejected, trials, ", ".join("%s %d" % kv for kv in kinds.items()), time.time() - t0))
