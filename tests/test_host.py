"""CPU tests of the host logic and of the C-ABI library surface (no compute calls: no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import CURVE_NAMES, ROOT
from oracle.pyref import phase1 as P
from oracle.pyref.curves import CURVES


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load_library()
    header = open(os.path.join(ROOT, "include", "ptau.h")).read()
    declared = set(re.findall(r"\b(ptau_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pkg._lib.DECLARED_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_params_sizes_match_oracle(pkg):
    for name in CURVE_NAMES:
        for (mode, idx, cs, pw) in ((0, 0, 4, 2), (0, 1, 4, 2), (0, 0, 1 << 10, 12), (0, 5, 1 << 10, 12), (0, 7, 1 << 10, 12),
                                    (0, 0, 1 << 20, 20), (0, 1, 1 << 20, 20), (0, 3, 1 << 18, 20), (0, 4, 1 << 18, 20), (1, 0, 1 << 20, 20)):
            q = P.Params(CURVES[name], mode, idx, cs, pw, cs)
            h = pkg.Phase1Parameters.new_chunk(name, mode, idx, cs, 0, pw, cs)
            assert (h.g1_chunk_size, h.other_chunk_size, h.accumulator_size, h.contribution_size, h.public_key_size,
                    h.hash_size) == (q.g1_chunk_size, q.other_chunk_size, q.accumulator_size, q.contribution_size,
                                     q.public_key_size, 64)
            assert h.powers_length == 1 << pw and h.powers_g1_length == (2 << pw) - 1
    # number of chunks as new_setup computes it (new_setup.rs:274-277)
    h = pkg.Phase1Parameters.new_chunk("bw6", 0, 0, 1 << 18, 0, 20, 1 << 18)
    assert -(-h.powers_g1_length // (1 << 18)) == 8


def test_host_blake2b(pkg, cref):
    import hashlib
    for n in (0, 1, 127, 128, 129, 256, 100003):
        d = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert pkg.calculate_hash(d) == hashlib.blake2b(d).digest() == cref.blake2b512(d)


def test_blake2b_portable_and_avx2_agree(pkg):
    """The compression functions of the host BLAKE2b (portable, AVX2, BMI2-compiled scalar, AVX-512) against hashlib, on
    block-boundary lengths (a form the CPU lacks falls back to the chosen one)."""
    import hashlib
    L = pkg.load_library()
    chosen = L.ptau_blake2b_implementation(-1)
    data = bytes((i * 131 + 7) & 0xFF for i in range(70001))
    try:
        for impl in (0, 1, 2, 3):
            L.ptau_blake2b_implementation(impl)
            for n in (0, 1, 127, 128, 129, 255, 256, 257, 4096, 70001):
                assert pkg.calculate_hash(data[:n]) == hashlib.blake2b(data[:n]).digest(), (impl, n)
    finally:
        L.ptau_blake2b_implementation(chosen)


def test_mode_parsers_and_errors(pkg):
    assert pkg.batch_exp_mode_from_str("auto") == pkg.BatchExpMode.Auto
    assert pkg.subgroup_check_mode_from_str("batched") == pkg.SubgroupCheckMode.Batched
    assert pkg.proving_system_from_str("groth16") == pkg.ProvingSystem.Groth16
    with pytest.raises(ValueError):
        pkg.curve_from_str("bn254")
    with pytest.raises(pkg.PtauError):   # Marlin geometry is not implemented: loud, not silent
        pkg.Phase1Parameters.new_chunk("bw6", 0, 0, 4, 1, 2, 4).accumulator_size


def test_no_cpu_fallback_without_gpu(pkg):
    """On a box without a GPU every compute entry point must fail with PTAU_E_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.apply_powers("bls12_377", 1, bytes(96), 1, 0, 5)
    assert e.value.code == -8


def test_product_never_links_the_oracle(pkg):
    out = subprocess.run(["ldd", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "ptau_oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "snark-setup-operator_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle/" not in txt.replace("test oracle", "") or f == "gen_constants.py", f
                assert "import oracle" not in txt and "from oracle" not in txt, f


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_kernel_bodies_on_host_match_golden(name, golden):
    """tests/native/host_emul runs the kernels' per-thread bodies (same templates) on the CPU."""
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built (python -c 'import __graft_entry__ as g; g.build()')")
    doc = golden[name]
    c = CURVES[name]
    cid = CURVE_NAMES.index(name)
    ch = doc["chunks"][0]
    p = P.Params(c, P.CHUNKED, 0, doc["chunk_size"], doc["power"], doc["batch_size"])
    challenge, response = bytes.fromhex(ch["challenge"]), bytes.fromhex(ch["response"])
    lay_in, lay_out = p.layout(False)[0], p.layout(True)[0]
    coeffs = {"tau_g1": "-", "tau_g2": "-", "alpha_g1": ch["alpha"], "beta_g1": ch["beta"]}
    for (nm, grp, off, cnt), (_, _, ooff, _) in zip(lay_in, lay_out):
        usz, csz = c.uncomp_size(grp), c.comp_size(grp)
        if nm == "beta_g2":
            args = ["apply", cid, grp, 1, 0, 0, 1, "1", ch["beta"]]
        else:
            args = ["apply", cid, grp, cnt, p.start, 0, 1, ch["tau"], coeffs[nm]]
        r = subprocess.run([exe] + [str(a) for a in args], input=challenge[off:off + cnt * usz], capture_output=True)
        assert r.returncode == 0
        assert r.stdout == response[ooff:ooff + cnt * csz], nm
        d = subprocess.run([exe, "decompress", str(cid), str(grp), str(cnt), "0", "1", "0", "0", "-"],
                           input=response[ooff:ooff + cnt * csz], capture_output=True)
        assert d.returncode == 0 and len(d.stdout) == cnt * usz


@pytest.mark.parametrize("name,group,n,c", [("bls12_377", 1, 300, 0), ("bls12_377", 1, 300, 2), ("bls12_377", 1, 37, 5), ("bls12_377", 2, 60, 4),
                                            ("bw6", 1, 40, 3), ("mnt4_753", 2, 12, 6), ("mnt6_753", 1, 20, 0)])
def test_bucket_msm_bodies_on_host(name, group, n, c, cref):
    """The bucket-MSM kernels' per-thread bodies (digits, counting sort, bucket sums, running-sum reduction,
    Horner), run serially on the CPU by host_emul, against the oracle's sum of scalar multiplications —
    plain sum and the shifted sum of power_pairs, with repeated points, negations, infinities and edge scalars."""
    import random
    import struct
    from oracle.pyref import serde
    from oracle.pyref.ec import Group
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    cv = CURVES[name]
    G = Group(cv, group)
    rng = random.Random(1000 + n)
    base = [G.mul(G.gen, rng.randrange(1, cv.r)) for _ in range(4)]
    base.append(G.neg(base[0]))
    pts = [base[i % 5] if i % 7 else None for i in range(n + 1)]
    data = serde.write_batch(G, pts, False)
    nl32 = (cv.r.bit_length() + 31) // 32
    scalars = [rng.randrange(cv.r) for _ in range(n)]
    scalars[0], scalars[1], scalars[2], scalars[3] = 0, 1, cv.r - 1, (1 << (cv.r.bit_length() - 1)) - 1
    sbytes = b"".join(struct.pack("<%dI" % nl32, *[(s >> (32 * j)) & 0xFFFFFFFF for j in range(nl32)]) for s in scalars)
    cid = CURVE_NAMES.index(name)
    usz = G.uncomp_size
    nl = cref.fr_limbs(name)
    exp_a = cref.msm(name, group, data, n, scalars, nl)
    exp_b = cref.msm(name, group, data[usz:], n, scalars, nl)
    r = subprocess.run([exe, "msm", str(cid), str(group), str(n), str(c), "0", "0", "0", "-"], input=data[:n * usz] + sbytes,
                       capture_output=True)
    assert r.returncode == 0 and r.stdout == exp_a
    r = subprocess.run([exe, "msm", str(cid), str(group), str(n), str(c), "0", "1", "0", "-"], input=data + sbytes,
                       capture_output=True)
    assert r.returncode == 0 and r.stdout == exp_a + exp_b


@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 1), ("bw6", 2), ("mnt4_753", 2), ("mnt6_753", 2)])
def test_endomorphism_subgroup_test_on_host(name, group):
    """Membership tests through an endomorphism — BLS12-377: phi on G1, psi on G2 (eprint 2021/1130); BW6-761:
    [a]P + [b]phi(P) == O for the norm-r element a + b phi of Z[phi]; MNT4/MNT6-753 G2: psi(P) == [t - 1]P (psi - [t - 1] has
    degree r) — the kernel bodies, run on the CPU, accept exactly
    the points the oracle's [r]P == O accepts: subgroup points, cofactor-only points, mixed points, small-order
    components, the point at infinity, for compressed and uncompressed encodings."""
    import random
    from oracle.pyref import serde
    from oracle.pyref.ec import Group
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    cv = CURVES[name]
    cid = str(CURVE_NAMES.index(name))
    G = Group(cv, group)
    F = G.F
    rng = random.Random(group)

    def rand_point():
        while True:
            x = tuple(rng.randrange(cv.q) for _ in range(F.deg))
            y = F.sqrt(F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b))
            if y is not None:
                return (x, y)
    pts = [G.gen, G.mul(G.gen, rng.randrange(1, cv.r)), None]
    for _ in range(3 if group == 1 else 2):
        P = rand_point()
        pts += [P, G.mul(P, cv.r), G.add(G.mul(P, cv.r), G.gen)]
    if group == 1 and name == "bls12_377":   # components of small order (the G1 cofactor is 2^92 * 3 * ...): r * (h / f) * P
        h = 0x170b5d44300000000000000000000000
        P = rand_point()
        pts += [G.mul(P, cv.r * (h // f)) for f in (2, 3, 4)]
    expect = bytes(0 if G.in_subgroup(P) else 8 for P in pts)
    assert 8 in expect and 0 in expect
    for comp in (0, 1):
        data = serde.write_batch(G, pts, bool(comp))
        r = subprocess.run([exe, "subgroup", cid, str(group), str(len(pts)), "0", str(comp), "0", "0", "-"], input=data,
                           capture_output=True)
        assert r.returncode == 0 and r.stdout == expect
    if group == 1 and name == "bls12_377":
        # the same verdicts from JACOBIAN representatives (x z^2, y z^3, z) with Z ignored: what the bucket sums of the
        # membership-fused MSM go through (kernels_impl.hpp endo_membership_lazy: the isomorphic-curve argument)
        many = pts + [G.mul(G.gen, rng.randrange(1, cv.r)) for _ in range(20)] + [rand_point() for _ in range(20)]
        exp2 = bytes(0 if G.in_subgroup(P) else 8 for P in many)
        r = subprocess.run([exe, "subgroupjac", cid, "1", str(len(many)), "0", "0", "0", "0", "-"], input=serde.write_batch(G, many, False),
                           capture_output=True)
        assert r.returncode == 0 and r.stdout == exp2
    # uncompressed point off the curve
    x, y = pts[0]
    off = serde.write_point(G, (x, F.add(y, F.one())), False)
    r = subprocess.run([exe, "subgroup", cid, str(group), "1", "0", "0", "0", "0", "-"], input=off, capture_output=True)
    assert r.stdout == bytes([2])


@pytest.mark.parametrize("name,group,n", [("bls12_377", 1, 96), ("bls12_377", 2, 48), ("bw6", 1, 40), ("bw6", 2, 40)])
def test_endomorphism_scalar_mul_bodies_on_host(name, group, n, cref):
    """The GLV (BLS12-377 G1, 2 x 128-bit), GLS (BLS12-377 G2, 4 x 64-bit, psi) and lattice-GLV (BW6-761 G1/G2,
    2 x 192-bit signed) window paths, run on the CPU by host_emul, against the oracle's plain double-and-add on random
    subgroup points, for scalars c * tau^(start+i) at several offsets — and the same with the endomorphism paths off."""
    import random
    from oracle.pyref import serde
    from oracle.pyref.ec import Group
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    cv = CURVES[name]
    cid = CURVE_NAMES.index(name)
    G = Group(cv, group)
    rng = random.Random(40 + group)
    pts = [G.mul(G.gen, rng.randrange(1, cv.r)) for _ in range(6)]
    data = serde.write_batch(G, [pts[i % 6] for i in range(n)], False)
    for start, tau, coeff in ((0, rng.randrange(cv.r), None), (12345, cv.r - 1, rng.randrange(cv.r)), (1 << 20, 2, None)):
        exp = cref.apply_powers(name, group, data, n, start, tau, coeff, False, True, nthreads=8)
        args = ["apply", str(cid), str(group), str(n), str(start), "0", "1", "%x" % tau, "-" if coeff is None else "%x" % coeff]
        r = subprocess.run([exe] + args, input=data, capture_output=True)
        assert r.returncode == 0 and r.stdout == exp
        r2 = subprocess.run([exe] + args, input=data, capture_output=True, env=dict(os.environ, PTAU_GLV="0"))
        assert r2.returncode == 0 and r2.stdout == exp


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_base_field_arithmetic_on_host(name):
    """The base-field class each curve's kernels are instantiated with (29-bit limbs for BLS12-377 / MNT, 32-bit for
    BW6-761), run on the CPU, against Python integers: add, sub, mul, square, neg, inverse, square root, sign class and
    the byte (de)serialisation — on edge values (0, 1, p-1, p-2, (p±1)/2, limb-boundary patterns) and random ones."""
    import random
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    cv = CURVES[name]
    p, B = cv.q, cv.fq_bytes
    rng = random.Random(5)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 29) - 1, 1 << 29, (1 << 32) - 1, 1 << 32,
            (1 << 58) - 1, (1 << 348) + 1, (1 << (p.bit_length() - 1)) - 1, 1 << (p.bit_length() - 1),
            int("1fffffff" * 12, 16) % p, int("ffffffff" * 11, 16) % p]
    vals = [(a, b) for a in edge for b in (0, 1, p - 1, edge[(edge.index(a) * 7 + 3) % len(edge)])]
    vals += [(rng.randrange(p), rng.randrange(p)) for _ in range(100)]
    vals += [(a * a % p, rng.randrange(p)) for a, _ in vals[-40:]]          # guaranteed residues
    data = b"".join(a.to_bytes(B, "little") + b.to_bytes(B, "little") for a, b in vals)
    # the emulator's generic reader consumes n * uncompressed-size bytes first: G1 uncompressed = 2 * B = one pair
    r = subprocess.run([exe, "field", str(CURVE_NAMES.index(name)), "1", str(len(vals)), "0", "0", "0", "0", "-"], input=data,
                       capture_output=True)
    assert r.returncode == 0, r.returncode
    rec = 8 * B + 2
    assert len(r.stdout) == rec * len(vals)
    half = (p - 1) // 2
    for i, (a, b) in enumerate(vals):
        o = r.stdout[i * rec:(i + 1) * rec]
        g = lambda k: int.from_bytes(o[k * B:(k + 1) * B], "little")
        assert g(0) == (a + b) % p and g(1) == (a - b) % p and g(2) == a * b % p and g(3) == a * a % p and g(4) == (-a) % p
        assert g(5) == (pow(a, -1, p) if a else 0)
        is_res = a == 0 or pow(a, half, p) == 1
        assert o[6 * B] == (1 if is_res else 0)
        root = int.from_bytes(o[6 * B + 1:7 * B + 1], "little")
        assert (root * root % p == a) if is_res else root == 0
        assert o[7 * B + 1] == (1 if a == 0 else (2 if a > half else 0))


def test_rust_shim_declarations_match_the_header():
    """rust-shim/src/lib.rs cannot be compiled here (no Rust toolchain): at least every `fn ptau_*` it declares must be a
    function of include/ptau.h with the same number of parameters, and the repr(C) structs must list the header's fields."""
    import re
    hdr = open(os.path.join(ROOT, "include", "ptau.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c_funcs = {}
    for m in re.finditer(r"\b(?:int|void|double|const char \*)\s*(ptau_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        c_funcs[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    rs = open(os.path.join(ROOT, "rust-shim", "src", "lib.rs")).read()
    ext = rs[rs.index('extern "C" {'):]
    ext = ext[:ext.index("\n}\n")]
    rust_funcs = {}
    for m in re.finditer(r"pub fn (ptau_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", ext, flags=re.S):
        args = m.group(2).strip()
        rust_funcs[m.group(1)] = 0 if not args else args.count(":")
    assert len(rust_funcs) >= 19
    for name, n in rust_funcs.items():
        assert name in c_funcs, "%s is not declared in include/ptau.h" % name
        assert c_funcs[name] == n, "%s: %d parameters in Rust, %d in C" % (name, n, c_funcs[name])
    for field in ("curve", "proving_system", "contribution_mode", "power", "chunk_index", "chunk_size", "batch_size"):
        assert re.search(r"pub %s: [iu](32|64)" % field, rs), field
    # and every symbol the header declares is exported by the library and listed in the ctypes mirror
    from conftest import load_package
    pkg = load_package()
    assert set(c_funcs) == set(pkg._lib.DECLARED_SYMBOLS)
    # the documented operator call sites do not assert subgroup membership for data the caller did not check
    # (contribute.rs:811-823 runs DEFAULT_CONTRIBUTE_CHECK_INPUT_CORRECTNESS = No on a downloaded challenge)
    for doc in ("rust-shim/operator.patch.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        calls = re.findall(r"phase1_cli::contribute\((.*?)\)[`;]", text, flags=re.S)
        assert calls, doc
        for call in calls:
            assert not re.search(r"\btrue\b", call), "%s passes input_in_subgroup = true at a documented call site" % doc
            assert re.search(r"input_in_subgroup\s*\*/\s*false", call), doc


def test_window_paths_on_small_order_and_cofactor_points_on_host(cref):
    """K1's window loop of BLS12-377 G1 runs in the carry-free twin of the base field (fpz.hpp): bounds are compile-time, the
    exceptional cases (a table multiple that is O, an addition that meets its own operand or its negative, a doubling of a
    2-torsion point) go through residue tests on unreduced values.  Points of order 2, 3, 4, 6, 7, 8, 13 and 16, cofactor-only
    and mixed points, the identity, against the oracle's double-and-add, for plain and edge scalars (exact path)."""
    import random
    from oracle.pyref import serde
    from oracle.pyref.ec import Group
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    name = "bls12_377"
    cv = CURVES[name]
    G = Group(cv, 1)
    rng = random.Random(77)
    x = 0x8508c00000000001
    h = (x - 1) ** 2 // 3
    assert (h * cv.r - (cv.q + 1)) ** 2 < 4 * cv.q      # Hasse: h r is the curve order
    F = G.F
    def random_curve_point():
        while True:
            xx = (rng.randrange(cv.q),)
            y = F.sqrt(F.add(F.mul(F.sqr(xx), xx), G.b))
            if y is not None:
                return (xx, y)
    def point_of_order(ell, k, m):      # ell^k || h; a point of exact order ell^m
        while True:
            T = G.mul(G.mul(random_curve_point(), cv.r), h // ell ** k)     # order ell^j, j <= k
            chain = []
            while T is not None:
                chain.append(T)
                T = G.mul(T, ell)
            if len(chain) >= m:
                return chain[len(chain) - m]
    small = {2: point_of_order(2, 92, 1), 4: point_of_order(2, 92, 2), 8: point_of_order(2, 92, 3), 16: point_of_order(2, 92, 4),
             3: point_of_order(3, 1, 1), 7: point_of_order(7, 2, 1), 13: point_of_order(13, 2, 1)}
    small[6] = G.add(small[2], small[3])
    for o, T in small.items():
        assert G.mul(T, o) is None and all(G.mul(T, o // f) is not None for f in (2, 3, 7, 13) if o % f == 0)
    assert small[2][1] == (0,)
    good = [G.mul(G.gen, rng.randrange(1, cv.r)) for _ in range(4)]
    cof = G.mul(random_curve_point(), cv.r)
    pts = [small[o] for o in sorted(small)] + [None, good[0], cof, G.add(good[1], small[3]), G.add(good[2], small[2]),
                                               G.add(good[3], cof), random_curve_point()]
    n = len(pts)
    data = serde.write_batch(G, pts, False)
    cid = CURVE_NAMES.index(name)
    for start, tau, coeff in ((0, rng.randrange(cv.r), None), (3, 1, 0x8888888888888888888888888888888888888888), (0, 1, 1 << 200),
                              (7, cv.r - 1, rng.randrange(cv.r)), (0, 2, 15), (1, 16, None), (5, 0, None), (0, rng.randrange(cv.r), 7)):
        exp = cref.apply_powers(name, 1, data, n, start, tau, coeff, False, False, nthreads=8)
        args = ["apply", str(cid), "1", str(n), str(start), "0", "0", "%x" % tau, "-" if coeff is None else "%x" % coeff]
        r = subprocess.run([exe] + args, input=data, capture_output=True, env=dict(os.environ, PTAU_GLV="0"))
        assert r.returncode == 0 and r.stdout == exp, (start, tau, coeff)


def test_carry_free_field_arithmetic_on_host():
    """fpz.hpp (the 14 x 28-bit carry-free twin of the BLS12-377 base field, K1's inner loop): conversion from / to the
    reduced field, limb-wise sums and differences, products of unnormalised operands, the quotient-estimate reduction and the
    residue tests, against Python integers on edge values (0, 1, p-1, limb-boundary patterns) and random ones."""
    import random
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    cv = CURVES["bls12_377"]
    p, B = cv.q, cv.fq_bytes
    rng = random.Random(11)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 28) - 1, 1 << 28, (1 << 29) - 1, (1 << 56) - 1, (1 << 364) - 1,
            1 << 364, (1 << 376) - 1, 1 << 376, int("fffffff" * 13, 16) % p, int("8000000" * 13, 16) % p, p - (1 << 28), p - (1 << 364)]
    vals = [(a, b) for a in edge for b in (0, 1, p - 1, a, (p - a) % p, edge[(edge.index(a) * 5 + 2) % len(edge)])]
    vals += [(rng.randrange(p), rng.randrange(p)) for _ in range(300)]
    data = b"".join(a.to_bytes(B, "little") + b.to_bytes(B, "little") for a, b in vals)
    r = subprocess.run([exe, "fieldz", "0", "1", str(len(vals)), "0", "0", "0", "0", "-"], input=data, capture_output=True)
    assert r.returncode == 0, r.returncode
    rec = 8 * B + 1
    assert len(r.stdout) == rec * len(vals)
    for i, (a, b) in enumerate(vals):
        o = r.stdout[i * rec:(i + 1) * rec]
        g = lambda k: int.from_bytes(o[k * B:(k + 1) * B], "little")
        assert g(0) == (a + b) % p and g(1) == (a - b) % p and g(2) == a * b % p and g(3) == a * a % p and g(4) == (-a) % p
        assert g(5) == (5 * a - 8 * b) % p
        assert g(6) == (62 * a + 32 * b) % p
        assert g(7) == (-4 * a * b) % p, (a, b)      # a b + 5 (32 p - a) b: two products, one reduction (ps_mul2)
        assert o[8 * B] == (1 | (2 if a == b else 0) | 4 | (8 if a == 0 else 0)), (a, b, o[8 * B])


@pytest.mark.parametrize("curve", ["mnt4_753", "mnt6_753"])
def test_gls_scalar_split_on_host(curve):
    """k1_lazy_a.hpp's gls2_split (the Euclidean split k = k0 + k1 N, N = |t - 1|, of K1's GLS loop on the MNT twists, through
    the reciprocal floor(2^768 / N) and at most three corrections): exact recomposition, both halves in range, on edge scalars
    (0, r - 1, multiples of N and their neighbours, all-ones words) and random ones."""
    import random
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    cv = CURVES[curve]
    r_ = cv.r
    # N from the generated constants (the twist's Frobenius eigenvalue is t - 1 modulo r)
    import re
    inc = open(os.path.join(ROOT, "snark-setup-operator_amd", "csrc", "curve_constants.inc")).read()
    tag = "Mnt4G2Cfg" if curve == "mnt4_753" else "Mnt6G2Cfg"
    body = inc[inc.index("struct " + tag):]
    m = re.search(r"PSI_N12\[12\]\s*=\s*\{([^}]*)\}", body)
    words = [int(w.strip().rstrip("u"), 0) for w in m.group(1).split(",") if w.strip()]
    N = sum(w << (32 * i) for i, w in enumerate(words))
    assert 1 << 376 <= N < 1 << 377
    rng = random.Random(5)
    ks = [0, 1, 2, N - 1, N, N + 1, 2 * N - 1, 2 * N, r_ - 1, r_ - 2, (r_ // N) * N, (r_ // N) * N - 1, (r_ // N) * N + 1 if (r_ // N) * N + 1 < r_ else 0,
          (1 << 752) - 1 if (1 << 752) - 1 < r_ else (1 << 751) - 1, int("ffffffff" * 23, 16) % r_]
    ks += [rng.randrange(r_) for _ in range(2000)]
    ks += [q * N + d for q in (rng.randrange(r_ // N) for _ in range(200)) for d in (0, 1, N - 1)]
    data = b"".join(k.to_bytes(96, "little") for k in ks)
    cid = "2" if curve == "mnt4_753" else "3"
    r = subprocess.run([exe, "gls2split", cid, "2", "0", str(len(ks)), "0", "0", "0", "-"], input=data, capture_output=True)
    assert r.returncode == 0, r.returncode
    assert len(r.stdout) == 96 * len(ks)
    for i, k in enumerate(ks):
        k0 = int.from_bytes(r.stdout[96 * i:96 * i + 48], "little")
        k1 = int.from_bytes(r.stdout[96 * i + 48:96 * i + 96], "little")
        assert k0 < N and k1 < 1 << 377 and k0 + k1 * N == k, (hex(k), hex(k0), hex(k1))


def test_file_level_tail_argument_checks_without_gpu(pkg, tmp_path):
    """phase1_cli::combine / transform_ratios in their file forms (control.rs:564-568, 587-591): argument and I/O errors are
    reported before anything touches a device — a list that names the wrong number of responses, a missing file, Full-mode
    parameters handed to combine, a device listed twice."""
    import ctypes
    L = pkg.load_library()
    prm0 = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, 4, 0, 3, 4)
    full = pkg.Phase1Parameters.new_full("bls12_377", 0, 3, 4)
    lst = tmp_path / "list"
    lst.write_text("a\nb\n")                      # 2 paths, the ceremony has 4 chunks
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.combine_file(str(lst), str(tmp_path / "combined"), prm0)
    assert e.value.code == -1 and "4 chunks" in str(e.value)
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.combine_file(str(tmp_path / "missing"), str(tmp_path / "combined"), prm0)
    assert e.value.code == -9
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.combine_file(str(lst), str(tmp_path / "combined"), full)
    assert e.value.code == -1
    lst.write_text("".join(str(tmp_path / ("r%d" % i)) + "\r\n\n" for i in range(4)))   # CRLF and blank lines tolerated
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.combine_file(str(lst), str(tmp_path / "combined"), prm0)
    assert e.value.code == -9 and "r0" in str(e.value)
    assert not os.path.exists(tmp_path / "combined")
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_file(str(tmp_path / "missing"), 0, full)
    assert e.value.code == -9
    buf = bytes(full.accumulator_size)
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_devices(buf, 0, full, [0, 0])
    assert e.value.code == -8          # a repeated ordinal is a legal slice list (ptau.h): the call gets as far as the missing device
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_devices(buf, 0, full, [0] * 17)
    assert e.value.code == -1
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_devices(buf[:-1], 0, full, [0])
    assert e.value.code == -2
    t = pkg._lib.ptau_timing()
    assert ctypes.sizeof(t) == 9 * 8 + 8 + 8 and hasattr(t, "ratio_checks") and hasattr(t, "pok_checks") and hasattr(t, "msm_tail_ms")


def test_host_hashing_of_several_lanes_runs_concurrently(pkg):
    """Several in-process lanes per GPU only help if one lane's BLAKE2b does not serialise another's: the
    library's hashing holds no lock (the one process-wide mutex guards context CREATION only).  Control = hashlib's BLAKE2b on the
    same buffers and threads (it releases the GIL and shares nothing): however many cores this box really grants, the library's
    digests on N threads must not take longer than the control's by more than the two implementations' single-thread ratio."""
    import hashlib
    import threading
    import time
    L = pkg.load_library()
    n = min(8, len(os.sched_getaffinity(0)))
    if n < 2:
        pytest.skip("needs >= 2 host cores")
    bufs = [bytes([k]) * (16 << 20) for k in range(n)]
    outs = [ctypes.create_string_buffer(64) for _ in range(n)]
    L.ptau_blake2b512(bufs[0], len(bufs[0]), outs[0])      # implementation choice + page-in

    def timed(fn, threads):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            if threads:
                th = [threading.Thread(target=fn, args=(k,)) for k in range(n)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            else:
                for k in range(n):
                    fn(k)
            best = min(best, time.perf_counter() - t0)
        return best
    ours = lambda k: L.ptau_blake2b512(bufs[k], len(bufs[k]), outs[k])
    ctrl = lambda k: hashlib.blake2b(bufs[k]).digest()
    seq_o, seq_c = timed(ours, False), timed(ctrl, False)
    par_o, par_c = timed(ours, True), timed(ctrl, True)
    assert [o.raw for o in outs] == [hashlib.blake2b(b).digest() for b in bufs]
    # scaling of ours (seq / par) at least 0.7 x the control's scaling
    assert seq_o / par_o > 0.7 * (seq_c / par_c), (seq_o, par_o, seq_c, par_c)


def test_only_context_creation_takes_a_process_wide_lock():
    """Lanes on DIFFERENT ordinals must never meet on a lock, lanes on one ordinal only for the device section: the library
    holds exactly one process-wide mutex (g_init_mu) and takes it in ensure_ctx only; everything else is per DeviceCtx."""
    import re
    src = open(os.path.join(ROOT, "snark-setup-operator_amd", "csrc", "capi.hip")).read()
    globals_ = re.findall(r"^std::mutex\s+(\w+);", src, flags=re.M)
    assert globals_ == ["g_init_mu"], globals_
    uses = [m.start() for m in re.finditer(r"g_init_mu", src)]
    body_start = src.index("int ensure_ctx(")
    body_end = src.index("\nint grow(")
    assert all(body_start < u < body_end or src[u - 11:u] == "std::mutex " for u in uses)
    # per-device state: the device mutex and the pairing mutex live in DeviceCtx
    ctx = src[src.index("struct DeviceCtx {"):src.index("DeviceCtx g_ctx[MAX_DEVICES];")]
    assert "std::mutex mu;" in ctx and "std::mutex pair_mu;" in ctx


def test_membership_fused_msm_plan_meets_the_security_parameter():
    """msm_plan with a bucket test cost (the batched subgroup check riding on the ratio check's MSM, and the probe of
    BatchExpMode::Auto): every window is a round of setup-utils' batched check with false-accept probability <= 2^-(c-1)
    (2^-(c-2) for the top window), so W (c - 1) - 1 >= 128 (SECURITY_PARAM) must hold for every size and cost, the
    coefficients are exactly W c - 1 bits wide (full windows, no carry out of the top one) and fit the PRF; the plain plan
    (cost 0) still serves 127-bit coefficients."""
    exe = os.path.join(ROOT, "tests", "native", "host_emul")
    if not os.path.exists(exe):
        pytest.skip("host_emul not built")
    def plan(n, cost, resident_k):
        r = subprocess.run([exe, "msmplan", "0", "1", str(n), str(cost), str(resident_k), "0", "0", "-"], capture_output=True, stdin=subprocess.DEVNULL)
        assert r.returncode == 0
        return [int(v) for v in r.stdout.split()]
    for n in (1, 2, 100, 4097, 1 << 13, 1 << 16, 1 << 18, (1 << 20) - 1, 1 << 20, 1 << 22):
        c, W, B, bits = plan(n, 0, 0)
        assert B == 1 << (c - 1) and W * c >= 127 + 1 and bits == 127
        for cost in (1, 46, 93, 170, 300, 565):
            for resident in (0, 65, 131):
                c, W, B, bits = plan(n, cost, resident)
                assert 3 <= c <= 16 and B == 1 << (c - 1)
                assert W * (c - 1) - 1 >= 128, (n, cost, c, W)
                assert bits == W * c - 1 and bits <= 200
    # more expensive tests push the plan towards fewer, larger buckets
    assert plan(1 << 20, 300, 131)[0] <= plan(1 << 20, 46, 131)[0]
