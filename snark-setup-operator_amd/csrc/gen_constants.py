"""Emit curve_constants.inc: 32-bit-limb Montgomery constants for the HIP kernels.

Input: the curve moduli / coefficients of SURVEY.md App. B (mirrored here, NOT imported from
oracle/: the product build must not depend on the test oracle).  Run by __graft_entry__.build().
The reference obtains these from ark-{bls12-377,bw6-761,mnt4-753,mnt6-753} 0.4.0
(/root/reference/Cargo.toml:42-49).
"""
import os

BLS_Q = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
BLS_R = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
BW6_Q = 0x0122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b
MNT_A = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888253786114353726529584385201591605722013126468931404347949840543007986327743462853720628051692141265303114721689601
MNT_B = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888458477323173057491593855069696241854796396165721416325350064441470418137846398469611935719059908164220784476160001
MNT4_B1 = 28798803903456388891410036793299405764940372360099938340752576406393880372126970068421383312482853541572780087363938442377933706865252053507077543420534380486492786626556269083255657125025963825610840222568694137138741554679540
MNT6_B1 = 11625908999541321152027340224010374716841167701783584648338908235410859267060079819722747939267925389062611062156601938166010098747920378738927832658133625454260115409075816187555055859490253375704728027944315501122723426879114
BLS_G2_B1 = 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906

# Base fields whose curves use the reduced-radix (2^29) field class.  Measured (MI355X, updates/s, 32-bit -> 29-bit limbs):
# BLS12-377 G1 26.1 M -> 31.4 M, G2 6.7 M -> 7.6 M; MNT4 G1 (2^18) 1.07 M -> 1.11 M; MNT6 G2 86.6 k -> 96.6 k;
# BW6-761 (2^18) 3.19 M -> 3.05 M — its 27-limb products go by reference through memory and are not MAC-bound, so it stays on 32.
RADIX29 = {"BlsFq", "Mnt4Fq", "Mnt6Fq", "Bw6Fq"}
# BW6-761 (round 2): 28 limbs of 29 bits — one more than ceil(761 / 29) — so that the element splits evenly over a lane pair
# (fpl.hpp); its single-lane kernels (verify path) use the same representation so both kinds of kernel exchange values limb for limb.
RADIX29_LIMBS = {"Bw6Fq": 28}
TOWER_NR = {"BlsFq": -5, "Mnt4Fq": 13}   # the quadratic non-residue of the curve's Fq2 (G2 of BLS12-377 / MNT4-753)
LAZY28 = {"BlsFq", "Bw6Fq"}      # base fields that also get a carry-free 28-bit twin (fpz.hpp; lane-split: fplz.hpp)
# ... and the 753-bit fields of the MNT cycle: 28 limbs of 28 bits over a lane pair (two more than the 26 x 29-bit reduced form),
# R = 2^784 >= 2^31 p — a != 0 curves: no common-Z table, Jacobian table entries, lazy_a.hpp
LAZY28_WIDE = {"Mnt4Fq", "Mnt6Fq"}


# The MNT twists' G2 generators: ark-mnt4-753 / ark-mnt6-753's constants are not recoverable offline, so the tree holds none
# and ptau_new_challenge refuses those curves (the host passes G2Affine::generator() through
# ptau_new_challenge_with_generators; the tests bring their own synthetic order-r point) — UNTIL the deferred pin has run:
# tools/pin_reference.sh writes pinned_generators.json (arkworks' own G2Affine::generator() bytes, uncompressed) next to this
# file, the next build emits them as Generators::MNT4_G2 / MNT6_G2 with HAVE_MNT_G2 = true, and the built-in form works.
def pinned_mnt_g2():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pinned_generators.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    g4, g6 = bytes.fromhex(d["mnt4_753"]["g2_uncompressed"]), bytes.fromhex(d["mnt6_753"]["g2_uncompressed"])
    if (len(g4), len(g6)) != (380, 570):
        raise SystemExit("pinned_generators.json: MNT4-753 / MNT6-753 G2 generators must be 380 / 570 bytes uncompressed")
    return g4, g6


BW6_G1_XY = (0x01075b020ea190c8b277ce98a477beaee6a0cfb7551b27f0ee05c54b85f56fc779017ffac15520ac11dbfcd294c2e746a17a54ce47729b905bd71fa0c9ea097103758f9a280ca27f6750dd0356133e82055928aca6af603f4088f3af66e5b43d,
             0x0058b84e0a6fc574e6fd637b45cc2a420f952589884c9ec61a7348d2a2e573a3265909f1af7e0dbac5b8fa1771b5b806cc685d31717a4c55be3fb90b6fc2cdd49f9df141b3053253b2b08119cad0fb93ad1cb2be0b20d2a1bafc8f2db4e95363)
BW6_G2_XY = (0x0110133241d9b816c852a82e69d660f9d61053aac5a7115f4c06201013890f6d26b41c5dab3da268734ec3f1f09feb58c5bbcae9ac70e7c7963317a300e1b6bace6948cb3cd208d700e96efbc2ad54b06410cf4fe1bf995ba830c194cd025f1c,
             0x0017c3357761369f8179eb10e4b6d2dc26b7cf9acec2181c81a78e2753ffe3160a1d86c80b95a59c94c97eb733293fef64f293dbd2c712b88906c170ffa823003ea96fcd504affc758aa2d3a3c5a02a591ec0594f9eac689eb70a16728c73b61)


def _cofactors():
    """(G1, G2) cofactors #E/r per curve, from the curve families' parametrisations (ark-*'s COFACTOR constants)."""
    x = 0x8508c00000000001
    assert (x ** 4 - x ** 2 + 1) == BLS_R and ((x - 1) ** 2 * BLS_R) // 3 + x == BLS_Q
    bls = ((x - 1) ** 2 // 3, (x ** 8 - 4 * x ** 7 + 5 * x ** 6 - 4 * x ** 4 + 6 * x ** 3 - 4 * x ** 2 - 4 * x + 13) // 9)
    # BW6-761 (eprint 2020/351): t = x^5 - 3x^4 + 3x^3 - x + 3 + 13 r, 4q = t^2 + 3y^2; E: y^2 = x^3 - 1 has q + 1 - t points,
    # the M-twist y^2 = x^3 + 4 has q + 1 - (3y + t)/2
    import math
    t = x ** 5 - 3 * x ** 4 + 3 * x ** 3 - x + 3 + 13 * BLS_Q
    y = math.isqrt((4 * BW6_Q - t * t) // 3)
    assert t * t + 3 * y * y == 4 * BW6_Q
    cands = [BW6_Q + 1 - t, BW6_Q + 1 + t, BW6_Q + 1 - (t + 3 * y) // 2, BW6_Q + 1 + (t + 3 * y) // 2,
             BW6_Q + 1 - (t - 3 * y) // 2, BW6_Q + 1 + (t - 3 * y) // 2]
    cof = sorted(o // BLS_Q for o in cands if o % BLS_Q == 0)
    assert len(cof) == 2 and cof[1] - cof[0] == 7
    bw6 = (cof[1], cof[0])   # ...4876 (G1), ...4869 (G2)
    # MNT cycle: #E4(Fq4) = r4 = q6, #E6(Fq6) = r6 = q4 (cofactor 1); twists over Fq2 / Fq3 from the traces
    t4, t6 = MNT_A + 1 - MNT_B, MNT_B + 1 - MNT_A
    o4 = MNT_A * MNT_A + 1 + (t4 * t4 - 2 * MNT_A)
    o6 = MNT_B ** 3 + 1 + (t6 ** 3 - 3 * MNT_B * t6)
    assert o4 % MNT_B == 0 and o6 % MNT_A == 0
    return {"bls12_377": bls, "bw6": bw6, "mnt4_753": (1, o4 // MNT_B), "mnt6_753": (1, o6 // MNT_A)}


COFACTORS = _cofactors()


def ec_mul_j0(P, k, p):
    """k*P on y^2 = x^3 + b over F_p (affine double-and-add; a = 0 so b is not needed)."""
    def add(A, B):
        if A is None:
            return B
        if B is None:
            return A
        (x1, y1), (x2, y2) = A, B
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            l = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            l = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (l * l - x1 - x2) % p
        return (x3, (l * (x1 - x3) - y1) % p)
    R = None
    for bit in bin(k)[2:]:
        R = add(R, R)
        if bit == "1":
            R = add(R, P)
    return R


def glv_beta_for(P, w, lam, p):
    """The cube root of unity beta in {w, w^2} with (beta x, y) = lam * (x, y)."""
    target = ec_mul_j0(P, lam, p)
    for beta in (w, w * w % p):
        if (beta * P[0] % p, P[1]) == target:
            return beta
    raise AssertionError("no cube root of unity matches lambda")


def glv_lattice(r, lam):
    """Two short vectors (a, b) with a + b*lam = 0 mod r (extended Euclid, GLV 2001)."""
    rs, ts = [r, lam], [0, 1]
    while rs[-1] * rs[-1] >= r:
        qq = rs[-2] // rs[-1]
        rs.append(rs[-2] - qq * rs[-1])
        ts.append(ts[-2] - qq * ts[-1])
    l = len(rs) - 1
    v1 = (rs[l], -ts[l])
    qq = rs[l - 1] // rs[l]
    c1, c2 = (rs[l - 1], -ts[l - 1]), (rs[l - 1] - qq * rs[l], -(ts[l - 1] - qq * ts[l]))
    v2 = c1 if c1[0] ** 2 + c1[1] ** 2 < c2[0] ** 2 + c2[1] ** 2 else c2
    return v1, v2


_LB = 32   # limb width of the field whose Montgomery constants are being emitted (set by field() / coeffs() / pairing_cfg())


def limbs(x, n, lb=None):
    lb = lb or _LB
    return ", ".join("0x%08xu" % ((x >> (lb * i)) & ((1 << lb) - 1)) for i in range(n))


def field(name, p, cubic_nr=None, sqrt_tables=False, lb=32, n_limbs=None, tower_nr=None):
    """Parameter struct of one prime field.  lb = 32: full 32-bit limbs (Fp, fp.hpp).  lb = 29: reduced radix (Fp29,
    fp29.hpp): ceil(bits/29) limbs of 29 bits, R = 2^(29 N), so that the 2N products of a product-scanning column fit a
    64-bit accumulator and the multiply-accumulate needs no carry instruction.  Exponents and the *32 arrays stay packed."""
    global _LB
    _LB = lb
    n = n_limbs or (p.bit_length() + lb - 1) // lb
    assert lb * n >= p.bit_length()
    n32 = (p.bit_length() + 31) // 32
    R = 1 << (lb * n)
    inv = (-pow(p, -1, 1 << lb)) % (1 << lb)
    s = []
    s.append("struct %s {" % name)
    s.append("  static constexpr int LB = %d;" % lb)
    s.append("  static constexpr int N = %d;" % n)
    s.append("  static constexpr int N32 = %d;" % n32)
    s.append("  static constexpr int BITS = %d;" % p.bit_length())
    s.append("  static constexpr int BYTES = %d;" % ((p.bit_length() + 7) // 8))
    s.append("  static constexpr uint32_t INV = 0x%08xu;" % inv)
    s.append("  static constexpr uint32_t MOD[%d] = {%s};" % (n, limbs(p, n)))
    s.append("  static constexpr uint32_t MOD32[%d] = {%s};   // packed" % (n32, limbs(p, n32, 32)))
    s.append("  static constexpr uint32_t R1[%d] = {%s};" % (n, limbs(R % p, n)))
    s.append("  static constexpr uint32_t R2[%d] = {%s};" % (n, limbs(R * R % p, n)))
    s.append("  static constexpr uint32_t HALF[%d] = {%s};  // (p-1)/2" % (n, limbs((p - 1) // 2, n)))
    s.append("  static constexpr uint32_t PM2[%d] = {%s};   // p-2, packed exponent" % (n32, limbs(p - 2, n32, 32)))
    S = ((p - 1) & -(p - 1)).bit_length() - 1
    T = (p - 1) >> S
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    s.append("  static constexpr int TS_S = %d;" % S)
    s.append("  static constexpr uint32_t QP1H[%d] = {%s};      // (p+1)/2 = 1/2, canonical limbs" % (n, limbs((p + 1) // 2, n)))
    s.append("  static constexpr uint32_t QP1H_E[%d] = {%s};    // (p+1)/2, packed exponent" % (n32, limbs((p + 1) // 2, n32, 32)))
    s.append("  static constexpr uint32_t TS_TM1H[%d] = {%s};   // (T-1)/2, packed exponent" % (n32, limbs((T - 1) // 2, n32, 32)))
    s.append("  static constexpr int TS_TM1H_BITS = %d;" % max(1, ((T - 1) // 2).bit_length()))
    if lb != 32:
        # inversion runs on the packed twin struct (same modulus, 32-bit limbs): see Fp29::inverse
        s.append("  using Twin32 = %s;" % name[:-3])
        e = 3 * lb * n - 64 * n32
        fix = pow(2, e, p) if e >= 0 else pow(pow(2, -e, p), -1, p)
        s.append("  static constexpr uint32_t INV_FIX[%d] = {%s};   // 2^(3*%d*N - 64*N32) mod p" % (n, limbs(fix, n), lb))
    if sqrt_tables:
        # Pohlig-Hellman data for the 2-Sylow part of the square root (fp.hpp sqrt_ex): g = z^T generates the
        # 2^S-torsion; a residue has a^T = (g^2)^d with d < 2^E, E = S-1, found W bits at a time from the bottom.
        #   SQRT_ZLOW[k]     = low Montgomery limb of zeta^k, zeta = (g^2)^(2^(E-W)) of order 2^W (lookup key)
        #   SQRT_GNEG[j][k]  = g^(-k 2^(W j)), Montgomery: strips window j from the residue and builds g^(-d)
        E = S - 1
        W = min(8, E) if E > 0 else 0      # 8-bit windows: S^2 / 16 squarings in the torsion stage (6-bit: S^2 / 12), 256-entry lookup
        L = (E + W - 1) // W if W else 0
        g = pow(z, T, p)
        s.append("  static constexpr int SQRT_W = %d;" % W)
        s.append("  static constexpr int SQRT_L = %d;" % L)
        if W:
            h = g * g % p
            zeta = pow(h, 1 << (E - W), p)
            zl = [pow(zeta, k, p) * R % p & ((1 << lb) - 1) for k in range(1 << W)]
            assert len(set(zl)) == len(zl), "low-limb collision in the sqrt lookup table"
            s.append("  static constexpr uint32_t SQRT_ZLOW[%d] = {%s};" % (1 << W, ", ".join("0x%08xu" % v for v in zl)))
            ginv = pow(g, -1, p)
            rows = []
            for j in range(L):
                base = pow(ginv, 1 << (W * j), p)
                acc = 1
                for k in range(1 << W):
                    rows.append("{%s}" % limbs(acc * R % p, n))
                    acc = acc * base % p
            s.append("  static constexpr uint32_t SQRT_GNEG[%d][%d] = {%s};" % (L << W, n, ",\n    ".join(rows)))
        else:
            s.append("  static constexpr uint32_t SQRT_ZLOW[1] = {0};")
            s.append("  static constexpr uint32_t SQRT_GNEG[1][%d] = {{0}};" % n)
    if tower_nr is not None and sqrt_tables:
        # square roots in Fq[u]/(u^2 - nr) (ext.hpp Fp2::sqrt): of the two candidates delta+ / delta- = nr (a1/2)^2 / delta+ exactly
        # one is a residue; when delta+ is not, nr delta+ is, and its (T-1)/2-th power is SQRT_NR_C times the one already computed
        assert pow(tower_nr % p, (p - 1) // 2, p) == p - 1
        s.append("  static constexpr int SQRT_NR = %d;" % tower_nr)
        s.append("  static constexpr uint32_t SQRT_NR_C[%d] = {%s};     // nr^((T-1)/2), Montgomery" % (n, limbs(pow(tower_nr % p, (T - 1) // 2, p) * R % p, n)))
        s.append("  static constexpr uint32_t SQRT_NR_INV[%d] = {%s};   // 1/nr, Montgomery" % (n, limbs(pow(tower_nr % p, -1, p) * R % p, n)))
    else:
        s.append("  static constexpr int SQRT_NR = 0;")
    if cubic_nr is not None:
        # Frobenius on Fq[u]/(u^3 - nr): (a0, a1, a2)^q = (a0, a1*c1, a2*c1^2), c1 = nr^((q-1)/3)
        c1 = pow(cubic_nr, (p - 1) // 3, p)
        s.append("  static constexpr uint32_t FROB3_C1[%d] = {%s};" % (n, limbs(c1 * R % p, n)))
        s.append("  static constexpr uint32_t FROB3_C2[%d] = {%s};" % (n, limbs(c1 * c1 % p * R % p, n)))
    s.append("};")
    return "\n".join(s), n, R


def mont(x, p, R):
    return x % p * R % p


def main():
    out = ["// GENERATED by gen_constants.py — do not edit.", "#pragma once", "#include <cstdint>", "namespace ptau {"]
    fields = {}
    for name, p in (("BlsFq", BLS_Q), ("BlsFr", BLS_R), ("Bw6Fq", BW6_Q), ("Mnt4Fq", MNT_A), ("Mnt6Fq", MNT_B)):
        txt, n, R = field(name, p, cubic_nr=11 if name == "Mnt6Fq" else None, sqrt_tables=name.endswith("Fq"), tower_nr=TOWER_NR.get(name))
        fields[name] = (p, n, R, 32)
        out.append(txt)
    # reduced-radix twins for the base fields of the curves listed in RADIX29 (the radix-32 structs stay: they are also the
    # SCALAR fields of BW6-761 / MNT6 / MNT4, and scalars are consumed as packed 32-bit words)
    for name, p in (("BlsFq", BLS_Q), ("Bw6Fq", BW6_Q), ("Mnt4Fq", MNT_A), ("Mnt6Fq", MNT_B)):
        if name in RADIX29:
            txt, n, R = field(name + "R29", p, cubic_nr=11 if name == "Mnt6Fq" else None, sqrt_tables=True, lb=29,
                              n_limbs=RADIX29_LIMBS.get(name), tower_nr=TOWER_NR.get(name))
            fields[name + "R29"] = (p, n, R, 29)
            out.append(txt)
    # carry-free ("lazy") twin of a base field for the inner loop of K1 (fpz.hpp): 28-bit limbs, one limb more than the
    # modulus needs, so that R = 2^(28 N) >= 2^15 p: sums and differences of a few elements stay valid operands of a product
    # (no carry chain, no conditional subtraction) and a product brings any such operand back below 2p.
    lazy_twins = []
    for name, p in ((nm, fields[nm][0]) for nm in sorted(LAZY28 | LAZY28_WIDE)):
        lb, n = 28, (p.bit_length() + 15 + 27) // 28
        if name in LAZY28_WIDE:
            assert n % 2 == 0 and n == 28
        Rz, R29 = 1 << (lb * n), fields[name + "R29"][2]
        assert Rz >= p << 15      # (BLS12-377: p = 1 mod 2^28, so m_i = -acc mod 2^28 and m_i p_0 is an addition)
        out.append("struct %sZ28 {" % name)
        out.append("  static constexpr int LB = %d;" % lb)
        out.append("  static constexpr int N = %d;" % n)
        out.append("  static constexpr uint32_t INV = 0x%08xu;" % ((-pow(p, -1, 1 << lb)) % (1 << lb)))
        out.append("  static constexpr uint32_t MOD[%d] = {%s};" % (n, limbs(p, n, lb)))
        out.append("  static constexpr uint32_t R1[%d] = {%s};" % (n, limbs(Rz % p, n, lb)))
        out.append("  static constexpr uint32_t R2[%d] = {%s};   // Rz^2 mod p: canonical -> Montgomery by one product" % (n, limbs(Rz * Rz % p, n, lb)))
        out.append("  static constexpr uint32_t FROM29[%d] = {%s};   // Rz^2 / R29 mod p: (x R29) -> (x Rz) by one product" % (n, limbs(Rz * Rz * pow(R29, -1, p) % p, n, lb)))
        out.append("  static constexpr uint32_t TO29[%d] = {%s};     // R29 mod p: (x Rz) -> (x R29)" % (n, limbs(R29 % p, n, lb)))
        out.append("  using P29 = %sR29;" % name)
        out.append("};")
        lazy_twins.append((name + "R29", name + "Z28"))
    out.append("template <class P> struct LazyTwin { using type = void; };")
    for a, b in lazy_twins:
        out.append("template <> struct LazyTwin<%s> { using type = %s; };" % (a, b))
    fq_of = lambda base: base + "R29" if base in RADIX29 else base   # the struct a curve over `base` is generated for
    out.append("using Bw6Fr = BlsFq;")
    out.append("using Mnt4Fr = Mnt6Fq;")
    out.append("using Mnt6Fr = Mnt4Fq;")

    def coeffs(cname, fname, deg, a, b, nr, glv=None, endo=None, glv2=None, cofactor=1, psi=None):
        global _LB
        p, n, R, _LB = fields[fname]
        s = ["struct %s {" % cname]
        s.append("  static constexpr bool HAS_GLV = %s;" % ("true" if glv else "false"))
        # endomorphism-based subgroup membership test (eprint 2021/1130):
        #   kind 1 (BLS12 G1): P in G1  <=>  (C0 x, y) == -[X^2] P          (C0 = the cube root of unity acting as -X^2)
        #   kind 2 (BLS12 G2): P in G2  <=>  (C0 conj(x), C1 conj(y)) == [X] P   (untwist-Frobenius-twist, C0, C1 in Fq)
        s.append("  static constexpr int ENDO_KIND = %d;" % (endo[0] if endo else 0))
        if endo:
            s.append("  static constexpr uint64_t ENDO_X = 0x%xull;" % endo[1])
            s.append("  static constexpr uint32_t ENDO_C0[%d] = {%s};" % (n, limbs(endo[2] * R % p, n)))
            s.append("  static constexpr uint32_t ENDO_C1[%d] = {%s};" % (n, limbs(endo[3] * R % p, n)))
            if fname.endswith("R29") and fname[:-3] in LAZY28:   # the same constants for the carry-free twin field (fpz.hpp)
                nz = (p.bit_length() + 15 + 27) // 28
                s.append("  static constexpr uint32_t ENDO_C0Z[%d] = {%s};" % (nz, limbs(endo[2] * (1 << (28 * nz)) % p, nz, 28)))
                s.append("  static constexpr uint32_t ENDO_C1Z[%d] = {%s};" % (nz, limbs(endo[3] * (1 << (28 * nz)) % p, nz, 28)))
            if endo[0] == 2:
                # psi^j (X, Y, Z) = (C0^j conj^j(X), C1^j conj^j(Y), conj^j(Z)), j = 1..3: rows (C0^j, C1^j) for the
                # 4-dimensional decomposition k = k0 + k1 X + k2 X^2 + k3 X^3 (psi acts as [X] on G2, r < X^4)
                rows = []
                for j in (1, 2, 3):
                    rows.append("{%s}" % limbs(pow(endo[2], j, p) * R % p, n))
                    rows.append("{%s}" % limbs(pow(endo[3], j, p) * R % p, n))
                s.append("  static constexpr uint32_t GLS_C[6][%d] = {%s};" % (n, ",\n    ".join(rows)))
                if fname.endswith("R29") and fname[:-3] in LAZY28:
                    # the same six constants for the carry-free twin field (fpz.hpp): 28-bit limbs, R = 2^(28 NZ)
                    nz = (p.bit_length() + 15 + 27) // 28
                    Rz = 1 << (28 * nz)
                    rows = []
                    for j in (1, 2, 3):
                        rows.append("{%s}" % limbs(pow(endo[2], j, p) * Rz % p, nz, 28))
                        rows.append("{%s}" % limbs(pow(endo[3], j, p) * Rz % p, nz, 28))
                    s.append("  static constexpr uint32_t GLS_CZ[6][%d] = {%s};" % (nz, ",\n    ".join(rows)))
        # Membership of a G2 point of an MNT twist through psi = twist o Frobenius o untwist (psi^2 - t psi + q = 0, psi = [q mod r]
        # = [t - 1] on the order-r subgroup): the endomorphism psi - [t - 1] has degree chi(t - 1) = q - t + 1 = r, so its kernel —
        # r points, containing the subgroup — IS the subgroup:  P in G2  <=>  psi(P) == [t - 1]P, exact for any curve point, with
        # |t - 1| ~ sqrt(r) (377 bits instead of 753).  psi(x, y) = (PSI_CX x^q, PSI_CY y^q), both constants in Fq.
        s.append("  static constexpr int PSI_KIND = %d;" % (1 if psi else 0))
        if psi:
            pcx, pcy, pn, pneg, nwords = psi
            s.append("  static constexpr uint32_t PSI_CX[%d] = {%s};" % (n, limbs(pcx * R % p, n)))
            s.append("  static constexpr uint32_t PSI_CY[%d] = {%s};" % (n, limbs(pcy * R % p, n)))
            s.append("  static constexpr uint32_t PSI_N[%d] = {%s};   // |t - 1|, packed words" % (nwords, limbs(pn, nwords, 32)))
            s.append("  static constexpr bool PSI_NEG = %s;            // t - 1 < 0" % ("true" if pneg else "false"))
            if fname.endswith("R29") and fname[:-3] in LAZY28_WIDE:
                # GLS on the twist (k1_lazy_a.hpp): psi = [t - 1] on G2, so k = k0 + k1 N (N = |t - 1|, Euclidean split: both halves
                # below 2^377) and k P = k0 P + k1 (+-psi(P)).  psi on a JACOBIAN entry: (CX X^q, CY Y^q, Z^q); Frobenius acts on
                # coefficient j as a multiplication by F_j = NR^(j (q - 1) / deg) (F = (1, -1) for Fq2).  Per-coefficient constants
                # in the carry-free field (28 limbs of 28 bits, R = 2^784), the sign of t - 1 folded into the Y row:
                #   KX_j = CX F_j,   KY_j = +-CY F_j,   KZ_j = F_j
                nz, Rz = 28, 1 << (28 * 28)
                f1 = pow(nr % p, (p - 1) // deg, p)
                F = [pow(f1, j, p) for j in range(deg)]
                sgn = p - 1 if pneg else 1
                rows = lambda vals: ",\n    ".join("{%s}" % limbs(v * Rz % p, nz, 28) for v in vals)
                s.append("  static constexpr uint32_t PSI_KXZ[%d][%d] = {%s};" % (deg, nz, rows([pcx * f % p for f in F])))
                s.append("  static constexpr uint32_t PSI_KYZ[%d][%d] = {%s};" % (deg, nz, rows([sgn * pcy * f % p for f in F])))
                s.append("  static constexpr uint32_t PSI_KZZ[%d][%d] = {%s};" % (deg, nz, rows(F)))
                assert pn.bit_length() <= 377 and pn.bit_length() > 352
                s.append("  static constexpr uint32_t PSI_N12[12] = {%s};   // N = |t - 1|" % limbs(pn, 12, 32))
                s.append("  static constexpr uint32_t PSI_RECIP[13] = {%s};   // floor(2^768 / N)" % limbs((1 << 768) // pn, 13, 32))
        s.append("  static constexpr int GLV_KIND = %d;   // 0 none, 1 half-size lambda (floor division), 2 lattice (Babai rounding)" % (1 if glv else (2 if glv2 else 0)))
        if glv2:
            # phi(x, y) = (BETA x, y) = LAMBDA (x, y); short basis (a1, b1), (a2, b2) of {(u, v): u + v LAMBDA = 0 mod r};
            # c_i = (k G_i) >> 384 approximates the Babai coefficients, k1 = k + E11 c1|a1| + E12 c2|a2|,
            # k2 = E21 c1|b1| + E22 c2|b2|  (signed), |k1|, |k2| < 2^191 (checked over edge and random scalars below).
            beta, lam, rr = glv2
            (a1, b1), (a2, b2) = glv_lattice(rr, lam)
            det = a1 * b2 - a2 * b1
            assert abs(det) == rr
            sg = lambda v: 1 if v >= 0 else -1
            s1, s2 = sg(b2) * sg(det), -sg(b1) * sg(det)
            g1, g2 = ((abs(b2) << 384) + rr // 2) // rr, ((abs(b1) << 384) + rr // 2) // rr
            e11, e12, e21, e22 = -s1 * sg(a1), -s2 * sg(a2), -s1 * sg(b1), -s2 * sg(b2)
            def decomp(k):
                c1, c2 = (k * g1) >> 384, (k * g2) >> 384
                return k + e11 * c1 * abs(a1) + e12 * c2 * abs(a2), e21 * c1 * abs(b1) + e22 * c2 * abs(b2)
            import random as _r
            rg = _r.Random(7)
            for k in [0, 1, 2, rr - 1, rr - 2, rr // 2, rr // 3, lam, lam - 1, lam + 1] + [rg.randrange(rr) for _ in range(2000)]:
                k1, k2 = decomp(k)
                assert (k1 + k2 * lam - k) % rr == 0 and abs(k1) < 1 << 191 and abs(k2) < 1 << 191
            for v in (a1, a2, b1, b2):
                assert abs(v) < 1 << 192
            assert g1 < 1 << 224 and g2 < 1 << 224
            s.append("  static constexpr uint32_t GLV_BETA[%d] = {%s};" % (n, limbs(beta * R % p, n)))
            if fname.endswith("R29") and fname[:-3] in LAZY28:   # the same constant in the carry-free twin field: a literal in the loop
                nz = (p.bit_length() + 15 + 27) // 28
                s.append("  static constexpr uint32_t GLV_BETAZ[%d] = {%s};" % (nz, limbs(beta * (1 << (28 * nz)) % p, nz, 28)))
            s.append("  static constexpr uint32_t GLV2_A1[6] = {%s};" % limbs(abs(a1), 6, 32))
            s.append("  static constexpr uint32_t GLV2_A2[6] = {%s};" % limbs(abs(a2), 6, 32))
            s.append("  static constexpr uint32_t GLV2_B1[6] = {%s};" % limbs(abs(b1), 6, 32))
            s.append("  static constexpr uint32_t GLV2_B2[6] = {%s};" % limbs(abs(b2), 6, 32))
            s.append("  static constexpr uint32_t GLV2_G1[7] = {%s};" % limbs(g1, 7, 32))
            s.append("  static constexpr uint32_t GLV2_G2[7] = {%s};" % limbs(g2, 7, 32))
            s.append("  static constexpr int GLV2_E11 = %d, GLV2_E12 = %d, GLV2_E21 = %d, GLV2_E22 = %d;" % (e11, e12, e21, e22))
            # Subgroup membership through the endomorphism: the lattice {(a, b): a + b LAMBDA = 0 mod r} is the prime ideal
            # (r, phi - LAMBDA) of Z[phi] (phi^2 + phi + 1 = 0), generated by an element of norm a^2 - a b + b^2 = r: the
            # endomorphism a + b phi then has degree r and its kernel — r points, containing the order-r subgroup on which
            # phi = LAMBDA — IS that subgroup.  So  P in G  <=>  [a]P + [b]phi(P) == O  for ANY curve point P, with
            # |a|, |b| ~ sqrt(r): half the doublings of [r]P == O and the same accept / reject set.
            sub = None
            for (va, vb) in ((a1, b1), (a2, b2), (a1 + a2, b1 + b2), (a1 - a2, b1 - b2)):
                if va * va - va * vb + vb * vb == rr:
                    sub = (va, vb)
                    break
            assert sub is not None and (sub[0] + sub[1] * lam) % rr == 0, "no norm-r lattice vector among the reduced basis"
            assert abs(sub[0]) < 1 << 191 and abs(sub[1]) < 1 << 191
            s.append("  static constexpr uint32_t SUB_A[6] = {%s};" % limbs(abs(sub[0]), 6, 32))
            s.append("  static constexpr uint32_t SUB_B[6] = {%s};" % limbs(abs(sub[1]), 6, 32))
            s.append("  static constexpr bool SUB_NEG_A = %s, SUB_NEG_B = %s;" % ("true" if sub[0] < 0 else "false", "true" if sub[1] < 0 else "false"))
        if glv:
            # phi(x, y) = (BETA x, y) = LAMBDA * (x, y) on the prime-order subgroup; k = k1 + k2*LAMBDA with
            # k2 = floor(k / LAMBDA), both below 2^127.  GLV_M = floor(2^256 / LAMBDA) for the reciprocal division.
            beta, lam = glv
            assert lam.bit_length() <= 127
            s.append("  static constexpr uint32_t GLV_BETA[%d] = {%s};" % (n, limbs(beta * R % p, n)))
            if fname.endswith("R29") and fname[:-3] in LAZY28:
                nz = (p.bit_length() + 15 + 27) // 28
                s.append("  static constexpr uint32_t GLV_BETAZ[%d] = {%s};" % (nz, limbs(beta * (1 << (28 * nz)) % p, nz, 28)))
            s.append("  static constexpr uint32_t GLV_LAMBDA[4] = {%s};" % limbs(lam, 4, 32))
            s.append("  static constexpr uint32_t GLV_M[5] = {%s};" % limbs((1 << 256) // lam, 5, 32))
        # #E / r (ark-*'s COFACTOR): G::rand and hash_to_g2 clear it (key generation, host side); tests/test_oracle.py
        # checks cofactor * r * P = O on random curve points
        cl = max(1, (cofactor.bit_length() + 31) // 32)
        s.append("  static constexpr int COFACTOR_BITS = %d;" % cofactor.bit_length())
        s.append("  static constexpr uint32_t COFACTOR[%d] = {%s};" % (cl, limbs(cofactor, cl, 32)))
        s.append("  static constexpr int DEG = %d;" % deg)
        s.append("  static constexpr bool A_IS_ZERO = %s;" % ("true" if not any(a) else "false"))
        # a = A_SMALL * u^A_SHIFT when exactly one coefficient is set and small (doubling uses an addition chain)
        nzc = [(i, v) for i, v in enumerate(a) if v]
        small = len(nzc) == 1 and nzc[0][1] < 64
        s.append("  static constexpr int A_SMALL = %d, A_SHIFT = %d;" % ((nzc[0][1], nzc[0][0]) if small else (0, 0)))
        s.append("  static constexpr int NR = %d;   // u^DEG = NR (small signed int)" % nr)
        for i in range(deg):
            s.append("  static constexpr uint32_t A%d[%d] = {%s};" % (i, n, limbs(mont(a[i], p, R), n)))
            s.append("  static constexpr uint32_t B%d[%d] = {%s};" % (i, n, limbs(mont(b[i], p, R), n)))
        s.append("};")
        return "\n".join(s)

    bls_x = 0x8508c00000000001
    bls_lambda = bls_x * bls_x - 1
    assert (bls_lambda * bls_lambda + bls_lambda + 1) % BLS_R == 0
    g = 2
    while pow(g, (BLS_Q - 1) // 3, BLS_Q) == 1:
        g += 1
    bls_beta = pow(g, 2 * (BLS_Q - 1) // 3, BLS_Q)   # the cube root of unity matching lambda = x^2 - 1 (tests check phi(G) = lambda G)
    # lambda^2 = -x^2 mod r is the eigenvalue of beta^2: the G1 membership test compares (beta^2 x, y) with -[x^2]P
    assert (bls_lambda * bls_lambda + bls_x * bls_x) % BLS_R == 0
    out.append(coeffs("BlsG1Cfg", fq_of("BlsFq"), 1, (0,), (1,), 0, glv=(bls_beta, bls_lambda),
                      endo=(1, bls_x, bls_beta * bls_beta % BLS_Q, 0), cofactor=COFACTORS["bls12_377"][0]))
    # psi(x, y) = (xi^((q-1)/3) conj(x), xi^((q-1)/2) conj(y)), xi = u (D-type twist, b' = b/xi); both constants lie in Fq.
    # Sound because q - x = r (x-1)^2/3 and gcd((x-1)^2/3, #E'(Fq2)/r) = 1 (checked in tests/test_oracle.py).
    def fq2_pow_u(e):   # u^e in Fq[u]/(u^2 + 5)
        r0, r1, b0, b1 = 1, 0, 0, 1
        while e:
            if e & 1:
                r0, r1 = (r0 * b0 - 5 * r1 * b1) % BLS_Q, (r0 * b1 + r1 * b0) % BLS_Q
            b0, b1 = (b0 * b0 - 5 * b1 * b1) % BLS_Q, 2 * b0 * b1 % BLS_Q
            e >>= 1
        return r0, r1
    psi_cx, psi_cy = fq2_pow_u((BLS_Q - 1) // 3), fq2_pow_u((BLS_Q - 1) // 2)
    assert psi_cx[1] == 0 and psi_cy[1] == 0
    out.append(coeffs("BlsG2Cfg", fq_of("BlsFq"), 2, (0, 0), (0, BLS_G2_B1), -5, endo=(2, bls_x, psi_cx[0], psi_cy[0]),
                      cofactor=COFACTORS["bls12_377"][1]))
    # BW6-761: both groups are j = 0 curves over Fq; one eigenvalue lambda (316 bits), the matching cube root of unity
    # per group (checked against the generators: phi(G) = lambda G)
    gq = 2
    while pow(gq, (BW6_Q - 1) // 3, BW6_Q) == 1:
        gq += 1
    bw6_w = pow(gq, (BW6_Q - 1) // 3, BW6_Q)
    hr = 2
    while pow(hr, (BLS_Q - 1) // 3, BLS_Q) == 1:
        hr += 1
    bw6_lam = pow(hr, 2 * (BLS_Q - 1) // 3, BLS_Q)
    if bw6_lam.bit_length() > 320:
        bw6_lam = bw6_lam * bw6_lam % BLS_Q
    out.append(coeffs("Bw6G1Cfg", fq_of("Bw6Fq"), 1, (0,), (BW6_Q - 1,), 0, glv2=(glv_beta_for(BW6_G1_XY, bw6_w, bw6_lam, BW6_Q), bw6_lam, BLS_Q),
                      cofactor=COFACTORS["bw6"][0]))
    out.append(coeffs("Bw6G2Cfg", fq_of("Bw6Fq"), 1, (0,), (4,), 0, glv2=(glv_beta_for(BW6_G2_XY, bw6_w, bw6_lam, BW6_Q), bw6_lam, BLS_Q),
                      cofactor=COFACTORS["bw6"][1]))
    out.append(coeffs("Mnt4G1Cfg", fq_of("Mnt4Fq"), 1, (2,), (MNT4_B1,), 0))
    out.append(coeffs("Mnt4G2Cfg", fq_of("Mnt4Fq"), 2, (26, 0), (0, MNT4_B1 * 13 % MNT_A), 13, cofactor=COFACTORS["mnt4_753"][1],
                      # quadratic twist by omega, omega^2 = u, u^2 = 13: psi_x = u^(1-q) = -1, psi_y = omega^(3(1-q)) = 13^((q-1)/4); t - 1 = q - r < 0
                      psi=(MNT_A - 1, pow(13, (MNT_A - 1) // 4, MNT_A), MNT_B - MNT_A, True, 24)))
    out.append(coeffs("Mnt6G1Cfg", fq_of("Mnt6Fq"), 1, (11,), (MNT6_B1,), 0))
    out.append(coeffs("Mnt6G2Cfg", fq_of("Mnt6Fq"), 3, (0, 0, 11), (MNT6_B1 * 11 % MNT_B, 0, 0), 11, cofactor=COFACTORS["mnt6_753"][1],
                      # omega^2 = u, u^3 = 11: psi_x = u^(1-q) = 11^(-(q-1)/3), psi_y = omega^(3(1-q)) = 11^(-(q-1)/2); t - 1 = q - r > 0
                      psi=(pow(pow(11, (MNT_B - 1) // 3, MNT_B), -1, MNT_B), pow(pow(11, (MNT_B - 1) // 2, MNT_B), -1, MNT_B), MNT_B - MNT_A, False, 24)))
    # ---- pairing tail (host): Fq^k = Fq[w]/(w^k - NU), untwist sign, Frobenius constant, hard exponent
    def cyclo(k, q):
        return {4: q * q + 1, 6: q * q - q + 1, 12: q ** 4 - q * q + 1}[k]

    def pairing_cfg(cname, fname, r, k, nu, sign, bls_x=None):
        global _LB
        p, n, R, _LB = fields[fname]
        assert (p - 1) % k == 0
        gamma1 = pow(nu % p, (p - 1) // k, p)
        hard = cyclo(k, p)
        assert hard % r == 0
        hard //= r
        hl = (hard.bit_length() + 31) // 32
        s = ["struct %s {" % cname]
        s.append("  static constexpr int K = %d;" % k)
        s.append("  static constexpr int NU = %d;        // w^K = NU" % nu)
        s.append("  static constexpr int SIGN = %d;      // untwist (x' w^(2 SIGN), y' w^(3 SIGN))" % sign)
        s.append("  static constexpr uint32_t GAMMA1[%d] = {%s};   // NU^((q-1)/K), Montgomery" % (n, limbs(gamma1 * R % p, n)))
        s.append("  static constexpr uint32_t NU_INV[%d] = {%s};   // 1/NU, Montgomery" % (n, limbs(pow(nu % p, -1, p) * R % p, n)))
        s.append("  static constexpr int HARD_BITS = %d;" % hard.bit_length())
        s.append("  static constexpr uint32_t HARD[%d] = {%s};     // Phi_K(q)/r" % (hl, limbs(hard, hl, 32)))
        # BLS12 family: Phi_12(q)/r = 1 + ((x - 1)^2 / 3) (x + q) (x^2 + q^2 - 1)  (x = 1 mod 3): the hard part of the final
        # exponentiation as one 125-bit and three 64-bit exponentiations with Frobenius maps in between — ~400 Fq^12 products
        # instead of the ~1900 of the bit-serial power (pairing_dev.hpp final_exp)
        s.append("  static constexpr int HARD_KIND = %d;   // 1: the BLS12 chain in x" % (1 if bls_x else 0))
        if bls_x:
            assert k == 12 and bls_x > 0 and (bls_x - 1) ** 2 % 3 == 0
            cc = (bls_x - 1) ** 2 // 3
            assert hard == 1 + cc * (bls_x + p) * (bls_x * bls_x + p * p - 1)
            cl = (cc.bit_length() + 31) // 32
            s.append("  static constexpr int HARD_C_BITS = %d;" % cc.bit_length())
            s.append("  static constexpr uint32_t HARD_C[%d] = {%s};   // (x - 1)^2 / 3" % (cl, limbs(cc, cl, 32)))
            s.append("  static constexpr int HARD_X_BITS = %d;" % bls_x.bit_length())
            s.append("  static constexpr uint32_t HARD_X[2] = {%s};   // x" % limbs(bls_x, 2, 32))
        else:
            s.append("  static constexpr int HARD_C_BITS = 0, HARD_X_BITS = 0;")
            s.append("  static constexpr uint32_t HARD_C[1] = {0}, HARD_X[1] = {0};")
        s.append("};")
        return "\n".join(s)

    out.append(pairing_cfg("BlsPairingCfg", fq_of("BlsFq"), BLS_R, 12, -5, +1, bls_x=0x8508c00000000001))
    out.append(pairing_cfg("Bw6PairingCfg", fq_of("Bw6Fq"), BLS_Q, 6, -4, -1))
    out.append(pairing_cfg("Mnt4PairingCfg", fq_of("Mnt4Fq"), MNT_B, 4, 13, -1))
    out.append(pairing_cfg("Mnt6PairingCfg", fq_of("Mnt6Fq"), MNT_A, 6, 11, -1))
    # ---- generators, arkworks uncompressed serialisation (x | y + sign flag): Phase1::initialization
    def ser(p, coords_x, coords_y):
        nb = (p.bit_length() + 7) // 8
        neg_y = tuple((-c) % p for c in coords_y)
        flag = 0x80 if tuple(reversed(coords_y)) > tuple(reversed(neg_y)) else 0
        b = bytearray()
        for c in coords_x:
            b += c.to_bytes(nb, "little")
        for c in coords_y:
            b += c.to_bytes(nb, "little")
        b[-1] |= flag
        return bytes(b)

    def gen_bytes(name, data):
        return "static constexpr unsigned char %s[%d] = {%s};" % (name, len(data), ", ".join(str(v) for v in data))

    BLS_G1 = ((0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef,),
              (0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6,))
    BLS_G2 = ((0x018480be71c785fec89630a2a3841d01c565f071203e50317ea501f557db6b9b71889f52bb53540274e3e48f7c005196,
               0x00ea6040e700403170dc5a51b1b140d5532777ee6651cecbe7223ece0799c9de5cf89984bff76fe6b26bfefa6ea16afe),
              (0x00690d665d446f7bd960736bcbb2efb4de03ed7274b49a58e458c282f832d204f2cf88886d8c7c2ef094094409fd4ddf,
               0x00f8169fd28355189e549da3151a70aa61ef11ac3d591bf12463b01acee304c24279b83f5e52270bd9a1cdd185eb8f93))
    BW6_G1 = ((0x01075b020ea190c8b277ce98a477beaee6a0cfb7551b27f0ee05c54b85f56fc779017ffac15520ac11dbfcd294c2e746a17a54ce47729b905bd71fa0c9ea097103758f9a280ca27f6750dd0356133e82055928aca6af603f4088f3af66e5b43d,),
              (0x0058b84e0a6fc574e6fd637b45cc2a420f952589884c9ec61a7348d2a2e573a3265909f1af7e0dbac5b8fa1771b5b806cc685d31717a4c55be3fb90b6fc2cdd49f9df141b3053253b2b08119cad0fb93ad1cb2be0b20d2a1bafc8f2db4e95363,))
    BW6_G2 = ((0x0110133241d9b816c852a82e69d660f9d61053aac5a7115f4c06201013890f6d26b41c5dab3da268734ec3f1f09feb58c5bbcae9ac70e7c7963317a300e1b6bace6948cb3cd208d700e96efbc2ad54b06410cf4fe1bf995ba830c194cd025f1c,),
              (0x0017c3357761369f8179eb10e4b6d2dc26b7cf9acec2181c81a78e2753ffe3160a1d86c80b95a59c94c97eb733293fef64f293dbd2c712b88906c170ffa823003ea96fcd504affc758aa2d3a3c5a02a591ec0594f9eac689eb70a16728c73b61,))
    MNT4_G1 = ((7790163481385331313124631546957228376128961350185262705123068027727518350362064426002432450801002268747950550964579198552865939244360469674540925037890082678099826733417900510086646711680891516503232107232083181010099241949569,),
               (6913648190367314284606685101150155872986263667483624713540251048208073654617802840433842931301128643140890502238233930290161632176167186761333725658542781350626799660920481723757654531036893265359076440986158843531053720994648,))
    MNT6_G1 = ((16364236387491689444759057944334173579070747473738339749093487337644739228935268157504218078126401066954815152892688541654726829424326599038522503517302466226143788988217410842672857564665527806044250003808514184274233938437290,),
               (4510127914410645922431074687553594593336087066778984214797709122300210966076979927285161950203037801392624582544098750667549188549761032654706830225743998064330900301346566408501390638273322467173741629353517809979540986561128,))
    out.append("struct Generators {")
    out.append("  " + gen_bytes("BLS_G1", ser(BLS_Q, *BLS_G1)))
    out.append("  " + gen_bytes("BLS_G2", ser(BLS_Q, *BLS_G2)))
    out.append("  " + gen_bytes("BW6_G1", ser(BW6_Q, *BW6_G1)))
    out.append("  " + gen_bytes("BW6_G2", ser(BW6_Q, *BW6_G2)))
    out.append("  " + gen_bytes("MNT4_G1", ser(MNT_A, *MNT4_G1)))
    out.append("  " + gen_bytes("MNT6_G1", ser(MNT_B, *MNT6_G1)))
    pinned = pinned_mnt_g2()
    out.append("  static constexpr bool HAVE_MNT_G2 = %s;   // pinned_generators.json (tools/pin_reference.sh) present at build time" % ("true" if pinned else "false"))
    out.append("  " + gen_bytes("MNT4_G2", pinned[0] if pinned else bytes(380)))
    out.append("  " + gen_bytes("MNT6_G2", pinned[1] if pinned else bytes(570)))
    out.append("};")
    out.append("}  // namespace ptau")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "curve_constants.inc")
    txt = "\n".join(out) + "\n"
    if not os.path.exists(path) or open(path).read() != txt:
        open(path, "w").write(txt)
    return path


if __name__ == "__main__":
    print(main())
