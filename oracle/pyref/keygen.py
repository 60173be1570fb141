"""derive_rng_from_seed / Phase1::key_generation / compute_g2_s on Python ints (TEST INFRASTRUCTURE — oracle only).

Restates, from memory of the un-vendored crates (PARITY UNPINNED — no reference-held vector exists for any of it):
  * setup_utils::derive_rng_from_seed — reference call site /root/reference/src/bin/contribute.rs:783-789:
    ChaCha20Rng::from_seed(first 32 bytes) of rand_chacha 0.3.1 (Cargo.lock:2927-2929); only the ChaCha20 block
    function itself is pinned (RFC 7539 / rand_chacha's own `test_chacha_true_values_a`, tests/test_keygen.py);
  * ark-ff 0.4.2 `Distribution<Fp>`: random u64 limbs ARE the Montgomery representation, top bits shaved, rejected
    while >= modulus;
  * ark-ec 0.4.2 `Distribution<Projective>`: x <- Fq::rand, greatest <- bool, cofactor cleared;
  * ark-ec `Affine::from_random_bytes`, ark-ff `from_random_bytes_with_flags`;
  * setup_utils::hash_to_g2 / compute_g2_s, phase1 `Phase1::key_generation` (inside phase1_cli::contribute,
    contribute.rs:811-823).
Independent of the C++ restatement in snark-setup-operator_amd/csrc/keygen.hip: the two must agree byte for byte.
"""
import hashlib

from . import serde
from .ec import Group

MASK32 = 0xFFFFFFFF


def _rotl(v, n):
    return ((v << n) | (v >> (32 - n))) & MASK32


def chacha20_block(key_words, counter):
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & MASK32, (counter >> 32) & MASK32, 0, 0]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & MASK32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & MASK32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = _rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(x, s)]


class ChaChaRng:
    """rand_chacha::ChaCha20Rng::from_seed: the keystream as little-endian u32 words, 64-bit block counter from 0."""

    def __init__(self, seed: bytes):
        assert len(seed) >= 32
        self.key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
        self.block, self.buf = 0, []

    def next_u32(self):
        if not self.buf:
            self.buf = chacha20_block(self.key, self.block)
            self.block += 1
        return self.buf.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)

    def gen_bool(self):
        return bool(self.next_u32() >> 31)

    def gen_u8(self):
        return self.next_u32() & 0xFF


def field_rand(rng: ChaChaRng, p: int) -> int:
    """Fp::rand -> CANONICAL value (the sampled limbs are the Montgomery form, R = 2^(64 N))."""
    n64 = (p.bit_length() + 63) // 64
    shave = 64 * n64 - p.bit_length()
    while True:
        raw = 0
        for i in range(n64):
            raw |= rng.next_u64() << (64 * i)
        raw &= (1 << (64 * n64 - shave)) - 1
        if raw < p:
            return raw * pow(1 << (64 * n64), -1, p) % p


def _cofactor(G: Group) -> int:
    c = G.curve
    x = 0x8508C00000000001
    if c.name == "bls12_377":
        return (x - 1) ** 2 // 3 if G.which == 1 else (x ** 8 - 4 * x ** 7 + 5 * x ** 6 - 4 * x ** 4 + 6 * x ** 3 - 4 * x ** 2 - 4 * x + 13) // 9
    if c.name == "bw6":
        base = 26642435879335816683987677701488073867751118270052650655942102502312977592501693353047140953112195348280268661194869
        return base + 7 if G.which == 1 else base
    if G.which == 1:
        return 1
    q, r = c.q, c.r
    t = q + 1 - r
    order = q * q + 1 + (t * t - 2 * q) if c.name == "mnt4_753" else q ** 3 + 1 + (t ** 3 - 3 * q * t)
    assert order % r == 0
    return order // r


def _point_from_x(G: Group, x, greatest: bool):
    F = G.F
    y = F.sqrt(F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b))
    if y is None:
        return None
    ny = F.neg(y)
    lo, hi = (y, ny) if F.cmp_key(y) <= F.cmp_key(ny) else (ny, y)
    return (x, hi if greatest else lo)


def group_rand(rng: ChaChaRng, G: Group):
    assert G.F.deg == 1
    while True:
        x = (field_rand(rng, G.curve.q),)
        greatest = rng.gen_bool()
        P = _point_from_x(G, x, greatest)
        if P is not None:
            return G.mul(P, _cofactor(G))


def from_random_bytes(G: Group, data: bytes):
    """Affine::from_random_bytes; returns ('none',) / ('inf',) / ('pt', P)."""
    n, d, q = G.curve.fq_bytes, G.F.deg, G.curve.q
    top_bits = q.bit_length() - 8 * (n - 1)
    coeffs, flags = [], 0
    for i in range(d):
        buf = bytearray(data[i * n:(i + 1) * n])
        if i == d - 1:
            flags = buf[-1] & 0xC0
        buf[-1] &= (1 << top_bits) - 1
        v = int.from_bytes(buf, "little")
        if v >= q:
            return ("none",)
        coeffs.append(v)
    x = tuple(coeffs)
    neg, inf = bool(flags & 0x80), bool(flags & 0x40)
    if neg and inf:
        return ("none",)
    if inf:
        return ("inf",) if not any(x) else ("none",)
    P = _point_from_x(G, x, not neg)
    return ("none",) if P is None else ("pt", P)


def hash_to_g2(G2: Group, digest: bytes):
    rng = ChaChaRng(digest[:32])
    cof = _cofactor(G2)
    while True:
        data = bytes(rng.gen_u8() for _ in range(G2.comp_size))
        r = from_random_bytes(G2, data)
        if r[0] != "pt":
            continue
        S = G2.mul(r[1], cof)
        if S is not None:
            return S


def compute_g2_s(curve, digest: bytes, g1_s, g1_s_x, personalization: int):
    G1, G2 = Group(curve, 1), Group(curve, 2)
    h = hashlib.blake2b(bytes([personalization]) + digest + serde.write_point(G1, g1_s, False) + serde.write_point(G1, g1_s_x, False),
                        digest_size=64).digest()
    return hash_to_g2(G2, h)


def key_generation(curve, seed: bytes, digest: bytes):
    """-> (tau, alpha, beta, public_key bytes, [g2_s x 3])."""
    G1, G2 = Group(curve, 1), Group(curve, 2)
    rng = ChaChaRng(seed[:32])
    sk = [field_rand(rng, curve.r) for _ in range(3)]
    g1s, g2s, bases = [], [], []
    for pers, x in enumerate(sk):
        s = group_rand(rng, G1)
        sx = G1.mul(s, x)
        h = compute_g2_s(curve, digest, s, sx, pers)
        g1s += [s, sx]
        g2s.append(G2.mul(h, x))
        bases.append(h)
    pk = serde.write_batch(G1, g1s, False) + serde.write_batch(G2, g2s, False)
    return sk[0], sk[1], sk[2], pk, bases
