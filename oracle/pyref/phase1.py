"""Chunked powers-of-tau phase 1 on Python ints (TEST INFRASTRUCTURE — oracle only; PARITY UNPINNED).

Restates `phase1::Phase1Parameters`, `Phase1::{initialization, computation, decompress,
aggregation}` and the `phase1_cli::{new_challenge, contribute, combine}` drivers of
nimiq/snark-setup@bd530da (un-vendored; Cargo.lock:2603-2605,2624-2626) from SURVEY.md §8(a) /
App. A.  Reference call sites:
  * parameters      /root/reference/src/utils.rs:326-352, src/bin/new_setup.rs:265-277
  * contribute      /root/reference/src/bin/contribute.rs:811-823
  * new_challenge   /root/reference/src/bin/new_setup.rs:105-109
  * combine         /root/reference/src/bin/control.rs:564-568
The private key (tau, alpha, beta) and the serialized public key are INPUTS here: key
generation depends on arkworks' `rand` internals and stays on the Rust host.
"""

import hashlib
from dataclasses import dataclass

from .curves import Curve
from .ec import Group
from . import serde

HASH_SIZE = 64
CHUNKED, FULL = 0, 1
GROTH16, MARLIN = 0, 1


def calculate_hash(data: bytes) -> bytes:
    """setup_utils::calculate_hash — BLAKE2b-512, unkeyed (verify_transcript.rs:344-345)."""
    return hashlib.blake2b(data, digest_size=64).digest()


BLANK_HASH = calculate_hash(b"")


@dataclass(frozen=True)
class Params:
    """Phase1Parameters::{new_chunk,new_full} (utils.rs:326-352)."""
    curve: Curve
    mode: int            # CHUNKED / FULL
    chunk_index: int
    chunk_size: int
    power: int
    batch_size: int
    proving_system: int = GROTH16

    def __post_init__(self):
        if self.proving_system != GROTH16:
            raise ValueError("only groth16 geometry is implemented")

    @property
    def powers_length(self):
        return 1 << self.power

    @property
    def powers_g1_length(self):
        return (self.powers_length << 1) - 1

    @property
    def start(self):
        return self.chunk_index * self.chunk_size if self.mode == CHUNKED else 0

    @property
    def g1_chunk_size(self):
        if self.mode == FULL:
            return self.powers_g1_length
        return max(0, min(self.chunk_size, self.powers_g1_length - self.start))

    @property
    def other_chunk_size(self):
        if self.mode == FULL:
            return self.powers_length
        return max(0, min(self.chunk_size, self.powers_length - self.start))

    def _sizes(self, compressed: bool):
        c = self.curve
        g1 = c.comp_size(1) if compressed else c.uncomp_size(1)
        g2 = c.comp_size(2) if compressed else c.uncomp_size(2)
        return g1, g2

    @property
    def public_key_size(self):
        g1, g2 = self._sizes(False)
        return 6 * g1 + 3 * g2

    @property
    def accumulator_size(self):
        g1, g2 = self._sizes(False)
        return HASH_SIZE + self.g1_chunk_size * g1 + self.other_chunk_size * (g2 + 2 * g1) + g2

    @property
    def contribution_size(self):
        g1, g2 = self._sizes(True)
        return (HASH_SIZE + self.g1_chunk_size * g1 + self.other_chunk_size * (g2 + 2 * g1) + g2
                + self.public_key_size)

    def num_chunks(self):
        return -(-self.powers_g1_length // self.chunk_size)

    def layout(self, compressed: bool):
        """Byte ranges [(name, group, offset, count)] after the 64-byte hash:
        tauG1 | tauG2 | alphaG1 | betaG1 | betaG2."""
        g1, g2 = self._sizes(compressed)
        n1, n2 = self.g1_chunk_size, self.other_chunk_size
        off = HASH_SIZE
        out = []
        for name, grp, cnt in (("tau_g1", 1, n1), ("tau_g2", 2, n2), ("alpha_g1", 1, n2),
                               ("beta_g1", 1, n2), ("beta_g2", 2, 1)):
            out.append((name, grp, off, cnt))
            off += cnt * (g1 if grp == 1 else g2)
        return out, off


def initialization(p: Params) -> bytes:
    """Phase1::initialization via phase1_cli::new_challenge (new_setup.rs:105-109):
    blank hash, then every element = the group generator, uncompressed."""
    G1, G2 = Group(p.curve, 1), Group(p.curve, 2)
    g1b = serde.write_point(G1, G1.gen, False)
    g2b = serde.write_point(G2, G2.gen, False)
    n1, n2 = p.g1_chunk_size, p.other_chunk_size
    out = BLANK_HASH + g1b * n1 + g2b * n2 + g1b * n2 + g1b * n2 + g2b
    assert len(out) == p.accumulator_size
    return out


def apply_powers(G: Group, data: bytes, compressed_in: bool, compressed_out: bool,
                 start: int, tau: int, coeff):
    """setup-utils `apply_powers` = read_batch -> generate_powers_of_tau -> batch_exp ->
    write_batch: out[i] = (coeff * tau^(start+i)) * in[i]."""
    r = G.r
    pts = serde.read_batch(G, data, compressed_in)
    out = []
    e = pow(tau, start, r)
    for P in pts:
        k = e if coeff is None else e * coeff % r
        out.append(G.mul(P, k))
        e = e * tau % r
    return serde.write_batch(G, out, compressed_out)


def computation(p: Params, inp: bytes, tau: int, alpha: int, beta: int,
                compressed_in: bool = False, compressed_out: bool = True) -> bytes:
    """Phase1::computation (inside phase1_cli::contribute, contribute.rs:811-823).
    Returns the element region only (no hash, no public key)."""
    G1, G2 = Group(p.curve, 1), Group(p.curve, 2)
    lay_in, end_in = p.layout(compressed_in)
    if compressed_in:
        assert len(inp) >= end_in
    else:
        assert len(inp) == p.accumulator_size == end_in
    out = b""
    s = p.start
    for (name, grp, off, cnt) in lay_in:
        G = G1 if grp == 1 else G2
        sz = G.comp_size if compressed_in else G.uncomp_size
        seg = inp[off:off + cnt * sz]
        if name == "beta_g2":
            pt = serde.read_point(G, seg, compressed_in)
            out += serde.write_point(G, G.mul(pt, beta % G.r), compressed_out)
            continue
        coeff = {"tau_g1": None, "tau_g2": None, "alpha_g1": alpha, "beta_g1": beta}[name]
        out += apply_powers(G, seg, compressed_in, compressed_out, s, tau, coeff)
    return out


def contribute(p: Params, challenge: bytes, tau: int, alpha: int, beta: int, public_key: bytes):
    """phase1_cli::contribute (contribute.rs:811-823) with the key pair as input.
    Returns (response, challenge_hash, response_hash)."""
    assert len(challenge) == p.accumulator_size
    assert len(public_key) == p.public_key_size
    ch = calculate_hash(challenge)
    body = computation(p, challenge, tau, alpha, beta)
    response = ch + body + public_key
    assert len(response) == p.contribution_size
    return response, ch, calculate_hash(response)


def decompress(p: Params, response: bytes) -> bytes:
    """Phase1::decompress (inside transform_pok_and_correctness, contribute.rs:968-986):
    compressed element region of a response -> uncompressed region of the next challenge.
    new_challenge = hash(response) || decompress(response)."""
    G1, G2 = Group(p.curve, 1), Group(p.curve, 2)
    lay, _ = p.layout(True)
    out = b""
    for (name, grp, off, cnt) in lay:
        G = G1 if grp == 1 else G2
        pts = serde.read_batch(G, response[off:off + cnt * G.comp_size], True)
        out += serde.write_batch(G, pts, False)
    return out


def new_challenge_from_response(p: Params, response: bytes) -> bytes:
    nc = calculate_hash(response) + decompress(p, response)
    assert len(nc) == p.accumulator_size
    return nc


def synthetic_public_key(p: Params, tau: int, alpha: int, beta: int, digest: bytes) -> bytes:
    """Stand-in for Phase1::key_generation's PublicKey: structurally valid
    (g1_s, x*g1_s pairs and x*g2_s), but the base points come from a BLAKE2b-derived scalar
    times the generator, NOT arkworks' `rand` — byte-compatibility with arkworks keygen
    is out of scope (keys are inputs at the boundary)."""
    G1, G2 = Group(p.curve, 1), Group(p.curve, 2)
    r = p.curve.r
    g1s, g2s = [], []
    for pers, x in enumerate((tau, alpha, beta)):
        k1 = int.from_bytes(hashlib.blake2b(bytes([pers]) + digest + b"g1", digest_size=64).digest(), "little") % r
        k2 = int.from_bytes(hashlib.blake2b(bytes([pers]) + digest + b"g2", digest_size=64).digest(), "little") % r
        s1 = G1.mul(G1.gen, k1 or 1)
        g1s += [s1, G1.mul(s1, x)]
        g2s.append(G2.mul(G2.mul(G2.gen, k2 or 1), x))
    return serde.write_batch(G1, g1s, False) + serde.write_batch(G2, g2s, False)


def synthetic_g2_s(p: Params, digest: bytes) -> bytes:
    """The three g2_s base points matching synthetic_public_key (stand-in for compute_g2_s / hash_to_g2)."""
    G2 = Group(p.curve, 2)
    out = []
    for pers in range(3):
        k2 = int.from_bytes(hashlib.blake2b(bytes([pers]) + digest + b"g2", digest_size=64).digest(), "little") % p.curve.r
        out.append(G2.mul(G2.gen, k2 or 1))
    return serde.write_batch(G2, out, False)


def aggregation(chunk_params, responses, full: Params) -> bytes:
    """Phase1::aggregation via phase1_cli::combine (control.rs:564-568; verify_transcript.rs:603-607):
    concatenate the per-chunk vectors of compressed responses into one full, UNCOMPRESSED
    accumulator (blank hash header region left as in the first chunk's decompressed form)."""
    G1, G2 = Group(full.curve, 1), Group(full.curve, 2)
    vecs = {"tau_g1": b"", "tau_g2": b"", "alpha_g1": b"", "beta_g1": b"", "beta_g2": b""}
    for cp, resp in zip(chunk_params, responses):
        lay, _ = cp.layout(True)
        for (name, grp, off, cnt) in lay:
            G = G1 if grp == 1 else G2
            if name == "beta_g2" and cp.chunk_index != 0:
                continue
            pts = serde.read_batch(G, resp[off:off + cnt * G.comp_size], True)
            vecs[name] += serde.write_batch(G, pts, False)
    out = bytes(HASH_SIZE) + vecs["tau_g1"] + vecs["tau_g2"] + vecs["alpha_g1"] + vecs["beta_g1"] + vecs["beta_g2"]
    assert len(out) == full.accumulator_size
    return out


def power_pairs(G: Group, pts, rng):
    """setup_utils::power_pairs / merge_pairs: random rho_i, (sum rho_i v_i, sum rho_i v_{i+1})."""
    a = b = None
    for i in range(len(pts) - 1):
        rho = rng.randrange(1, G.r)
        a = G.add(a, G.mul(pts[i], rho))
        b = G.add(b, G.mul(pts[i + 1], rho))
    return a, b


def verify_chunk_ratios(p: Params, response: bytes, tau_g2_check=None, rng=None) -> bool:
    """check_power_ratios part of Phase1::verification (contribute.rs:968-986 with ratio_check): every vector
    of the response must be geometric with the ratio of (tauG2[j], tauG2[j+1]) resp. (tauG1[j], tauG1[j+1])."""
    import random
    from .pairing import Pairing
    rng = rng or random.Random()
    Pg = Pairing(p.curve)
    G1, G2 = Pg.G1, Pg.G2
    vecs = {}
    for (name, grp, off, cnt) in p.layout(True)[0]:
        G = G1 if grp == 1 else G2
        vecs[name] = serde.read_batch(G, response[off:off + cnt * G.comp_size], True)
    g2_pair = tuple(vecs["tau_g2"][:2]) if len(vecs["tau_g2"]) >= 2 else tau_g2_check
    assert g2_pair is not None
    for name in ("tau_g1", "alpha_g1", "beta_g1"):
        if len(vecs[name]) >= 2 and not Pg.check_same_ratio(power_pairs(G1, vecs[name], rng), g2_pair):
            return False
    if len(vecs["tau_g2"]) >= 2 and len(vecs["tau_g1"]) >= 2:
        if not Pg.check_same_ratio(tuple(vecs["tau_g1"][:2]), power_pairs(G2, vecs["tau_g2"], rng)):
            return False
    return True
