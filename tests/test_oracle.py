"""CPU tests of the oracle itself: constants, the two independent restatements against each other,
and both against the committed golden fixtures.  (Parity with arkworks is UNPINNED — SURVEY §8c.)"""
import hashlib
import random

import pytest

from conftest import CURVE_NAMES
from oracle.pyref import phase1 as P, serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group


def _is_probable_prime(n, rounds=12):
    if n < 4:
        return n in (2, 3)
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    rng = random.Random(1)
    for _ in range(rounds):
        a = rng.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_curve_constants(name):
    c = CURVES[name]
    assert _is_probable_prime(c.q) and _is_probable_prime(c.r)
    for w in (1, 2):
        G = Group(c, w)
        assert G.on_curve(G.gen)
        assert G.in_subgroup(G.gen)
        assert G.mul(G.gen, 1) == G.gen


def test_bls12_377_parametrisation():
    x = 0x8508c00000000001
    c = CURVES["bls12_377"]
    assert c.r == x ** 4 - x ** 2 + 1
    assert c.q == (x - 1) ** 2 * c.r // 3 + x
    assert CURVES["bw6"].r == c.q            # BW6-761 is built over BLS12-377's base field
    assert CURVES["mnt4_753"].r == CURVES["mnt6_753"].q and CURVES["mnt6_753"].r == CURVES["mnt4_753"].q


def test_sizes_match_survey():
    # SURVEY.md §8(a) row a1
    p = P.Params(CURVES["bls12_377"], P.CHUNKED, 0, 1 << 20, 20, 1 << 20)
    assert (p.accumulator_size, p.contribution_size) == (503316736, 251659552)
    p = P.Params(CURVES["bls12_377"], P.CHUNKED, 1, 1 << 20, 20, 1 << 20)
    assert (p.accumulator_size, p.contribution_size, p.g1_chunk_size, p.other_chunk_size) == (100663456, 50332912, (1 << 20) - 1, 0)
    p = P.Params(CURVES["bw6"], P.CHUNKED, 0, 1 << 18, 20, 1 << 18)
    assert (p.accumulator_size, p.contribution_size, p.num_chunks()) == (201326848, 100665184, 8)
    p = P.Params(CURVES["mnt4_753"], P.CHUNKED, 0, 1 << 16, 18, 1 << 16)
    assert (p.accumulator_size, p.contribution_size) == (62259644, 31132134)
    p = P.Params(CURVES["mnt6_753"], P.CHUNKED, 0, 1 << 16, 18, 1 << 16)
    assert (p.accumulator_size, p.contribution_size) == (74711674, 37358719)
    p = P.Params(CURVES["bls12_377"], P.FULL, 0, 1 << 20, 20, 1 << 20)
    assert p.accumulator_size == 603979936
    assert [CURVES[n].g1 and P.Params(CURVES[n], P.FULL, 0, 4, 2, 4).public_key_size for n in CURVE_NAMES] == [1152, 1728, 2280, 2850]


def test_blank_hash_and_blake2b(cref):
    # RFC 7693 "abc" vector and the empty-input hash the reference's initial challenge carries
    assert P.BLANK_HASH.hex().startswith("786a02f742015903c6c6fd852552d272")
    abc = "ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d17d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923"
    assert cref.blake2b512(b"abc").hex() == abc
    rng = random.Random(3)
    for n in (0, 1, 63, 64, 127, 128, 129, 255, 256, 257, 1000, 4096 + 17):
        d = bytes(rng.randrange(256) for _ in range(n))
        assert cref.blake2b512(d) == hashlib.blake2b(d).digest()


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_serde_roundtrip_and_sign_flag(name):
    c = CURVES[name]
    rng = random.Random(5)
    for w in (1, 2):
        G = Group(c, w)
        pts = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(3)] + [None]
        for Pt in pts:
            for comp in (True, False):
                b = serde.write_point(G, Pt, comp)
                assert len(b) == (G.comp_size if comp else G.uncomp_size)
                assert serde.read_point(G, b, comp) == Pt
            if Pt is not None:
                bn = serde.write_point(G, G.neg(Pt), True)
                bp = serde.write_point(G, Pt, True)
                assert bn[:-1] == bp[:-1] and (bn[-1] ^ bp[-1]) == 0x80   # sign-flag involution
        assert serde.write_point(G, None, True)[-1] == 0x40


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_golden_vs_c_oracle(name, golden, cref):
    """The C restatement reproduces every committed fixture (made by the Python restatement)."""
    doc = golden[name]
    c = CURVES[name]
    for ch in doc["chunks"]:
        tau, alpha, beta = (int(ch[k], 16) for k in ("tau", "alpha", "beta"))
        challenge, response = bytes.fromhex(ch["challenge"]), bytes.fromhex(ch["response"])
        prm = cref.params(name, 0, ch["chunk_index"], doc["chunk_size"], doc["power"])
        assert prm["accumulator_size"] == ch["accumulator_size"] == len(challenge)
        assert prm["contribution_size"] == ch["contribution_size"] == len(response)
        assert prm["g1"] == ch["g1_chunk_size"] and prm["other"] == ch["other_chunk_size"]
        assert cref.blake2b512(challenge).hex() == ch["challenge_hash"] == response[:64].hex()
        assert cref.blake2b512(response).hex() == ch["response_hash"]
        out = cref.computation(name, 0, ch["chunk_index"], doc["chunk_size"], doc["power"], challenge, tau, alpha, beta, nthreads=2)
        body_end = ch["contribution_size"] - ch["public_key_size"]
        assert out[64:body_end] == response[64:body_end]
        # decompress(response) reproduces the next challenge
        p = P.Params(c, P.CHUNKED, ch["chunk_index"], doc["chunk_size"], doc["power"], doc["batch_size"])
        nc = b""
        for (nm, grp, off, cnt) in p.layout(True)[0]:
            sz = cref.point_size(name, grp, True)
            nc += cref.decompress(name, grp, response[off:off + cnt * sz], cnt)
        assert cref.blake2b512(cref.blake2b512(response) + nc).hex() == ch["new_challenge_hash"]


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_update_invariant(name, golden):
    """new[i] = tau^(start+i) * old[i], checked with the Python group law."""
    doc = golden[name]
    c = CURVES[name]
    ch = doc["chunks"][1]          # tauG1-only chunk: start = chunk_size
    p = P.Params(c, P.CHUNKED, 1, doc["chunk_size"], doc["power"], doc["batch_size"])
    G = Group(c, 1)
    tau = int(ch["tau"], 16)
    old = serde.read_batch(G, bytes.fromhex(ch["challenge"])[64:64 + p.g1_chunk_size * G.uncomp_size], False)
    new = serde.read_batch(G, bytes.fromhex(ch["response"])[64:64 + p.g1_chunk_size * G.comp_size], True)
    for i, (o, n) in enumerate(zip(old, new)):
        assert G.mul(o, pow(tau, p.start + i, c.r)) == n


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_random_cross_check(name, cref):
    """Python vs C restatements on random points / scalars, both groups, incl. infinity and edge scalars."""
    c = CURVES[name]
    rng = random.Random(11)
    for w in (1, 2):
        G = Group(c, w)
        pts = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(3)] + [None]
        data = serde.write_batch(G, pts, False)
        for tau, coeff, start in ((rng.randrange(c.r), rng.randrange(c.r), rng.randrange(1 << 30)), (1, None, 0),
                                  (c.r - 1, 1, 3), (0, None, 0)):
            exp = P.apply_powers(G, data, False, True, start, tau, coeff)
            assert cref.apply_powers(name, w, data, len(pts), start, tau, coeff) == exp
