"""CPU tests of the host-side key path: derive_rng_from_seed's ChaCha20 stream against published
vectors, Phase1::key_generation / compute_g2_s against the independent Python restatement (oracle/pyref/keygen.py), the
proofs of knowledge of a generated key, and the cofactors the sampling clears.  No GPU involved.

Only the ChaCha20 block function is pinned to published data; everything above it restates arkworks' sampling from memory
(PARITY UNPINNED — see keygen.hip) and these tests check the two restatements against each other and the keys for validity.
"""
import hashlib
import random

import pytest

from conftest import CURVE_NAMES
from oracle.pyref import keygen as K
from oracle.pyref import phase1 as P
from oracle.pyref import serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group

# RFC 7539 appendix A.1 test vectors #1 and #2 (all-zero key and nonce, block counter 0 and 1); the first sixteen words
# are also rand_chacha 0.3.1's own test_chacha_true_values_a (ChaCha20Rng::from_seed([0; 32]))
ZERO_KEY_BLOCK0 = bytes.fromhex(
    "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
    "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
ZERO_KEY_BLOCK1 = bytes.fromhex(
    "9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed"
    "29b721769ce64e43d57133b074d839d531ed1f28510afb45ace10a1f4b794d6f")
# RFC 7539 appendix A.1 test vector #3: key = 00..00 01, nonce 0, block counter 1
KEY1_BLOCK1 = bytes.fromhex(
    "3aeb5224ecf849929b9d828db1ced4dd832025e8018b8160b82284f3c949aa5a"
    "8eca00bbb4a73bdad192b5c42f73f2fd4e273644c8b36125a64addeb006c13a0")


def _words(b):
    return [int.from_bytes(b[i:i + 4], "little") for i in range(0, len(b), 4)]


def test_chacha20_published_vectors(pkg):
    zero = bytes(32)
    assert pkg.phase1_cli.derive_rng_words(zero, 32) == _words(ZERO_KEY_BLOCK0 + ZERO_KEY_BLOCK1)
    assert pkg.phase1_cli.derive_rng_words(zero, 16, word_offset=16) == _words(ZERO_KEY_BLOCK1)
    assert pkg.phase1_cli.derive_rng_words(zero, 5, word_offset=14) == _words(ZERO_KEY_BLOCK0 + ZERO_KEY_BLOCK1)[14:19]
    key1 = bytes(31) + b"\x01"
    assert pkg.phase1_cli.derive_rng_words(key1, 16, word_offset=16) == _words(KEY1_BLOCK1)
    # the oracle's generator is the same stream
    rng = K.ChaChaRng(zero)
    assert [rng.next_u32() for _ in range(32)] == _words(ZERO_KEY_BLOCK0 + ZERO_KEY_BLOCK1)
    assert K.chacha20_block([0] * 7 + [0x01000000], 1) == _words(KEY1_BLOCK1)
    # only the first 32 seed bytes count (derive_rng_from_seed)
    seed = hashlib.blake2b(b"seed", digest_size=64).digest()
    assert pkg.phase1_cli.derive_rng_words(seed, 8) == pkg.phase1_cli.derive_rng_words(seed[:32], 8)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_cofactors(name):
    """cofactor * r annihilates random curve points of both groups (the constants keygen clears; gen_constants.py)."""
    c = CURVES[name]
    rng = random.Random(11)
    for which in (1, 2):
        G = Group(c, which)
        cof = K._cofactor(G)
        found = 0
        while found < 2:
            x = tuple(rng.randrange(c.q) for _ in range(G.F.deg))
            pt = K._point_from_x(G, x, True)
            if pt is None:
                continue
            found += 1
            assert G.mul(G.mul(pt, cof), c.r) is None
            if cof > 1:
                assert G.mul(pt, c.r) is not None or G.mul(pt, cof) is None   # a random point is (almost surely) outside the subgroup


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_key_generation_matches_python_restatement(name, pkg):
    c = CURVES[name]
    seed = hashlib.blake2b(b"keygen seed " + name.encode(), digest_size=64).digest()
    digest = hashlib.blake2b(b"challenge digest " + name.encode(), digest_size=64).digest()
    key = pkg.phase1_cli.key_generation(name, seed, digest)
    tau, alpha, beta, pk, bases = K.key_generation(c, seed, digest)
    assert (key.tau, key.alpha, key.beta) == (tau, alpha, beta)
    assert key.public_key == pk
    G2 = Group(c, 2)
    g2_s = pkg.phase1_cli.compute_g2_s(name, digest, key.public_key)
    assert g2_s == serde.write_batch(G2, bases, False)
    # every base point is in the prime-order subgroup and the three proofs of knowledge hold
    for b in bases:
        assert G2.in_subgroup(b)
    u1, u2 = c.uncomp_size(1), c.uncomp_size(2)
    for x in range(3):
        g1_pair = (pk[2 * x * u1:(2 * x + 1) * u1], pk[(2 * x + 1) * u1:(2 * x + 2) * u1])
        g2_pair = (g2_s[x * u2:(x + 1) * u2], pk[6 * u1 + x * u2:6 * u1 + (x + 1) * u2])
        assert pkg.phase1_cli.check_same_ratio(name, g1_pair, g2_pair)
    # another digest moves the G2 side only; another seed moves everything
    key2 = pkg.phase1_cli.key_generation(name, seed, hashlib.blake2b(b"other", digest_size=64).digest())
    assert (key2.tau, key2.alpha, key2.beta) == (tau, alpha, beta)
    assert key2.public_key[:6 * u1] == pk[:6 * u1] and key2.public_key[6 * u1:] != pk[6 * u1:]
    key3 = pkg.phase1_cli.key_generation(name, bytes(reversed(seed)), digest)
    assert key3.tau != tau and key3.public_key != pk


def test_verify_pok_derives_g2_s(pkg):
    """ptau_verify_pok with g2_s = NULL derives the compute_g2_s points itself (BLS12-377, chunk 1: the proofs of knowledge
    alone), accepts the generated key and rejects a key generated for another challenge digest."""
    name = "bls12_377"
    c = CURVES[name]
    p = P.Params(c, P.CHUNKED, 1, 4, 2, 4)
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 1, 4, 0, 2, 4)
    challenge = P.initialization(p)
    digest = P.calculate_hash(challenge)
    seed = bytes(range(64))
    key = pkg.phase1_cli.key_generation(name, seed, digest)
    body = P.computation(p, challenge, key.tau, key.alpha, key.beta)
    response = digest + body + key.public_key
    assert len(response) == prm.contribution_size
    pkg.phase1_cli.verify_pok(challenge, response, None, prm)
    wrong = pkg.phase1_cli.key_generation(name, seed, bytes(64))
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verify_pok(challenge, digest + body + wrong.public_key, None, prm)
    assert e.value.code == -6
