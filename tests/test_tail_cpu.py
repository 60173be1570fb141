"""CPU tests of the host-side tail of the path: new_challenge (initialization) and the proof-of-knowledge /
before-after pairing checks (ptau_verify_pok) — no GPU involved."""
import pytest

from conftest import CURVE_NAMES
from oracle.pyref import phase1 as P
from oracle.pyref.curves import CURVES


def _oracle_generators(c):
    from oracle.pyref import serde
    from oracle.pyref.ec import Group
    G1, G2 = Group(c, 1), Group(c, 2)
    return serde.write_point(G1, G1.gen, False), serde.write_point(G2, G2.gen, False)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_new_challenge_matches_oracle(name, pkg, tmp_path):
    """Phase1::initialization.  BLS12-377 / BW6-761: built-in generators.  MNT4 / MNT6: the product holds no G2 generator
    (arkworks' constants are not recoverable offline) — the built-in form refuses, the caller-supplied form is exact."""
    c = CURVES[name]
    g1b, g2b = _oracle_generators(c)
    mnt = name.startswith("mnt")
    for idx in (0, 1):
        prm = pkg.Phase1Parameters.new_chunk(name, 0, idx, 4, 0, 2, 4)
        exp = P.initialization(P.Params(c, P.CHUNKED, idx, 4, 2, 4))
        assert pkg.phase1_cli.initialization_with_generators(prm, g1b, g2b) == exp
        if mnt:
            with pytest.raises(pkg.PtauError) as e:
                pkg.phase1_cli.initialization(prm)
            assert e.value.code == -10
        else:
            assert pkg.phase1_cli.initialization(prm) == exp
    assert pkg.phase1_cli.generator(name, 1) == g1b
    if mnt:
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.generator(name, 2)
        assert e.value.code == -10
        return
    assert pkg.phase1_cli.generator(name, 2) == g2b
    f, fh = str(tmp_path / "challenge"), str(tmp_path / "challenge.hash")
    prm = pkg.Phase1Parameters.new_full(name, 0, 2, 4)
    pkg.new_challenge(f, fh, prm)
    exp = P.initialization(P.Params(c, P.FULL, 0, 4, 2, 4))
    assert open(f, "rb").read() == exp
    assert open(fh, "rb").read() == P.calculate_hash(exp)
    assert exp[:64].hex().startswith("786a02f742015903c6c6fd852552d272")   # blank hash (new_setup.rs:200-201 context)
    with pytest.raises(pkg.PtauError):      # create_new semantics
        pkg.new_challenge(f, fh, prm)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_verify_pok(name, pkg, golden):
    """The golden responses carry a structurally valid public key (oracle.synthetic_public_key); its g2_s base
    points are supplied the way the Rust host would supply compute_g2_s."""
    doc = golden[name]
    c = CURVES[name]
    for ch in doc["chunks"]:
        p = P.Params(c, P.CHUNKED, ch["chunk_index"], doc["chunk_size"], doc["power"], doc["batch_size"])
        prm = pkg.Phase1Parameters.new_chunk(name, 0, ch["chunk_index"], doc["chunk_size"], 0, doc["power"], doc["batch_size"])
        challenge, response = bytes.fromhex(ch["challenge"]), bytes.fromhex(ch["response"])
        g2_s = P.synthetic_g2_s(p, P.calculate_hash(challenge))
        pkg.phase1_cli.verify_pok(challenge, response, g2_s, prm)
        if ch["chunk_index"] == 0:
            # a response computed with a DIFFERENT tau than the one proven in the public key must be rejected
            tau, alpha, beta = (int(ch[k], 16) for k in ("tau", "alpha", "beta"))
            body = P.computation(p, challenge, (tau + 1) % c.r, alpha, beta)
            forged = response[:64] + body + response[64 + len(body):]
            with pytest.raises(pkg.PtauError) as e:
                pkg.phase1_cli.verify_pok(challenge, forged, g2_s, prm)
            assert e.value.code == -6
        # wrong g2_s (swap tau and alpha bases) breaks the proof of knowledge
        u2 = c.uncomp_size(2)
        swapped = g2_s[u2:2 * u2] + g2_s[:u2] + g2_s[2 * u2:]
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.verify_pok(challenge, response, swapped, prm)
        assert e.value.code == -6
