"""check_same_ratio: the host pairing tail of the product (C ABI) against the Python oracle pairing.
Runs on the CPU (the pairing tail is host code by design)."""
import random

import pytest

from conftest import CURVE_NAMES
from oracle.pyref import serde
from oracle.pyref.curves import CURVES
from oracle.pyref.pairing import Pairing


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_oracle_pairing_bilinear(name):
    c = CURVES[name]
    Pg = Pairing(c)
    rng = random.Random(3)
    a, b = rng.randrange(1, c.r), rng.randrange(1, c.r)
    e1 = Pg.pairing(Pg.G1.gen, Pg.G2.gen)
    assert e1 != Pg.K.one()
    assert Pg.pairing(Pg.G1.mul(Pg.G1.gen, a), Pg.G2.mul(Pg.G2.gen, b)) == Pg.K.pow(e1, a * b % c.r)
    assert Pg.K.pow(e1, c.r) == Pg.K.one()


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_check_same_ratio_matches_oracle(name, pkg):
    c = CURVES[name]
    Pg = Pairing(c)
    G1, G2 = Pg.G1, Pg.G2
    rng = random.Random(17)
    P = G1.mul(G1.gen, rng.randrange(1, c.r))
    Q = G2.mul(G2.gen, rng.randrange(1, c.r))
    s = rng.randrange(2, c.r)
    w1 = lambda pt: serde.write_point(G1, pt, False)
    w2 = lambda pt: serde.write_point(G2, pt, False)
    cases = [((P, G1.mul(P, s)), (Q, G2.mul(Q, s)), True),
             ((P, G1.mul(P, s)), (Q, G2.mul(Q, s + 1)), False),
             ((P, P), (Q, Q), True),
             ((P, G1.mul(P, s)), (G2.mul(Q, s), Q), False),
             ((P, None), (Q, Q), False)]
    for g1p, g2p, expect in cases:
        assert Pg.check_same_ratio(g1p, g2p) is expect
        got = pkg.phase1_cli.check_same_ratio(name, (w1(g1p[0]), w1(g1p[1])), (w2(g2p[0]), w2(g2p[1])))
        assert got is expect
    # non-canonical coordinate -> PTAU_E_POINT
    bad = bytearray(w1(P))
    bad[:c.fq_bytes] = c.q.to_bytes(c.fq_bytes, "little")
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.check_same_ratio(name, (bytes(bad), w1(P)), (w2(Q), w2(Q)))
    assert e.value.code == -3
