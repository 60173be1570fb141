"""GPU parity tests of the verify path (transform_pok_and_correctness element checks + decompress),
through the C ABI, against the oracle and the golden fixtures."""
import random

import pytest

from conftest import CURVE_NAMES
from oracle.pyref import phase1 as P, serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group

pytestmark = pytest.mark.gpu


def _params(pkg, name, doc, idx):
    return pkg.Phase1Parameters.new_chunk(name, 0, idx, doc["chunk_size"], 0, doc["power"], doc["batch_size"])


def _off_subgroup_point(G, rng):
    """A curve point outside the prime-order subgroup (exists when the cofactor is > 1)."""
    F = G.F
    while True:
        x = tuple(rng.randrange(G.curve.q) for _ in range(F.deg))
        rhs = F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b)
        y = F.sqrt(rhs)
        if y is not None and not G.in_subgroup((x, y)):
            return (x, y)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_verify_golden_chunks(name, pkg, golden, cref):
    doc = golden[name]
    for ch in doc["chunks"]:
        prm = _params(pkg, name, doc, ch["chunk_index"])
        challenge, response = bytes.fromhex(ch["challenge"]), bytes.fromhex(ch["response"])
        nc = pkg.phase1_cli.verification_and_decompress(challenge, response, prm)
        assert pkg.calculate_hash(nc).hex() == ch["new_challenge_hash"]
        assert nc[:64] == bytes.fromhex(ch["response_hash"])
        # the oracle's decompress gives the same bytes
        p = P.Params(CURVES[name], P.CHUNKED, ch["chunk_index"], doc["chunk_size"], doc["power"], doc["batch_size"])
        assert nc == P.new_challenge_from_response(p, response)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_verify_rejects_bad_inputs(name, pkg, golden):
    doc = golden[name]
    ch = doc["chunks"][0]
    c = CURVES[name]
    prm = _params(pkg, name, doc, 0)
    challenge, response = bytes.fromhex(ch["challenge"]), bytearray.fromhex(ch["response"])
    # 1. broken hash chain
    bad_ch = bytearray(challenge)
    bad_ch[100] ^= 1
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(bytes(bad_ch), bytes(response), prm)
    assert e.value.code == -7
    # 2. point at infinity in the response (OnlyNonZero is the verifier's default output check)
    G1 = Group(c, 1)
    r2 = bytearray(response)
    r2[64:64 + G1.comp_size] = serde.write_point(G1, None, True)
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(challenge, bytes(r2), prm)
    assert e.value.code == -4
    # 3. element outside the prime-order subgroup (G1 where the cofactor allows it, else G2)
    rng = random.Random(99)
    grp = 2 if name.startswith("mnt") else 1
    G = Group(c, grp)
    bad_pt = serde.write_point(G, _off_subgroup_point(G, rng), True)
    p = P.Params(c, P.CHUNKED, 0, doc["chunk_size"], doc["power"], doc["batch_size"])
    off = {nm: o for (nm, g, o, cnt) in p.layout(True)[0]}["tau_g1" if grp == 1 else "tau_g2"]
    r3 = bytearray(response)
    r3[off + G.comp_size:off + 2 * G.comp_size] = bad_pt
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(challenge, bytes(r3), prm)
    assert e.value.code == -5
    # ... and accepted when the subgroup check is switched off (SubgroupCheckMode::No)
    nc = pkg.phase1_cli.verification_and_decompress(challenge, bytes(r3), prm, subgroup_check_mode=pkg.SubgroupCheckMode.No)
    assert len(nc) == prm.accumulator_size
    # 4. wrong sizes
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(challenge + b"\0", bytes(response), prm)
    assert e.value.code == -2


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_subgroup_check_api(name, pkg):
    c = CURVES[name]
    rng = random.Random(5)
    for grp in (1, 2):
        G = Group(c, grp)
        good = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(5)] + [None]
        for comp in (False, True):
            pkg.phase1_cli.subgroup_check(name, grp, serde.write_batch(G, good, comp), len(good), compressed_in=comp)
        if not (name.startswith("mnt") and grp == 1):
            bad = good[:3] + [_off_subgroup_point(G, rng)] + good[3:]
            for comp in (False, True):
                with pytest.raises(pkg.PtauError) as e:
                    pkg.phase1_cli.subgroup_check(name, grp, serde.write_batch(G, bad, comp), len(bad), compressed_in=comp)
                assert e.value.code == -5
        # uncompressed point that is not on the curve: y + 1
        x, y = good[0]
        off = serde.write_point(G, (x, G.F.add(y, G.F.one())), False)
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.subgroup_check(name, grp, off, 1, compressed_in=False)
        assert e.value.code == -3


def test_transform_pok_and_correctness_files(pkg, golden, tmp_path):
    name = "bls12_377"
    doc = golden[name]
    ch = doc["chunks"][0]
    prm = _params(pkg, name, doc, 0)
    f = {k: str(tmp_path / k) for k in ("challenge", "challenge.hash", "response", "response.hash", "new_challenge", "new_challenge.hash")}
    open(f["challenge"], "wb").write(bytes.fromhex(ch["challenge"]))
    open(f["response"], "wb").write(bytes.fromhex(ch["response"]))
    pkg.transform_pok_and_correctness(f["challenge"], f["challenge.hash"], pkg.CheckForCorrectness.No, f["response"],
                                      f["response.hash"], pkg.CheckForCorrectness.OnlyNonZero, f["new_challenge"],
                                      f["new_challenge.hash"], pkg.SubgroupCheckMode.Auto, False, prm)
    assert open(f["challenge.hash"], "rb").read().hex() == ch["challenge_hash"]
    assert open(f["response.hash"], "rb").read().hex() == ch["response_hash"]
    assert open(f["new_challenge.hash"], "rb").read().hex() == ch["new_challenge_hash"]
    nc = open(f["new_challenge"], "rb").read()
    assert pkg.calculate_hash(nc).hex() == ch["new_challenge_hash"]


def test_contribute_verify_round_trip_2pow12(pkg, cref):
    """BASELINE config #1 shape (BLS12-377, power 12, chunk 2^10): contribute then verify chunk 0 and a
    tauG1-only chunk; the verifier's new challenge must be the decompressed response (oracle-checked on a sample)."""
    name = "bls12_377"
    c = CURVES[name]
    rng = random.Random(12)
    for idx in (0, 5):
        p = P.Params(c, P.CHUNKED, idx, 1 << 10, 12, 1 << 10)
        prm = pkg.Phase1Parameters.new_chunk(name, 0, idx, 1 << 10, 0, 12, 1 << 10)
        ch0 = P.initialization(p)
        key = pkg.PrivateKey(rng.randrange(c.r), rng.randrange(c.r), rng.randrange(c.r), bytes(prm.public_key_size))
        resp = bytearray(pkg.phase1_cli.computation(ch0, prm, key))
        resp[:64] = pkg.calculate_hash(ch0)
        nc = pkg.phase1_cli.verification_and_decompress(ch0, bytes(resp), prm)
        assert nc[:64] == pkg.calculate_hash(bytes(resp))
        ref = cref.computation(name, 0, idx, 1 << 10, 12, ch0, key.tau, key.alpha, key.beta, nthreads=8)
        end = prm.contribution_size - prm.public_key_size
        assert bytes(resp[64:end]) == ref[64:end]
        G1 = Group(c, 1)
        n1 = p.g1_chunk_size
        assert nc[64:64 + n1 * 96] == cref.decompress(name, 1, bytes(resp[64:64 + n1 * 48]), n1, nthreads=8)


@pytest.mark.parametrize("name,group,n", [("bls12_377", 1, 1000), ("bls12_377", 2, 200), ("bw6", 1, 150), ("bw6", 2, 150),
                                          ("mnt4_753", 1, 100), ("mnt4_753", 2, 40), ("mnt6_753", 1, 100), ("mnt6_753", 2, 24)])
def test_msm_and_power_pairs(name, group, n, pkg, cref):
    """dense_multiexp with explicit scalars, and power_pairs with its documented coefficient generator,
    both bit-exact against the oracle's serial sum of scalar multiplications."""
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(31 + group)
    base = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(5)]
    pts = [base[i % 5] if i % 11 else None for i in range(n)]       # repeated points + some infinities
    data = serde.write_batch(G, pts, False)
    nl = cref.fr_limbs(name)
    scalars = [rng.randrange(c.r) for _ in range(n)]
    scalars[0], scalars[1] = 0, 1
    exp = cref.msm(name, group, data, n, scalars, nl)
    assert pkg.phase1_cli.dense_multiexp(name, group, data, scalars) == exp
    # empty and single-element sums
    assert pkg.phase1_cli.dense_multiexp(name, group, b"", []) == serde.write_point(G, None, False)
    assert pkg.phase1_cli.dense_multiexp(name, group, data[:G.uncomp_size], [7]) == cref.msm(name, group, data[:G.uncomp_size], 1, [7], nl)
    # power_pairs: same rho on the vector and on its shift by one
    seed = bytes(range(16))
    rho = [pkg.phase1_cli.power_pairs_coefficient(seed, 1000 + i, c.r.bit_length()) for i in range(n - 1)]
    a, b = pkg.phase1_cli.power_pairs(name, group, data, n, seed=seed, index_base=1000)
    assert a == cref.msm(name, group, data, n - 1, rho, nl)
    assert b == cref.msm(name, group, data[G.uncomp_size:], n - 1, rho, nl)


def test_power_pairs_ratio_property_full_size(pkg):
    """Size-independent property on a 2^18-point geometric vector v_i = s^i G: power_pairs gives (A, B) with
    B = s*A — the relation check_same_ratio tests with pairings — here checked by one more GPU scalar-mul."""
    name, n = "bls12_377", 1 << 18
    c = CURVES[name]
    G = Group(c, 1)
    s = 0x1234567890abcdef1234567890abcdef % c.r
    gen = serde.write_point(G, G.gen, False) * n
    v = pkg.phase1_cli.apply_powers(name, 1, gen, n, 0, s, None, False, False)
    a, b = pkg.phase1_cli.power_pairs(name, 1, v, n)        # fresh OS entropy
    sa = pkg.phase1_cli.apply_powers(name, 1, a, 1, 1, s, None, False, False)    # s^1 * A
    assert sa == b and a != serde.write_point(G, None, False)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_verify_with_ratio_check(name, pkg, golden):
    """ratio_check = true (the operator's default, contribute.rs:146-147): honest chunks pass, a response whose
    elements are valid subgroup points but NOT a geometric sequence is rejected with PTAU_E_INVALID_RATIO.
    The boolean outcome matches the oracle's own check_power_ratios (independent random coefficients)."""
    doc = golden[name]
    c = CURVES[name]
    G1, G2 = Group(c, 1), Group(c, 2)
    ch0, ch1 = doc["chunks"]
    prm0, prm1 = _params(pkg, name, doc, 0), _params(pkg, name, doc, 1)
    p0 = P.Params(c, P.CHUNKED, 0, doc["chunk_size"], doc["power"], doc["batch_size"])
    p1 = P.Params(c, P.CHUNKED, 1, doc["chunk_size"], doc["power"], doc["batch_size"])
    chal0, resp0 = bytes.fromhex(ch0["challenge"]), bytes.fromhex(ch0["response"])
    nc0 = pkg.phase1_cli.verification_and_decompress(chal0, resp0, prm0, ratio_check=True)
    assert pkg.calculate_hash(nc0).hex() == ch0["new_challenge_hash"]
    assert P.verify_chunk_ratios(p0, resp0, rng=random.Random(1))
    # tauG1-only chunk: needs (tauG2[0], tauG2[1]) of the same contribution
    off_g2 = {nm: o for (nm, g, o, cnt) in p0.layout(False)[0]}["tau_g2"]
    tau_g2_check = nc0[off_g2:off_g2 + 2 * G2.uncomp_size]
    chal1, resp1 = bytes.fromhex(ch1["challenge"]), bytes.fromhex(ch1["response"])
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(chal1, resp1, prm1, ratio_check=True)
    assert e.value.code == -1
    nc1 = pkg.phase1_cli.verification_and_decompress(chal1, resp1, prm1, ratio_check=True, tau_g2_check=tau_g2_check)
    assert pkg.calculate_hash(nc1).hex() == ch1["new_challenge_hash"]
    assert P.verify_chunk_ratios(p1, resp1, tau_g2_check=tuple(serde.read_batch(G2, tau_g2_check, False)), rng=random.Random(2))
    # tamper: swap two tauG1 elements (still in the subgroup, no longer geometric)
    bad = bytearray(resp0)
    o = 64
    a, b = bad[o + G1.comp_size:o + 2 * G1.comp_size], bad[o + 2 * G1.comp_size:o + 3 * G1.comp_size]
    bad[o + G1.comp_size:o + 2 * G1.comp_size], bad[o + 2 * G1.comp_size:o + 3 * G1.comp_size] = b, a
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(chal0, bytes(bad), prm0, ratio_check=True)
    assert e.value.code == -6
    assert not P.verify_chunk_ratios(p0, bytes(bad), rng=random.Random(3))
    # the same tampering passes when the ratio check is skipped (--skip-ratio-check)
    pkg.phase1_cli.verification_and_decompress(chal0, bytes(bad), prm0, ratio_check=False)
    # tamper the G2 vector the same way
    off2 = {nm: o for (nm, g, o, cnt) in p0.layout(True)[0]}["tau_g2"]
    bad2 = bytearray(resp0)
    s = G2.comp_size
    bad2[off2 + s:off2 + 2 * s], bad2[off2 + 2 * s:off2 + 3 * s] = bad2[off2 + 2 * s:off2 + 3 * s], bad2[off2 + s:off2 + 2 * s]
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.verification_and_decompress(chal0, bytes(bad2), prm0, ratio_check=True)
    assert e.value.code == -6


@pytest.mark.parametrize("name", ["bls12_377", "mnt6_753"])
def test_round_combine_and_transform_ratios(name, pkg):
    """A whole (tiny) round: new_challenge -> contribute every chunk with one key -> combine -> transform_ratios
    (verify_transcript.rs:596-653 / control.rs:510-591).  combine is bit-exact vs the oracle's aggregation."""
    c = CURVES[name]
    power, cs = 3, 4
    rng = random.Random(77)
    key_scalars = [rng.randrange(1, c.r) for _ in range(3)]
    full = pkg.Phase1Parameters.new_full(name, 0, power, cs)
    n_chunks = -(-full.powers_g1_length // cs)
    responses, oparams = [], []
    for i in range(n_chunks):
        prm = pkg.Phase1Parameters.new_chunk(name, 0, i, cs, 0, power, cs)
        ch = pkg.phase1_cli.initialization(prm)
        key = pkg.PrivateKey(*key_scalars, bytes(prm.public_key_size))
        resp = bytearray(pkg.phase1_cli.computation(ch, prm, key))
        resp[:64] = pkg.calculate_hash(ch)
        responses.append(bytes(resp))
        oparams.append(P.Params(c, P.CHUNKED, i, cs, power, cs))
    combined = pkg.combine(responses, cs, full)
    assert combined == P.aggregation(oparams, responses, P.Params(c, P.FULL, 0, cs, power, cs))
    pkg.transform_ratios(combined, pkg.CheckForCorrectness.Full, full)
    # a combined accumulator with two elements of alpha_g1 swapped is not geometric any more
    G1, G2 = Group(c, 1), Group(c, 2)
    off = 64 + full.powers_g1_length * G1.uncomp_size + full.powers_length * G2.uncomp_size
    bad = bytearray(combined)
    u = G1.uncomp_size
    bad[off + u:off + 2 * u], bad[off + 2 * u:off + 3 * u] = bad[off + 2 * u:off + 3 * u], bad[off + u:off + 2 * u]
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_ratios(bytes(bad), pkg.CheckForCorrectness.No, full)
    assert e.value.code == -6
    # chunks contributed with DIFFERENT keys do not combine into a valid accumulator
    prm1 = pkg.Phase1Parameters.new_chunk(name, 0, 1, cs, 0, power, cs)
    other = pkg.PrivateKey(key_scalars[0] + 1, key_scalars[1], key_scalars[2], bytes(prm1.public_key_size))
    r1 = bytearray(pkg.phase1_cli.computation(pkg.phase1_cli.initialization(prm1), prm1, other))
    mixed = pkg.combine([responses[0], bytes(r1)] + responses[2:], cs, full)
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_ratios(mixed, pkg.CheckForCorrectness.No, full)
    assert e.value.code == -6


def test_e2e_phase1_config1_files(pkg, cref, tmp_path):
    """BASELINE configs[0] — e2e/e2ephase1.sh BLS12-377 --powers 12 --chunk-size 10 — through the FILE-level entry points
    the operator calls: new_challenge, two rounds of contribute -> transform_pok_and_correctness on all 8 chunks
    (4 full + 4 tauG1-only), then combine + transform_ratios on the last responses.  Hash chain and `.hash` side files
    checked at every step; every response is bit-exact vs the C oracle's Phase1::computation."""
    name, power, cs = "bls12_377", 12, 1 << 10
    rng = random.Random(2024)
    full = pkg.Phase1Parameters.new_full(name, 0, power, cs)
    n_chunks = -(-full.powers_g1_length // cs)
    assert n_chunks == 8
    keys = [[rng.randrange(1, CURVES[name].r) for _ in range(3)] for _ in range(2)]
    final_responses = []
    tau_g2_check = [None, None]
    for i in range(n_chunks):
        prm = pkg.Phase1Parameters.new_chunk(name, 0, i, cs, 0, power, cs)
        f = lambda s: str(tmp_path / ("chunk%d_%s" % (i, s)))
        pkg.new_challenge(f("challenge0"), f("challenge0.hash"), prm)
        ch = open(f("challenge0"), "rb").read()
        assert len(ch) == prm.accumulator_size and ch[:64].hex().startswith("786a02f742015903c6c6fd852552d272")
        assert open(f("challenge0.hash"), "rb").read() == pkg.calculate_hash(ch)
        for rnd in range(2):
            key = pkg.PrivateKey(*keys[rnd], (bytes(range(256)) * 12)[:prm.public_key_size])
            c, r, nc = f("challenge%d" % rnd), f("response%d" % rnd), f("challenge%d" % (rnd + 1))
            pkg.contribute(c, c + ".h", r, r + ".hash", pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, key)
            chal, resp = open(c, "rb").read(), open(r, "rb").read()
            assert resp[:64] == pkg.calculate_hash(chal) == open(c + ".h", "rb").read()
            assert open(r + ".hash", "rb").read() == pkg.calculate_hash(resp)
            assert resp[-prm.public_key_size:] == key.public_key
            ref = cref.computation(name, 0, i, cs, power, chal, key.tau, key.alpha, key.beta, nthreads=8)
            end = prm.contribution_size - prm.public_key_size
            assert resp[64:end] == ref[64:end]
            with pytest.raises(pkg.PtauError) as e:      # create_new semantics: the stale response is never overwritten
                pkg.contribute(c, c + ".h", r, r + ".hash", pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, key)
            assert e.value.code == -9
            kw = {}
            if prm.other_chunk_size == 0:
                kw["tau_g2_check"] = tau_g2_check[rnd]
            pkg.transform_pok_and_correctness(c, c + ".vh", pkg.CheckForCorrectness.No, r, r + ".vh",
                                              pkg.CheckForCorrectness.OnlyNonZero, nc, nc + ".hash", pkg.SubgroupCheckMode.Auto,
                                              True, prm, **kw)
            new_ch = open(nc, "rb").read()
            assert new_ch[:64] == pkg.calculate_hash(resp) == open(r + ".vh", "rb").read()
            assert open(nc + ".hash", "rb").read() == pkg.calculate_hash(new_ch)
            if i == 0:    # (tauG2[0], tauG2[1]) of this contribution, for the tauG1-only chunks
                off = 64 + prm.g1_chunk_size * 96
                tau_g2_check[rnd] = new_ch[off:off + 2 * 192]
        final_responses.append(open(f("response1"), "rb").read())
    combined = pkg.combine(final_responses, cs, full)
    assert len(combined) == full.accumulator_size
    pkg.transform_ratios(combined, pkg.CheckForCorrectness.Full, full)
    # the combined accumulator is [ (tau0*tau1)^i G ]: element 1 of tauG1 is (tau0*tau1) G
    t = keys[0][0] * keys[1][0] % CURVES[name].r
    G1 = Group(CURVES[name], 1)
    assert combined[64 + 96:64 + 192] == serde.write_point(G1, G1.mul(G1.gen, t), False)


@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 1)])
def test_contribute_check_input_in_group(name, group, pkg, cref):
    """CheckForCorrectness on the contribute side (--force-correctness-checks upgrades it to Full, contribute.rs:129-133):
    OnlyInGroup / Full run the membership test over the input vector before the update and reject a curve point outside
    the prime-order subgroup; with No (the operator's default) a vector of subgroup points gives the oracle's bytes."""
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(8)
    good = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(6)]
    tau = rng.randrange(c.r)
    data = serde.write_batch(G, good, False)
    exp = cref.apply_powers(name, group, data, 6, 3, tau, None, False, True)
    for chk in (pkg.CheckForCorrectness.No, pkg.CheckForCorrectness.Full):
        assert pkg.phase1_cli.apply_powers(name, group, data, 6, 3, tau, check_input=chk) == exp
    bad = serde.write_batch(G, good[:2] + [_off_subgroup_point(G, rng)] + good[2:], False)
    for chk in (pkg.CheckForCorrectness.OnlyInGroup, pkg.CheckForCorrectness.Full):
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.apply_powers(name, group, bad, 7, 3, tau, check_input=chk)
        assert e.value.code == -5
    inf = serde.write_batch(G, good[:1] + [None], False)
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.apply_powers(name, group, inf, 2, 0, tau, check_input=pkg.CheckForCorrectness.OnlyNonZero)
    assert e.value.code == -4


def test_power_pairs_across_msm_slices(pkg):
    """More points than one bucket pass takes (MSM_SLICE_POINTS = 2^22) and than one tau-update launch takes (2^20): the
    geometric vector v_i = s^i G of 2^22 + 5 points still reduces to (A, B) with B = s*A, and a single swapped pair of
    elements in the LAST slice breaks the relation."""
    name, n = "bls12_377", (1 << 22) + 5
    c = CURVES[name]
    G = Group(c, 1)
    s = 0x1234567890abcdef1234567890abcdef % c.r
    gen = serde.write_point(G, G.gen, False) * n
    v = pkg.phase1_cli.apply_powers(name, 1, gen, n, 0, s, None, False, False)
    del gen
    a, b = pkg.phase1_cli.power_pairs(name, 1, v, n)
    assert pkg.phase1_cli.apply_powers(name, 1, a, 1, 1, s, None, False, False) == b and a != serde.write_point(G, None, False)
    bad = bytearray(v)
    o = (n - 3) * 96
    bad[o:o + 96], bad[o + 96:o + 192] = bad[o + 96:o + 192], bad[o:o + 96]
    a2, b2 = pkg.phase1_cli.power_pairs(name, 1, bytes(bad), n)
    assert pkg.phase1_cli.apply_powers(name, 1, a2, 1, 1, s, None, False, False) != b2
