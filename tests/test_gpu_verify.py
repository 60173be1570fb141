"""GPU parity tests of the verify path (transform_pok_and_correctness element checks + decompress),
through the C ABI, against the oracle and the golden fixtures."""
import random

import pytest

from conftest import CURVE_NAMES, initial_challenge
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
    """The ONE file-level call named after phase1_cli::transform_pok_and_correctness checks the proofs of knowledge and the
    chunk-0 before/after ratios itself (Phase1::verification steps 1-2), not only the elements."""
    name = "bls12_377"
    doc = golden[name]
    ch = doc["chunks"][0]
    prm = _params(pkg, name, doc, 0)
    p = P.Params(CURVES[name], P.CHUNKED, 0, doc["chunk_size"], doc["power"], doc["batch_size"])
    challenge, response = bytes.fromhex(ch["challenge"]), bytes.fromhex(ch["response"])
    g2_s = P.synthetic_g2_s(p, P.calculate_hash(challenge))   # the golden key's base points, as the Rust host would pass compute_g2_s
    f = {k: str(tmp_path / k) for k in ("challenge", "challenge.hash", "response", "response.hash", "new_challenge", "new_challenge.hash")}
    open(f["challenge"], "wb").write(challenge)
    open(f["response"], "wb").write(response)
    pkg.transform_pok_and_correctness(f["challenge"], f["challenge.hash"], pkg.CheckForCorrectness.No, f["response"],
                                      f["response.hash"], pkg.CheckForCorrectness.OnlyNonZero, f["new_challenge"],
                                      f["new_challenge.hash"], pkg.SubgroupCheckMode.Auto, False, prm, g2_s=g2_s)
    assert open(f["challenge.hash"], "rb").read().hex() == ch["challenge_hash"]
    assert open(f["response.hash"], "rb").read().hex() == ch["response_hash"]
    assert open(f["new_challenge.hash"], "rb").read().hex() == ch["new_challenge_hash"]
    nc = open(f["new_challenge"], "rb").read()
    assert pkg.calculate_hash(nc).hex() == ch["new_challenge_hash"]
    # forged proof of knowledge: the elements are those of an honest response, the public key proves ANOTHER tau
    tau, alpha, beta = (int(ch[k], 16) for k in ("tau", "alpha", "beta"))
    other_pk = P.synthetic_public_key(p, (tau + 1) % CURVES[name].r, alpha, beta, P.calculate_hash(challenge))
    forged = response[:-len(other_pk)] + other_pk
    for k, (g, code) in enumerate(((g2_s, -6), (None, -6))):   # caller-supplied and library-derived g2_s both reject
        fr, fn = str(tmp_path / ("forged%d" % k)), str(tmp_path / ("forged%d.nc" % k))
        open(fr, "wb").write(forged)
        with pytest.raises(pkg.PtauError) as e:
            pkg.transform_pok_and_correctness(f["challenge"], fr + ".ch", pkg.CheckForCorrectness.No, fr, fr + ".h",
                                              pkg.CheckForCorrectness.OnlyNonZero, fn, fn + ".h", pkg.SubgroupCheckMode.Auto,
                                              False, prm, g2_s=g)
        assert e.value.code == code
    # the golden (synthetic) key is NOT one the library's own compute_g2_s derivation accepts: g2_s = None rejects it too
    fn = str(tmp_path / "derived.nc")
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_pok_and_correctness(f["challenge"], fn + ".ch", pkg.CheckForCorrectness.No, f["response"], fn + ".rh",
                                          pkg.CheckForCorrectness.OnlyNonZero, fn, fn + ".h", pkg.SubgroupCheckMode.Auto, False, prm)
    assert e.value.code == -6
    # check_input is honoured (--force-correctness-checks): a challenge with a point at infinity / off the subgroup is refused
    G1 = Group(CURVES[name], 1)
    bad_ch = bytearray(challenge)
    bad_ch[64 + 96:64 + 192] = serde.write_point(G1, None, False)
    fc, fn = str(tmp_path / "bad.challenge"), str(tmp_path / "bad.nc")
    open(fc, "wb").write(bytes(bad_ch))
    bad_resp = pkg.calculate_hash(bytes(bad_ch)) + response[64:]
    fr = str(tmp_path / "bad.response")
    open(fr, "wb").write(bad_resp)
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_pok_and_correctness(fc, fc + ".h", pkg.CheckForCorrectness.Full, fr, fr + ".h",
                                          pkg.CheckForCorrectness.OnlyNonZero, fn, fn + ".h", pkg.SubgroupCheckMode.Auto, False, prm, g2_s=g2_s)
    assert e.value.code == -4
    bad_ch[64 + 96:64 + 192] = serde.write_point(G1, _off_subgroup_point(G1, random.Random(4)), False)
    open(fc, "wb").write(bytes(bad_ch))
    open(fr, "wb").write(pkg.calculate_hash(bytes(bad_ch)) + response[64:])
    for chk, code in ((pkg.CheckForCorrectness.OnlyInGroup, -5), (pkg.CheckForCorrectness.Full, -5)):
        fn = str(tmp_path / ("bad%d.nc" % int(chk)))
        with pytest.raises(pkg.PtauError) as e:
            pkg.transform_pok_and_correctness(fc, fc + ".h2", chk, fr, fr + ".h2", pkg.CheckForCorrectness.OnlyNonZero, fn, fn + ".h",
                                              pkg.SubgroupCheckMode.Auto, False, prm, g2_s=g2_s)
        assert e.value.code == code


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
    # without tau_g2_check an honest tauG1-only chunk still verifies (its own files hold no G2 pair of ratio tau: the
    # power ratios of such a chunk are left to transform_ratios on the combined accumulator, control.rs:587-591)
    assert pkg.phase1_cli.verification_and_decompress(chal1, resp1, prm1, ratio_check=True)[64:] == P.new_challenge_from_response(p1, resp1)[64:]
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
        ch = initial_challenge(pkg, name, prm)
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
    r1 = bytearray(pkg.phase1_cli.computation(initial_challenge(pkg, name, prm1), prm1, other))
    mixed = pkg.combine([responses[0], bytes(r1)] + responses[2:], cs, full)
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_ratios(mixed, pkg.CheckForCorrectness.No, full)
    assert e.value.code == -6


def test_e2e_phase1_config1_files(pkg, cref, tmp_path):
    """BASELINE configs[0] — e2e/e2ephase1.sh BLS12-377 --powers 12 --chunk-size 10 — through the FILE-level entry points
    the operator calls, with the reference's own argument lists: new_challenge, two rounds of contribute ->
    transform_pok_and_correctness (ratio_check = true, the operator default) on all 8 chunks (4 full + 4 tauG1-only), then
    combine + transform_ratios on the last responses.  Round 0 contributes from a SEED (derive_rng_from_seed +
    Phase1::key_generation inside the library), round 1 with a key pair made by ptau_key_generation on the host side; the
    verifier derives compute_g2_s itself and checks every proof of knowledge.  Hash chain and `.hash` side files checked at
    every step; every response is bit-exact vs the C oracle's Phase1::computation."""
    name, power, cs = "bls12_377", 12, 1 << 10
    full = pkg.Phase1Parameters.new_full(name, 0, power, cs)
    n_chunks = -(-full.powers_g1_length // cs)
    assert n_chunks == 8
    seeds = [bytes([7 + rnd]) * 64 for rnd in range(2)]
    final_responses = []
    taus = [None, None]
    for i in range(n_chunks):
        prm = pkg.Phase1Parameters.new_chunk(name, 0, i, cs, 0, power, cs)
        f = lambda s: str(tmp_path / ("chunk%d_%s" % (i, s)))
        pkg.new_challenge(f("challenge0"), f("challenge0.hash"), prm)
        ch = open(f("challenge0"), "rb").read()
        assert len(ch) == prm.accumulator_size and ch[:64].hex().startswith("786a02f742015903c6c6fd852552d272")
        assert open(f("challenge0.hash"), "rb").read() == pkg.calculate_hash(ch)
        for rnd in range(2):
            c, r, nc = f("challenge%d" % rnd), f("response%d" % rnd), f("challenge%d" % (rnd + 1))
            chal = open(c, "rb").read()
            key = pkg.phase1_cli.key_generation(name, seeds[rnd], pkg.calculate_hash(chal))
            taus[rnd] = key.tau
            if rnd == 0:
                pkg.phase1_cli.contribute_seeded(c, c + ".h", r, r + ".hash", pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, seeds[rnd])
            else:
                pkg.contribute(c, c + ".h", r, r + ".hash", pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, key)
            resp = open(r, "rb").read()
            assert resp[:64] == pkg.calculate_hash(chal) == open(c + ".h", "rb").read()
            assert open(r + ".hash", "rb").read() == pkg.calculate_hash(resp)
            assert resp[-prm.public_key_size:] == key.public_key
            ref = cref.computation(name, 0, i, cs, power, chal, key.tau, key.alpha, key.beta, nthreads=8)
            end = prm.contribution_size - prm.public_key_size
            assert resp[64:end] == ref[64:end]
            with pytest.raises(pkg.PtauError) as e:      # create_new semantics: the stale response is never overwritten
                pkg.contribute(c, c + ".h", r, r + ".hash", pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, key)
            assert e.value.code == -9
            pkg.transform_pok_and_correctness(c, c + ".vh", pkg.CheckForCorrectness.No, r, r + ".vh",
                                              pkg.CheckForCorrectness.OnlyNonZero, nc, nc + ".hash", pkg.SubgroupCheckMode.Auto,
                                              True, prm)
            new_ch = open(nc, "rb").read()
            assert new_ch[:64] == pkg.calculate_hash(resp) == open(r + ".vh", "rb").read()
            assert open(nc + ".hash", "rb").read() == pkg.calculate_hash(new_ch)
        final_responses.append(open(f("response1"), "rb").read())
    combined = pkg.combine(final_responses, cs, full)
    assert len(combined) == full.accumulator_size
    pkg.transform_ratios(combined, pkg.CheckForCorrectness.Full, full)
    # the combined accumulator is [ (tau0*tau1)^i G ]: element 1 of tauG1 is (tau0*tau1) G
    t = taus[0] * taus[1] % CURVES[name].r
    G1 = Group(CURVES[name], 1)
    assert combined[64 + 96:64 + 192] == serde.write_point(G1, G1.mul(G1.gen, t), False)


@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 1), ("bw6", 2), ("mnt4_753", 2), ("mnt6_753", 2)])
def test_batch_exp_exact_on_any_curve_point(name, group, pkg, cref):
    """arkworks' batch_exp returns [k]P for ANY curve point, and contribute's default is CheckForCorrectness::No
    (contribute.rs:816-819): with unchecked input the update must be exact on points OUTSIDE the prime-order subgroup too,
    where the endomorphisms are not [lambda] / [x].  Auto and Direct give the oracle's bytes on a vector that mixes
    subgroup points, cofactor-only points and mixed points; the endomorphism path (membership established by check_input,
    or asserted by the caller) gives the same bytes on subgroup points."""
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(31 + group)
    good = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(5)]
    off = [_off_subgroup_point(G, rng) for _ in range(3)]
    cof_only = G.mul(off[0], c.r)                  # order divides the cofactor
    assert cof_only is not None and not G.in_subgroup(off[1])
    pts = [good[0], off[0], good[1], cof_only, off[1], good[2], G.add(good[3], cof_only), off[2], None]
    n = len(pts)
    data = serde.write_batch(G, pts, False)
    tau, coeff = rng.randrange(c.r), rng.randrange(1, c.r)
    for cout in (True, False):
        exp = cref.apply_powers(name, group, data, n, 5, tau, coeff, False, cout)
        for mode in (pkg.BatchExpMode.Auto, pkg.BatchExpMode.Direct, pkg.BatchExpMode.BatchInversion):
            got = pkg.phase1_cli.apply_powers(name, group, data, n, 5, tau, coeff, False, cout, batch_exp_mode=mode)
            assert got == exp, (name, group, mode, cout)
    # subgroup points: every path agrees with the oracle, including the asserted-membership one
    gdata = serde.write_batch(G, good, False)
    exp = cref.apply_powers(name, group, gdata, 5, 9, tau, None, False, True)
    for mode, chk in ((pkg.BatchExpMode.Auto, pkg.CheckForCorrectness.No), (pkg.BatchExpMode.Auto, pkg.CheckForCorrectness.Full),
                      (int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP, pkg.CheckForCorrectness.No),
                      (pkg.BatchExpMode.Direct, pkg.CheckForCorrectness.Full)):
        assert pkg.phase1_cli.apply_powers(name, group, gdata, 5, 9, tau, None, False, True, check_input=chk, batch_exp_mode=mode) == exp
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.apply_powers(name, group, gdata, 5, 9, tau, batch_exp_mode=7)
    assert e.value.code == -1


@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 1), ("mnt4_753", 2), ("mnt6_753", 2)])
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


def test_cold_start_from_several_threads(tmp_path):
    """The FIRST ptau_* calls of a process arrive from three threads at once on one device (in-process lanes of the operator,
    contribute.rs:467-492): context creation is serialised inside the library, every lane gets the right bytes."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "cold.py"
    script.write_text('''
import sys, threading, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import torch; torch.cuda.is_available()
from conftest import load_package
pkg = load_package()
gen = pkg.phase1_cli.generator("bls12_377", 1)
n = 4096
data = gen * n
out = [None] * 3
def lane(k):
    out[k] = pkg.phase1_cli.apply_powers("bls12_377", 1, data, n, 17 * k, 0x1234567 + k)
th = [threading.Thread(target=lane, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
again = [pkg.phase1_cli.apply_powers("bls12_377", 1, data, n, 17 * k, 0x1234567 + k) for k in range(3)]
assert out == again and len(set(out)) == 3
print("ok", hashlib.blake2b(b"".join(out)).hexdigest()[:16])
''' % (ROOT, ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 1), ("bw6", 2)])
def test_small_order_points_through_the_window_loops(name, group, pkg, cref):
    """The window loops of these groups run in the carry-free fields (fpz.hpp, fplz.hpp): the exceptional cases — a table
    multiple that is O, an addition that meets its own operand or its negative, the doubling of a 2-torsion point — are
    decided by residue tests on unreduced values (over a lane pair for G2 and BW6-761).  Points of every small prime-power
    order the cofactor offers, sums of them with subgroup points, and the identity, against the oracle's double-and-add."""
    from oracle.pyref import keygen as K
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(90 + group)
    h = K._cofactor(G)
    F = G.F
    def random_curve_point():
        while True:
            x = tuple(rng.randrange(c.q) for _ in range(F.deg))
            y = F.sqrt(F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b))
            if y is not None:
                return (x, y)
    small = []
    for ell in (2, 3, 5, 7, 11, 13, 17, 19):
        k = 0
        while h % ell ** (k + 1) == 0:
            k += 1
        if not k:
            continue
        for _ in range(4):   # a point of order ell^j, j >= 1, and its multiples down to order ell
            T = G.mul(G.mul(random_curve_point(), c.r), h // ell ** k)
            chain = []
            while T is not None and len(chain) < 64:
                chain.append(T)
                T = G.mul(T, ell)
            if chain:
                small += chain[-3:]
                break
    if not small:   # BLS12-377 G2: the cofactor has no prime factor below 2 * 10^5 — cofactor-only points of large order instead
        small = [G.mul(random_curve_point(), c.r) for _ in range(2)]
    good = [G.mul(G.gen, rng.randrange(1, c.r)) for _ in range(3)]
    pts = small + [None] + [G.add(good[i % 3], small[i % len(small)]) for i in range(3)] + good
    n = len(pts)
    data = serde.write_batch(G, pts, False)
    for start, tau, coeff in ((0, rng.randrange(c.r), None), (3, 1, int("8" * 60, 16) % c.r), (0, 1, 1 << 200), (7, c.r - 1, rng.randrange(c.r)),
                              (0, 2, 15), (5, 0, None)):
        exp = cref.apply_powers(name, group, data, n, start, tau, coeff, False, False)
        for mode in (pkg.BatchExpMode.Auto, pkg.BatchExpMode.Direct):
            got = pkg.phase1_cli.apply_powers(name, group, data, n, start, tau, coeff, False, False, batch_exp_mode=mode)
            assert got == exp, (name, group, start, tau, coeff, mode)


@pytest.mark.parametrize("name", ["bls12_377", "bw6", "mnt4_753"])
def test_batched_subgroup_check_through_the_msm_buckets(name, pkg):
    """SubgroupCheckMode::Auto on more than 2^12 elements (and ::Batched) with the ratio check on: the membership test runs on
    the bucket sums of the ratio check's own MSM (capi.hip run_verify_vector, msm_plan) — setup-utils' batched check with
    SECURITY_PARAM = 128.  An honest 2^13 chunk passes and reports no per-element membership time; one element outside the
    subgroup — first, middle, last of a vector, in G1 and G2, a cofactor-only point, and a cancelling pair P, -P — is
    rejected with the same code as the exact test (Direct), which is also run on the same bytes."""
    c = CURVES[name]
    rng = random.Random(31)
    lg = 13
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, n, 0, lg, n)
    p = P.Params(c, P.CHUNKED, 0, n, lg, n)
    ch0 = initial_challenge(pkg, name, prm)
    key = pkg.PrivateKey(rng.randrange(1, c.r), rng.randrange(1, c.r), rng.randrange(1, c.r), bytes(prm.public_key_size))
    resp = bytearray(pkg.phase1_cli.computation(ch0, prm, key))
    resp[:64] = pkg.calculate_hash(ch0)
    Mode = pkg.SubgroupCheckMode
    for mode in (Mode.Auto, Mode.Batched, Mode.Direct):
        nc = pkg.phase1_cli.verification_and_decompress(ch0, bytes(resp), prm, subgroup_check_mode=mode, ratio_check=True)
        assert nc[:64] == pkg.calculate_hash(bytes(resp))
        tm = pkg.phase1_cli.last_timing(0)
        assert tm["ratio_checks"] == 4
        if mode == Mode.Direct or name.startswith("mnt"):
            assert tm["subgroup_ms"] > 0          # MNT G1 has cofactor 1: its (on-curve) test stays per element
        if mode != Mode.Direct and not name.startswith("mnt"):
            assert tm["subgroup_ms"] == 0.0       # every vector went through the buckets
    offs = {nm: (g, o, cnt) for (nm, g, o, cnt) in p.layout(True)[0]}
    cases = []
    groups = (2,) if name.startswith("mnt") else (1, 2)
    for grp in groups:
        G = Group(c, grp)
        bad = _off_subgroup_point(G, rng)
        vecs = ["tau_g2"] if grp == 2 else ["tau_g1", "alpha_g1", "beta_g1"]
        for v in vecs:
            _, o, cnt = offs[v]
            for pos in (0, cnt // 2 + 3, cnt - 1):
                cases.append((G, o, [(pos, bad)]))
        _, o, cnt = offs[vecs[0]]
        cof = G.mul(bad, c.r)                       # a point of the cofactor subgroup alone
        if cof is not None:
            cases.append((G, o, [(17, cof)]))
        cases.append((G, o, [(100, bad), (4000, G.neg(bad))]))            # P and -P: cancel only if they share a bucket in every window
        cases.append((G, o, [(cnt - 2, bad), (cnt - 1, G.neg(bad))]))
    for G, o, subs in cases:
        r2 = bytearray(resp)
        for pos, pt in subs:
            r2[o + pos * G.comp_size:o + (pos + 1) * G.comp_size] = serde.write_point(G, pt, True)
        for mode in (Mode.Auto, Mode.Batched, Mode.Direct):
            with pytest.raises(pkg.PtauError) as e:
                pkg.phase1_cli.verification_and_decompress(ch0, bytes(r2), prm, subgroup_check_mode=mode, ratio_check=True)
            assert e.value.code == -5, (G.group, subs[0][0], mode, e.value)
        # without the ratio check there is no MSM to ride on: the exact test runs, same verdict
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.verification_and_decompress(ch0, bytes(r2), prm, ratio_check=False)
        assert e.value.code == -5
    # the library keeps working after the rejected calls (a tail left pending by a failed call is discarded)
    nc = pkg.phase1_cli.verification_and_decompress(ch0, bytes(resp), prm, ratio_check=True)
    assert nc[:64] == pkg.calculate_hash(bytes(resp))


@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 1), ("mnt4_753", 2), ("mnt6_753", 2)])
def test_auto_probe_batched_on_long_vectors(name, group, pkg, cref):
    """BatchExpMode::Auto on more than 2^12 unchecked elements probes the vector with the batched membership test (random
    bucket sums, each tested exactly; capi.hip run_batched_subgroup_dev) and takes the endomorphism path only when it passes.
    A genuine vector gives the bytes of Direct and of the oracle; with ONE element outside the subgroup anywhere in it (first,
    middle, last; a cofactor-only point) the probe must fail and the plain path's exact bytes come out — compared with Direct
    over the whole vector and with the oracle around the planted element."""
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(77 + group)
    n = 4096 + 200
    usz = G.uncomp_size
    gen = serde.write_point(G, G.gen, False)
    s = rng.randrange(2, c.r)
    vec = pkg.phase1_cli.apply_powers(name, group, gen * n, n, 0, s, None, False, False, batch_exp_mode=pkg.BatchExpMode.Direct)
    tau, coeff = rng.randrange(c.r), rng.randrange(1, c.r)
    exp = pkg.phase1_cli.apply_powers(name, group, vec, n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Direct)
    got = pkg.phase1_cli.apply_powers(name, group, vec, n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Auto)
    assert pkg.phase1_cli.last_timing(0)["subgroup_ms"] > 0      # the probe ran
    assert got == exp
    csz = G.comp_size
    assert got[:8 * csz] == cref.apply_powers(name, group, vec[:8 * usz], 8, 11, tau, coeff, False, True)
    off = _off_subgroup_point(G, rng)
    for pos, pt in ((0, off), (n // 2 + 1, off), (n - 1, off), (1234, G.mul(off, c.r))):
        bad = bytearray(vec)
        bad[pos * usz:(pos + 1) * usz] = serde.write_point(G, pt, False)
        exp = pkg.phase1_cli.apply_powers(name, group, bytes(bad), n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Direct)
        got = pkg.phase1_cli.apply_powers(name, group, bytes(bad), n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Auto)
        assert got == exp, (name, group, pos)
        lo = max(0, min(pos - 1, n - 3))
        assert got[lo * csz:(lo + 3) * csz] == cref.apply_powers(name, group, bytes(bad[lo * usz:(lo + 3) * usz]), 3, 11 + lo, tau, coeff, False, True)
    # input that is not even a vector of curve points: the speculative endomorphism launch is thrown away and the plain path
    # reports what Direct reports
    x, y = G.mul(G.gen, 5)
    off_curve = serde.write_point(G, (x, G.F.add(y, G.F.one())), False)
    non_canonical = b"\xff" * (usz - 1) + b"\x3f"
    bad = bytearray(vec)
    bad[777 * usz:778 * usz] = non_canonical
    for mode in (pkg.BatchExpMode.Auto, pkg.BatchExpMode.Direct):
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.apply_powers(name, group, bytes(bad), n, 11, tau, coeff, False, True, batch_exp_mode=mode)
        assert e.value.code == -3, (name, group, mode, e.value)
    # an uncompressed (x, y) that is not on the curve is not looked at under CheckForCorrectness::No (deserialize_unchecked): the
    # plain formulas run on it; Auto's probe sees it is no curve point and must end on the same plain path
    bad[777 * usz:778 * usz] = off_curve
    assert pkg.phase1_cli.apply_powers(name, group, bytes(bad), n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Auto) == \
        pkg.phase1_cli.apply_powers(name, group, bytes(bad), n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Direct)
    inf = bytearray(vec)
    inf[5 * usz:6 * usz] = serde.write_point(G, None, False)
    for mode in (pkg.BatchExpMode.Auto, pkg.BatchExpMode.Direct):
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.apply_powers(name, group, bytes(inf), n, 11, tau, coeff, False, True, check_input=pkg.CheckForCorrectness.OnlyNonZero,
                                        batch_exp_mode=mode)
        assert e.value.code == -4
    # ... and an infinity element is a legal input when nobody asked for non-zero elements: same bytes on both paths
    assert pkg.phase1_cli.apply_powers(name, group, bytes(inf), n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Auto) == \
        pkg.phase1_cli.apply_powers(name, group, bytes(inf), n, 11, tau, coeff, False, True, batch_exp_mode=pkg.BatchExpMode.Direct)


@pytest.mark.parametrize("name,group", [(c, g) for c in ("bls12_377", "bw6", "mnt4_753", "mnt6_753") for g in (1, 2)])
def test_direct_windows_on_structured_scalars(name, group, pkg):
    """K1's window loops on scalars whose signed 4-bit digits are chosen, not random: a positive digit after the first addition
    (0x11, 0x21, 0x7777), a negative one (0x1f, 0x9999), single digits, the top of the range — [k]G for 64 copies of a point
    through BatchExpMode::Direct (plain windows) and through the endomorphism loops (membership asserted), against the Python
    oracle's scalar multiplication.  Round 4 found the BLS12-377 G2 plain-window kernel wrong on exactly the first pattern while
    every random-scalar parity test passed through the GLS loop; this pins each loop's digit handling on every group."""
    c = CURVES[name]
    G = Group(c, group)
    csz = G.comp_size
    gen = serde.write_point(G, G.gen, False)
    n = 64
    for k in (1, 7, 8, 9, 0x10, 0x11, 0x12, 0x1f, 0x21, 0x88, 0x89, 0x111, 0x7777, 0x9999, 0xfedcba9876543210, c.r - 1, c.r - 2, c.r // 3):
        want = serde.write_point(G, G.mul(G.gen, k), True)
        for mode in (pkg.BatchExpMode.Direct, int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP):
            got = pkg.phase1_cli.apply_powers(name, group, gen * n, n, 0, 1, k, False, True, batch_exp_mode=mode)
            assert got == want * n, (name, group, hex(k), int(mode))
