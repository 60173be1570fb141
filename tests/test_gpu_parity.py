"""GPU parity tests: every call goes through the C ABI (libptau_hip.so) and is compared bit-for-bit
with the CPU oracle / the committed golden fixtures.  Run with -m gpu on an MI355X."""
import ctypes
import os
import random

import pytest

from conftest import CURVE_NAMES
from oracle.pyref import phase1 as P, serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group

pytestmark = pytest.mark.gpu


def _key(pkg, ch):
    return pkg.PrivateKey(int(ch["tau"], 16), int(ch["alpha"], 16), int(ch["beta"], 16), bytes.fromhex(ch["public_key"]))


def _rand_points(G, rng, n, with_inf=True):
    base = [G.mul(G.gen, rng.randrange(1, G.r)) for _ in range(4)]
    pts = []
    for i in range(n):
        pts.append(base[i % 4] if i % 7 else G.add(base[i % 4], base[(i + 1) % 4]))
    if with_inf and n > 2:
        pts[n // 2] = None
    return pts


def test_native_library_loaded(pkg):
    import torch
    assert torch.cuda.is_available()
    L = pkg.load_library()
    assert L.ptau_device_count() >= 1
    name = ctypes.create_string_buffer(128)
    cu = ctypes.c_int()
    mem = ctypes.c_uint64()
    assert L.ptau_device_info(0, name, 128, ctypes.byref(cu), ctypes.byref(mem)) == 0
    assert b"gfx950" in name.value, name.value
    with open("/proc/self/maps") as f:
        assert "libptau_hip.so" in f.read()


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_golden_chunk_contribution(name, pkg, golden, cref):
    """Phase1::computation on both golden chunks: identical to the fixture and to the C oracle."""
    doc = golden[name]
    for ch in doc["chunks"]:
        prm = pkg.Phase1Parameters.new_chunk(name, 0, ch["chunk_index"], doc["chunk_size"], 0, doc["power"], doc["batch_size"])
        challenge, response = bytes.fromhex(ch["challenge"]), bytes.fromhex(ch["response"])
        out = pkg.phase1_cli.computation(challenge, prm, _key(pkg, ch))
        end = ch["contribution_size"] - ch["public_key_size"]
        assert out[64:end] == response[64:end]
        ref = cref.computation(name, 0, ch["chunk_index"], doc["chunk_size"], doc["power"], challenge,
                               int(ch["tau"], 16), int(ch["alpha"], 16), int(ch["beta"], 16))
        assert out[64:end] == ref[64:end]


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_contribute_file_hash_chain(name, pkg, golden, tmp_path):
    """phase1_cli::contribute file semantics: .hash side files, response = hash | elements | pk."""
    doc = golden[name]
    ch = doc["chunks"][0]
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, doc["chunk_size"], 0, doc["power"], doc["batch_size"])
    f = {k: str(tmp_path / k) for k in ("challenge", "challenge.hash", "response", "response.hash")}
    open(f["challenge"], "wb").write(bytes.fromhex(ch["challenge"]))
    pkg.contribute(f["challenge"], f["challenge.hash"], f["response"], f["response.hash"],
                   pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, _key(pkg, ch))
    assert open(f["response"], "rb").read() == bytes.fromhex(ch["response"])
    assert open(f["challenge.hash"], "rb").read().hex() == ch["challenge_hash"]
    assert open(f["response.hash"], "rb").read().hex() == ch["response_hash"]
    # create_new semantics: a second run must refuse to overwrite the response
    with pytest.raises(pkg.PtauError) as e:
        pkg.contribute(f["challenge"], f["challenge.hash"], f["response"], f["response.hash"],
                       pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, _key(pkg, ch))
    assert e.value.code == -9
    # wrong challenge length -> PTAU_E_LEN, like the reference's size assertion
    open(f["challenge"], "ab").write(b"\0")
    os.remove(f["response"])
    with pytest.raises(pkg.PtauError) as e:
        pkg.contribute(f["challenge"], f["challenge.hash"], f["response"], f["response.hash"],
                       pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, _key(pkg, ch))
    assert e.value.code == -2


@pytest.mark.parametrize("name,group,n", [("bls12_377", 1, 777), ("bls12_377", 2, 300), ("bw6", 1, 300), ("bw6", 2, 290),
                                          ("mnt4_753", 1, 270), ("mnt4_753", 2, 130), ("mnt6_753", 1, 270), ("mnt6_753", 2, 70)])
def test_apply_powers_random(name, group, n, pkg, cref):
    """Ragged sizes (not multiples of the block / inversion-group sizes), infinity inside, large start,
    all four (compressed_in, compressed_out) combinations."""
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(1000 + group)
    pts = _rand_points(G, rng, n)
    unc = serde.write_batch(G, pts, False)
    tau, coeff, start = rng.randrange(c.r), rng.randrange(c.r), (1 << 27) + 12345
    exp_c = cref.apply_powers(name, group, unc, n, start, tau, coeff, False, True, nthreads=8)
    got_c = pkg.phase1_cli.apply_powers(name, group, unc, n, start, tau, coeff, False, True)
    assert got_c == exp_c
    exp_u = cref.decompress(name, group, exp_c, n, nthreads=8)
    assert pkg.phase1_cli.apply_powers(name, group, unc, n, start, tau, coeff, False, False) == exp_u
    # compressed input (sqrt on device), no coefficient
    m = min(n, 64)
    comp_in = exp_c[:m * G.comp_size]
    exp2 = cref.apply_powers(name, group, comp_in, m, 3, tau, None, True, True, nthreads=8)
    assert pkg.phase1_cli.apply_powers(name, group, comp_in, m, 3, tau, None, True, True) == exp2
    assert pkg.phase1_cli.apply_powers(name, group, comp_in, m, 3, tau, None, True, False) == cref.decompress(name, group, exp2, m)


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_edge_cases(name, pkg, cref):
    c = CURVES[name]
    for group in (1, 2):
        G = Group(c, group)
        g = serde.write_point(G, G.gen, False)
        # empty vector
        assert pkg.phase1_cli.apply_powers(name, group, b"", 0, 0, 5) == b""
        # single element; tau = 0 with start 0 -> 0^0 = 1; tau = 1; tau = r-1
        for tau, coeff, start in ((0, None, 0), (1, None, 9), (c.r - 1, c.r - 1, 1), (0, None, 5)):
            exp = cref.apply_powers(name, group, g, 1, start, tau, coeff)
            assert pkg.phase1_cli.apply_powers(name, group, g, 1, start, tau, coeff) == exp
        # all-infinity input stays infinity
        inf = serde.write_point(G, None, False) * 3
        assert pkg.phase1_cli.apply_powers(name, group, inf, 3, 0, 7) == serde.write_point(G, None, True) * 3
        # non-canonical coordinate (x = q) is rejected
        bad = bytearray(g)
        qb = c.q.to_bytes(c.fq_bytes, "little")
        bad[:c.fq_bytes] = qb
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.apply_powers(name, group, bytes(bad), 1, 0, 7)
        assert e.value.code == -3
        # CheckForCorrectness::OnlyNonZero rejects infinity
        with pytest.raises(pkg.PtauError) as e:
            pkg.phase1_cli.apply_powers(name, group, inf, 3, 0, 7, check_input=pkg.CheckForCorrectness.OnlyNonZero)
        assert e.value.code == -4


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_decompress_parity(name, pkg, cref, golden):
    doc = golden[name]
    c = CURVES[name]
    ch = doc["chunks"][0]
    p = P.Params(c, P.CHUNKED, 0, doc["chunk_size"], doc["power"], doc["batch_size"])
    response = bytes.fromhex(ch["response"])
    nc = b""
    for (nm, grp, off, cnt) in p.layout(True)[0]:
        sz = c.comp_size(grp)
        seg = response[off:off + cnt * sz]
        got = pkg.phase1_cli.decompress(name, grp, seg, cnt)
        assert got == cref.decompress(name, grp, seg, cnt)
        nc += got
    assert pkg.calculate_hash(pkg.calculate_hash(response) + nc).hex() == ch["new_challenge_hash"]
    # an x that is not on the curve must be rejected (x = 0 is off-curve for none/all? search one)
    G = Group(c, 1)
    for xv in range(1, 50):
        x = G.F.from_base(xv)
        rhs = G.F.add(G.F.add(G.F.mul(G.F.sqr(x), x), G.F.mul(G.a, x)), G.b)
        if not G.F.is_square(rhs):
            bad = xv.to_bytes(c.fq_bytes, "little")
            with pytest.raises(pkg.PtauError) as e:
                pkg.phase1_cli.decompress(name, 1, bad, 1)
            assert e.value.code == -3
            break


def test_full_size_bls12_377_tau_g1(pkg, cref):
    """BASELINE config #2 shape: 2^20 BLS12-377 G1 points.  Size-independent properties:
    (1) sampled outputs equal the oracle on the same inputs, (2) the update composes:
    update(update(v, a), b) == update(v, a*b) as a hash of the whole vector, (3) uncompressed and
    compressed outputs decode to the same points on a sample."""
    name, n = "bls12_377", 1 << 20
    c = CURVES[name]
    G = Group(c, 1)
    rng = random.Random(2020)
    s, a, b = (rng.randrange(c.r) for _ in range(3))
    gen = serde.write_point(G, G.gen, False) * n
    v0 = pkg.phase1_cli.apply_powers(name, 1, gen, n, 0, s, None, False, False)          # [s^i G]
    v1c = pkg.phase1_cli.apply_powers(name, 1, v0, n, 0, a, None, False, True)
    t = pkg.phase1_cli.last_timing()
    print("2^20 G1 update: scalar_mul %.1f ms, normalize %.1f ms, h2d %.1f ms, d2h %.1f ms" %
          (t["scalar_mul_ms"], t["normalize_ms"], t["h2d_ms"], t["d2h_ms"]))
    v1u = pkg.phase1_cli.apply_powers(name, 1, v0, n, 0, a, None, False, False)
    idx = sorted(rng.sample(range(n), 48)) + [0, 1, n - 1]
    for i in idx:
        one = v0[i * 96:(i + 1) * 96]
        exp = cref.apply_powers(name, 1, one, 1, i, a, None, False, True)
        assert v1c[i * 48:(i + 1) * 48] == exp, i
        assert cref.decompress(name, 1, exp, 1) == v1u[i * 96:(i + 1) * 96], i
    v2 = pkg.phase1_cli.apply_powers(name, 1, v1u, n, 0, b, None, False, True)
    v12 = pkg.phase1_cli.apply_powers(name, 1, v0, n, 0, a * b % c.r, None, False, True)
    assert pkg.calculate_hash(v2) == pkg.calculate_hash(v12)


@pytest.mark.parametrize("name,group,log2n", [("bls12_377", 1, 20), ("bls12_377", 2, 20), ("bw6", 1, 18), ("bw6", 2, 18),
                                              ("mnt4_753", 1, 16), ("mnt4_753", 2, 16), ("mnt6_753", 1, 16), ("mnt6_753", 2, 16)])
def test_full_size_round_trip(name, group, log2n, pkg, cref):
    """BASELINE-size vectors (configs 2-4: 2^20 BLS12-377, 2^18 BW6-761, 2^16 MNT4 / MNT6, G1 and G2) through size-independent properties:
    a tau-update followed by the tau^-1-update at the same offset is the identity on the bytes, compressed output is the
    uncompressed output with y dropped (same x bytes, flags moved), and the first / last 128 elements are bit-exact
    vs the C oracle."""
    from oracle.pyref import serde
    from oracle.pyref.curves import CURVES
    from oracle.pyref.ec import Group
    c = CURVES[name]
    G = Group(c, group)
    n, start = 1 << log2n, 3 * (1 << log2n)
    u, cs = G.uncomp_size, G.comp_size
    s, tau = 0x1234567890abcdef1234567890abcdef % c.r, (c.r // 7) * 3 + 11
    gen = serde.write_point(G, G.gen, False) * n
    v = pkg.phase1_cli.apply_powers(name, group, gen, n, 0, s, None, False, False)          # a genuine accumulator vector
    w = pkg.phase1_cli.apply_powers(name, group, v, n, start, tau, None, False, False)
    back = pkg.phase1_cli.apply_powers(name, group, w, n, start, pow(tau, -1, c.r), None, False, False)
    assert back == v
    wc = pkg.phase1_cli.apply_powers(name, group, v, n, start, tau, None, False, True)
    if name in ("bls12_377", "bw6") or group == 2:   # the endomorphism path (membership asserted: v is [s^i G]) writes the same bytes as the exact one (MNT twists: GLS through psi = [t - 1])
        endo = int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP
        assert pkg.calculate_hash(pkg.phase1_cli.apply_powers(name, group, v, n, start, tau, None, False, True, batch_exp_mode=endo)) == pkg.calculate_hash(wc)
    fb = c.fq_bytes
    for i in (0, 1, n // 2, n - 1):                      # compressed = x bytes of the uncompressed form, sign flag from y
        xu, xc = bytearray(w[i * u:i * u + cs]), bytearray(wc[i * cs:(i + 1) * cs])
        flag = w[i * u + u - 1] & 0x80
        xu[-1] |= flag
        assert xu == xc
    k = 128
    assert w[:k * u] == cref.apply_powers(name, group, v[:k * u], k, start, tau, None, False, False, nthreads=8)
    assert w[-k * u:] == cref.apply_powers(name, group, v[-k * u:], k, start + n - k, tau, None, False, False, nthreads=8)


@pytest.mark.parametrize("name,group", [("bls12_377", 1), ("bls12_377", 2), ("bw6", 2), ("mnt4_753", 2)])
def test_update_passes_points_at_infinity_through(name, group, pkg, cref):
    """Points at infinity inside a vector (CheckForCorrectness::No lets them through, as the reference's read_batch with
    Validate::No does) stay at infinity and do not disturb their neighbours — on the lane-pair G2 kernels the infinity
    flag lives in one lane's bytes and must reach both lanes."""
    from oracle.pyref import serde
    from oracle.pyref.curves import CURVES
    from oracle.pyref.ec import Group
    c = CURVES[name]
    G = Group(c, group)
    rng = random.Random(77)
    pts = [None if i in (0, 3, 4, 9) else G.mul(G.gen, rng.randrange(1, c.r)) for i in range(10)]
    data = serde.write_batch(G, pts, False)
    tau = rng.randrange(c.r)
    for cout in (True, False):
        got = pkg.phase1_cli.apply_powers(name, group, data, 10, 5, tau, None, False, cout)
        assert got == cref.apply_powers(name, group, data, 10, 5, tau, None, False, cout)
    inf_c = serde.write_point(G, None, True)
    assert got is not None and pkg.phase1_cli.apply_powers(name, group, data, 10, 5, tau, None, False, True)[:G.comp_size] == inf_c


@pytest.mark.parametrize("name", ["bls12_377", "bw6"])
def test_full_mode_contribute_and_verify(name, pkg, tmp_path):
    """ContributionMode::Full (beacon application and the final check of verify_transcript: /root/reference/src/bin/
    verify_transcript.rs:675-696, 745-776, src/bin/control.rs:781-808): one accumulator holds every power — 2^(p+1) - 1
    tauG1 elements, 2^p of the others.  Phase1::computation, transform_pok_and_correctness (file level, proofs of knowledge
    and the chunk-0-style before/after checks, ratio check) and the result's independence of batch_size, against the
    Python oracle."""
    from oracle.pyref import keygen as K
    c = CURVES[name]
    power = 3
    p = P.Params(c, P.FULL, 0, 1 << power, power, 4)
    prm = pkg.Phase1Parameters.new_full(name, 0, power, 4)
    assert prm.accumulator_size == p.accumulator_size and prm.g1_chunk_size == (2 << power) - 1 and prm.other_chunk_size == 1 << power
    ch0 = P.initialization(p)
    assert pkg.phase1_cli.initialization(prm) == ch0
    f = lambda s: str(tmp_path / s)
    open(f("c0"), "wb").write(ch0)
    seed = bytes(range(100, 164))
    pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h"), f("r0"), f("r0.h"), pkg.CheckForCorrectness.Full, pkg.BatchExpMode.Auto, prm, seed)
    tau, alpha, beta, pk, _ = K.key_generation(c, seed, P.calculate_hash(ch0))
    resp = open(f("r0"), "rb").read()
    assert resp == P.calculate_hash(ch0) + P.computation(p, ch0, tau, alpha, beta) + pk
    # batch_size is a memory knob of the reference (it iterates batch_size slices): the bytes do not depend on it
    for bs in (1, 3, 1 << 10):
        prm_b = pkg.Phase1Parameters.new_full(name, 0, power, bs)
        key = pkg.PrivateKey(tau, alpha, beta, pk)
        assert pkg.phase1_cli.computation(ch0, prm_b, key)[64:-len(pk)] == resp[64:-len(pk)]
    pkg.transform_pok_and_correctness(f("c0"), f("v.c.h"), pkg.CheckForCorrectness.Full, f("r0"), f("v.r.h"), pkg.CheckForCorrectness.Full,
                                      f("c1"), f("c1.h"), pkg.SubgroupCheckMode.Direct, True, prm)
    nc = open(f("c1"), "rb").read()
    assert nc == P.new_challenge_from_response(p, resp)
    assert open(f("c1.h"), "rb").read() == P.calculate_hash(nc)
    # Full mode runs the before/after checks: a response whose beta_g2 was moved by another factor is refused
    G2 = Group(c, 2)
    end = p.contribution_size - p.public_key_size
    bad_b = serde.write_point(G2, G2.mul(G2.gen, (beta + 1) % c.r), True)
    bad = resp[:end - G2.comp_size] + bad_b + resp[end:]
    open(f("bad"), "wb").write(bad)
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_pok_and_correctness(f("c0"), f("b.c.h"), pkg.CheckForCorrectness.No, f("bad"), f("b.r.h"), pkg.CheckForCorrectness.OnlyNonZero,
                                          f("bad.nc"), f("bad.nc.h"), pkg.SubgroupCheckMode.Auto, False, prm)
    assert e.value.code == -6
    # the accumulator a Full-mode contribution leaves passes the aggregate ratio check
    pkg.transform_ratios(nc, pkg.CheckForCorrectness.Full, prm)


def test_full_size_verify_chunk_config5(pkg, cref, tmp_path):
    """BASELINE configs[4] shape, one chunk: verification_and_decompress of chunk 0 of the 2^20 BLS12-377 ceremony (3 * 2^20 G1
    + 2^20 + 1 G2 elements, ratio check on, proofs of knowledge on).  The new challenge is checked against the C oracle's
    decompress on sampled ranges of ALL FIVE vectors, the hash chain against BLAKE2b of the files, and a single swapped pair
    deep inside a vector is caught by the ratio check."""
    name, lg = "bls12_377", 20
    n = 1 << lg
    c = CURVES[name]
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, n, 0, lg, n)
    f = lambda s: str(tmp_path / s)
    pkg.new_challenge(f("c0"), f("c0.h"), prm)
    seed = bytes(range(7, 71))
    pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.h"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, seed)
    pkg.transform_pok_and_correctness(f("c0"), f("v.c.h"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.h"), pkg.CheckForCorrectness.OnlyNonZero,
                                      f("c1"), f("c1.h"), pkg.SubgroupCheckMode.Auto, True, prm)
    resp, nc = open(f("r0"), "rb").read(), open(f("c1"), "rb").read()
    assert nc[:64] == pkg.calculate_hash(resp) == open(f("v.r.h"), "rb").read()
    assert open(f("c1.h"), "rb").read() == pkg.calculate_hash(nc)
    g1c, g1u, g2c, g2u = 48, 96, 96, 192
    vecs = [(1, n, g1c, g1u), (2, n, g2c, g2u), (1, n, g1c, g1u), (1, n, g1c, g1u), (2, 1, g2c, g2u)]
    oc, ou, k = 64, 64, 192
    for grp, cnt, cs, us in vecs:
        for lo in sorted({0, max(0, cnt // 2 - k // 2), max(0, cnt - k)}):
            m = min(k, cnt - lo)
            exp = cref.decompress(name, grp, resp[oc + lo * cs:oc + (lo + m) * cs], m)
            assert nc[ou + lo * us:ou + (lo + m) * us] == exp, (grp, lo)
        oc += cnt * cs
        ou += cnt * us
    assert oc == prm.contribution_size - prm.public_key_size and ou == prm.accumulator_size
    # tamper deep inside alpha_g1: swap elements 600000 and 600001 (both still valid subgroup points)
    off = 64 + n * g1c + n * g2c + 600000 * g1c
    bad = bytearray(resp)
    bad[off:off + g1c], bad[off + g1c:off + 2 * g1c] = bad[off + g1c:off + 2 * g1c], bad[off:off + g1c]
    open(f("bad"), "wb").write(bytes(bad))
    del bad, resp, nc
    with pytest.raises(pkg.PtauError) as e:
        pkg.transform_pok_and_correctness(f("c0"), f("b.c.h"), pkg.CheckForCorrectness.No, f("bad"), f("b.r.h"), pkg.CheckForCorrectness.OnlyNonZero,
                                          f("bad.nc"), f("bad.nc.h"), pkg.SubgroupCheckMode.Auto, True, prm)
    assert e.value.code == -6
