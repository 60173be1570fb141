"""GPU parity tests of the whole-transcript tail through the FILE-level entry points, with the reference's argument lists:
phase1_cli::combine(response_list, combined, &chunk_parameters) (control.rs:564-568, verify_transcript.rs:603-607), the beacon
contribution in Full mode seeded by the 32-byte beacon hash (verify_transcript.rs:675-696, control.rs:781-808; all-zero in
e2e/e2ephase1.sh:33-34), transform_pok_and_correctness(Full) (verify_transcript.rs:745-776) and
transform_ratios(file, check, &full_parameters) (verify_transcript.rs:811-822, control.rs:587-591) — every artefact
against oracle/pyref; plus the e2e script's MNT shapes (e2e/e2ephase1.sh:24-25) and malformed public keys."""
import os
import random

import pytest

from conftest import ROOT, initial_challenge
from oracle.pyref import keygen as K
from oracle.pyref import phase1 as P, serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group

pytestmark = pytest.mark.gpu

BEACON = bytes(32)   # e2e/e2ephase1.sh:33-34


def _write_initial(pkg, name, prm, path):
    ch = initial_challenge(pkg, name, prm)
    open(path, "wb").write(ch)
    open(path + ".hash", "wb").write(pkg.calculate_hash(ch))
    return ch


def _round_of_chunks(pkg, name, power, cs, seed, d, verify=None, cref=None):
    """Every chunk of one round contributed from ONE seed (contribute.rs:783-789) through the file-level calls; returns the
    response paths, the chunk parameters (ours, oracle's) and the round's private key."""
    c = CURVES[name]
    full = pkg.Phase1Parameters.new_full(name, 0, power, cs)
    n_chunks = -(-full.powers_g1_length // cs)
    paths, oparams, key = [], [], None
    for i in range(n_chunks):
        prm = pkg.Phase1Parameters.new_chunk(name, 0, i, cs, 0, power, cs)
        f = lambda s: os.path.join(d, "chunk%d.%s" % (i, s))
        ch = _write_initial(pkg, name, prm, f("challenge"))
        pkg.phase1_cli.contribute_seeded(f("challenge"), f("challenge.h"), f("response"), f("response.hash"), pkg.CheckForCorrectness.No,
                                         pkg.BatchExpMode.Auto, prm, seed)
        resp = open(f("response"), "rb").read()
        k = pkg.phase1_cli.key_generation(name, seed, pkg.calculate_hash(ch))
        key = key or (k.tau, k.alpha, k.beta)
        assert (k.tau, k.alpha, k.beta) == key          # the same seed -> the same private key in every chunk
        assert resp[:64] == pkg.calculate_hash(ch) and resp[-prm.public_key_size:] == k.public_key
        if cref is not None:
            ref = cref.computation(name, 0, i, cs, power, ch, k.tau, k.alpha, k.beta, nthreads=8)
            end = prm.contribution_size - prm.public_key_size
            assert resp[64:end] == ref[64:end], "chunk %d differs from the C oracle" % i
        if verify is not None and i in verify:
            pkg.transform_pok_and_correctness(f("challenge"), f("vc.hash"), pkg.CheckForCorrectness.No, f("response"), f("vr.hash"),
                                              pkg.CheckForCorrectness.OnlyNonZero, f("new_challenge"), f("new_challenge.hash"),
                                              pkg.SubgroupCheckMode.Auto, True, prm)
            nc = open(f("new_challenge"), "rb").read()
            assert nc[:64] == pkg.calculate_hash(resp) and open(f("new_challenge.hash"), "rb").read() == pkg.calculate_hash(nc)
        paths.append(f("response"))
        oparams.append(P.Params(c, P.CHUNKED, i, cs, power, cs))
    return paths, oparams, key, full


@pytest.mark.parametrize("name", ["bls12_377", "bw6"])
def test_beacon_chain_file_level(name, pkg, tmp_path):
    """tiny ceremony -> combine FILE -> Full-mode contribute with the all-zero beacon seed -> transform_pok_and_correctness
    (Full) -> transform_ratios FILE: verify_transcript.rs:596-696, 745-776, 811-822, every artefact vs oracle/pyref."""
    c = CURVES[name]
    power, cs = 3, 4
    d = str(tmp_path)
    f = lambda s: os.path.join(d, s)
    paths, oparams, key, full = _round_of_chunks(pkg, name, power, cs, bytes([9]) * 64, d)
    prm0 = pkg.Phase1Parameters.new_chunk(name, 0, 0, cs, 0, power, cs)
    open(f("response_list"), "w").write("".join(p + "\n" for p in paths))
    pkg.phase1_cli.combine_file(f("response_list"), f("combined"), prm0)
    combined = open(f("combined"), "rb").read()
    pfull = P.Params(c, P.FULL, 0, cs, power, cs)
    assert combined == P.aggregation(oparams, [open(p, "rb").read() for p in paths], pfull)
    assert combined[:64] == bytes(64)                   # set_len + aggregation: the header is never written (ptau.h)
    with pytest.raises(pkg.PtauError) as e:             # create_new: the caller removes a stale file first (control.rs:560)
        pkg.phase1_cli.combine_file(f("response_list"), f("combined"), prm0)
    assert e.value.code == -9
    with pytest.raises(pkg.PtauError) as e:             # a list that does not name every chunk
        open(f("short_list"), "w").write(paths[0] + "\n")
        pkg.phase1_cli.combine_file(f("short_list"), f("combined2"), prm0)
    assert e.value.code == -1
    pkg.phase1_cli.transform_ratios_file(f("combined"), pkg.CheckForCorrectness.Full, full)
    pkg.phase1_cli.transform_ratios_file(f("combined"), pkg.CheckForCorrectness.No, full, devices=[0])
    # the beacon: phase1_cli::contribute(COMBINED, COMBINED_HASH, .._VERIFIED_POK_AND_CORRECTNESS, .., rng = beacon)
    pkg.phase1_cli.contribute_seeded(f("combined"), f("combined.hash"), f("beacon_response"), f("beacon_response.hash"),
                                     pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, full, BEACON)
    digest = P.calculate_hash(combined)
    assert open(f("combined.hash"), "rb").read() == digest
    tau, alpha, beta, pk, _ = K.key_generation(c, BEACON, digest)
    want = digest + P.computation(pfull, combined, tau, alpha, beta) + pk
    got = open(f("beacon_response"), "rb").read()
    assert got == want
    assert open(f("beacon_response.hash"), "rb").read() == P.calculate_hash(want)      # the transcript's final hash (:717-729)
    pkg.transform_pok_and_correctness(f("combined"), f("v.combined.hash"), pkg.CheckForCorrectness.No, f("beacon_response"),
                                      f("v.beacon_response.hash"), pkg.CheckForCorrectness.OnlyNonZero, f("beacon_new_challenge"),
                                      f("beacon_new_challenge.hash"), pkg.SubgroupCheckMode.Auto, True, full)
    tm = pkg.phase1_cli.last_timing(0)
    assert tm["ratio_checks"] == 4 and tm["pok_checks"] == 8
    nc = open(f("beacon_new_challenge"), "rb").read()
    assert nc == P.new_challenge_from_response(pfull, want)
    assert open(f("beacon_new_challenge.hash"), "rb").read() == P.calculate_hash(nc)
    pkg.phase1_cli.transform_ratios_file(f("beacon_new_challenge"), pkg.CheckForCorrectness.No, full)
    # element 1 of tauG1 is now (tau_round * tau_beacon) G
    G1 = Group(c, 1)
    u = G1.uncomp_size
    assert nc[64 + u:64 + 2 * u] == serde.write_point(G1, G1.mul(G1.gen, key[0] * tau % c.r), False)
    # a tampered accumulator file is refused by the file-level check
    bad = bytearray(nc)
    bad[64 + 2 * u:64 + 3 * u], bad[64 + 3 * u:64 + 4 * u] = bad[64 + 3 * u:64 + 4 * u], bad[64 + 2 * u:64 + 3 * u]
    open(f("tampered"), "wb").write(bytes(bad))
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_file(f("tampered"), pkg.CheckForCorrectness.No, full)
    assert e.value.code == -6
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_devices(bytes(bad), pkg.CheckForCorrectness.No, full, [0])
    assert e.value.code == -6


@pytest.mark.parametrize("name", ["bls12_377", "bw6"])
def test_transform_ratios_in_process_slices(name, pkg, tmp_path):
    """ptau_transform_ratios_devices with MORE THAN ONE slice (control.rs:587-591 on a node with several GPUs; here one GPU
    listed three times, its slices queue on the device lock): power-4 accumulator, tauG1 = 31 points = 30 pairs in slices
    [0,10) [10,20) [20,30) (points 0..10, 10..20, 20..30: neighbours overlap by one element), the other vectors 15 pairs in
    slices of 5.  The honest accumulator is accepted; a swap ON a slice boundary, a swap inside the last slice and a foreign
    point in the last slice are rejected by the summed partial pairs; check_input = Full rejects an off-subgroup element of
    slice 2 with the membership code; buffer and FILE forms."""
    c = CURVES[name]
    power = 4
    d = str(tmp_path)
    f = lambda s: os.path.join(d, s)
    full = pkg.Phase1Parameters.new_full(name, 0, power, 1 << power)
    _write_initial(pkg, name, full, f("c0"))
    pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                     full, bytes([3]) * 64)
    pkg.transform_pok_and_correctness(f("c0"), f("v.c0.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r0.hash"),
                                      pkg.CheckForCorrectness.OnlyNonZero, f("acc"), f("acc.hash"), pkg.SubgroupCheckMode.Auto, True, full)
    acc = open(f("acc"), "rb").read()
    pfull = P.Params(c, P.FULL, 0, 1 << power, power, 1 << power)
    assert acc == P.new_challenge_from_response(pfull, open(f("r0"), "rb").read())
    G1, G2 = Group(c, 1), Group(c, 2)
    u1, u2 = G1.uncomp_size, G2.uncomp_size
    n1, n2 = full.powers_g1_length, full.powers_length
    assert (n1, n2) == (31, 16)
    off = {"tau_g1": 64, "tau_g2": 64 + n1 * u1, "alpha_g1": 64 + n1 * u1 + n2 * u2, "beta_g1": 64 + n1 * u1 + n2 * u2 + n2 * u1}
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 7):
        pkg.phase1_cli.transform_ratios_devices(acc, pkg.CheckForCorrectness.No, full, devs)
        pkg.phase1_cli.transform_ratios_devices(acc, pkg.CheckForCorrectness.Full, full, devs)
    pkg.phase1_cli.transform_ratios_file(f("acc"), pkg.CheckForCorrectness.Full, full, devices=[0, 0, 0])

    def swapped(vec, usz, i, j):
        b = bytearray(acc)
        o = off[vec]
        b[o + i * usz:o + (i + 1) * usz], b[o + j * usz:o + (j + 1) * usz] = acc[o + j * usz:o + (j + 1) * usz], acc[o + i * usz:o + (i + 1) * usz]
        return bytes(b)
    cases = [
        ("tau_g1 boundary 0|1 (element 10 is shared by slices 0 and 1)", swapped("tau_g1", u1, 10, 11)),
        ("tau_g1 boundary 1|2", swapped("tau_g1", u1, 19, 20)),
        ("tau_g1 inside the last slice", swapped("tau_g1", u1, 25, 27)),
        ("tau_g1 last two elements", swapped("tau_g1", u1, 29, 30)),
        ("tau_g2 boundary 0|1", swapped("tau_g2", u2, 5, 6)),
        ("tau_g2 inside the last slice", swapped("tau_g2", u2, 12, 14)),
        ("alpha_g1 boundary 1|2", swapped("alpha_g1", u1, 9, 10)),
        ("beta_g1 inside the last slice", swapped("beta_g1", u1, 13, 15)),
    ]
    for tag, bad in cases:
        for devs in ([0], [0, 0, 0]):
            with pytest.raises(pkg.PtauError) as e:
                pkg.phase1_cli.transform_ratios_devices(bad, pkg.CheckForCorrectness.No, full, devs)
            assert e.value.code == -6, (tag, devs)
    open(f("bad"), "wb").write(cases[0][1])
    with pytest.raises(pkg.PtauError) as e:
        pkg.phase1_cli.transform_ratios_file(f("bad"), pkg.CheckForCorrectness.No, full, devices=[0, 0, 0])
    assert e.value.code == -6
    # an element of slice 2 that is on the curve but outside the prime-order subgroup
    rng = random.Random(4)
    for vec, G, usz, idx in (("tau_g1", G1, u1, 26), ("tau_g2", G2, u2, 13)):
        pt = serde.write_point(G, _off_subgroup(G, rng), False)
        b = bytearray(acc)
        b[off[vec] + idx * usz:off[vec] + (idx + 1) * usz] = pt
        for devs in ([0], [0, 0, 0]):
            with pytest.raises(pkg.PtauError) as e:
                pkg.phase1_cli.transform_ratios_devices(bytes(b), pkg.CheckForCorrectness.Full, full, devs)
            assert e.value.code == -5, (vec, devs)
            with pytest.raises(pkg.PtauError) as e:      # unchecked input: the ratio check itself refuses it
                pkg.phase1_cli.transform_ratios_devices(bytes(b), pkg.CheckForCorrectness.No, full, devs)
            assert e.value.code == -6, (vec, devs)


@pytest.mark.parametrize("name,cs", [("mnt4_753", 1 << 12), ("mnt6_753", 1 << 8)])
def test_e2e_script_mnt_shapes_file_level(name, cs, pkg, cref, tmp_path):
    """The shapes e2e/e2ephase1.sh:24-25 really runs — MNT4-753 --powers 12 --chunk-size 12 (2 chunks), MNT6-753 --powers 12
    --chunk-size 8 (32 chunks: 16 full + 16 tauG1-only) — through the file-level entry points with caller-supplied
    generators: contribute from a seed (bit-exact vs the C oracle), transform_pok_and_correctness with ratio_check on the
    first / last full and tauG1-only chunks, combine FILE vs the oracle's aggregation on sampled ranges, transform_ratios FILE."""
    c = CURVES[name]
    power = 12
    d = str(tmp_path)
    f = lambda s: os.path.join(d, s)
    full = pkg.Phase1Parameters.new_full(name, 0, power, cs)
    n_chunks = -(-full.powers_g1_length // cs)
    half = n_chunks // 2
    verify = {0, half - 1, half, n_chunks - 1}
    paths, oparams, key, full = _round_of_chunks(pkg, name, power, cs, bytes([3]) * 64, d, verify=verify, cref=cref)
    # a tauG1-only chunk: ratio_check has no G2 pair of its own (ratio_checks == 0 says "not performed") unless the caller
    # hands over (tauG2[0], tauG2[1]) of the same contribution
    i = n_chunks - 1
    prm = pkg.Phase1Parameters.new_chunk(name, 0, i, cs, 0, power, cs)
    assert pkg.phase1_cli.last_timing(0)["ratio_checks"] == 0
    G1, G2 = Group(c, 1), Group(c, 2)
    nc0 = open(os.path.join(d, "chunk0.new_challenge"), "rb").read()
    prm0 = pkg.Phase1Parameters.new_chunk(name, 0, 0, cs, 0, power, cs)
    off2 = 64 + prm0.g1_chunk_size * G1.uncomp_size
    pair = nc0[off2:off2 + 2 * G2.uncomp_size]
    g = lambda s: os.path.join(d, "chunk%d.%s" % (i, s))
    pkg.transform_pok_and_correctness(g("challenge"), g("vc2.hash"), pkg.CheckForCorrectness.No, g("response"), g("vr2.hash"),
                                      pkg.CheckForCorrectness.OnlyNonZero, g("new_challenge2"), g("new_challenge2.hash"),
                                      pkg.SubgroupCheckMode.Auto, True, prm, tau_g2_check=pair)
    assert pkg.phase1_cli.last_timing(0)["ratio_checks"] == 1
    open(f("response_list"), "w").write("".join(p + "\n" for p in paths))
    pkg.phase1_cli.combine_file(f("response_list"), f("combined"), prm0)
    combined = open(f("combined"), "rb").read()
    assert len(combined) == full.accumulator_size and combined[:64] == bytes(64)
    # sampled against the oracle's decompression: head and tail of tauG1, head of tauG2, the lone betaG2
    r0, rl = open(paths[0], "rb").read(), open(paths[-1], "rb").read()
    u1, c1, u2, c2 = G1.uncomp_size, G1.comp_size, G2.uncomp_size, G2.comp_size
    dec = lambda G, b, n: serde.write_batch(G, serde.read_batch(G, b, True), False)
    assert combined[64:64 + 4 * u1] == dec(G1, r0[64:64 + 4 * c1], 4)
    nl = oparams[-1].g1_chunk_size
    end1 = 64 + full.powers_g1_length * u1
    assert combined[end1 - 3 * u1:end1] == dec(G1, rl[64 + (nl - 3) * c1:64 + nl * c1], 3)
    assert combined[end1:end1 + 2 * u2] == dec(G2, r0[64 + cs * c1:64 + cs * c1 + 2 * c2], 2)
    lay = {nm: o for (nm, grp, o, cnt) in oparams[0].layout(True)[0]}
    assert combined[-u2:] == dec(G2, r0[lay["beta_g2"]:lay["beta_g2"] + c2], 1)
    pkg.phase1_cli.transform_ratios_file(f("combined"), pkg.CheckForCorrectness.No, full)
    # tauG1[1] = tau G
    assert combined[64 + u1:64 + 2 * u1] == serde.write_point(G1, G1.mul(G1.gen, key[0]), False)


def test_malformed_public_key_is_refused(pkg, tmp_path):
    """The PublicKey at the tail of a response is the contributor's bytes: an off-curve point, a point outside the
    prime-order subgroup, or a bad caller-supplied g2_s never reaches the pairing (PTAU_E_POINT / PTAU_E_NOT_IN_SUBGROUP)."""
    name, power, cs = "bls12_377", 3, 8
    c = CURVES[name]
    G1, G2 = Group(c, 1), Group(c, 2)
    d = str(tmp_path)
    f = lambda s: os.path.join(d, s)
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, cs, 0, power, cs)
    _write_initial(pkg, name, prm, f("challenge"))
    pkg.phase1_cli.contribute_seeded(f("challenge"), f("challenge.h"), f("response"), f("response.hash"), pkg.CheckForCorrectness.No,
                                     pkg.BatchExpMode.Auto, prm, bytes([5]) * 64)
    resp = open(f("response"), "rb").read()
    pk_off = prm.contribution_size - prm.public_key_size
    u1, u2 = G1.uncomp_size, G2.uncomp_size
    rng = random.Random(11)

    def run(tag, response, g2_s=None):
        open(f(tag), "wb").write(response)
        pkg.transform_pok_and_correctness(f("challenge"), f(tag + ".ch"), pkg.CheckForCorrectness.No, f(tag), f(tag + ".rh"),
                                          pkg.CheckForCorrectness.OnlyNonZero, f(tag + ".nc"), f(tag + ".nc.hash"),
                                          pkg.SubgroupCheckMode.Auto, True, prm, g2_s=g2_s)
    run("honest", resp)
    x, y = serde.read_point(G1, resp[pk_off:pk_off + u1], False)
    cases = [
        ("g1_off_curve", pk_off, serde.write_point(G1, (x, G1.F.add(y, G1.F.one())), False), -3),
        ("g1_off_subgroup", pk_off + u1, serde.write_point(G1, _off_subgroup(G1, rng), False), -5),
        ("g2_off_subgroup", pk_off + 6 * u1 + u2, serde.write_point(G2, _off_subgroup(G2, rng), False), -5),
    ]
    for tag, off, pt, code in cases:
        bad = bytearray(resp)
        bad[off:off + len(pt)] = pt
        with pytest.raises(pkg.PtauError) as e:
            run(tag, bytes(bad))
        assert e.value.code == code, tag
        with pytest.raises(pkg.PtauError) as e:      # the standalone (host) form validates too
            pkg.phase1_cli.verify_pok(open(f("challenge"), "rb").read(), bytes(bad), None, prm)
        assert e.value.code == code, tag
    # caller-supplied g2_s: honest points pass, one off the subgroup is refused
    g2_s = pkg.phase1_cli.compute_g2_s(name, pkg.calculate_hash(open(f("challenge"), "rb").read()), resp[pk_off:])
    run("honest_g2s", resp, g2_s=g2_s)
    bad_s = bytearray(g2_s)
    bad_s[u2:2 * u2] = serde.write_point(G2, _off_subgroup(G2, rng), False)
    with pytest.raises(pkg.PtauError) as e:
        run("bad_g2s", resp, g2_s=bytes(bad_s))
    assert e.value.code == -5


def _off_subgroup(G, rng):
    F = G.F
    while True:
        x = tuple(rng.randrange(G.curve.q) for _ in range(F.deg))
        rhs = F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b)
        y = F.sqrt(rhs)
        if y is not None and not G.in_subgroup((x, y)):
            return (x, y)


def test_abi_caller_whole_transcript_tail(tmp_path):
    """The same tail driven by the plain-C caller (no Python between caller and library): every chunk, combine FILE, beacon,
    Full-mode verification, transform_ratios FILE over every device — hashes vs oracle/pyref."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "native", "abi_caller")
    if not os.path.exists(exe):
        pytest.skip("abi_caller not built (run __graft_entry__.build())")
    name, power, cs = "bls12_377", 3, 4
    c = CURVES[name]
    seed = bytes(range(40, 104))
    r = subprocess.run([exe, str(tmp_path), str(c.curve_id), str(power), str(cs), "0", seed.hex(), "tail"], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, PTAU_TEST_DEVICES="0,0"))
    assert r.returncode == 0, r.stderr
    out = dict(l.split("=", 1) for l in r.stdout.strip().splitlines())
    assert out["tail"] == "ok" and out["second_combine_rc"] == "-9" and out["ratio_slices"] == "2"
    rd = lambda s: open(os.path.join(str(tmp_path), s), "rb").read()
    pfull = P.Params(c, P.FULL, 0, cs, power, cs)
    n_chunks = pfull.num_chunks()
    oparams = [P.Params(c, P.CHUNKED, i, cs, power, cs) for i in range(n_chunks)]
    combined = rd("combined")
    assert combined == P.aggregation(oparams, [rd("chunk%d.response" % i) for i in range(n_chunks)], pfull)
    digest = P.calculate_hash(combined)
    assert out["combined_hash"] == digest.hex()
    tau, alpha, beta, pk, _ = K.key_generation(c, BEACON, digest)
    want = digest + P.computation(pfull, combined, tau, alpha, beta) + pk
    assert rd("combined_verified_pok_and_correctness") == want and out["beacon_response_hash"] == P.calculate_hash(want).hex()
    nc = rd("combined_verified_pok_and_correctness_new_challenge")
    assert nc == P.new_challenge_from_response(pfull, want) and out["beacon_new_challenge_hash"] == P.calculate_hash(nc).hex()


@pytest.mark.gpu
def test_concurrent_lanes_write_the_same_files(pkg, tmp_path):
    """Six in-process lanes on one GPU (the operator's --max-in-process-lane, contribute.rs:467-492), verifying and contributing
    at the same time, three rounds: every lane's new challenge / response and every .hash file is byte-identical to what one
    lane alone produces.  The device sections of the lanes interleave; their device-to-host copies are queued and not waited
    for, output buffers and MSM-tail slots are reused from lane to lane (capi.hip CopyTrack, bufev, TailSlot) — a missing
    dependency there would show as a corrupted file here."""
    import hashlib
    import os
    import threading
    name, lg = "bls12_377", 17
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, n, 0, lg, n)
    f = lambda s: str(tmp_path / s)
    seed0, seed1 = b"lanes seed 0".ljust(64, b"."), b"lanes seed 1".ljust(64, b".")
    pkg.new_challenge(f("c0"), f("c0.hash"), prm)
    pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm, seed0, device=0)
    digest = lambda p: hashlib.blake2b(open(p, "rb").read()).digest()

    def verify(tag):
        pkg.transform_pok_and_correctness(f("c0"), f(tag + ".c.hash"), pkg.CheckForCorrectness.No, f("r0"), f(tag + ".r.hash"),
                                          pkg.CheckForCorrectness.OnlyNonZero, f(tag + ".c1"), f(tag + ".c1.hash"), pkg.SubgroupCheckMode.Auto,
                                          True, prm, device=0)

    def contribute(tag):
        pkg.phase1_cli.contribute_seeded(f("ref.c1"), f(tag + ".h"), f(tag + ".r1"), f(tag + ".r1.hash"), pkg.CheckForCorrectness.No,
                                         pkg.BatchExpMode.Auto, prm, seed1, device=0)
    verify("ref")
    contribute("refc")
    want_v = [digest(f("ref.c1")), digest(f("ref.c1.hash")), digest(f("ref.c.hash")), digest(f("ref.r.hash"))]
    want_c = [digest(f("refc.r1")), digest(f("refc.r1.hash")), digest(f("refc.h"))]
    assert open(f("ref.c1"), "rb").read(64) == open(f("ref.r.hash"), "rb").read()
    errors = []
    for rnd in range(3):
        def lane(k):
            try:
                tag = "L%d_%d" % (rnd, k)
                if k % 2 == 0:
                    verify(tag)
                    got = [digest(f(tag + ".c1")), digest(f(tag + ".c1.hash")), digest(f(tag + ".c.hash")), digest(f(tag + ".r.hash"))]
                    if got != want_v:
                        errors.append((rnd, k, "verify"))
                    for s in (".c1", ".c1.hash", ".c.hash", ".r.hash"):
                        os.remove(f(tag + s))
                else:
                    contribute(tag)
                    got = [digest(f(tag + ".r1")), digest(f(tag + ".r1.hash")), digest(f(tag + ".h"))]
                    if got != want_c:
                        errors.append((rnd, k, "contribute"))
                    for s in (".r1", ".r1.hash", ".h"):
                        os.remove(f(tag + s))
            except Exception as e:   # noqa: BLE001 - reported below, with the lane
                errors.append((rnd, k, repr(e)))
        th = [threading.Thread(target=lane, args=(k,)) for k in range(6)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    assert not errors, errors


def test_writers_wait_for_another_lanes_copy_and_timings_stay_with_their_call(pkg, tmp_path):
    """A verification queues the device-to-host copies of its decompressed vectors and releases the device WITHOUT waiting for
    them (capi.hip CopyTrack); the next holder of the device lock — whatever entry point — must order its writes to the two
    output staging buffers behind those copies (claim_out_buffer).  PTAU_TEST_COPY_DELAY_MS holds every copy back by 120 ms, so
    the copies certainly outlive the verification's device section; a second lane (contribute / decompress / apply_powers on
    the same device, started at the same time, i.e. waiting for the lock) then runs while they are still queued.  The new
    challenge must equal the undisturbed one.  Also: each lane reads ITS OWN call's record from ptau_last_timing — the
    verification 4 ratio checks + 8 proof-of-knowledge checks, the other lane none (the 'ratio check not performed' signal of
    ptau.h must not come from somebody else's call)."""
    import hashlib
    import threading
    name, lg = "bls12_377", 14
    n = 1 << lg
    c = CURVES[name]
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, n, 0, lg, n)
    f = lambda s: str(tmp_path / s)
    pkg.new_challenge(f("c0"), f("c0.hash"), prm)
    pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm,
                                     b"copy-order".ljust(64, b"."), device=0)
    digest = lambda p: hashlib.blake2b(open(p, "rb").read()).digest()

    def verify(tag):
        pkg.transform_pok_and_correctness(f("c0"), f(tag + ".c.hash"), pkg.CheckForCorrectness.No, f("r0"), f(tag + ".r.hash"),
                                          pkg.CheckForCorrectness.OnlyNonZero, f(tag + ".c1"), f(tag + ".c1.hash"), pkg.SubgroupCheckMode.Auto,
                                          True, prm, device=0)
        return pkg.phase1_cli.last_timing(0)
    tm = verify("ref")
    assert tm["ratio_checks"] == 4 and tm["pok_checks"] == 8
    want = digest(f("ref.c1"))
    resp = open(f("r0"), "rb").read()
    G1 = Group(c, 1)
    comp_g1 = resp[64:64 + n * G1.comp_size]
    unc_g1 = open(f("ref.c1"), "rb").read()[64:64 + n * G1.uncomp_size]

    def other_contribute(tag):
        pkg.phase1_cli.contribute_seeded(f("ref.c1"), f(tag + ".h"), f(tag + ".r1"), f(tag + ".r1.hash"), pkg.CheckForCorrectness.No,
                                         pkg.BatchExpMode.Auto, prm, b"second".ljust(64, b"."), device=0)

    def other_decompress(tag):
        assert pkg.phase1_cli.decompress(name, 1, comp_g1, n, device=0) == unc_g1

    def other_apply_powers(tag):
        pkg.phase1_cli.apply_powers(name, 1, unc_g1, n, 0, 12345, None, device=0)
    os.environ["PTAU_TEST_COPY_DELAY_MS"] = "120"
    try:
        for rnd, other in enumerate((other_contribute, other_decompress, other_apply_powers, other_contribute)):
            got, errors = {}, []

            def lane_v():
                try:
                    got["v"] = verify("d%d" % rnd)
                except Exception as e:   # noqa: BLE001
                    errors.append(repr(e))

            def lane_o():
                try:
                    other("o%d" % rnd)
                    got["o"] = pkg.phase1_cli.last_timing(0)
                except Exception as e:   # noqa: BLE001
                    errors.append(repr(e))
            tv, to = threading.Thread(target=lane_v), threading.Thread(target=lane_o)
            tv.start()
            to.start()
            tv.join()
            to.join()
            assert not errors, errors
            assert digest(f("d%d.c1" % rnd)) == want, "round %d (%s): the new challenge was overwritten under its copy" % (rnd, other.__name__)
            assert digest(f("d%d.c1.hash" % rnd)) == digest(f("ref.c1.hash"))
            assert got["v"]["ratio_checks"] == 4 and got["v"]["pok_checks"] == 8, got["v"]
            assert got["o"]["ratio_checks"] == 0 and got["o"]["pok_checks"] == 0 and got["o"]["points"] > 0, got["o"]
            assert got["v"]["d2h_ms"] > 0 and got["v"]["decompress_ms"] > 0
    finally:
        del os.environ["PTAU_TEST_COPY_DELAY_MS"]
    assert digest(f("o0.r1")) == digest(f("o3.r1"))      # and the lane that came second was not disturbed either


def test_prefetch_events_are_read_before_they_are_recorded_again(pkg, tmp_path):
    """Regression check for the event race fixed in round 3 (capi.hip run_verify_vector / run_apply_powers_host): the brackets
    of the NEXT vector's host-to-device prefetch (pev[0], pev[1]) are re-recorded for every vector, so the elapsed time of
    vector k's copy must be read — behind a host wait for pev[1] — BEFORE vector k + 1's prefetch re-records them; reading
    after is hipErrorNotReady (the call fails) or the time of another copy.  Both prefetch paths (verification: compressed
    response vectors; contribution: uncompressed challenge vectors), five passes each: every call succeeds and its h2d_ms is
    finite, non-negative and below the call's wall time."""
    import math
    import time
    name, lg = "bls12_377", 13
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk(name, 0, 0, n, 0, lg, n)
    f = lambda s: str(tmp_path / s)
    pkg.new_challenge(f("c0"), f("c0.hash"), prm)
    for k in range(5):
        t0 = time.perf_counter()
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h%d" % k), f("r%d" % k), f("r%d.hash" % k), pkg.CheckForCorrectness.No,
                                         pkg.BatchExpMode.Auto, prm, bytes([k + 1]) * 64, device=0)
        wall = (time.perf_counter() - t0) * 1e3
        tm = pkg.phase1_cli.last_timing(0)
        assert math.isfinite(tm["h2d_ms"]) and 0 <= tm["h2d_ms"] < wall and tm["kernel_ms"] > 0, tm
        t0 = time.perf_counter()
        pkg.transform_pok_and_correctness(f("c0"), f("v%d.c.hash" % k), pkg.CheckForCorrectness.No, f("r%d" % k), f("v%d.r.hash" % k),
                                          pkg.CheckForCorrectness.OnlyNonZero, f("v%d.c1" % k), f("v%d.c1.hash" % k), pkg.SubgroupCheckMode.Auto,
                                          True, prm, device=0)
        wall = (time.perf_counter() - t0) * 1e3
        tm = pkg.phase1_cli.last_timing(0)
        assert math.isfinite(tm["h2d_ms"]) and 0 <= tm["h2d_ms"] < wall and tm["kernel_ms"] > 0 and tm["decompress_ms"] > 0, tm
