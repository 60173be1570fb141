"""The C ABI driven by a plain C program (tests/native/abi_caller.c) — no Python between the caller and the library, as the
Rust FFI shim would drive it: new_challenge -> contribute from a seed -> transform_pok_and_correctness (ratio check, proofs
of knowledge derived and checked inside) -> a second contribution on the endomorphism path.  Every artefact it leaves on
disk is compared with the oracle's."""
import os
import subprocess

import pytest

from conftest import ROOT
from oracle.pyref import keygen as K
from oracle.pyref import phase1 as P
from oracle.pyref.curves import CURVES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,power,cs,idx", [("bls12_377", 4, 8, 0), ("bw6", 3, 4, 1)])
def test_round_trip_through_the_c_abi_without_python(name, power, cs, idx, tmp_path, cref):
    exe = os.path.join(ROOT, "tests", "native", "abi_caller")
    if not os.path.exists(exe):
        pytest.skip("abi_caller not built (run __graft_entry__.build())")
    c = CURVES[name]
    seed = bytes(range(40, 104))
    r = subprocess.run([exe, str(tmp_path), str(c.curve_id), str(power), str(cs), str(idx), seed.hex()], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    out = dict(l.split("=", 1) for l in r.stdout.strip().splitlines())
    p = P.Params(c, P.CHUNKED, idx, cs, power, cs)
    assert int(out["accumulator_size"]) == p.accumulator_size and int(out["contribution_size"]) == p.contribution_size
    assert out["second_contribute_rc"] == "-9"          # create_new: the existing response is never overwritten
    rd = lambda f: open(os.path.join(str(tmp_path), f), "rb").read()
    ch0 = rd("challenge0")
    assert ch0 == P.initialization(p) and out["challenge0_hash"] == P.calculate_hash(ch0).hex()
    # round 0: the key pair of the seed (Python restatement of derive_rng_from_seed + key_generation), exact update
    tau, alpha, beta, pk, _ = K.key_generation(c, seed, P.calculate_hash(ch0))
    resp0 = P.calculate_hash(ch0) + P.computation(p, ch0, tau, alpha, beta) + pk
    assert rd("response0") == resp0
    assert out["response0_hash"] == P.calculate_hash(resp0).hex() == out["verify_response0_hash"]
    assert out["verify_challenge0_hash"] == P.calculate_hash(ch0).hex()
    ch1 = rd("challenge1")
    assert ch1 == P.new_challenge_from_response(p, resp0) and out["challenge1_hash"] == P.calculate_hash(ch1).hex()
    # round 1 (endomorphism path asserted on a verified challenge): same bytes as the oracle's exact arithmetic
    seed1 = bytes([seed[0] ^ 0x5A]) + seed[1:]
    t1, a1, b1, pk1, _ = K.key_generation(c, seed1, P.calculate_hash(ch1))
    body = cref.computation(name, 0, idx, cs, power, ch1, t1, a1, b1)
    end = p.contribution_size - p.public_key_size
    resp1 = rd("response1")
    assert resp1[:64] == P.calculate_hash(ch1) and resp1[64:end] == body[64:end] and resp1[end:] == pk1
    assert out["response1_hash"] == P.calculate_hash(resp1).hex()
