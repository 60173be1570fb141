#!/usr/bin/env python3
"""A stand-in for tools/pin_reference.sh's output made by the IN-REPO ORACLE — same JSON schema, NOT a pin.

Purpose: to exercise tests/test_arkworks_pins.py (the consumer of the deferred pin) before a Rust toolchain exists, so that
the day the real file arrives the tests that read it are known to work.  The file says what made it ("generator"), is
written wherever the caller asks (never to tests/golden/arkworks_pins.json — the tests refuse a stand-in there) and is
selected through PTAU_PINS_FILE.  Mirrors rust-shim/pin/src/main.rs step for step, with oracle/pyref in place of
phase1-cli.

    python3 tools/pins_from_oracle.py out.json [curve power chunk_size]...
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.pyref import keygen as K          # noqa: E402
from oracle.pyref import phase1 as P, serde   # noqa: E402
from oracle.pyref.curves import CURVES        # noqa: E402
from oracle.pyref.ec import Group             # noqa: E402

EMBED_LIMIT = 48 * 1024
SEED = bytes((i * 7 + 1) & 0xFF for i in range(64))
BEACON = bytes(32)


def entry(data: bytes):
    e = {"len": len(data), "blake2b": P.calculate_hash(data).hex(), "head64": data[:64].hex()}
    if len(data) <= EMBED_LIMIT:
        e["hex"] = data.hex()
    return e


def contribute(p, challenge, seed):
    digest = P.calculate_hash(challenge)
    tau, alpha, beta, pk, _ = K.key_generation(p.curve, seed, digest)
    return digest + P.computation(p, challenge, tau, alpha, beta) + pk, pk


def ceremony(name, power, chunk_size):
    c = CURVES[name]
    full = P.Params(c, P.FULL, 0, chunk_size, power, chunk_size)
    n_chunks = -(-full.powers_g1_length // chunk_size)
    chunks, cparams, responses = [], [], []
    for i in range(n_chunks):
        p = P.Params(c, P.CHUNKED, i, chunk_size, power, chunk_size)
        ch = P.initialization(p)
        resp, pk = contribute(p, ch, SEED)
        nc = P.new_challenge_from_response(p, resp)
        chunks.append({"chunk_index": i, "accumulator_size": p.accumulator_size, "contribution_size": p.contribution_size,
                       "public_key_size": p.public_key_size, "g1_chunk_size": p.g1_chunk_size, "other_chunk_size": p.other_chunk_size,
                       "challenge": entry(ch), "challenge_hash_file": P.calculate_hash(ch).hex(),
                       "response": entry(resp), "response_hash_file": P.calculate_hash(resp).hex(), "public_key": pk.hex(),
                       "new_challenge": entry(nc), "new_challenge_hash_file": P.calculate_hash(nc).hex()})
        cparams.append(p)
        responses.append(resp)
    combined = P.aggregation(cparams, responses, full)
    bresp, _ = contribute(full, combined, BEACON)
    bnc = P.new_challenge_from_response(full, bresp)
    fch = P.initialization(full)
    fresp, _ = contribute(full, fch, SEED)
    return {"curve": name, "power": power, "chunk_size": chunk_size, "batch_size": chunk_size, "n_chunks": n_chunks,
            "powers_length": full.powers_length, "powers_g1_length": full.powers_g1_length,
            "full_accumulator_size": full.accumulator_size, "full_contribution_size": full.contribution_size, "chunks": chunks,
            "combined": entry(combined), "combined_hash_file": P.calculate_hash(combined).hex(),
            "beacon_response": entry(bresp), "beacon_response_hash_file": P.calculate_hash(bresp).hex(),
            "beacon_new_challenge": entry(bnc), "beacon_new_challenge_hash_file": P.calculate_hash(bnc).hex(),
            "full_challenge": entry(fch), "full_challenge_hash_file": P.calculate_hash(fch).hex(),
            "full_response": entry(fresp), "full_response_hash_file": P.calculate_hash(fresp).hex()}


def generators():
    out = {}
    for name, c in CURVES.items():
        G1, G2 = Group(c, 1), Group(c, 2)
        out[name] = {"g1_uncompressed": serde.write_point(G1, G1.gen, False).hex(), "g2_uncompressed": serde.write_point(G2, G2.gen, False).hex(),
                     "g1_compressed": serde.write_point(G1, G1.gen, True).hex(), "g2_compressed": serde.write_point(G2, G2.gen, True).hex()}
    return out


def main(argv):
    out_path = argv[1]
    if os.path.abspath(out_path) == os.path.join(ROOT, "tests", "golden", "arkworks_pins.json"):
        raise SystemExit("refusing to write an oracle-made stand-in where the real pin belongs")
    specs = [(argv[i], int(argv[i + 1]), int(argv[i + 2])) for i in range(2, len(argv) - 2, 3)] or [("bls12_377", 4, 8)]
    doc = {"generator": "tools/pins_from_oracle.py (IN-REPO ORACLE stand-in — NOT a pin of the reference)", "stand_in": True,
           "seed": SEED.hex(), "beacon": BEACON.hex(), "batch_exp_mode": "direct", "subgroup_check_mode": "auto", "ratio_check": True,
           "generators": generators(), "cases": [ceremony(*s) for s in specs]}
    json.dump(doc, open(out_path, "w"))
    print("wrote", out_path, [(c["curve"], c["power"]) for c in doc["cases"]])


if __name__ == "__main__":
    main(sys.argv)
