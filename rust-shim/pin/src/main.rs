//! Deferred pin: runs the REFERENCE's own phase-1 path — phase1-cli / phase1 / setup-utils at the
//! operator's pins, arkworks 0.4 — on tiny ceremonies and prints one JSON document with the hash of every artefact (and the
//! artefact itself when it is small).  `tools/pin_reference.sh` stores it as `tests/golden/arkworks_pins.json`;
//! `tests/test_arkworks_pins.py` compares `libptau_hip.so` with it byte for byte.
//!
//! Every call below has the argument list of the operator's call site it is taken from:
//!   new_challenge                  src/bin/new_setup.rs:95-109
//!   contribute (seeded)            src/bin/contribute.rs:783-823
//!   transform_pok_and_correctness  src/bin/contribute.rs:968-986
//!   combine                        src/bin/control.rs:564-568, src/bin/verify_transcript.rs:603-607
//!   transform_ratios               src/bin/control.rs:587-591, src/bin/verify_transcript.rs:646-653, 811-822
//!   the beacon                     src/bin/verify_transcript.rs:675-696, 745-776 (all-zero beacon: e2e/e2ephase1.sh:33-34)
//! Parameters are built as src/utils.rs:326-352 builds them.
//!
//! Usage: `ptau-pin [work_dir] > arkworks_pins.json` (work_dir defaults to ./pin_work; it is created and left behind).
use ark_bls12_377::Bls12_377;
use ark_bw6_761::BW6_761;
use ark_ec::pairing::Pairing;
use ark_ec::AffineRepr;
use ark_mnt4_753::MNT4_753;
use ark_mnt6_753::MNT6_753;
use ark_serialize::CanonicalSerialize;
use blake2::{Blake2b512, Digest};
use phase1::{ContributionMode, Phase1Parameters, ProvingSystem};
use serde_json::{json, Map, Value};
use setup_utils::{
    derive_rng_from_seed, from_slice, upgrade_correctness_check_config, BatchExpMode, SubgroupCheckMode,
    DEFAULT_CONTRIBUTE_CHECK_INPUT_CORRECTNESS, DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS, DEFAULT_VERIFY_CHECK_OUTPUT_CORRECTNESS,
};
use std::fs;
use std::io::Write;
use std::path::{Path, PathBuf};

/// Artefacts up to this size are embedded whole (hex) beside their hash, so that a mismatch can be located, not just seen.
const EMBED_LIMIT: usize = 48 * 1024;
/// The contributor's seed: `derive_rng_from_seed(&exposed_seed[..])` (contribute.rs:783-789) reads its first 32 bytes.
const SEED: [u8; 64] = {
    let mut s = [0u8; 64];
    let mut i = 0;
    while i < 64 {
        s[i] = (i as u8).wrapping_mul(7).wrapping_add(1);
        i += 1;
    }
    s
};
/// e2e/e2ephase1.sh:33-34
const BEACON: [u8; 32] = [0u8; 32];

fn hash_hex(bytes: &[u8]) -> String { hex::encode(Blake2b512::digest(bytes)) }

fn file_entry(path: &Path) -> Value {
    let bytes = fs::read(path).unwrap_or_else(|e| panic!("read {}: {}", path.display(), e));
    let mut m = Map::new();
    m.insert("len".into(), json!(bytes.len()));
    m.insert("blake2b".into(), json!(hash_hex(&bytes)));
    m.insert("head64".into(), json!(hex::encode(&bytes[..bytes.len().min(64)])));
    if bytes.len() <= EMBED_LIMIT {
        m.insert("hex".into(), json!(hex::encode(&bytes)));
    }
    Value::Object(m)
}

/// The 64 raw bytes phase1-cli writes next to an artefact (`*.hash`, src/utils.rs:264-276 hex-encodes them).
fn hash_file_hex(path: &Path) -> String { hex::encode(fs::read(path).unwrap_or_else(|e| panic!("read {}: {}", path.display(), e))) }

fn rm(path: &Path) { let _ = fs::remove_file(path); }

fn s(p: &Path) -> &str { p.to_str().expect("utf-8 path") }

struct Files { dir: PathBuf }
impl Files {
    fn p(&self, name: &str) -> PathBuf { self.dir.join(name) }
}

/// One tiny ceremony on curve `E`: every chunk of round 0 (new_challenge -> contribute -> verify), combine, transform_ratios,
/// the beacon in Full mode, its verification, transform_ratios on the result; and the same contribution in Full mode from a
/// Full-mode initial challenge.
fn ceremony<E: Pairing>(curve: &str, power: usize, chunk_size: usize, work: &Path) -> Value {
    let dir = work.join(format!("{}_p{}_c{}", curve, power, chunk_size));
    let _ = fs::remove_dir_all(&dir);
    fs::create_dir_all(&dir).expect("create work dir");
    let f = Files { dir };
    let batch_size = chunk_size; // new_setup.rs:95-102 passes chunk_size as batch_size
    let proving_system = ProvingSystem::Groth16;
    let full = Phase1Parameters::<E>::new_full(proving_system, power, batch_size);
    let n_chunks = (full.powers_g1_length + chunk_size - 1) / chunk_size; // new_setup.rs:274-277
    let mut chunks = Vec::new();
    let mut list = fs::File::create(f.p("response_list")).expect("create response_list");
    for i in 0..n_chunks {
        let parameters = Phase1Parameters::<E>::new_chunk(ContributionMode::Chunked, i, chunk_size, proving_system, power, batch_size);
        let (ch, chh) = (f.p(&format!("c{}.challenge", i)), f.p(&format!("c{}.challenge.hash", i)));
        let (re, reh) = (f.p(&format!("c{}.response", i)), f.p(&format!("c{}.response.hash", i)));
        let (nc, nch) = (f.p(&format!("c{}.new_challenge", i)), f.p(&format!("c{}.new_challenge.hash", i)));
        let (vch, vrh) = (f.p(&format!("c{}.v.challenge.hash", i)), f.p(&format!("c{}.v.response.hash", i)));
        for p in [&ch, &chh, &re, &reh, &nc, &nch, &vch, &vrh] { rm(p); }
        // new_setup.rs:105-109
        phase1_cli::new_challenge(s(&ch), s(&chh), &parameters);
        // contribute.rs:783-823 — the operator overwrites challenge.hash; here the contribute-side hash goes to its own file
        let rng = derive_rng_from_seed(&SEED[..]);
        let c_chh = f.p(&format!("c{}.contribute.challenge.hash", i));
        rm(&c_chh);
        phase1_cli::contribute(s(&ch), s(&c_chh), s(&re), s(&reh),
                               upgrade_correctness_check_config(DEFAULT_CONTRIBUTE_CHECK_INPUT_CORRECTNESS, false),
                               BatchExpMode::Direct, &parameters, rng);
        // contribute.rs:968-986
        phase1_cli::transform_pok_and_correctness(s(&ch), s(&vch),
                                                  upgrade_correctness_check_config(DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS, false),
                                                  s(&re), s(&vrh),
                                                  upgrade_correctness_check_config(DEFAULT_VERIFY_CHECK_OUTPUT_CORRECTNESS, false),
                                                  s(&nc), s(&nch), SubgroupCheckMode::Auto, true, &parameters);
        let response = fs::read(&re).expect("read response");
        let pk = &response[response.len() - parameters.public_key_size..];
        chunks.push(json!({
            "chunk_index": i,
            "accumulator_size": parameters.accumulator_size,
            "contribution_size": parameters.contribution_size,
            "public_key_size": parameters.public_key_size,
            "g1_chunk_size": parameters.g1_chunk_size,
            "other_chunk_size": parameters.other_chunk_size,
            "challenge": file_entry(&ch),
            "challenge_hash_file": hash_file_hex(&chh),
            "response": file_entry(&re),
            "response_hash_file": hash_file_hex(&reh),
            "public_key": hex::encode(pk),
            "new_challenge": file_entry(&nc),
            "new_challenge_hash_file": hash_file_hex(&nch),
        }));
        writeln!(list, "{}", s(&re)).expect("write response_list"); // control.rs:547-557: "{}\n" per chunk
    }
    drop(list);
    // control.rs:560-568 / verify_transcript.rs:596-607
    let parameters0 = Phase1Parameters::<E>::new_chunk(ContributionMode::Chunked, 0, chunk_size, proving_system, power, batch_size);
    let combined = f.p("combined");
    rm(&combined);
    phase1_cli::combine(s(&f.p("response_list")), s(&combined), &parameters0);
    // control.rs:587-591
    phase1_cli::transform_ratios(s(&combined), DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS, &full);
    // verify_transcript.rs:675-696: the beacon, rng = derive_rng_from_seed(&from_slice(&beacon_hash))
    let (cbh, bre, breh) = (f.p("combined.hash"), f.p("combined_verified_pok_and_correctness"), f.p("combined_verified_pok_and_correctness.hash"));
    for p in [&cbh, &bre, &breh] { rm(p); }
    let rng = derive_rng_from_seed(&from_slice(&BEACON));
    phase1_cli::contribute(s(&combined), s(&cbh), s(&bre), s(&breh),
                           upgrade_correctness_check_config(DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS, false),
                           BatchExpMode::Direct, &full, rng);
    // verify_transcript.rs:745-776
    let (bnc, bnch) = (f.p("combined_verified_pok_and_correctness_new_challenge"), f.p("combined_verified_pok_and_correctness_new_challenge.hash"));
    let (v1, v2) = (f.p("b.v.challenge.hash"), f.p("b.v.response.hash"));
    for p in [&bnc, &bnch, &v1, &v2] { rm(p); }
    phase1_cli::transform_pok_and_correctness(s(&combined), s(&v1),
                                              upgrade_correctness_check_config(DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS, false),
                                              s(&bre), s(&v2),
                                              upgrade_correctness_check_config(DEFAULT_VERIFY_CHECK_OUTPUT_CORRECTNESS, false),
                                              s(&bnc), s(&bnch), SubgroupCheckMode::Auto, true, &full);
    // verify_transcript.rs:811-822
    phase1_cli::transform_ratios(s(&bnc), upgrade_correctness_check_config(DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS, false), &full);
    // Full mode from a Full-mode initial challenge (the shape `control apply-beacon` sees after a one-chunk ceremony)
    let (fch, fchh, fre, freh) = (f.p("full.challenge"), f.p("full.challenge.hash"), f.p("full.response"), f.p("full.response.hash"));
    let fcch = f.p("full.contribute.challenge.hash");
    for p in [&fch, &fchh, &fre, &freh, &fcch] { rm(p); }
    phase1_cli::new_challenge(s(&fch), s(&fchh), &full);
    let rng = derive_rng_from_seed(&SEED[..]);
    phase1_cli::contribute(s(&fch), s(&fcch), s(&fre), s(&freh),
                           upgrade_correctness_check_config(DEFAULT_CONTRIBUTE_CHECK_INPUT_CORRECTNESS, false),
                           BatchExpMode::Direct, &full, rng);
    json!({
        "curve": curve,
        "power": power,
        "chunk_size": chunk_size,
        "batch_size": batch_size,
        "n_chunks": n_chunks,
        "powers_length": full.powers_length,
        "powers_g1_length": full.powers_g1_length,
        "full_accumulator_size": full.accumulator_size,
        "full_contribution_size": full.contribution_size,
        "chunks": chunks,
        "combined": file_entry(&combined),
        "combined_hash_file": hash_file_hex(&cbh),
        "beacon_response": file_entry(&bre),
        "beacon_response_hash_file": hash_file_hex(&breh),
        "beacon_new_challenge": file_entry(&bnc),
        "beacon_new_challenge_hash_file": hash_file_hex(&bnch),
        "full_challenge": file_entry(&fch),
        "full_challenge_hash_file": hash_file_hex(&fchh),
        "full_response": file_entry(&fre),
        "full_response_hash_file": hash_file_hex(&freh),
    })
}

fn generators<E: Pairing>() -> Value {
    let (mut g1, mut g2) = (Vec::new(), Vec::new());
    E::G1Affine::generator().serialize_uncompressed(&mut g1).expect("serialize G1 generator");
    E::G2Affine::generator().serialize_uncompressed(&mut g2).expect("serialize G2 generator");
    let (mut c1, mut c2) = (Vec::new(), Vec::new());
    E::G1Affine::generator().serialize_compressed(&mut c1).expect("serialize G1 generator");
    E::G2Affine::generator().serialize_compressed(&mut c2).expect("serialize G2 generator");
    json!({"g1_uncompressed": hex::encode(g1), "g2_uncompressed": hex::encode(g2),
           "g1_compressed": hex::encode(c1), "g2_compressed": hex::encode(c2)})
}

fn main() {
    let work = PathBuf::from(std::env::args().nth(1).unwrap_or_else(|| "pin_work".to_string()));
    fs::create_dir_all(&work).expect("create work dir");
    let mut cases = Vec::new();
    // powers 4..=8 on the two curves whose generators the library holds; chunk_size = 2^(power-1): 4 chunks, two with all five
    // vectors and two tauG1-only, the last one short by one element (powers_g1_length = 2^(power+1) - 1)
    for power in 4..=8usize {
        cases.push(ceremony::<Bls12_377>("bls12_377", power, 1 << (power - 1), &work));
    }
    for power in 4..=6usize {
        cases.push(ceremony::<BW6_761>("bw6", power, 1 << (power - 1), &work));
    }
    // the shapes e2e/e2ephase1.sh:24-25 runs, scaled to seconds: MNT4-753 one chunk size = 2^power, MNT6-753 small chunks
    cases.push(ceremony::<MNT4_753>("mnt4_753", 4, 1 << 4, &work));
    cases.push(ceremony::<MNT6_753>("mnt6_753", 4, 1 << 2, &work));
    let doc = json!({
        "generator": "rust-shim/pin (phase1-cli / phase1 / setup-utils @ nimiq/snark-setup bd530da9804b628107e29ccd31098ed061d1cd66, arkworks 0.4)",
        "seed": hex::encode(SEED),
        "beacon": hex::encode(BEACON),
        "batch_exp_mode": "direct",
        "subgroup_check_mode": "auto",
        "ratio_check": true,
        "generators": {
            "bls12_377": generators::<Bls12_377>(),
            "bw6": generators::<BW6_761>(),
            "mnt4_753": generators::<MNT4_753>(),
            "mnt6_753": generators::<MNT6_753>(),
        },
        "cases": cases,
    });
    println!("{}", serde_json::to_string_pretty(&doc).expect("serialize JSON"));
}
