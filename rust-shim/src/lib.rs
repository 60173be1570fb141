//! Rust binding of `libptau_hip.so` (C ABI: `include/ptau.h`) for nimiq/snark-setup-operator.
//!
//! The operator reaches the ceremony math through five `phase1_cli` functions
//! (`src/bin/contribute.rs:811-823, 968-986`, `src/bin/new_setup.rs:105-109`, `src/bin/control.rs:564-568, 587-591`,
//! `src/bin/verify_transcript.rs:466-484, 603-607, 812-822`).  [`phase1_cli`] below offers the same five names with the
//! same argument lists (file paths, `CheckForCorrectness`, `BatchExpMode`, `SubgroupCheckMode`, `ratio_check`,
//! parameters) and the same error behaviour — they **panic** with the library's message, which the operator's
//! `spawn_quiet(..).join()` turns into `ContributeError::FailedRunningContributeError` / `…VerificationError`
//! exactly as today (`contribute.rs:809-856, 987-1010`).
//!
//! `rng` becomes a seed: `derive_rng_from_seed(&seed)` (`contribute.rs:783-789`) is the only way the operator builds its
//! generator, and the library re-derives the same ChaCha20 stream (`ptau_contribute_file_seeded`).  A caller that wants
//! arkworks' own key generation bit for bit passes the key pair instead ([`phase1_cli::contribute_with_key`]) and the
//! `compute_g2_s` points to the verifier (`g2_s` argument) — see INTEGRATION.md §3.
#![allow(clippy::too_many_arguments)]

use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};
use std::path::Path;

/// `ptau_params` — the ceremony `Parameters` JSON (`src/data_structs.rs:123-131`) + chunk index.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtauParams {
    pub curve: i32,             // 0 bls12_377, 1 bw6, 2 mnt4_753, 3 mnt6_753 (contribute.rs:1156-1211)
    pub proving_system: i32,    // 0 groth16, 1 marlin
    pub contribution_mode: i32, // 0 chunked, 1 full
    pub power: u32,
    pub chunk_index: u64,
    pub chunk_size: u64,
    pub batch_size: u64,
}

/// `ptau_sizes` — the `Phase1Parameters` fields the operator reads (`new_setup.rs:274-277`, `utils.rs:526-532`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtauSizes {
    pub powers_length: u64,
    pub powers_g1_length: u64,
    pub g1_chunk_size: u64,
    pub other_chunk_size: u64,
    pub accumulator_size: u64,
    pub contribution_size: u64,
    pub public_key_size: u64,
    pub hash_size: u64,
    pub start: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtauTiming {
    pub h2d_ms: f64,
    pub kernel_ms: f64,
    pub d2h_ms: f64,
    pub scalar_mul_ms: f64,
    pub normalize_ms: f64,
    pub points: u64,
    pub decompress_ms: f64,
    pub subgroup_ms: f64,
    pub msm_ms: f64,
    pub ratio_checks: u32,
    pub pok_checks: u32,
    /// latency of the MSM tails that ran beside the next vector's kernels (verify path)
    pub msm_tail_ms: f64,
}

/// What one `transform_pok_and_correctness` call checked (from that call's own `PtauTiming`).
#[derive(Clone, Copy, Debug)]
pub struct VerifyReport {
    /// power-ratio (same-ratio pairing) checks run; 0 with `ratio_check = true` = NOT performed on this chunk
    pub ratio_checks: u32,
    /// proofs of knowledge + chunk-0 before/after checks run
    pub pok_checks: u32,
    /// device time of the call's kernels, ms
    pub device_ms: f64,
}

pub const PTAU_BATCH_EXP_INPUT_IN_SUBGROUP: c_int = 0x100;

extern "C" {
    pub fn ptau_strerror(code: c_int) -> *const c_char;
    pub fn ptau_device_count() -> c_int;
    pub fn ptau_device_info(device: c_int, name: *mut c_char, name_len: usize, compute_units: *mut c_int, hbm_bytes: *mut u64) -> c_int;
    pub fn ptau_params_sizes(p: *const PtauParams, out: *mut PtauSizes) -> c_int;
    pub fn ptau_point_size(curve: c_int, group: c_int, compressed: c_int) -> c_int;
    pub fn ptau_fr_limbs(curve: c_int) -> c_int;
    pub fn ptau_blake2b512(data: *const u8, len: usize, out: *mut u8);
    pub fn ptau_apply_powers(curve: c_int, group: c_int, input: *const u8, compressed_in: c_int, output: *mut u8, compressed_out: c_int, n: u64, start: u64, tau: *const u64, coeff: *const u64, check_input: c_int, batch_exp_mode: c_int, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_phase1_computation(p: *const PtauParams, input: *const u8, in_len: usize, output: *mut u8, out_len: usize, tau: *const u64, alpha: *const u64, beta: *const u64, check_input: c_int, batch_exp_mode: c_int, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_contribute_file(p: *const PtauParams, challenge_path: *const c_char, challenge_hash_path: *const c_char, response_path: *const c_char, response_hash_path: *const c_char, tau: *const u64, alpha: *const u64, beta: *const u64, public_key: *const u8, public_key_len: usize, check_input: c_int, batch_exp_mode: c_int, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_contribute_file_seeded(p: *const PtauParams, challenge_path: *const c_char, challenge_hash_path: *const c_char, response_path: *const c_char, response_hash_path: *const c_char, seed: *const u8, seed_len: usize, check_input: c_int, batch_exp_mode: c_int, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_transform_pok_and_correctness_file(p: *const PtauParams, challenge_path: *const c_char, challenge_hash_path: *const c_char, check_input: c_int, response_path: *const c_char, response_hash_path: *const c_char, check_output: c_int, new_challenge_path: *const c_char, new_challenge_hash_path: *const c_char, subgroup_check_mode: c_int, ratio_check: c_int, g2_s: *const u8, tau_g2_check: *const u8, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_new_challenge_file(p: *const PtauParams, challenge_path: *const c_char, challenge_hash_path: *const c_char, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_new_challenge_with_generators(p: *const PtauParams, g1_generator: *const u8, g2_generator: *const u8, out: *mut u8, out_len: usize, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_combine(full: *const PtauParams, chunk_size: u64, responses: *const *const u8, response_lens: *const usize, n_chunks: u64, combined: *mut u8, combined_len: usize, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_transform_ratios(full: *const PtauParams, combined: *const u8, combined_len: usize, check_input: c_int, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_combine_file(chunk0: *const PtauParams, response_list_path: *const c_char, combined_path: *const c_char, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_transform_ratios_file(full: *const PtauParams, combined_path: *const c_char, check_input: c_int, device: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_transform_ratios_devices(full: *const PtauParams, combined: *const u8, combined_len: usize, check_input: c_int, devices: *const c_int, n_devices: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_transform_ratios_file_devices(full: *const PtauParams, combined_path: *const c_char, check_input: c_int, devices: *const c_int, n_devices: c_int, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_key_generation(curve: c_int, seed: *const u8, seed_len: usize, digest: *const u8, tau: *mut u64, alpha: *mut u64, beta: *mut u64, public_key: *mut u8, public_key_len: usize, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_compute_g2_s(curve: c_int, digest: *const u8, public_key: *const u8, public_key_len: usize, g2_s_out: *mut u8, err: *mut c_char, err_len: usize) -> c_int;
    pub fn ptau_last_timing(device: c_int, out: *mut PtauTiming) -> c_int;
    pub fn ptau_device_busy_ms(device: c_int) -> f64;
}

/// The selectors the operator parses from its CLI (`contribute.rs:38-43, 134-145`), with the C ABI's values.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum CheckForCorrectness { No = 0, OnlyNonZero = 1, OnlyInGroup = 2, Full = 3 }
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum BatchExpMode { Auto = 0, Direct = 1, BatchInversion = 2 }
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum SubgroupCheckMode { Auto = 0, Direct = 1, Batched = 2, No = 3 }

/// `curve_kind` of the ceremony (`contribute.rs:1156-1211`, `verify_transcript.rs:876-887`).
pub fn curve_id(curve_kind: &str) -> i32 {
    match curve_kind {
        "bls12_377" => 0,
        "bw6" => 1,
        "mnt4_753" => 2,
        "mnt6_753" => 3,
        other => panic!("unsupported curve_kind {}", other),
    }
}

/// `create_parameters_for_chunk` / `create_full_parameters` (`src/utils.rs:326-352`) as the ABI struct.
pub fn parameters_for_chunk(curve_kind: &str, proving_system: i32, power: usize, batch_size: usize, chunk_size: usize, chunk_index: usize) -> PtauParams {
    PtauParams { curve: curve_id(curve_kind), proving_system, contribution_mode: 0, power: power as u32,
                 chunk_index: chunk_index as u64, chunk_size: chunk_size as u64, batch_size: batch_size as u64 }
}
pub fn full_parameters(curve_kind: &str, proving_system: i32, power: usize, batch_size: usize) -> PtauParams {
    PtauParams { curve: curve_id(curve_kind), proving_system, contribution_mode: 1, power: power as u32,
                 chunk_index: 0, chunk_size: 1u64 << power, batch_size: batch_size as u64 }
}

/// GPU for a pipeline task: `contribute.rs:245-250` clones one `Contribute` per task with `clone_with_index(i)`.
pub fn device_for_task(task_index: Option<usize>) -> i32 {
    let n = unsafe { ptau_device_count() };
    assert!(n > 0, "ptau: no usable gfx950 device (the library has no CPU fallback)");
    (task_index.unwrap_or(0) % n as usize) as i32
}

fn cpath(p: &str) -> CString { CString::new(Path::new(p).to_str().expect("utf-8 path")).expect("path without NUL") }

/// Turns a non-zero return code into the panic the reference's functions raise.
fn check(what: &str, rc: c_int, err: &[c_char; 256]) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(err.as_ptr()) }.to_string_lossy().into_owned();
        let name = unsafe { CStr::from_ptr(ptau_strerror(rc)) }.to_string_lossy().into_owned();
        panic!("{} failed: {} ({}): {}", what, name, rc, msg);
    }
}

/// Same names, argument order and panic-on-failure behaviour as the `phase1_cli` crate the operator imports
/// (`contribute.rs:29-31`): `use ptau_hip::phase1_cli;` under `#[cfg(feature = "hip")]` is the whole patch.
pub mod phase1_cli {
    use super::*;

    /// `phase1_cli::contribute` (`contribute.rs:811-823`) — `rng` is `derive_rng_from_seed(seed)`.
    /// `input_in_subgroup`: pass `false` at the operator's call sites.  A downloaded challenge is data of unknown origin
    /// to a contributor; the reference runs `CheckForCorrectness::No` on it and returns `[k]P` for whatever bytes arrived,
    /// and with `false` so does this call (`BatchExpMode::Auto` probes the vectors itself and takes the endomorphism
    /// kernels only when the probe passes).  `true` is legitimate ONLY for a caller that ran `CheckForCorrectness::Full`
    /// on the same bytes itself: on a point outside the prime-order subgroup the asserted path returns a different point.
    pub fn contribute(challenge_filename: &str, challenge_hash_filename: &str, response_filename: &str, response_hash_filename: &str,
                      check_input_correctness: CheckForCorrectness, batch_exp_mode: BatchExpMode, parameters: &PtauParams,
                      seed: &[u8], input_in_subgroup: bool, device: i32) {
        let mut err = [0 as c_char; 256];
        let mode = batch_exp_mode as c_int | if input_in_subgroup { PTAU_BATCH_EXP_INPUT_IN_SUBGROUP } else { 0 };
        let (c, ch, r, rh) = (cpath(challenge_filename), cpath(challenge_hash_filename), cpath(response_filename), cpath(response_hash_filename));
        let rc = unsafe { ptau_contribute_file_seeded(parameters, c.as_ptr(), ch.as_ptr(), r.as_ptr(), rh.as_ptr(), seed.as_ptr(), seed.len(),
                                                      check_input_correctness as c_int, mode, device, err.as_mut_ptr(), err.len()) };
        check("phase1_cli::contribute", rc, &err);
    }

    /// The same with arkworks' own `Phase1::key_generation` output: `tau/alpha/beta` = `key.tau.into_bigint().0` …,
    /// `public_key` = the `PublicKey` serialized uncompressed.
    pub fn contribute_with_key(challenge_filename: &str, challenge_hash_filename: &str, response_filename: &str, response_hash_filename: &str,
                               check_input_correctness: CheckForCorrectness, batch_exp_mode: BatchExpMode, parameters: &PtauParams,
                               tau: &[u64], alpha: &[u64], beta: &[u64], public_key: &[u8], input_in_subgroup: bool, device: i32) {
        let mut err = [0 as c_char; 256];
        let mode = batch_exp_mode as c_int | if input_in_subgroup { PTAU_BATCH_EXP_INPUT_IN_SUBGROUP } else { 0 };
        let (c, ch, r, rh) = (cpath(challenge_filename), cpath(challenge_hash_filename), cpath(response_filename), cpath(response_hash_filename));
        let rc = unsafe { ptau_contribute_file(parameters, c.as_ptr(), ch.as_ptr(), r.as_ptr(), rh.as_ptr(), tau.as_ptr(), alpha.as_ptr(), beta.as_ptr(),
                                               public_key.as_ptr(), public_key.len(), check_input_correctness as c_int, mode, device,
                                               err.as_mut_ptr(), err.len()) };
        check("phase1_cli::contribute", rc, &err);
    }

    /// `phase1_cli::transform_pok_and_correctness` (`contribute.rs:968-986`), the reference's eleven arguments; `g2_s`:
    /// `None` (derived by the library) or arkworks' three `compute_g2_s` points, uncompressed; `tau_g2_check`: `None`, or
    /// `tauG2[0] | tauG2[1]` (uncompressed) of the same contribution so that `ratio_check` covers a tauG1-only chunk.
    /// Returns what THIS call checked (`ptau_last_timing` is per calling thread, so the record is this call's whatever other
    /// lanes did on the same GPU meanwhile): `ratio_checks == 0` under `ratio_check = true` means the chunk holds no tauG2
    /// pair and no `tau_g2_check` was given — its tauG1 power ratios are then left to `transform_ratios` on the combined
    /// accumulator (`control.rs:587-591`).  A verifier that must not accept such a chunk unchecked asserts on it.
    pub fn transform_pok_and_correctness(challenge_filename: &str, challenge_hash_filename: &str, check_input_correctness: CheckForCorrectness,
                                         response_filename: &str, response_hash_filename: &str, check_output_correctness: CheckForCorrectness,
                                         new_challenge_filename: &str, new_challenge_hash_filename: &str, subgroup_check_mode: SubgroupCheckMode,
                                         ratio_check: bool, parameters: &PtauParams, g2_s: Option<&[u8]>, tau_g2_check: Option<&[u8]>, device: i32) -> VerifyReport {
        let mut err = [0 as c_char; 256];
        let (c, ch, r, rh, n, nh) = (cpath(challenge_filename), cpath(challenge_hash_filename), cpath(response_filename),
                                     cpath(response_hash_filename), cpath(new_challenge_filename), cpath(new_challenge_hash_filename));
        let rc = unsafe { ptau_transform_pok_and_correctness_file(parameters, c.as_ptr(), ch.as_ptr(), check_input_correctness as c_int, r.as_ptr(), rh.as_ptr(),
                                                                  check_output_correctness as c_int, n.as_ptr(), nh.as_ptr(), subgroup_check_mode as c_int,
                                                                  ratio_check as c_int, g2_s.map_or(std::ptr::null(), |s| s.as_ptr()),
                                                                  tau_g2_check.map_or(std::ptr::null(), |s| s.as_ptr()), device,
                                                                  err.as_mut_ptr(), err.len()) };
        check("phase1_cli::transform_pok_and_correctness", rc, &err);
        let mut t: PtauTiming = unsafe { std::mem::zeroed() };
        unsafe { ptau_last_timing(device, &mut t) };   // same thread as the call above: this call's record
        VerifyReport { ratio_checks: t.ratio_checks, pok_checks: t.pok_checks, device_ms: t.kernel_ms }
    }

    /// `phase1_cli::new_challenge` (`new_setup.rs:105-109`).  MNT4-753 / MNT6-753: use [`new_challenge_with_generators`].
    pub fn new_challenge(challenge_filename: &str, challenge_hash_filename: &str, parameters: &PtauParams) {
        let mut err = [0 as c_char; 256];
        let (c, ch) = (cpath(challenge_filename), cpath(challenge_hash_filename));
        let rc = unsafe { ptau_new_challenge_file(parameters, c.as_ptr(), ch.as_ptr(), err.as_mut_ptr(), err.len()) };
        check("phase1_cli::new_challenge", rc, &err);
    }

    /// `Phase1::initialization` into a buffer with `E::G1Affine::generator()` / `E::G2Affine::generator()` serialized
    /// uncompressed by the caller: bit-exact with arkworks on every curve by construction.
    pub fn new_challenge_with_generators(parameters: &PtauParams, g1_generator: &[u8], g2_generator: &[u8]) -> Vec<u8> {
        let mut sz = PtauSizes::default();
        assert_eq!(unsafe { ptau_params_sizes(parameters, &mut sz) }, 0, "invalid parameters");
        let mut out = vec![0u8; sz.accumulator_size as usize];
        let mut err = [0 as c_char; 256];
        let rc = unsafe { ptau_new_challenge_with_generators(parameters, g1_generator.as_ptr(), g2_generator.as_ptr(), out.as_mut_ptr(), out.len(),
                                                             err.as_mut_ptr(), err.len()) };
        check("phase1_cli::new_challenge", rc, &err);
        out
    }

    /// `phase1_cli::combine(response_list_filename, combined_filename, &parameters)` (`control.rs:564-568`,
    /// `verify_transcript.rs:603-607`) — the reference's three arguments: `parameters` = `create_parameters_for_chunk(.., 0)`;
    /// the library maps the listed responses and CREATES `combined_filename` (nothing passes through the Rust heap).
    pub fn combine(response_list_filename: &str, combined_filename: &str, parameters: &PtauParams, device: i32) {
        let mut err = [0 as c_char; 256];
        let (l, c) = (cpath(response_list_filename), cpath(combined_filename));
        let rc = unsafe { ptau_combine_file(parameters, l.as_ptr(), c.as_ptr(), device, err.as_mut_ptr(), err.len()) };
        check("phase1_cli::combine", rc, &err);
    }

    /// `phase1_cli::transform_ratios(combined_filename, check_input_correctness, &full_parameters)` (`control.rs:587-591`,
    /// `verify_transcript.rs:646-653, 811-822`).  `devices`: one ordinal, or several — the operator is ONE process, so the
    /// final ratio check reaches the node's GPUs through host threads inside the library (`ptau_transform_ratios_file_devices`).
    pub fn transform_ratios(combined_filename: &str, check_input_correctness: CheckForCorrectness, full_parameters: &PtauParams, devices: &[i32]) {
        let mut err = [0 as c_char; 256];
        let c = cpath(combined_filename);
        let rc = unsafe { ptau_transform_ratios_file_devices(full_parameters, c.as_ptr(), check_input_correctness as c_int, devices.as_ptr(),
                                                             devices.len() as c_int, err.as_mut_ptr(), err.len()) };
        check("phase1_cli::transform_ratios", rc, &err);
    }

    /// Buffer-level forms (tests, hosts that already hold the accumulator in memory).
    pub fn combine_buffers(responses: &[&[u8]], chunk_size: usize, full_parameters: &PtauParams, device: i32) -> Vec<u8> {
        let mut sz = PtauSizes::default();
        assert_eq!(unsafe { ptau_params_sizes(full_parameters, &mut sz) }, 0, "invalid parameters");
        let ptrs: Vec<*const u8> = responses.iter().map(|r| r.as_ptr()).collect();
        let lens: Vec<usize> = responses.iter().map(|r| r.len()).collect();
        let mut out = vec![0u8; sz.accumulator_size as usize];
        let mut err = [0 as c_char; 256];
        let rc = unsafe { ptau_combine(full_parameters, chunk_size as u64, ptrs.as_ptr(), lens.as_ptr(), responses.len() as u64, out.as_mut_ptr(), out.len(),
                                       device, err.as_mut_ptr(), err.len()) };
        check("phase1_cli::combine", rc, &err);
        out
    }
    pub fn transform_ratios_buffer(combined: &[u8], check_input_correctness: CheckForCorrectness, full_parameters: &PtauParams, devices: &[i32]) {
        let mut err = [0 as c_char; 256];
        let rc = unsafe { ptau_transform_ratios_devices(full_parameters, combined.as_ptr(), combined.len(), check_input_correctness as c_int,
                                                        devices.as_ptr(), devices.len() as c_int, err.as_mut_ptr(), err.len()) };
        check("phase1_cli::transform_ratios", rc, &err);
    }
}

/// Every usable device ordinal: what `transform_ratios` takes when the whole node serves the final check.
pub fn all_devices() -> Vec<i32> { (0..unsafe { ptau_device_count() }).collect() }

/// `processor_data` for `ContributedData` (`contribute.rs:858-864`): the GPU that did the work.
pub fn processor_name(device: i32) -> String {
    let mut name = [0 as c_char; 128];
    let (mut cus, mut hbm) = (0 as c_int, 0u64);
    if unsafe { ptau_device_info(device, name.as_mut_ptr(), name.len(), &mut cus, &mut hbm) } != 0 {
        return String::from("unknown GPU");
    }
    format!("{} ({} CUs, {} GiB HBM)", unsafe { CStr::from_ptr(name.as_ptr()) }.to_string_lossy(), cus, hbm >> 30)
}
