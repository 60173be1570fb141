/* ptau.h — C ABI of the MI355X-native chunked powers-of-tau engine (libptau_hip.so).
 *
 * Drop-in boundary for the ONE hot path of nimiq/snark-setup-operator: the per-chunk phase-1
 * tau-update and its verification.  Each entry point names the reference interface it replaces
 * (file:line in /root/reference; the bodies live in the un-vendored crates phase1-cli / phase1 /
 * setup-utils @ nimiq/snark-setup bd530da — SURVEY.md §8).  INTEGRATION.md shows the Rust
 * `extern "C"` binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the library never retains a caller pointer after return;
 *   - scalars (tau, alpha, beta, coefficients) are CANONICAL (non-Montgomery) little-endian
 *     64-bit limbs, ptau_fr_limbs(curve) of them  (= ark-ff BigInt<N>.0);
 *   - byte formats are arkworks-0.4 canonical serialisation;
 *   - return 0 on success, a negative PTAU_E_* code otherwise; a message is copied into the
 *     caller's `err` buffer when one is given.  Nothing aborts, exits or throws across the ABI:
 *     the Rust shim turns a non-zero code into `panic!(msg)` inside `spawn_quiet`, preserving the
 *     reference's error behaviour (src/bin/contribute.rs:809-856);
 *   - `device` is the HIP device ordinal (the operator's task index % ngpu,
 *     src/bin/contribute.rs:245-250).  Calls on distinct devices may run concurrently; calls on
 *     one device are serialised internally.
 *   - There is NO CPU fallback: every compute entry point fails with PTAU_E_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef PTAU_H
#define PTAU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* curve_kind dispatch — src/bin/contribute.rs:1156-1211, src/bin/verify_transcript.rs:876-887 */
enum { PTAU_CURVE_BLS12_377 = 0, PTAU_CURVE_BW6_761 = 1, PTAU_CURVE_MNT4_753 = 2, PTAU_CURVE_MNT6_753 = 3 };
enum { PTAU_G1 = 1, PTAU_G2 = 2 };
/* phase1::ContributionMode / ProvingSystem — src/utils.rs:326-352 */
enum { PTAU_MODE_CHUNKED = 0, PTAU_MODE_FULL = 1 };
enum { PTAU_PROVING_GROTH16 = 0, PTAU_PROVING_MARLIN = 1 };
/* setup_utils::CheckForCorrectness — src/bin/contribute.rs:38-43,816-819,971-980 */
enum { PTAU_CHECK_NO = 0, PTAU_CHECK_ONLY_NON_ZERO = 1, PTAU_CHECK_ONLY_IN_GROUP = 2, PTAU_CHECK_FULL = 3 };
/* setup_utils::SubgroupCheckMode — src/bin/contribute.rs:140-145.
 *   DIRECT  : the exact per-element membership test (endomorphism tests on BLS12-377 / BW6-761 / the MNT twists — same
 *             accept/reject set as [r]P == O);
 *   BATCHED, and AUTO on more than 2^12 elements (setup-utils check_subgroup: BATCH_SIZE = 1 << 12): the batched test with
 *             SECURITY_PARAM = 128 — random bucket sums, each tested exactly; a vector holding an element outside the subgroup
 *             passes with probability < 2^-128.  Here the buckets are those of the power-ratio check's own random linear
 *             combination (ptau_verify_chunk / transform_pok_and_correctness with ratio_check != 0), so the test rides on work
 *             already done; without a ratio check, on vectors of <= 2^12 elements, and in every other entry point
 *             (ptau_subgroup_check, check_input, transform_ratios) the exact test runs — never anything weaker than the
 *             reference's choice;
 *   NO      : no membership test. */
enum { PTAU_SUBGROUP_AUTO = 0, PTAU_SUBGROUP_DIRECT = 1, PTAU_SUBGROUP_BATCHED = 2, PTAU_SUBGROUP_NO = 3 };
/* setup_utils::BatchExpMode — src/bin/contribute.rs:134-139.  arkworks' batch_exp returns [k]P for ANY curve point, so:
 *   DIRECT                 : plain signed-window multiplication, exact on the whole curve;
 *   AUTO / BATCH_INVERSION : the same bytes, faster where the library can establish that every input element lies in the
 *                            prime-order subgroup — then the endomorphism-accelerated multiplication runs (GLV on BLS12-377
 *                            G1 / BW6-761, GLS on BLS12-377 G2 and the MNT twists), equal to [k]P on the subgroup, ~1.4x
 *                            faster.  Membership is established (a) by check_input >= PTAU_CHECK_ONLY_IN_GROUP in the same
 *                            call, or (b) by a PROBE the library runs by itself: on vectors of more than 2^12 elements the
 *                            batched membership test (random bucket sums, each tested exactly; a vector with an element
 *                            outside the subgroup passes with probability < 2^-128 — setup-utils' SubgroupCheckMode::Batched,
 *                            SECURITY_PARAM = 128), whose bucket tests run beside the endomorphism kernel; on short BLS12-377
 *                            G2 vectors the exact test.  When the probe fails, or the input is not even valid points, the
 *                            plain path runs (and reports bad points as it always did): the bytes are the plain path's
 *                            either way;
 *   | PTAU_BATCH_EXP_INPUT_IN_SUBGROUP : the CALLER asserts membership (the challenge came out of a verifier that ran the
 *                            subgroup checks, which is what the coordinator hands a contributor): the endomorphism path runs
 *                            without a probe.  On an element outside the subgroup the result would differ from
 *                            the reference's — never set it for data of unknown origin. */
enum { PTAU_BATCH_EXP_AUTO = 0, PTAU_BATCH_EXP_DIRECT = 1, PTAU_BATCH_EXP_BATCH_INVERSION = 2, PTAU_BATCH_EXP_INPUT_IN_SUBGROUP = 0x100 };

enum {
    PTAU_OK = 0,
    PTAU_E_ARG = -1,               /* bad enum / null pointer */
    PTAU_E_LEN = -2,               /* buffer length != accumulator_size / contribution_size */
    PTAU_E_POINT = -3,             /* non-canonical coordinate or x not on the curve */
    PTAU_E_POINT_AT_INFINITY = -4, /* CheckForCorrectness::OnlyNonZero / Full */
    PTAU_E_NOT_IN_SUBGROUP = -5,   /* CheckForCorrectness::OnlyInGroup / Full, subgroup check */
    PTAU_E_INVALID_RATIO = -6,     /* check_same_ratio failed */
    PTAU_E_HASH_CHAIN = -7,        /* response[0..64) != BLAKE2b(challenge) */
    PTAU_E_DEVICE = -8,            /* no usable gfx950 device / HIP error */
    PTAU_E_IO = -9,                /* file open / size / create_new failure */
    PTAU_E_UNSUPPORTED = -10       /* e.g. Marlin geometry */
};

/* The ceremony `Parameters` JSON (src/data_structs.rs:123-131) + chunk index:
 * what create_parameters_for_chunk / create_full_parameters consume (src/utils.rs:326-352). */
typedef struct ptau_params {
    int32_t curve;             /* PTAU_CURVE_* */
    int32_t proving_system;    /* PTAU_PROVING_* */
    int32_t contribution_mode; /* PTAU_MODE_* */
    uint32_t power;
    uint64_t chunk_index;
    uint64_t chunk_size;
    uint64_t batch_size;
} ptau_params;

/* Fields of Phase1Parameters the reference reads (src/bin/new_setup.rs:274-277, src/utils.rs:526-532). */
typedef struct ptau_sizes {
    uint64_t powers_length, powers_g1_length;
    uint64_t g1_chunk_size, other_chunk_size;
    uint64_t accumulator_size, contribution_size, public_key_size, hash_size;
    uint64_t start; /* global exponent index of element 0 */
} ptau_sizes;

/* Per-call device timings (HIP events on the library's stream), for the reference's
 * contribution_duration accounting (src/bin/contribute.rs:857-870) and bench.py. */
typedef struct ptau_timing {
    double h2d_ms, kernel_ms, d2h_ms; /* summed over the call */
    double scalar_mul_ms, normalize_ms; /* the two kernels of the tau-update */
    uint64_t points;                  /* group elements processed */
    double decompress_ms, subgroup_ms, msm_ms; /* verify path: sqrt, subgroup checks, bucket MSMs (parts of kernel_ms) */
    /* verification calls: how many same-ratio (pairing) checks the call ran — power-ratio checks of check_power_ratios and
     * proofs of knowledge / chunk-0 before-after checks.  ratio_checks == 0 after a call with ratio_check != 0 means "NOT
     * PERFORMED": a chunk without tauG2 elements and no tau_g2_check (see ptau_verify_chunk). */
    uint32_t ratio_checks, pok_checks;
    /* verify path: latency of the bucket MSMs' tails (bucket reduction, row sums, Horner, normalisation: a few waves), which
     * run on a second stream beside the next vector's kernels.  kernel_ms / msm_ms hold only the part that was not hidden. */
    double msm_tail_ms;
} ptau_timing;

const char *ptau_strerror(int code);
int ptau_device_count(void);
/* Device name / CU count for processor_data (src/bin/contribute.rs:858-864). */
int ptau_device_info(int device, char *name, size_t name_len, int *compute_units, uint64_t *hbm_bytes);

/* Phase1Parameters::{new_chunk,new_full} geometry — src/utils.rs:326-352 */
int ptau_params_sizes(const ptau_params *p, ptau_sizes *out);
int ptau_point_size(int curve, int group, int compressed);
int ptau_fr_limbs(int curve);

/* setup_utils::calculate_hash — src/bin/verify_transcript.rs:344-345, src/utils.rs:618-623 (host CPU) */
void ptau_blake2b512(const uint8_t *data, size_t len, uint8_t out[64]);
/* Which compression function the host BLAKE2b uses: 0 portable, 1 AVX2, 2 BMI2-compiled scalar, 3 AVX-512 (timed against each
 * other on first use).  force = 0..3 selects one if the CPU has it (test hook: all are checked against hashlib), force < 0
 * only reports. */
int ptau_blake2b_implementation(int force);

/* setup-utils apply_powers = read_batch -> generate_powers_of_tau -> batch_exp -> write_batch
 * (inside phase1_cli::contribute, src/bin/contribute.rs:811-823):
 *     out[i] = (coeff * tau^(start+i)) * in[i],  i in [0, n)
 * `in`/`out` are HOST buffers of n serialized points. coeff may be NULL. */
int ptau_apply_powers(int curve, int group, const uint8_t *in, int compressed_in, uint8_t *out,
                      int compressed_out, uint64_t n, uint64_t start, const uint64_t *tau,
                      const uint64_t *coeff, int check_input, int batch_exp_mode, int device, char *err, size_t err_len);

/* Same, on DEVICE-resident buffers (hipMalloc / torch CUDA tensors): the kernel-only path
 * bench.py times.  Runs on the library's stream for `device` and synchronises before returning. */
int ptau_apply_powers_dev(int curve, int group, const void *d_in, int compressed_in, void *d_out,
                          int compressed_out, uint64_t n, uint64_t start, const uint64_t *tau,
                          const uint64_t *coeff, int check_input, int batch_exp_mode, int device, char *err, size_t err_len);

/* Phase1::computation — inside phase1_cli::contribute (src/bin/contribute.rs:811-823):
 * `in` = whole challenge (accumulator_size, uncompressed), `out` = whole response buffer
 * (contribution_size); fills out[64 .. contribution_size - public_key_size).  The host keeps
 * hashing, key generation and the public-key write. */
int ptau_phase1_computation(const ptau_params *p, const uint8_t *in, size_t in_len, uint8_t *out,
                            size_t out_len, const uint64_t *tau, const uint64_t *alpha,
                            const uint64_t *beta, int check_input, int batch_exp_mode, int device, char *err, size_t err_len);

/* phase1_cli::contribute with the key pair as INPUT (the Rust host ran Phase1::key_generation itself; see
 * ptau_contribute_file_seeded for the reference's own seed-driven form) — src/bin/contribute.rs:811-823.  Reads `challenge_path` (must be exactly
 * accumulator_size), writes challenge_hash_path (64 raw bytes), creates response_path
 * (create_new semantics) = hash(challenge) | updated elements | public_key, writes
 * response_hash_path. */
int ptau_contribute_file(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path,
                         const char *response_path, const char *response_hash_path, const uint64_t *tau,
                         const uint64_t *alpha, const uint64_t *beta, const uint8_t *public_key,
                         size_t public_key_len, int check_input, int batch_exp_mode, int device, char *err,
                         size_t err_len);

/* Phase1::decompress for one vector — inside phase1_cli::transform_pok_and_correctness
 * (src/bin/contribute.rs:968-986): n compressed points -> n uncompressed points (host buffers). */
int ptau_decompress(int curve, int group, const uint8_t *in, uint8_t *out, uint64_t n, int check_output,
                    int device, char *err, size_t err_len);

/* setup-utils check_subgroup (SubgroupCheckMode::Direct) + on-curve test for n HOST-resident points —
 * CheckForCorrectness::OnlyInGroup of read_batch (src/bin/contribute.rs:971-980).
 * Fails with PTAU_E_NOT_IN_SUBGROUP / PTAU_E_POINT. */
int ptau_subgroup_check(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n, int device,
                        char *err, size_t err_len);

/* Phase1::verification + Phase1::decompress on buffers — inside phase1_cli::transform_pok_and_correctness
 * (src/bin/contribute.rs:968-986):
 *   - response[0..64) must equal BLAKE2b(challenge)             -> PTAU_E_HASH_CHAIN
 *   - check_input (CheckForCorrectness of the CHALLENGE elements, --force-correctness-checks, contribute.rs:133,971-974):
 *     non-zero / on the curve and in the prime-order subgroup      -> PTAU_E_POINT_AT_INFINITY / _NOT_IN_SUBGROUP / _POINT
 *   - every response element: canonical, on the curve, non-zero per check_output, in the prime-order
 *     subgroup unless subgroup_check_mode == PTAU_SUBGROUP_NO      -> PTAU_E_POINT / _AT_INFINITY / _NOT_IN_SUBGROUP
 *   - verify_pok != 0: the three proofs of knowledge of the PublicKey at the tail of `response` and, for chunk 0 / Full
 *     mode, the before/after checks (ptau_verify_pok).  g2_s = the three compute_g2_s points (tau, alpha, beta;
 *     UNCOMPRESSED) or NULL: derived by the library from BLAKE2b(challenge) and the public key (ptau_compute_g2_s).
 *     verify_pok == 0 leaves them out (element checks only: tests, and callers that run ptau_verify_pok themselves).
 *   - ratio_check (contribute.rs:146-147,954,984): check_power_ratios of every vector — power_pairs on the GPU,
 *     check_same_ratio on the host — against (tauG2[j], tauG2[j+1]) / (tauG1[j], tauG1[j+1]) of this chunk.  A chunk
 *     without tauG2 elements (start >= 2^power) has no G2 pair of ratio tau in its own files: its tauG1 power ratios are
 *     checked when the caller supplies tau_g2_check = two UNCOMPRESSED G2 points (tauG2[0], tauG2[1]) of the same
 *     contribution, and are otherwise left to ptau_transform_ratios on the combined accumulator (src/bin/control.rs:587-591)
 *     — never an error for an honest response                          -> PTAU_E_INVALID_RATIO
 *   - new_challenge = BLAKE2b(response) | decompressed elements (accumulator_size bytes). */
int ptau_verify_chunk(const ptau_params *p, const uint8_t *challenge, size_t challenge_len, const uint8_t *response,
                      size_t response_len, uint8_t *new_challenge, size_t new_challenge_len, int check_input,
                      int check_output, int subgroup_check_mode, int ratio_check, int verify_pok, const uint8_t *g2_s,
                      const uint8_t *tau_g2_check, int device, char *err, size_t err_len);

/* phase1_cli::transform_pok_and_correctness, file level (src/bin/contribute.rs:968-986), the reference's eleven
 * arguments in the reference's order: hash chain, CheckForCorrectness of the challenge, PROOFS OF KNOWLEDGE and chunk-0
 * before/after checks (always), element checks, ratio checks, decompression; writes the three 64-byte `.hash` side files
 * and creates new_challenge_path (create_new semantics).  g2_s: NULL (derived by the library, as the reference does) or
 * the three compute_g2_s points computed by the Rust host (uncompressed), which removes the one step whose bit-exactness
 * with arkworks cannot be pinned offline (ptau_compute_g2_s).
 * tau_g2_check: NULL, or the two UNCOMPRESSED G2 points (tauG2[0], tauG2[1]) of the same contribution (the head of the
 * chunk-0 new challenge): lets ratio_check cover a chunk that holds no tauG2 elements of its own (start >= 2^power).
 * Without it such a chunk's tauG1 power ratios are NOT checked here (ptau_last_timing().ratio_checks == 0 says so) and
 * are left to ptau_transform_ratios_file on the combined accumulator. */
int ptau_transform_pok_and_correctness_file(const ptau_params *p, const char *challenge_path,
                                            const char *challenge_hash_path, int check_input, const char *response_path,
                                            const char *response_hash_path, int check_output,
                                            const char *new_challenge_path, const char *new_challenge_hash_path,
                                            int subgroup_check_mode, int ratio_check, const uint8_t *g2_s,
                                            const uint8_t *tau_g2_check, int device, char *err, size_t err_len);

/* setup_utils::dense_multiexp — sum_i scalars[i] * in[i] -> one serialized point (host buffers; scalars are
 * n x ptau_fr_limbs(curve) canonical limbs).  Used inside merge_pairs / power_pairs. */
int ptau_msm(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n, const uint64_t *scalars,
             uint8_t *out, int compressed_out, int device, char *err, size_t err_len);

/* setup_utils::power_pairs (inside check_power_ratios; reference call sites src/bin/verify_transcript.rs:812-822,
 * src/bin/control.rs:587-591): with random coefficients rho_i,
 *     out_a = sum_{i<n-1} rho_i * in[i],   out_b = sum_{i<n-1} rho_i * in[i+1]      (both UNCOMPRESSED).
 * rho_i comes from a counter-based generator keyed by `seed` (16 bytes; NULL = fresh OS entropy) and the GLOBAL
 * element index index_base + i, so slices of one vector processed on different GPUs use one rho stream (row C1). */
int ptau_power_pairs(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n, const uint8_t seed[16],
                     uint64_t index_base, uint8_t *out_a, uint8_t *out_b, int device, char *err, size_t err_len);

/* setup_utils::check_same_ratio((a, b), (c, d)) — e(a, d) == e(b, c) — reference call sites
 * src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591 (inside transform_ratios / verification).
 * a, b: G1 points, c, d: G2 points, all UNCOMPRESSED.  This single-check form runs on the HOST CPU (it needs no GPU: the
 * checker of the GPU kernel in the tests, and the standalone ptau_verify_pok); the verification entry points use the batch form.
 * Returns PTAU_OK when the ratios agree, PTAU_E_INVALID_RATIO when they do not. */
int ptau_check_same_ratio(int curve, const uint8_t *g1_a, const uint8_t *g1_b, const uint8_t *g2_c,
                          const uint8_t *g2_d, char *err, size_t err_len);

/* setup_utils::check_same_ratio for n independent checks in ONE launch of the pairing kernel (K5 on the GPU: Miller loop and
 * final exponentiation in HIP, an Fq^k element spread over k lanes, one workgroup per check) — what the verification entry
 * points use for the proofs of knowledge, the chunk-0 before/after checks and the power-ratio checks (reference call sites
 * src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591, src/bin/contribute.rs:968-986).
 * checks = n x (g1_a | g1_b | g2_c | g2_d), UNCOMPRESSED; results[i] = 1 same ratio, 0 different, -1 malformed point.
 * gt_out (optional, test hook): per check the reduced pairing product e(a, d) * e(-b, c) as k x ceil(bits(q)/32) canonical
 * little-endian 32-bit words (coefficients of w^0 .. w^(k-1) in Fq[w]/(w^k - nu)); the value 1 <=> same ratio. */
int ptau_check_same_ratio_batch(int curve, const uint8_t *checks, uint64_t n, int *results, uint32_t *gt_out, int device,
                                char *err, size_t err_len);

/* Phase1::initialization via phase1_cli::new_challenge (src/bin/new_setup.rs:105-109, verify_transcript.rs:322-326):
 * blank BLAKE2b hash, then every element = the group generator, uncompressed (host only, no GPU needed).
 * BLS12-377 and BW6-761 use the built-in ark-bls12-377 / ark-bw6-761 generators.  MNT4-753 / MNT6-753 return
 * PTAU_E_UNSUPPORTED in a build without pinned constants: their arkworks G2 generator constants are not recoverable offline
 *  and an accumulator built on any other point would not be the reference's initial challenge — use the
 * _with_generators form, or run tools/pin_reference.sh (it stores arkworks' own G2Affine::generator() bytes as
 * csrc/pinned_generators.json) and rebuild: the built-in form then serves all four curves. */
int ptau_new_challenge(const ptau_params *p, uint8_t *out, size_t out_len, char *err, size_t err_len);
int ptau_new_challenge_file(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path,
                            char *err, size_t err_len);
/* The same with the generators supplied by the caller (uncompressed bytes of G1Affine::generator() and
 * G2Affine::generator()): bit-exact with the reference on every curve by construction. */
int ptau_new_challenge_with_generators(const ptau_params *p, const uint8_t *g1_generator, const uint8_t *g2_generator,
                                       uint8_t *out, size_t out_len, char *err, size_t err_len);
/* Built-in generator (uncompressed); PTAU_E_UNSUPPORTED for the G2 of MNT4-753 / MNT6-753. */
int ptau_generator(int curve, int group, uint8_t *out, size_t out_len, char *err, size_t err_len);

/* Phase1::aggregation via phase1_cli::combine (src/bin/control.rs:564-568, verify_transcript.rs:603-607):
 * n_chunks compressed chunk responses -> one Full-mode UNCOMPRESSED accumulator (64 zero bytes | tauG1 |
 * tauG2 | alphaG1 | betaG1 | betaG2 of chunk 0).  `full` must be Full-mode parameters. */
int ptau_combine(const ptau_params *full, uint64_t chunk_size, const uint8_t *const *responses,
                 const size_t *response_lens, uint64_t n_chunks, uint8_t *combined, size_t combined_len, int device,
                 char *err, size_t err_len);

/* Phase1::aggregate_verification via phase1_cli::transform_ratios (src/bin/control.rs:587-591,
 * verify_transcript.rs:646-653, 812-822): full-length check_power_ratios of tauG1, tauG2, alphaG1, betaG1 and
 * the betaG2 consistency check on a combined accumulator.  -> PTAU_E_INVALID_RATIO */
int ptau_transform_ratios(const ptau_params *full, const uint8_t *combined, size_t combined_len, int check_input,
                          int device, char *err, size_t err_len);

/* phase1_cli::combine(response_list_filename, combined_filename, &parameters) in the reference's FILE form and argument
 * order (src/bin/control.rs:564-568, src/bin/verify_transcript.rs:603-607): `chunk0` = the parameters of chunk 0
 * (create_parameters_for_chunk(.., 0): curve, power, chunk_size, batch_size); response_list_path = a text file with one
 * chunk-response path per line, in chunk order (control.rs:547-557 writes it); combined_path is CREATED (create_new
 * semantics — the caller removes a stale one first, control.rs:560) with the Full-mode accumulator_size and filled with
 * the decompressed vectors.  Its first 64 bytes stay zero: the reference creates the file with set_len and
 * Phase1::aggregation writes the element vectors only (recollection of phase1-cli/src/combine.rs @ bd530da — UNPINNED;
 * BLAKE2b(combined) is what the beacon contribution that follows puts in its response header, verify_transcript.rs:675-696).
 * The responses are mapped, never read into the heap. */
int ptau_combine_file(const ptau_params *chunk0, const char *response_list_path, const char *combined_path, int device,
                      char *err, size_t err_len);

/* phase1_cli::transform_ratios(combined_filename, check_input_correctness, &full_parameters), file form
 * (src/bin/control.rs:587-591, src/bin/verify_transcript.rs:646-653, 811-822): the file is mapped read-only (it may be a
 * `combined` file or the new challenge of the beacon verification — both are Full-mode accumulators).  */
int ptau_transform_ratios_file(const ptau_params *full, const char *combined_path, int check_input, int device, char *err,
                               size_t err_len);

/* The same check spread over SEVERAL devices from ONE process (the operator is one process with lane threads,
 * src/bin/contribute.rs:467-492): every vector's pair indices are cut into n_devices contiguous slices
 * (overlapping by one element), slice k is reduced by power_pairs on devices[k] on its own host thread under ONE rho stream
 * (one seed, coefficients keyed by the global index), the n_devices partial pairs per vector are summed and the five
 * pairing checks run once, on devices[0].  Same accept / reject set as ptau_transform_ratios.  check_input: each slice is
 * membership-tested on its device.  An ordinal may be listed more than once (its slices queue on that device's lock):
 * {0, 0, 0} cuts every vector into three slices on one GPU. */
int ptau_transform_ratios_devices(const ptau_params *full, const uint8_t *combined, size_t combined_len, int check_input,
                                  const int *devices, int n_devices, char *err, size_t err_len);
int ptau_transform_ratios_file_devices(const ptau_params *full, const char *combined_path, int check_input,
                                       const int *devices, int n_devices, char *err, size_t err_len);

/* Phase1::verification steps (1)-(2) (inside transform_pok_and_correctness, src/bin/contribute.rs:968-986):
 * the three proofs of knowledge in the PublicKey at the tail of `response`, and for chunk 0 / Full mode the
 * before/after checks (tauG1[1], alphaG1[0], betaG1[0], tauG2[1], betaG2 moved by exactly the public key).
 * g2_s = the three compute_g2_s points (tau, alpha, beta), UNCOMPRESSED, or NULL (derived: ptau_compute_g2_s). */
int ptau_verify_pok(const ptau_params *p, const uint8_t *challenge, size_t challenge_len, const uint8_t *response,
                    size_t response_len, const uint8_t *g2_s, char *err, size_t err_len);

/* ---- keys (host CPU) ------------------------------------------------------------------------------
 * setup_utils::derive_rng_from_seed (src/bin/contribute.rs:783-789) = ChaCha20Rng::from_seed(seed[0..32)) of rand_chacha
 * 0.3.1 (Cargo.lock:2927-2929): the RFC 7539 block function with a 64-bit block counter from 0, stream 0, output consumed
 * as little-endian 32-bit words in order.  ptau_chacha20_rng_words exposes that word stream (words [offset, offset+n)),
 * checked against the published ChaCha20 vectors. */
int ptau_chacha20_rng_words(const uint8_t seed[32], uint64_t word_offset, uint32_t *out, size_t n_words);

/* Phase1::key_generation(rng, digest) with rng = derive_rng_from_seed(seed) (src/bin/contribute.rs:783-789, 811-823):
 * tau, alpha, beta <- Fr::rand; for x in (tau, alpha, beta), personalization 0, 1, 2: g1_s <- G1::rand, g1_s_x = x g1_s,
 * g2_s = compute_g2_s(digest, g1_s, g1_s_x, personalization), g2_s_x = x g2_s.  Outputs the scalars as canonical
 * little-endian limbs (ptau_fr_limbs each) and the PublicKey, uncompressed: tau_g1 (s, s_x), alpha_g1, beta_g1, then
 * tau_g2, alpha_g2, beta_g2 (public_key_size bytes).  `digest` = BLAKE2b(challenge), 64 bytes.
 * PARITY UNPINNED: the sampling order and rejection rules restate ark-ff / ark-ec 0.4.2 and rand 0.8 from memory (DESIGN.md
 * §2); keys are valid whatever the sampling, but equal the reference's only if that recollection is exact. */
int ptau_key_generation(int curve, const uint8_t *seed, size_t seed_len, const uint8_t digest[64], uint64_t *tau,
                        uint64_t *alpha, uint64_t *beta, uint8_t *public_key, size_t public_key_len, char *err, size_t err_len);

/* The PrivateKey half alone (the draws that precede anything bound to the digest): lets a contribution start on the
 * GPU while BLAKE2b(challenge) is still being computed. */
int ptau_private_key_from_seed(int curve, const uint8_t *seed, size_t seed_len, uint64_t *tau, uint64_t *alpha, uint64_t *beta,
                               char *err, size_t err_len);

/* setup_utils::compute_g2_s for the three key components of a serialized PublicKey (inside Phase1::verification and
 * key_generation): g2_s[x] = hash_to_g2(BLAKE2b(personalization x | digest | g1_s | g1_s_x)) — ChaCha20Rng seeded by the
 * first 32 bytes of that hash, G2Affine::from_random_bytes until a point is found, cofactor cleared.
 * out = 3 UNCOMPRESSED G2 points.  Same parity caveat as ptau_key_generation. */
int ptau_compute_g2_s(int curve, const uint8_t digest[64], const uint8_t *public_key, size_t public_key_len, uint8_t *g2_s_out,
                      char *err, size_t err_len);

/* phase1_cli::contribute in the reference's own form (src/bin/contribute.rs:783-789, 811-823): the key pair comes from
 * `seed` through derive_rng_from_seed + Phase1::key_generation (the private key is drawn first and the device starts at
 * once; the public key, which is bound to BLAKE2b(challenge), is made while the GPU works). */
int ptau_contribute_file_seeded(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path,
                                const char *response_path, const char *response_hash_path, const uint8_t *seed, size_t seed_len,
                                int check_input, int batch_exp_mode, int device, char *err, size_t err_len);

/* Cumulative kernel time (HIP events, ms) of every call on `device` since the library was loaded: the numerator of a
 * GPU-busy fraction over a wall-clock interval (several in-process lanes).  < 0 on a bad ordinal. */
double ptau_device_busy_ms(int device);

/* Timings and check counts of the most recent compute call THE CALLING THREAD made on `device` (all zero if it made none).
 * The record belongs to the call, not to the device: several in-process lanes (threads) may share a GPU, and each reads its
 * own call's numbers — in particular the `ratio_checks == 0` ("power ratios NOT checked here") signal of a verification —
 * whatever the other lanes did meanwhile.  Read it on the thread that made the call (the operator runs every phase1_cli call
 * inside one spawn_quiet closure, src/bin/contribute.rs:809-856, 966-1025: read it in that closure). */
int ptau_last_timing(int device, ptau_timing *out);

#ifdef __cplusplus
}
#endif
#endif /* PTAU_H */
