/* CPU oracle for the chunked powers-of-tau hot path — TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference (nimiq/snark-setup-operator) holds no golden vectors for this
 * path and its arithmetic lives in un-vendored crates (nimiq/snark-setup@bd530da, arkworks 0.4);
 * this file restates the algorithm from SURVEY.md §8(a)/App. A and is cross-checked against the
 * independent Python big-int restatement in oracle/pyref and the fixtures in tests/golden.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this
 * library.  The product (libptau_hip.so) never does.
 *
 * Scalars are canonical (non-Montgomery) little-endian 64-bit limbs, ceil(fr_bits/64) of them.
 * All functions return 0 on success, a negative ORACLE_E_* code otherwise; none aborts.
 */
#ifndef PTAU_ORACLE_H
#define PTAU_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_BLS12_377 = 0, ORACLE_BW6_761 = 1, ORACLE_MNT4_753 = 2, ORACLE_MNT6_753 = 3 };
enum { ORACLE_E_ARG = -1, ORACLE_E_LEN = -2, ORACLE_E_POINT = -3, ORACLE_E_NOT_ON_CURVE = -4,
       ORACLE_E_INFINITY = -5, ORACLE_E_NOT_IN_SUBGROUP = -6 };

/* setup_utils::calculate_hash (reference: src/bin/verify_transcript.rs:344-345): BLAKE2b-512. */
void oracle_blake2b512(const uint8_t *data, size_t len, uint8_t out[64]);

/* Byte sizes. group: 1 = G1, 2 = G2. */
int oracle_point_size(int curve, int group, int compressed);
int oracle_fr_limbs(int curve);

/* Phase1Parameters geometry (reference: src/utils.rs:326-352). mode 0 = chunked, 1 = full.
 * out[0..5] = g1_chunk_size, other_chunk_size, accumulator_size, contribution_size,
 * public_key_size, start index. */
int oracle_params(int curve, int mode, uint64_t chunk_index, uint64_t chunk_size, uint32_t power,
                  uint64_t out[6]);

/* setup-utils apply_powers: out[i] = (coeff * tau^(start+i)) * in[i], i in [0,n).
 * coeff may be NULL.  nthreads >= 1 splits the index range. */
int oracle_apply_powers(int curve, int group, const uint8_t *in, int compressed_in, uint8_t *out,
                        int compressed_out, uint64_t n, uint64_t start, const uint64_t *tau,
                        const uint64_t *coeff, int nthreads);

/* Phase1::computation (reference call site: src/bin/contribute.rs:811-823): element region only;
 * `in`/`out` point at the whole challenge / response buffers (64-byte hash header skipped,
 * public key tail untouched). */
int oracle_computation(int curve, int mode, uint64_t chunk_index, uint64_t chunk_size, uint32_t power,
                       const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                       const uint64_t *tau, const uint64_t *alpha, const uint64_t *beta, int nthreads);

/* Phase1::decompress for one vector (reference call site: src/bin/contribute.rs:968-986). */
int oracle_decompress(int curve, int group, const uint8_t *in, uint8_t *out, uint64_t n, int nthreads);

/* Generic scalar multiplication of serialized points by per-point scalars (k limbs each,
 * canonical LE); used for subgroup checks (scalar = r) and MSM cross-checks. */
int oracle_scalar_mul(int curve, int group, const uint8_t *in, int compressed_in, uint8_t *out,
                      int compressed_out, uint64_t n, const uint64_t *scalars, int limbs_per_scalar,
                      int shared_scalar);

/* sum_i scalars[i] * in[i] -> one serialized point. */
int oracle_msm(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n,
               const uint64_t *scalars, int limbs_per_scalar, uint8_t *out, int compressed_out);

#ifdef __cplusplus
}
#endif
#endif
