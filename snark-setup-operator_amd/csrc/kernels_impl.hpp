// HIP kernels of the chunked powers-of-tau hot path, templated over one group of one curve; each curve's translation
// unit instantiates them (k_<curve>.hip) and capi.hip orchestrates them behind the C ABI (include/ptau.h).
//
// Replaces setup-utils' apply_powers / batch_exp / read_batch / write_batch, Phase1::decompress, check_subgroup and
// dense_multiexp / power_pairs — reference call sites /root/reference/src/bin/contribute.rs:811-823, 968-986,
// src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591.
//
//   K1  window_scalar_mul_kernel : thread i derives e_i = c * tau^(start+i) from a tau^(2^k) table, reads point i, builds
//                                  its window table 1P..8P in HBM and runs the signed 4-bit window multiplication —
//                                  GLV (2 half-length scalars, phi) on BLS12-377 G1 and BW6-761, GLS (4 x 64-bit, psi) on
//                                  BLS12-377 G2 with every Fq2 element spread over a lane pair — Jacobian result to scratch;
//       normalize_kernel         : Montgomery-trick batched inversion, affine conversion, arkworks serialisation.
//   K2  decompress_kernel        : compressed -> uncompressed (square root).
//   K3  endo_subgroup_kernel     : prime-order membership through phi / psi (BLS12-377); other curves: K1 with the group
//                                  order as scalar + expect_infinity_kernel.
//   K4  msm_* / reduce_rows      : bucket-method multi-scalar multiplication (plain sum and the shifted sum of power_pairs).
// The per-thread bodies are host+device so that tests/native/host_emul.hip can run the SAME code on the CPU against the
// oracle (test infrastructure only: the product has no CPU path).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ec.hpp"
#include "fpz.hpp"
#include "fplz.hpp"

#ifndef PTAU_G1_MIN_WAVES
#define PTAU_G1_MIN_WAVES 2
#endif
#ifndef PTAU_K1_G1_MIN_WAVES
// tau-update of a 13-limb G1: 0 = as PTAU_G1_MIN_WAVES.  With 29-bit limbs two waves (247 VGPRs, 64 B of spills) run at
// 31.4 M updates/s and three waves (168 VGPRs, 384 B of spills, 6x the store traffic) at 31.1 M; with 32-bit limbs three
// waves were 4 % ahead (26.1 vs 25.1 M).
#define PTAU_K1_G1_MIN_WAVES 0
#endif
#ifndef PTAU_LANE_MIN_WAVES
#define PTAU_LANE_MIN_WAVES 2
#endif
#ifndef PTAU_MIN_WAVES_WIDE
#define PTAU_MIN_WAVES_WIDE 2
#endif
#ifndef PTAU_SPLIT_MIN_WAVES
#define PTAU_SPLIT_MIN_WAVES 2   // limb-split (FpL) kernels: waves per SIMD they are compiled for
#endif

namespace ptau {

// Error bits accumulated in a device word by the kernels.
enum : int { ERRBIT_NONCANONICAL = 1, ERRBIT_NOT_ON_CURVE = 2, ERRBIT_INFINITY = 4, ERRBIT_NOT_IN_SUBGROUP = 8 };

constexpr int SCALAR_TABLE_ENTRIES = 64;   // tau^(2^k), k < 64, preceded by the base coefficient

// How a point is laid over the lanes of a wavefront: one lane per point; an Fq2 element over a lane pair, one coefficient
// per lane (Fp2L, ext.hpp: BLS12-377 G2); every base-field element over a lane pair, half of the limbs per lane (FpL,
// fpl.hpp: the 26-28-limb fields of MNT4/MNT6-753 and BW6-761).  The last two run two threads per point.
enum : int { LANES_ONE = 0, LANES_FQ2_PAIR = 1, LANES_LIMB_SPLIT = 2 };

// A carry-free twin field usable with one lane per element (FpZ): 14 limbs, p = 1 mod 2^28.
template <class P>
struct LazyOneLane { static constexpr bool value = P::N <= 14 && P::MOD[0] == 1u; };
template <>
struct LazyOneLane<void> { static constexpr bool value = false; };

// Host+device view of one group's types.
template <class FqP, class FrP, class Cfg, int LANE_MODE_ = LANES_ONE>
struct GroupTypes {
    static constexpr int LANE_MODE = LANE_MODE_;
    // Fp (32-bit limbs) or Fp29 (29-bit limbs), as the parameter struct was generated; FpL: the Fp29 element over two lanes
    using Fq = std::conditional_t<LANE_MODE == LANES_LIMB_SPLIT, FpL<FqP>, FieldOf<FqP>>;
    using Fr = Fp<FrP>;        // scalars are consumed as packed 32-bit words
    using F = std::conditional_t<Cfg::DEG == 1, Fq,
              std::conditional_t<Cfg::DEG == 2, std::conditional_t<LANE_MODE == LANES_FQ2_PAIR, Fp2L<Fq, Cfg::NR>, Fp2<Fq, Cfg::NR>>,
                                 Fp3<Fq, Cfg::NR>>>;
    static constexpr int LPP = F::LANES;   // lanes (threads) per point
    using J = Jac<F, Cfg>;
    using A = Affine<F>;
    static constexpr int COMP = F::BYTES;
    static constexpr int UNCOMP = 2 * F::BYTES;
    // Waves per SIMD the kernels are compiled for (0 = the compiler's choice).  Groups whose point operations are
    // inlined (12-limb G1, lane-pair G2) are held to 256 registers = two waves: left alone, the membership kernel of
    // BLS12-377 G1 took 494 (VGPRs + AGPRs) and ran one wave per SIMD.  Point operations over the wide fields
    // go through out-of-line callees the bound does not reach (single-lane Fq2 / Fq3 callees keep their 512 registers and
    // one wave per SIMD), but the 24-limb prime-field groups' callees fit 248 VGPRs, so bounding the kernel body (276
    // otherwise) is what gets BW6-761 and MNT G1 to two waves.
    static constexpr bool SMALL = F::DEG == 1 && F::POINT_OPS_INLINE;   // a point operation inlines at <= 256 registers
    static constexpr int K1_MIN_WAVES = (LPP == 1 && SMALL) ? PTAU_K1_G1_MIN_WAVES : 0;   // 0: same as MIN_WAVES
    // K1's per-point window table with a COMMON Z (no inversion): the eight multiples are rescaled to one denominator Zc =
    // prod Z_e (~50 products), which makes them AFFINE points of the isomorphic curve y^2 = x^3 + b Zc^6.  On a j = 0 curve
    // neither the doubling nor the mixed addition involves b, and (x, y) -> (beta x, y) is the same endomorphism there, so
    // the whole window loop runs on that curve with mixed additions (7M + 4S instead of 11M + 5S, two coordinates per
    // table entry instead of three) and the result (X, Y, Z) maps back as (X, Y, Z Zc).  Exact for every curve point.
    // (A per-point inversion for a truly affine table was measured first: the binary inversion diverges across lanes and
    // cost ~400 products per point — 5 % slower than the Jacobian table.)
#ifndef PTAU_AFFINE_TABLE
#define PTAU_AFFINE_TABLE 1
#endif
    static constexpr bool AFFINE_TABLE = PTAU_AFFINE_TABLE && Cfg::A_IS_ZERO && F::DEG <= 2 && F::POINT_OPS_INLINE;
    // K1's window loop in the carry-free twin of the base field (fpz.hpp): BLS12-377 G1, and G2 with Fq2 over a lane pair.
    // The point is converted once, table and loop run there, the accumulator is converted back before the result store.
#ifndef PTAU_LAZY_LOOP
#define PTAU_LAZY_LOOP 1
#endif
    using LazyP = typename LazyTwin<FqP>::type;
    static constexpr bool LAZY_LOOP = PTAU_LAZY_LOOP && AFFINE_TABLE && !std::is_void<LazyP>::value &&
                                      ((LANE_MODE == LANES_ONE && F::DEG == 1 && LazyOneLane<LazyP>::value) ||
                                       (LANE_MODE == LANES_FQ2_PAIR && F::DEG == 2 && LazyOneLane<LazyP>::value) ||
                                       (LANE_MODE == LANES_LIMB_SPLIT && F::DEG == 1));
    static constexpr int MIN_WAVES = LANE_MODE == LANES_LIMB_SPLIT ? PTAU_SPLIT_MIN_WAVES : LPP == 2 ? PTAU_LANE_MIN_WAVES : SMALL ? PTAU_G1_MIN_WAVES : (PTAU_MIN_WAVES_WIDE > 0 ? PTAU_MIN_WAVES_WIDE : 1);
};

// table layout: [0] = base coefficient c*tau^start, [1+k] = tau^(2^k); all Montgomery, Fr::N limbs
template <class Fr>
PTAU_HD Fr scalar_for_index(const uint32_t *__restrict__ table, uint64_t idx) {
    Fr e;
#pragma unroll
    for (int i = 0; i < Fr::N; i++) e.l[i] = table[i];
    for (int k = 0; idx != 0; k++, idx >>= 1) {
        if (idx & 1) {
            Fr t;
#pragma unroll
            for (int i = 0; i < Fr::N; i++) t.l[i] = table[(1 + k) * Fr::N + i];
            e = Fr::mul(e, t);
        }
    }
    return e;
}

// MSB-first double-and-add with mixed additions (ark-ec sw_double_and_add_affine).
template <class G, int NL>
PTAU_HD typename G::J scalar_mul_bits(const typename G::A &p, const uint32_t (&k)[NL], int top_bit) {
    using J = typename G::J;
    J acc = J::infinity();
    for (int i = top_bit; i >= 0; i--) {
        acc = J::dbl(acc);
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) w = (j == (i >> 5)) ? k[j] : w;   // register select, no scratch
        if ((w >> (i & 31)) & 1u) acc = J::add_mixed(acc, p);
    }
    return acc;
}

// Per-thread bodies are host+device so tests/native/host_emul.hip can run the SAME code on the
// CPU (test infrastructure only; the product has no CPU path).  Each returns its error bits.
template <class G>
PTAU_HD int read_point_any(const uint8_t *in, int compressed_in, uint64_t i, typename G::A &p) {
    using J = typename G::J;
    int bad = 0;
    if (compressed_in) {
        int rc = read_compressed<typename G::F, typename J::CfgT>(in + i * G::COMP, p);
        if (rc == 1) bad |= ERRBIT_NONCANONICAL;
        if (rc == 2) bad |= ERRBIT_NOT_ON_CURVE;
    } else {
        if (!read_uncompressed<typename G::F>(in + i * G::UNCOMP, p)) bad |= ERRBIT_NONCANONICAL;
    }
    return bad;
}

template <class G>
PTAU_HD int scalar_mul_thread(const uint8_t *in, int compressed_in, typename G::J *scratch, uint64_t i,
                              const uint32_t *table, int check_input) {
    using Fr = typename G::Fr;
    using A = typename G::A;
    using J = typename G::J;
    A p;
    int bad = read_point_any<G>(in, compressed_in, i, p);
    if ((check_input & 1) && p.inf) bad |= ERRBIT_INFINITY;
    if (bad) {
        scratch[i] = J::infinity();
        return bad;
    }
    Fr e = scalar_for_index<Fr>(table, i);
    uint32_t k[Fr::N];
    e.to_canonical(k);
    scratch[i] = scalar_mul_bits<G, Fr::N>(p, k, Fr::Params::BITS - 1);
    return 0;
}

template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
scalar_mul_kernel(const uint8_t *__restrict__ in, int compressed_in, typename G::J *__restrict__ scratch,
                  uint64_t n, const uint32_t *__restrict__ table, int check_input, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = scalar_mul_thread<G>(in, compressed_in, scratch, i, table, check_input);
    if (bad) atomicOr(err, bad);
}

// ---- windowed variant: signed 4-bit digits, per-point table of 1P..8P (Jacobian) in HBM ------------
// The update is VARIABLE-base (every element has its own base and its own scalar), so the window
// table is per point: 8 Jacobian multiples = 8 x 3|F| bytes, 1.2 GB for 2^20 BLS12-377 G1 points —
// cheap on a 288 GB part and the live working set (resident waves only) sits in the Infinity Cache.
// Digits: d_w = nibble_w(k + 0x88..8) - 8 in [-8, 7], so any digit is available MSB-first without a
// carry pass.  Table layout is entry-major, table[e * n + i], so lanes holding the same digit coalesce.
// Where a thread's scalar comes from.
// SCALAR_SUBGROUP_ENDO: no scalar at all — the two fixed half-length multipliers (a, b) of the endomorphism membership test
// [a]P + [b]phi(P) == O (lattice-GLV curves: BW6-761; gen_constants.py SUB_A / SUB_B), through the GLV window loop
// SCALAR_FIXED_PSI: SCALAR_FIXED with the scalar |t - 1| of the psi membership test (MNT twists); a kernel that can compares
// its result with psi(P) itself (k1_lazy_a.hpp) and Launch::psi_check then launches nothing
enum : int { SCALAR_TAU_TABLE = 0, SCALAR_FIXED = 1, SCALAR_ARRAY = 2, SCALAR_PRF = 3, SCALAR_SUBGROUP_ENDO = 4, SCALAR_FIXED_PSI = 5 };
struct ScalarSource {
    int mode;
    const uint32_t *ptr;      // TAU_TABLE: the tau^(2^k) table; FIXED: NL canonical limbs; ARRAY: n x NL canonical limbs
    uint64_t seed0, seed1;    // PRF: 128-bit seed
    uint64_t index_base;      // PRF: global index of element 0 (slices of one vector agree across GPUs)
    int prf_bits = 0;         // PRF: width of rho in bits (0 = PRF_BITS; up to 255 for the membership-fused MSM, see msm_plan)
};

// The random coefficient rho_idx of merge_pairs / power_pairs.  setup-utils draws Fr::rand from thread_rng — rand 0.8's
// ChaCha12 generator; here the SAME cipher as a counter-mode PRF, so that any slice of a vector, on any GPU, sees the same
// rho for the same global index: one ChaCha12 block per element, key = the per-verification 128-bit secret (from the OS)
// followed by a fixed domain tag, block counter = the element's global index; rho = the low `bits` bits of words 0..7 of the
// block (PRF_BITS = 127: words 0..3 with the top bit cleared).  ~600 plain ALU operations per element, against the ~1400 of
// converting the point.
PTAU_HD uint32_t prf_rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
PTAU_HD void prf_rho(uint64_t seed0, uint64_t seed1, uint64_t idx, int bits, uint32_t (&out)[8]) {
    const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                            (uint32_t)seed0, (uint32_t)(seed0 >> 32), (uint32_t)seed1, (uint32_t)(seed1 >> 32),
                            0x75617470u /* "ptau" */, 0x6f68722du /* "-rho" */, 0x72657020u, 0x78646e69u,
                            (uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define PTAU_QR(a, b, c, d)                                   \
    x[a] += x[b]; x[d] = prf_rotl(x[d] ^ x[a], 16);          \
    x[c] += x[d]; x[b] = prf_rotl(x[b] ^ x[c], 12);          \
    x[a] += x[b]; x[d] = prf_rotl(x[d] ^ x[a], 8);           \
    x[c] += x[d]; x[b] = prf_rotl(x[b] ^ x[c], 7);
#pragma unroll
    for (int r = 0; r < 6; r++) {   // ChaCha12: six double rounds
        PTAU_QR(0, 4, 8, 12) PTAU_QR(1, 5, 9, 13) PTAU_QR(2, 6, 10, 14) PTAU_QR(3, 7, 11, 15)
        PTAU_QR(0, 5, 10, 15) PTAU_QR(1, 6, 11, 12) PTAU_QR(2, 7, 8, 13) PTAU_QR(3, 4, 9, 14)
    }
#undef PTAU_QR
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int lo = 32 * i;
        const uint32_t m = bits >= lo + 32 ? 0xffffffffu : (bits <= lo ? 0u : ((1u << (bits - lo)) - 1u));
        out[i] = (x[i] + s[i]) & m;
    }
}

// Number of random bits in a power_pairs coefficient.  setup-utils draws full-width Fr::rand; the check
// sum_i rho_i (v_{i+1} - tau v_i) == O is equally sound with 127-bit rho (a cheating vector passes with
// probability 2^-127: distinct 127-bit values stay distinct mod r > 2^252), and the MSM is half as long.
constexpr int PRF_BITS = 127;   // + the sign-recoding carry = 128 = 8 full windows of 16 bits at 2^20 points
// Widest rho the membership-fused MSM asks for (msm_plan with bucket_test_cost > 0 at >= 4096 points: c >= 5, W c - 1 <= 164);
// every scalar field here has more than 252 bits, so the values stay distinct and canonical.
constexpr int PRF_MAX_BITS = 200;

// The scalar of element i under a ScalarSource, as NL canonical 32-bit limbs.
template <class Fr>
PTAU_HD void thread_scalar(const ScalarSource &src, uint64_t i, uint32_t (&k)[Fr::N]) {
    constexpr int NL = Fr::N;
    if (src.mode == SCALAR_FIXED || src.mode == SCALAR_FIXED_PSI) {   // one public scalar for every point (subgroup check)
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = src.ptr[j];
    } else if (src.mode == SCALAR_ARRAY) {   // explicit per-point scalars (MSM with caller-supplied coefficients)
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = src.ptr[i * NL + j];
    } else if (src.mode == SCALAR_PRF) {     // random linear combination coefficients
        static_assert(NL >= 8 && Fr::Params::BITS > PRF_MAX_BITS, "scalar field too small for the PRF width");
        uint32_t rho[8];
        prf_rho(src.seed0, src.seed1, src.index_base + i, src.prf_bits ? src.prf_bits : PRF_BITS, rho);
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = j < 8 ? rho[j] : 0u;
    } else {
        Fr ek = scalar_for_index<Fr>(src.ptr, i);
        ek.to_canonical(k);
    }
}

// GLV decomposition for curves with the endomorphism phi(x, y) = (beta x, y) (BLS12-377 G1):
// k = k1 + k2 * lambda with k2 = floor(k / lambda) through a 2^256 reciprocal; k1, k2 < 2^127.
// Valid ONLY for points of the prime-order subgroup (phi = [lambda] there): callers select it for the
// tau-update / MSM of verified elements, never for the subgroup check itself.
template <class Cfg>
PTAU_HD void glv_decompose(const uint32_t (&k)[8], uint32_t (&k1)[4], uint32_t (&k2)[4]) {
    // qhat = (k * M) >> 256, M = floor(2^256 / lambda) (5 limbs)
    uint32_t prod[13];
    {
        uint64_t acc = 0, carry_hi = 0;
        for (int col = 0; col < 13; col++) {
            uint64_t lo = acc, hi = carry_hi;
            acc = 0;
            carry_hi = 0;
            for (int i = 0; i < 8; i++) {
                int j = col - i;
                if (j < 0 || j >= 5) continue;
                uint64_t t = (uint64_t)k[i] * Cfg::GLV_M[j];
                lo += t;
                if (lo < t) hi++;
            }
            prod[col] = (uint32_t)lo;
            acc = (lo >> 32) | (hi << 32);
            carry_hi = hi >> 32;
        }
    }
    uint32_t qh[5] = {prod[8], prod[9], prod[10], prod[11], prod[12]};
    // rem = k - qhat * lambda   (fits 5 limbs: < 3 lambda < 2^128)
    uint32_t ql[5];
    {
        uint64_t acc = 0, carry_hi = 0;
        for (int col = 0; col < 5; col++) {
            uint64_t lo = acc, hi = carry_hi;
            acc = 0;
            carry_hi = 0;
            for (int i = 0; i < 5; i++) {
                int j = col - i;
                if (j < 0 || j >= 4) continue;
                uint64_t t = (uint64_t)qh[i] * Cfg::GLV_LAMBDA[j];
                lo += t;
                if (lo < t) hi++;
            }
            ql[col] = (uint32_t)lo;
            acc = (lo >> 32) | (hi << 32);
            carry_hi = hi >> 32;
        }
    }
    uint32_t rem[5];
    uint32_t br = 0;
    for (int i = 0; i < 5; i++) rem[i] = subb(k[i], ql[i], br);
    // at most two corrections
    for (int it = 0; it < 3; it++) {
        uint32_t t[5];
        uint32_t b2 = 0;
        for (int i = 0; i < 5; i++) t[i] = subb(rem[i], i < 4 ? Cfg::GLV_LAMBDA[i] : 0u, b2);
        if (b2) break;   // rem < lambda
        for (int i = 0; i < 5; i++) rem[i] = t[i];
        uint32_t c = 1;
        for (int i = 0; i < 5; i++) qh[i] = addc(qh[i], 0u, c);
    }
    for (int i = 0; i < 4; i++) { k1[i] = rem[i]; k2[i] = qh[i]; }
}

// ---- small fixed-size big-integer helpers for the scalar decompositions (compile-time sizes, register arrays) ----
template <int NA, int NB>
PTAU_HD void bn_mul(const uint32_t (&a)[NA], const uint32_t (&b)[NB], uint32_t (&out)[NA + NB]) {
#pragma unroll
    for (int i = 0; i < NA + NB; i++) out[i] = 0;
#pragma unroll
    for (int i = 0; i < NA; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;   // <= 2^64 - 1
            out[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        out[i + NB] = carry;
    }
}
// acc (two's complement, N limbs) += sign * t (NT limbs, non-negative)
template <int N, int NT>
PTAU_HD void bn_addsub(uint32_t (&acc)[N], const uint32_t (&t)[NT], int sign) {
    uint32_t c = 0;
    if (sign > 0) {
#pragma unroll
        for (int i = 0; i < N; i++) acc[i] = addc(acc[i], i < NT ? t[i] : 0u, c);
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) acc[i] = subb(acc[i], i < NT ? t[i] : 0u, c);
    }
}
// |acc| -> out (low NO limbs), returns the sign
template <int N, int NO>
PTAU_HD bool bn_abs(uint32_t (&acc)[N], uint32_t (&out)[NO]) {
    const bool neg = (acc[N - 1] >> 31) != 0;
    if (neg) {
        uint32_t c = 1;
#pragma unroll
        for (int i = 0; i < N; i++) acc[i] = addc(~acc[i], 0u, c);
    }
#pragma unroll
    for (int i = 0; i < NO; i++) out[i] = acc[i];
    return neg;
}

// Lattice GLV decomposition (BW6-761, both groups): k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^191, signs returned.
// c_i = (k G_i) >> 384 approximate the Babai rounding coefficients of the short basis (gen_constants.py checks the
// bound over edge and random scalars).  Valid ONLY for points of the prime-order subgroup.
template <class Cfg>
PTAU_HD void glv2_decompose(const uint32_t (&k)[12], uint32_t (&k1)[6], uint32_t (&k2)[6], bool &neg1, bool &neg2) {
    uint32_t g1[7], g2[7], a1[6], a2[6], b1[6], b2[6];
#pragma unroll
    for (int i = 0; i < 7; i++) { g1[i] = Cfg::GLV2_G1[i]; g2[i] = Cfg::GLV2_G2[i]; }
#pragma unroll
    for (int i = 0; i < 6; i++) { a1[i] = Cfg::GLV2_A1[i]; a2[i] = Cfg::GLV2_A2[i]; b1[i] = Cfg::GLV2_B1[i]; b2[i] = Cfg::GLV2_B2[i]; }
    uint32_t p1[19], p2[19], c1[7], c2[7];
    bn_mul<12, 7>(k, g1, p1);
    bn_mul<12, 7>(k, g2, p2);
#pragma unroll
    for (int i = 0; i < 7; i++) { c1[i] = p1[12 + i]; c2[i] = p2[12 + i]; }
    uint32_t t11[13], t12[13], t21[13], t22[13];
    bn_mul<7, 6>(c1, a1, t11);
    bn_mul<7, 6>(c2, a2, t12);
    bn_mul<7, 6>(c1, b1, t21);
    bn_mul<7, 6>(c2, b2, t22);
    uint32_t acc[14];
#pragma unroll
    for (int i = 0; i < 14; i++) acc[i] = i < 12 ? k[i] : 0u;
    bn_addsub<14, 13>(acc, t11, Cfg::GLV2_E11);
    bn_addsub<14, 13>(acc, t12, Cfg::GLV2_E12);
    neg1 = bn_abs<14, 6>(acc, k1);
#pragma unroll
    for (int i = 0; i < 14; i++) acc[i] = 0;
    bn_addsub<14, 13>(acc, t21, Cfg::GLV2_E21);
    bn_addsub<14, 13>(acc, t22, Cfg::GLV2_E22);
    neg2 = bn_abs<14, 6>(acc, k2);
}

// 4-dimensional decomposition for BLS12 G2 (Galbraith-Scott): psi = twist^-1 o Frobenius o twist acts as [X] on the
// prime-order subgroup and r = X^4 - X^2 + 1 < X^4, so the base-X digits of k, k = k0 + k1 X + k2 X^2 + k3 X^3 with
// 0 <= k_j < X < 2^64, give k P = sum_j k_j psi^j(P): 64 doublings instead of 253.  Three restoring divisions by the
// 64-bit constant X, bit-serial (a few thousand integer operations against ~3 M for the multiplication itself).
// Valid ONLY for points of the prime-order subgroup, like the G1 GLV path.
template <class Cfg>
PTAU_HD void gls4_decompose(const uint32_t (&k)[8], uint64_t (&d)[4]) {
    constexpr uint64_t X = Cfg::ENDO_X;
    uint64_t q[4];
#pragma unroll
    for (int l = 0; l < 4; l++) q[l] = (uint64_t)k[2 * l] | ((uint64_t)k[2 * l + 1] << 32);
    for (int j = 0; j < 3; j++) {
        uint64_t rem = 0, nq[4];
#pragma unroll
        for (int l = 3; l >= 0; l--) {
            uint64_t w = q[l], out = 0;
            for (int b = 0; b < 64; b++) {
                const uint64_t top = rem >> 63;
                rem = (rem << 1) | (w >> 63);
                w <<= 1;
                out <<= 1;
                if (top || rem >= X) {   // true value rem + 2^64 top is below 2X: one subtraction, wrap-around exact
                    rem -= X;
                    out |= 1;
                }
            }
            nq[l] = out;
        }
        d[j] = rem;
#pragma unroll
        for (int l = 0; l < 4; l++) q[l] = nq[l];
    }
    d[3] = q[0];
}

constexpr int WINDOW_BITS = 4;
constexpr int WINDOW_ENTRIES = 8;

// Result store in the STANDARD Jacobian layout (X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1 for Fq2) whatever the lane mapping,
// so normalize_kernel of the single-lane group type reads what a lane-pair kernel wrote.
template <class G>
PTAU_HD void store_jac(typename G::J *scratch, uint64_t i, const typename G::J &v) {
    if constexpr (G::LANE_MODE == LANES_LIMB_SPLIT) {
        // every base-field coefficient in turn, this lane's H limbs at their place among the N of the single-lane element
        using L = typename G::Fq;
        constexpr int D = G::F::DEG;
        uint32_t *dst = reinterpret_cast<uint32_t *>(scratch) + i * (uint64_t)(3 * D * L::N) + (lane_parity() ? L::H : 0);
        auto put = [&](const L &c, int slot) {
#pragma unroll
            for (int j = 0; j < L::H; j++) dst[slot * L::N + j] = c.l[j];
        };
        auto put_f = [&](const typename G::F &f, int slot) {
            if constexpr (D == 1) put(f, slot);
            else if constexpr (D == 2) { put(f.c0, slot); put(f.c1, slot + 1); }
            else { put(f.c0, slot); put(f.c1, slot + 1); put(f.c2, slot + 2); }
        };
        put_f(v.X, 0);
        put_f(v.Y, D);
        put_f(v.Z, 2 * D);
    } else if constexpr (G::LPP == 2) {
        typename G::Fq *base = reinterpret_cast<typename G::Fq *>(scratch) + i * 6 + lane_parity();
        base[0] = v.X.c;
        base[2] = v.Y.c;
        base[4] = v.Z.c;
    } else {
        scratch[i] = v;
    }
}

// The carry-free field of a group (named only under G::LAZY_LOOP): the loop's accumulator type, conversion from / to the
// group's reduced field, and this lane's limbs of an element.
template <class K, bool PAIR>
struct LazyPairNrm { using type = void; };
template <class K>
struct LazyPairNrm<K, true> { using type = typename K::template E<1, 2>; };

template <class G>
struct LazyOf {
    using F = typename G::F;
    static constexpr bool PAIR = F::DEG == 2;                            // Fq2 over a lane pair, FpZ coefficients
    static constexpr bool SPLIT = G::LANE_MODE == LANES_LIMB_SPLIT;      // a wide prime field split over a lane pair
    using LZ = std::conditional_t<SPLIT, FpLZ<typename G::LazyP>, FpZ<typename G::LazyP>>;
    using K = std::conditional_t<PAIR, Fp2Z<LZ, G::J::CfgT::NR>, LZ>;
    using Acc = std::conditional_t<PAIR, LazyJac2<K>, LazyJac<LZ>>;
    using AX = typename Acc::AX;
    using AY = typename Acc::AY;
    static constexpr int N = SPLIT ? LZ::N / 2 : LZ::N;                  // limbs held by one lane
    // a converted element (a product: limbs normalised, value below 2p), and a product of the group's field
    using C = std::conditional_t<PAIR, typename LazyPairNrm<K, PAIR>::type, typename LZ::Nrm>;
    PTAU_HD static C to_lazy(const F &x) {
        if constexpr (PAIR) return {LZ::from29(x.c)};
        else return LZ::from29(x);
    }
    template <class T>
    PTAU_HD static F from_lazy(const T &v) {
        if constexpr (PAIR) return {LZ::to29(v.c)};
        else return LZ::to29(v);
    }
    template <class T>
    PTAU_HD static T load(const uint32_t *w) {   // this lane's N limbs of one coordinate
        T r;
#pragma unroll
        for (int j = 0; j < N; j++) {
            if constexpr (PAIR) r.c.l[j] = w[j];
            else r.l[j] = w[j];
        }
        return r;
    }
    template <class T>
    PTAU_HD static void store(uint32_t *w, const T &v) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            if constexpr (PAIR) w[j] = v.c.l[j];
            else w[j] = v.l[j];
        }
    }
    PTAU_HD static void store_zero(uint32_t *w) {
#pragma unroll
        for (int j = 0; j < N; j++) w[j] = 0;
    }
    PTAU_HD static bool all_lanes(bool mine) {   // true in every lane that holds a part of this point
        if constexpr (PAIR) return K::both(mine);
        else if constexpr (SPLIT) return LZ::both(mine);
        else return mine;
    }
};

// The same address as far as the hardware goes, a NEW one as far as the optimiser goes: a re-read through it is not merged
// with an earlier read of the same words (whose registers may then die early) and stays where it is written.
PTAU_HD const uint32_t *reread_ptr(const uint32_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(p));
#endif
    return p;
}

template <class G, bool LAZY>
struct LoopAcc {
    using type = typename G::J;
    using zc_type = char;   // unused
};
template <class G>
struct LoopAcc<G, true> {
    using type = typename LazyOf<G>::Acc;
    using L = LazyOf<G>;
    using zc_type = decltype(L::K::mul(std::declval<typename L::C>(), std::declval<typename L::C>()));   // a product
};

template <class G>
PTAU_HD int window_scalar_mul_thread(const uint8_t *in, int compressed_in, typename G::J *scratch,
                                     typename G::J *wtab, uint64_t n, uint64_t i, const ScalarSource &src,
                                     int check_input, int use_glv = 0) {
    using Fr = typename G::Fr;
    using F = typename G::F;
    using A = typename G::A;
    using J = typename G::J;
    constexpr int NL = Fr::N;
    constexpr int NW = NL * 8;   // nibbles in the limb array (k + 0x88..8 never carries out: BITS <= 32*NL - 3)
    static_assert(Fr::Params::BITS <= 32 * NL - 3, "scalar field leaves no headroom for the digit offset");
    A p;
    int bad = read_point_any<G>(in, compressed_in, i, p);
    if ((check_input & 1) && p.inf) bad |= ERRBIT_INFINITY;
    if (bad || p.inf) {
        store_jac<G>(scratch, i, J::infinity());
        return bad;
    }
    // table: e -> (e+1) * P   (lane-pair groups: one half-width Jacobian per lane, the pair adjacent)
    const uint64_t slot = i * G::LPP + (G::LPP == 2 ? lane_parity() : 0u), nslots = n * G::LPP;
    F zc = F::one();   // the table's common denominator (AFFINE_TABLE)
    using L = LazyOf<G>;                                   // named only: instantiated under LAZY_LOOP
    using AccT = typename LoopAcc<G, G::LAZY_LOOP>::type;   // the window loop's accumulator
    typename LoopAcc<G, G::LAZY_LOOP>::zc_type zcz{};       // ... and the common denominator in its field
    if constexpr (G::LAZY_LOOP) {
        // The whole table in the carry-free field: the point converted once, seven mixed additions, the running product of the
        // Z's, then every entry rescaled to the common denominator as below (same construction, same exceptional cases).
        // Entry e keeps X | Y as 2 x N limbs per lane from the start of its slot; the Z's stay in private memory.
        using K = typename L::K;
        using KM = typename LoopAcc<G, true>::zc_type;
        static_assert(sizeof(J) >= 2 * L::N * sizeof(uint32_t), "a converted entry must fit the Jacobian slot");
        auto slot_w = [&](int e) { return reinterpret_cast<uint32_t *>(&wtab[(uint64_t)e * nslots + slot]); };
        // Liveness (see LazyJac2::add_src): an addition keeps up to seven field elements alive across its product calls, and
        // the callee-saved registers hold eight.  So nothing else lives in registers over the table's additions: P itself is
        // parked in entry 0's slot (unscaled; rescaled last) and read back where the formula consumes it, and the running
        // product of the Z's is taken in a pass of its own.
        uint32_t *const w0 = slot_w(0);
        decltype(AccT{}.Zc) zs[WINDOW_ENTRIES - 1];
        KM pre[WINDOW_ENTRIES - 1];
        {
            const typename L::C px = L::to_lazy(p.x), py = L::to_lazy(p.y);
            L::store(w0, px);
            L::store(w0 + L::N, py);
            struct ParkedP {   // (x, y) of P from entry 0's slot, re-read for every addition (never held in registers across them)
                const uint32_t *w;
                PTAU_HD typename L::C coord(int c) const {
                    const uint32_t *q = reread_ptr(w + c * L::N);
                    typename L::C v;
#pragma unroll
                    for (int j = 0; j < L::N; j++) {
                        if constexpr (L::PAIR) v.c.l[j] = q[j];
                        else v.l[j] = q[j];
                    }
                    return v;
                }
                PTAU_HD typename L::AX x() const { return coord(0); }
                PTAU_HD typename L::AY y() const { return coord(1); }
                PTAU_HD typename L::AX x_again() const { return coord(0); }
                PTAU_HD typename L::AY y_again() const { return coord(1); }
            };
            AccT cur = {px, py, K::one()};
#pragma nounroll
            for (int e = 1; e < WINDOW_ENTRIES; e++) {
                cur = AccT::add_src(cur, ParkedP{w0});
                uint32_t *w = slot_w(e);
                L::store(w, cur.X);
                L::store(w + L::N, cur.Y);
                zs[e - 1] = cur.Zc;
            }
        }
        KM zl = K::one();
#pragma nounroll
        for (int e = 1; e < WINDOW_ENTRIES; e++) {
            pre[e - 1] = zl;
            if (!K::is_zero(zs[e - 1])) zl = K::mul(zl, zs[e - 1]);
        }
        // Fq2: a REAL common denominator Zc conj(Zc), so that psi keeps entries on it (see below)
        auto suf = [&] {
            if constexpr (L::PAIR) return K::conj(zl);
            else return KM(K::one());
        }();
#pragma nounroll
        for (int e = WINDOW_ENTRIES - 1; e >= 1; e--) {
            uint32_t *w = slot_w(e);
            const auto ze = zs[e - 1];
            if (K::is_zero(ze)) {   // the multiple is O: (0, 0) marks it
                L::store_zero(w);
                L::store_zero(w + L::N);
            } else {
                const KM c = K::mul(pre[e - 1], suf);
                suf = K::mul(suf, ze);
                const auto c2 = K::sqr(c);
                const KM x = K::mul(L::template load<decltype(AccT{}.X)>(w), c2);
                const KM y = K::mul(L::template load<decltype(AccT{}.Y)>(w + L::N), K::mul(c2, c));
                L::store(w, x);
                L::store(w + L::N, y);
            }
        }
        {   // entry 0 = P itself (Z = 1): c = Zc
            if constexpr (L::PAIR) zcz = K::mul(zl, K::conj(zl));
            else zcz = zl;
            const auto c2 = K::sqr(zcz);
            L::store(w0, KM(K::mul(L::template load<typename L::C>(w0), c2)));
            L::store(w0 + L::N, KM(K::mul(L::template load<typename L::C>(w0 + L::N), K::mul(c2, zcz))));
        }
    }
    J cur = J::infinity();
    if constexpr (!G::LAZY_LOOP) {
        cur = J::from_affine(p);
        wtab[slot] = cur;
    }
    auto put_entry = [&](int e, const F &x, const F &y) {   // affine table entry e: (x, y) in the X and Y slots
        J *t = &wtab[(uint64_t)e * nslots + slot];
        t->X = x;
        t->Y = y;
    };
    if constexpr (G::LAZY_LOOP) {
        // table built above
    } else if constexpr (G::AFFINE_TABLE) {
        // Jacobian multiples first, the running product of their Z's on the side; then, last to first, entry e is scaled by
        // c_e = Zc / Z_e = (product of the Z's before it) * (product of those after it): (X c^2, Y c^3) written to its X and
        // Y slots.  (0, 0), which is on no curve with b != 0, marks the point at infinity — a multiple of a small-order
        // point outside the subgroup can be O; such entries are left out of the product.
        F pre[WINDOW_ENTRIES - 1];
#pragma nounroll
        for (int e = 1; e < WINDOW_ENTRIES; e++) {
            cur = J::add_mixed(cur, p);
            wtab[(uint64_t)e * nslots + slot] = cur;
            pre[e - 1] = zc;
            if (!cur.Z.is_zero()) zc = F::mul(zc, cur.Z);
        }
        // Fq2 (BLS12-377 G2): the common denominator is made REAL — Zc * conj(Zc), in Fq — so that psi, which conjugates Z,
        // maps table entries to points with the same denominator (the GLS path applies psi^j to the entries on the fly)
        F suf = F::one();
        if constexpr (F::DEG == 2) suf = F::conj(zc);
#pragma nounroll
        for (int e = WINDOW_ENTRIES - 1; e >= 1; e--) {
            J q = wtab[(uint64_t)e * nslots + slot];
            if (q.Z.is_zero()) {
                q.X = F::zero();
                q.Y = F::zero();
            } else {
                const F c = F::mul(pre[e - 1], suf);
                suf = F::mul(suf, q.Z);
                const F c2 = F::sqr(c);
                q.X = F::mul(q.X, c2);
                q.Y = F::mul(q.Y, F::mul(c2, c));
            }
            put_entry(e, q.X, q.Y);
        }
        {   // entry 0 = P itself (Z = 1): c = Zc
            if constexpr (F::DEG == 2) zc = F::mul(zc, F::conj(zc));
            const F c2 = F::sqr(zc);
            put_entry(0, F::mul(p.x, c2), F::mul(p.y, F::mul(c2, zc)));
        }
    } else {
        for (int e = 1; e < WINDOW_ENTRIES; e++) {
            cur = J::add_mixed(cur, p);
            wtab[(uint64_t)e * nslots + slot] = cur;
        }
    }
    // acc += (+-) table entry `mag` (with the GLV endomorphism applied when `endo`): a mixed addition on an affine table
    auto add_entry = [&](AccT &acc, int mag, bool negate, bool endo, const auto &beta) {
        if constexpr (G::LAZY_LOOP) {
            // the entry: x (times beta under the endomorphism: a literal of the carry-free field) and y (negated for a negative
            // digit), read ONCE at the top of the addition (add_src) and re-read only on its rare exceptional join
            using K = typename L::K;
            using KM = typename LoopAcc<G, true>::zc_type;
            struct Entry {
                const uint32_t *w;
                bool endo, negate;
                PTAU_HD static KM coord(const uint32_t *q) {
                    KM v;
#pragma unroll
                    for (int j = 0; j < L::N; j++) {
                        if constexpr (L::PAIR) v.c.l[j] = q[j];
                        else v.l[j] = q[j];
                    }
                    return v;
                }
                PTAU_HD typename L::AX xform(KM v) const {
                    if constexpr (!L::PAIR && J::CfgT::GLV_KIND != 0) {
                        if (endo) {
                            typename L::LZ::template E<1, 1> b;
#pragma unroll
                            for (int j = 0; j < L::N; j++) {
                                if constexpr (L::SPLIT) b.l[j] = L::LZ::mine(J::CfgT::GLV_BETAZ, j);
                                else b.l[j] = J::CfgT::GLV_BETAZ[j];
                            }
                            v = K::mul(v, b);
                        }
                    }
                    return v;
                }
                PTAU_HD typename L::AY yform(const KM &v) const {
                    typename L::AY r = v;
                    if (negate) r = K::neg(v);
                    return r;
                }
                KM x0, y0;   // as loaded at the top
                PTAU_HD typename L::AX x() const { return xform(x0); }
                PTAU_HD typename L::AY y() const { return yform(y0); }
                PTAU_HD static KM coord_again(const uint32_t *q0) { return coord(reread_ptr(q0)); }   // not merged with the read at the top
                PTAU_HD typename L::AX x_again() const { return xform(coord_again(w)); }
                PTAU_HD typename L::AY y_again() const { return yform(coord_again(w + L::N)); }
            };
            const uint32_t *w = reinterpret_cast<const uint32_t *>(&wtab[(uint64_t)(mag - 1) * nslots + slot]);
            const Entry ent{w, endo, negate, Entry::coord(w), Entry::coord(w + L::N)};
            if (L::all_lanes(K::limbs_all_zero_local(ent.x0) && K::limbs_all_zero_local(ent.y0))) return;   // (0, 0) marks a multiple that is O
            (void)beta;
            if constexpr (L::PAIR) {
                // BLS12-377 G2, plain windows (BatchExpMode::Direct, the fallback of Auto): the entry stays in registers for the
                // rare join as well.  With the re-reading form ROCm 7.2's clang MISCOMPILES this kernel: the register allocator's
                // copies of x0 land at the top of the join block of `if (negate)` in yform, AHEAD of that block's
                // `s_or_b64 exec, exec, s[0:1]`, so they run under the branch's mask - for a positive digit under EXEC = 0, i.e.
                // not at all, and every addition after the first (whose result comes from the join) used stale registers
                // (tests/test_gpu_verify.py::test_direct_windows_on_structured_scalars).  Not a source error, and
                // any other change of register pressure hides it, so the BUILD is gated instead: the Makefile disassembles
                // every object and refuses that pattern (tools/isa_exec_prologue_check.py).  This form does not produce it.
                const typename L::AX ex = ent.x();
                const typename L::AY ey = ent.y();
                acc = AccT::add_mixed(acc, ex, ey);
            } else {
                acc = AccT::add_src(acc, ent);
            }
        } else if constexpr (G::AFFINE_TABLE) {
            const J *t = &wtab[(uint64_t)(mag - 1) * nslots + slot];
            A q;
            q.x = t->X;
            q.y = t->Y;
            q.inf = q.x.is_zero() && q.y.is_zero();
            if (endo) q.x = F::mul(q.x, beta);
            if (negate) q.y = F::neg(q.y);
            acc = J::add_mixed(acc, q);
        } else {
            J ent = wtab[(uint64_t)(mag - 1) * nslots + slot];
            if (endo) ent.X = F::mul(ent.X, beta);
            if (negate) ent.Y = F::neg(ent.Y);
            acc = J::add(acc, ent);
        }
    };
    // leaving the loop: back from the isomorphic curve (Z Zc) and, for the carry-free accumulator, back to the reduced field
    auto finish = [&](const AccT &acc) -> J {
        if constexpr (G::LAZY_LOOP) {
            const auto z = L::K::mul(acc.Zc, zcz);
            return {L::from_lazy(acc.X), L::from_lazy(acc.Y), L::from_lazy(z)};
        } else {
            J r = acc;
            if constexpr (G::AFFINE_TABLE) r.Z = F::mul(r.Z, zc);
            return r;
        }
    };
    auto dbl_acc = [&](AccT &acc) { acc = AccT::dbl(acc); };
    const auto no_beta = [&] {
        if constexpr (G::LAZY_LOOP) return L::K::zero();
        else return F::zero();
    }();
    uint32_t k[NL];
    if (src.mode != SCALAR_SUBGROUP_ENDO) {
        thread_scalar<Fr>(src, i, k);
    } else {
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = 0;
    }
    using Cfg = typename J::CfgT;
    if constexpr (Cfg::GLV_KIND != 0) {
        if (use_glv) {
            // k = k1 + k2*lambda, k*P = k1*P + k2*phi(P), phi(X, Y, Z) = (beta X, Y, Z): two half-length scalars over
            // ONE table, interleaved signed 4-bit windows.  HL = limbs of a half (128 bits BLS12-377, 192 bits BW6-761).
            constexpr int HL = Cfg::GLV_KIND == 1 ? 4 : 6;
            uint32_t kk[2][HL];
            bool ng[2] = {false, false};
            if constexpr (Cfg::GLV_KIND == 1) {
                glv_decompose<Cfg>(k, kk[0], kk[1]);
            } else {
                if (src.mode == SCALAR_SUBGROUP_ENDO) {   // membership test: fixed (a, b) with a + b lambda = 0 mod r, norm r
#pragma unroll
                    for (int j = 0; j < HL; j++) { kk[0][j] = Cfg::SUB_A[j]; kk[1][j] = Cfg::SUB_B[j]; }
                    ng[0] = Cfg::SUB_NEG_A;
                    ng[1] = Cfg::SUB_NEG_B;
                } else {
                    glv2_decompose<Cfg>(k, kk[0], kk[1], ng[0], ng[1]);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t c2 = 0;
#pragma unroll
                for (int j = 0; j < HL; j++) kk[h][j] = addc(kk[h][j], 0x88888888u, c2);   // halves < 2^(32 HL - 1): no carry out
            }
            const auto beta = [&] {   // (the carry-free loop takes beta as a literal where it is used: Entry::x)
                if constexpr (G::LAZY_LOOP) return 0;
                else return F::load_const(Cfg::GLV_BETA);
            }();
            AccT acc = AccT::infinity();
            for (int w = HL * 8 - 1; w >= 0; w--) {
                if (!acc.is_inf()) {
#pragma nounroll
                    for (int d = 0; d < WINDOW_BITS; d++) dbl_acc(acc);
                }
#pragma nounroll
                for (int h = 0; h < 2; h++) {
                    uint32_t limb = 0;
#pragma unroll
                    for (int j = 0; j < HL; j++) limb = (j == (w >> 3)) ? (h ? kk[1][j] : kk[0][j]) : limb;
                    int dgt = (int)((limb >> ((w & 7) * 4)) & 15u) - 8;
                    if (dgt != 0) {
                        int mag = dgt < 0 ? -dgt : dgt;
                        add_entry(acc, mag, (dgt < 0) != (h ? ng[1] : ng[0]), h != 0, beta);
                    }
                }
            }
            store_jac<G>(scratch, i, finish(acc));
            return 0;
        }
    }
    if constexpr (Cfg::ENDO_KIND == 2 && NL == 8) {
        if (use_glv) {
            // k P = sum_j k_j psi^j(P): four 64-bit sub-scalars, interleaved signed 4-bit windows over ONE table of
            // multiples of P; psi^j is applied to the table entry on the fly (two Fq2-by-Fq products + conjugations).
            using B = typename F::Base;
            uint64_t dj[4];
            gls4_decompose<Cfg>(k, dj);
            uint64_t lo[4];
            uint32_t hi[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {   // + 0x8_8888_8888_8888_8888: 17 nibbles, every digit = nibble - 8
                lo[h] = dj[h] + 0x8888888888888888ull;
                hi[h] = 8u + (lo[h] < dj[h] ? 1u : 0u);
            }
            AccT acc = AccT::infinity();
            for (int w = 16; w >= 0; w--) {
                if (!acc.is_inf()) {
#pragma nounroll
                    for (int d = 0; d < WINDOW_BITS; d++) dbl_acc(acc);
                }
#pragma nounroll
                for (int h = 0; h < 4; h++) {
                    uint64_t l64 = 0;
                    uint32_t h32 = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) { l64 = (j == h) ? lo[j] : l64; h32 = (j == h) ? hi[j] : h32; }
                    const uint32_t nib = w == 16 ? h32 : (uint32_t)(l64 >> (4 * w)) & 15u;
                    const int dgt = (int)nib - 8;
                    if (dgt != 0) {
                        const int mag = dgt < 0 ? -dgt : dgt;
                        if constexpr (G::LAZY_LOOP) {
                            // psi^h of the table entry: conjugate for odd h, then one Fq constant per coordinate; both coordinates are
                            // read at the top (one memory latency), transformed where LazyJac2::add_src takes them and re-read only on
                            // its rare exceptional join
                            using K = typename L::K;
                            using KM = typename LoopAcc<G, true>::zc_type;
                            struct GlsEntry {
                                const uint32_t *tw;
                                int h;
                                bool negate;
                                KM x0, y0;
                                PTAU_HD static KM load(const uint32_t *q) {
                                    KM v;
#pragma unroll
                                    for (int j = 0; j < L::N; j++) v.c.l[j] = q[j];
                                    return v;
                                }
                                PTAU_HD KM psi(KM v, int c) const {
                                    if (h) {
                                        typename L::LZ::template E<1, 1> k;
#pragma unroll
                                        for (int j = 0; j < L::N; j++) k.l[j] = Cfg::GLS_CZ[2 * (h - 1) + c][j];
                                        decltype(K::conj(v)) vc = v;
                                        if (h & 1) vc = K::conj(v);
                                        v = K::mul_base(vc, k);
                                    }
                                    return v;
                                }
                                PTAU_HD typename L::AY ysign(const KM &v) const {
                                    typename L::AY r = v;
                                    if (negate) r = K::neg(v);
                                    return r;
                                }
                                PTAU_HD typename L::AX x() const { return psi(x0, 0); }
                                PTAU_HD typename L::AY y() const { return ysign(psi(y0, 1)); }
                                PTAU_HD static KM load_again(const uint32_t *q0) { return load(reread_ptr(q0)); }   // not merged with the read at the top
                                PTAU_HD typename L::AX x_again() const { return psi(load_again(tw), 0); }
                                PTAU_HD typename L::AY y_again() const { return ysign(psi(load_again(tw + L::N), 1)); }
                            };
                            const uint32_t *tw = reinterpret_cast<const uint32_t *>(&wtab[(uint64_t)(mag - 1) * nslots + slot]);
                            const GlsEntry ent{tw, h, dgt < 0, GlsEntry::load(tw), GlsEntry::load(tw + L::N)};
                            // (0, 0) marks a multiple that is O
                            if (!L::all_lanes(K::limbs_all_zero_local(ent.x0) && K::limbs_all_zero_local(ent.y0))) acc = AccT::add_src(acc, ent);
                        } else {
                        J ent = wtab[(uint64_t)(mag - 1) * nslots + slot];   // affine table: Z slot unused (real common Z)
                        if (h) {
                            B cx, cy;
#pragma unroll
                            for (int j = 0; j < B::N; j++) { cx.l[j] = Cfg::GLS_C[2 * (h - 1)][j]; cy.l[j] = Cfg::GLS_C[2 * (h - 1) + 1][j]; }
                            if (h & 1) {   // odd powers of psi conjugate
                                ent.X = F::conj(ent.X);
                                ent.Y = F::conj(ent.Y);
                                if constexpr (!G::AFFINE_TABLE) ent.Z = F::conj(ent.Z);
                            }
                            ent.X = F::mul_base(ent.X, cx);
                            ent.Y = F::mul_base(ent.Y, cy);
                        }
                        if (dgt < 0) ent.Y = F::neg(ent.Y);
                        if constexpr (G::AFFINE_TABLE) {
                            A q;
                            q.x = ent.X;
                            q.y = ent.Y;
                            q.inf = q.x.is_zero() && q.y.is_zero();
                            acc = J::add_mixed(acc, q);
                        } else {
                            acc = J::add(acc, ent);
                        }
                        }
                    }
                }
            }
            store_jac<G>(scratch, i, finish(acc));
            return 0;
        }
    }
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) k[j] = addc(k[j], 0x88888888u, c);
    AccT acc = AccT::infinity();
    for (int w = NW - 1; w >= 0; w--) {
        if (!acc.is_inf()) {
#pragma nounroll
            for (int d = 0; d < WINDOW_BITS; d++) dbl_acc(acc);
        }
        uint32_t limb = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) limb = (j == (w >> 3)) ? k[j] : limb;   // register select, no scratch
        int dgt = (int)((limb >> ((w & 7) * 4)) & 15u) - 8;
        if (dgt != 0) {
            int mag = dgt < 0 ? -dgt : dgt;
            add_entry(acc, mag, dgt < 0, false, no_beta);
        }
    }
    store_jac<G>(scratch, i, finish(acc));
    return 0;
}

// CIN is a template parameter: the square root behind a compressed read is a register-hungry callee and a kernel's
// VGPR allocation is the maximum over its call graph — the uncompressed instance (the tau-update of a challenge)
// does not carry it.
template <class G, int CIN>
__global__ void __launch_bounds__(256, G::K1_MIN_WAVES ? G::K1_MIN_WAVES : G::MIN_WAVES)
window_scalar_mul_kernel(const uint8_t *__restrict__ in, typename G::J *__restrict__ scratch,
                         typename G::J *__restrict__ wtab, uint64_t n, ScalarSource src, int check_input,
                         int use_glv, int *__restrict__ err) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G::LPP;
    if (i >= n) return;
    int bad = window_scalar_mul_thread<G>(in, CIN, scratch, wtab, n, i, src, check_input, use_glv);
    if (bad) atomicOr(err, bad);
}

// Sum of Jacobian points: thread t adds elements t, t+T, t+2T, ... (< n) into out[t]  (one tree level)
template <class G>
PTAU_HD void reduce_sum_thread(const typename G::J *in, typename G::J *out, uint64_t n, uint64_t T, uint64_t t) {
    using J = typename G::J;
    J acc = J::infinity();
    for (uint64_t i = t; i < n; i += T) acc = J::add(acc, in[i]);
    out[t] = acc;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
reduce_sum_kernel(const typename G::J *__restrict__ in, typename G::J *__restrict__ out, uint64_t n, uint64_t T) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    reduce_sum_thread<G>(in, out, n, T, t);
}

// ---- K4: multi-scalar multiplication, bucket method ------------------------------------------------------
// Replaces setup-utils' dense_multiexp / merge_pairs / power_pairs (bellman-style Pippenger on the CPU;
// reference call sites /root/reference/src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591).
//   msm_prepare    : thread i converts point i to Montgomery affine ONCE, cuts its scalar into W signed c-bit
//                    digits d in [-2^(c-1), 2^(c-1)] and counts the bucket (w, |d|) it will land in;
//   exclusive scan : bucket counts -> offsets (capi.hip, one workgroup);
//   msm_scatter    : counting sort of the point indices by (window, bucket);
//   msm_accumulate / msm_combine : bucket sums by mixed additions (P_i, or P_{i+1} for the shifted sum of
//                    power_pairs — both sums share digits, sort and the affine array), a bucket cut into runs
//                    of <= MSM_RUN entries so that a crowded bucket is still summed in parallel;
//   msm_bucket_reduce : per window sum_b b * bucket[b] by running sums over segments of SEG buckets;
//   reduce_rows, msm_horner : segment partials -> window sums -> sum_w 2^(c w) R_w.
// Every sum is over group elements, so the affine result is canonical whatever the addition order.
struct MsmPlan {
    int c;            // window bits
    int W;            // windows: W * c >= scalar bits + 1 (the signed recoding carries one bit up)
    uint32_t B;       // buckets per window = 2^(c-1)
    uint32_t seg;     // buckets per bucket-reduce thread
};

// Window width for m points and `scalar_bits`-bit scalars (host side).  Cost model: W windows, each m mixed additions
// plus ~3 addition-equivalents per bucket for the running-sum reduction; a plan whose TOP window holds fewer than
// c - 2 scalar bits is rejected when another one exists (its few buckets would each collect m / 2^t points: the run
// splitting keeps that parallel, but the combine pass then walks long run lists).  127-bit power_pairs coefficients
// at 2^20 points give c = 16: 8 full windows, 32 points per bucket.
// bucket_test_cost > 0 selects the plan of the MEMBERSHIP-FUSED MSM of the verify path (capi.hip, run_verify_vector): every
// bucket sum of every window is tested for prime-order-subgroup membership, which makes each window one round of the batched
// subgroup check of setup-utils (SubgroupCheckMode::Batched, SECURITY_PARAM = 128): a point outside the subgroup lands, in
// each window independently, in a uniformly random bucket (or in none, digit 0) and that window's test passes only if its image
// in the cofactor group is cancelled by what the other points left in exactly that bucket: probability <= 2^-(c-1) per
// window: the signed digit is uniform over the 2^c residues whatever the carry from below, and at most two of them (+m, -m
// of the one bucket holding the cancelling image; or the single residue 0 when every bucket is clean without the point) let
// the window pass.  The top window holds one random bit less (2^-(c-2)).  The plan therefore takes W = ceil(129 / (c - 1))
// windows of scalars with W c - 1 random bits (`bits`) — false-accept probability <= 2^-(W (c-1) - 1) <= 2^-128 — which are
// also valid (longer) coefficients of the random linear combination, and weighs a bucket with the cost of its test
// (bucket_test_cost, in mixed additions).  Both sums of power_pairs are tested: A holds points 0 .. n-2, B points 1 .. n-1,
// each placed by an independent rho.
constexpr uint32_t MSM_RUN = 64, MSM_RUN_MAX = 128;   // entries per accumulate thread: the host picks the length of a launch in this range (capi.hip)
struct MsmPlanEx : MsmPlan { int bits; };
// resident_runs (fused plans): accumulate threads resident on the device at once.  A bucket's partial sums are added up by
// ONE thread of the combine pass (a full addition each, ~3.6 mixed additions of time), so few large buckets mean a long
// latency-bound combine: that latency, times the lanes it leaves idle, joins the cost.
inline MsmPlanEx msm_plan(uint64_t m, int scalar_bits, int force_c = 0, int bucket_test_cost = 0, uint64_t resident_runs = 0) {
    MsmPlanEx best{};
    double best_cost = -1;
    bool best_balanced = false;
    for (int c = 2; c <= 16; c++) {
        if (force_c && c != force_c) continue;
        int W = (scalar_bits + 1 + c - 1) / c, bits = scalar_bits;
        bool balanced;
        if (bucket_test_cost > 0) {
            if (c < 3) continue;
            W = (129 + c - 2) / (c - 1);
            bits = W * c - 1;
            if (bits > PRF_MAX_BITS) continue;
            balanced = true;
        } else {
            const int top_bits = scalar_bits + 1 - (W - 1) * c;
            balanced = top_bits >= c - 2;
        }
        double cost = (double)W * ((double)m + (3.0 + bucket_test_cost) * (double)(1u << (c - 1)));
        if (bucket_test_cost > 0 && resident_runs) {
            const double total = (double)W * (double)m, R = (double)resident_runs;
            const double k = total > R * MSM_RUN_MAX ? (double)(uint64_t)((total + R * MSM_RUN_MAX - 1) / (R * MSM_RUN_MAX)) : 1.0;
            double L = total / (k * R);
            L = L < MSM_RUN ? MSM_RUN : (L > MSM_RUN_MAX ? MSM_RUN_MAX : L);
            const double parts = (double)m / (double)(1u << (c - 1)) / L + 1.0;
            cost += parts * 3.6 * R;
        }
        if (best_cost < 0 || (balanced && !best_balanced) || (balanced == best_balanced && cost < best_cost)) {
            best.c = c;
            best.W = W;
            best.B = 1u << (c - 1);
            best.seg = best.B < 8 ? best.B : 8;
            best.bits = bits;
            best_cost = cost;
            best_balanced = balanced;
        }
    }
    return best;
}

PTAU_HD uint32_t atomic_inc_u32(uint32_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1u);
#else
    return (*p)++;
#endif
}

// The MSM's affine array of a one-lane prime-field group with a carry-free twin (BLS12-377 G1) is kept IN that field's
// Montgomery form — x | y as 14 limbs of 28 bits, then the flag word — so that the bucket accumulation runs the carry-free
// mixed addition without converting every point it meets (msm_prepare converts once per point).
template <class G>
constexpr bool msm_aff_lazy() {
    if constexpr (G::LAZY_LOOP) return G::F::DEG == 1 && G::LANE_MODE == LANES_ONE;
    else return false;
}
template <class G>
constexpr size_t msm_aff_bytes() {
    if constexpr (msm_aff_lazy<G>()) return (2 * LazyOf<G>::N + 1) * sizeof(uint32_t);
    else return sizeof(typename G::A);
}

// digit entry: 0 = nothing to add; else bit 31 = negative, low bits = magnitude in [1, 2^(c-1)]
template <class G>
PTAU_HD int msm_prepare_thread(const uint8_t *in, int compressed_in, typename G::A *aff, uint32_t *dig,
                               uint32_t *counts, uint64_t n_points, uint64_t n_scalars, uint64_t i,
                               const ScalarSource &src, const MsmPlan &pl) {
    using Fr = typename G::Fr;
    using F = typename G::F;
    constexpr int NL = Fr::N;
    typename G::A p;
    const bool on_curve_too = compressed_in == 2;   // uncompressed input that nobody has looked at yet (the probe of Auto)
    int bad = read_point_any<G>(in, on_curve_too ? 0 : compressed_in, i, p);
    if (!bad && on_curve_too && !p.inf && !(F::sqr(p.y) == curve_rhs<F, typename G::J::CfgT>(p.x))) bad |= ERRBIT_NOT_ON_CURVE;
    if (bad) p.inf = true;
    if (p.inf) { p.x = F::zero(); p.y = F::zero(); }
    if constexpr (msm_aff_lazy<G>()) {
        using L = LazyOf<G>;
        uint32_t *w = reinterpret_cast<uint32_t *>(aff) + i * (2 * L::N + 1);
        L::store(w, L::to_lazy(p.x));
        L::store(w + L::N, L::to_lazy(p.y));
        w[2 * L::N] = p.inf ? 1u : 0u;
    } else {
        aff[i] = p;
    }
    if (i >= n_scalars) return bad;
    uint32_t k[NL];
    thread_scalar<Fr>(src, i, k);
    const uint32_t mask = (1u << pl.c) - 1u;
    uint32_t carry = 0;
    for (int w = 0; w < pl.W; w++) {
        const int bit = w * pl.c, li = bit >> 5, sh = bit & 31;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) {   // register select, no scratch
            lo = (j == li) ? k[j] : lo;
            hi = (j == li + 1) ? k[j] : hi;
        }
        uint32_t v = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & mask;
        v += carry;
        uint32_t e = 0;
        if (v > pl.B) {            // negative digit v - 2^c, carry one into the next window
            v = (1u << pl.c) - v;
            carry = 1;
            if (v) e = v | 0x80000000u;
        } else {
            carry = 0;
            if (v) e = v;
        }
        dig[(uint64_t)w * n_scalars + i] = e;
        if (e) atomic_inc_u32(&counts[(uint64_t)w * pl.B + (e & 0x7fffffffu) - 1]);
    }
    return bad;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
msm_prepare_kernel(const uint8_t *__restrict__ in, int compressed_in, typename G::A *__restrict__ aff,
                   uint32_t *__restrict__ dig, uint32_t *__restrict__ counts, uint64_t n_points, uint64_t n_scalars,
                   ScalarSource src, MsmPlan pl, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    int bad = msm_prepare_thread<G>(in, compressed_in, aff, dig, counts, n_points, n_scalars, i, src, pl);
    if (bad) atomicOr(err, bad);
}

PTAU_HD void msm_scatter_thread(const uint32_t *dig, uint32_t *cursor, uint32_t *sorted, uint64_t n, uint64_t i,
                                const MsmPlan &pl) {
    for (int w = 0; w < pl.W; w++) {
        const uint32_t e = dig[(uint64_t)w * n + i];
        if (!e) continue;
        const uint32_t pos = atomic_inc_u32(&cursor[(uint64_t)w * pl.B + (e & 0x7fffffffu) - 1]);
        sorted[pos] = (uint32_t)i | (e & 0x80000000u);
    }
}

template <int DUMMY>
__global__ void __launch_bounds__(256)
msm_scatter_kernel(const uint32_t *__restrict__ dig, uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted,
                   uint64_t n, MsmPlan pl) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    msm_scatter_thread(dig, cursor, sorted, n, i, pl);
}

// Bucket sums in two steps so that one crowded bucket (a skewed scalar distribution, a short top window) cannot
// serialise the pass and every thread has the same amount of work: the sorted entries are cut into runs of run_len
// consecutive entries, whatever buckets they belong to;
//   msm_accumulate : one thread per RUN — finds the bucket of its first entry by bisection of the offsets, adds +-P over its
//                    entries, writing a partial sum whenever a bucket ends inside the run
//                    (shift = 1 adds P_{i+1} instead of P_i: the shifted sum of power_pairs);
//   msm_combine    : one thread per bucket — sums the bucket's partials (three or four at 2^20 points).

// Affine point `idx` of an array written in the STANDARD (single-lane) layout — Affine<Fp2> = x.c0 x.c1 y.c0 y.c1 inf —
// as the group's own Affine (lane-pair groups pick their coefficient).
template <class G>
PTAU_HD typename G::A load_affine_std(const typename G::A *aff, uint64_t idx) {
    if constexpr (G::LANE_MODE == LANES_LIMB_SPLIT) {
        // standard layout: x | y as DEG x N limbs each, then the flag word; this lane takes its H limbs of every coefficient
        using L = typename G::Fq;
        using F = typename G::F;
        constexpr int D = F::DEG;
        constexpr size_t WORDS = 2 * D * L::N + 1;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(aff) + idx * WORDS;
        const uint32_t *mine = w + (lane_parity() ? L::H : 0);
        auto get = [&](int slot) {
            L c;
#pragma unroll
            for (int j = 0; j < L::H; j++) c.l[j] = mine[slot * L::N + j];
            return c;
        };
        typename G::A q;
        if constexpr (D == 1) { q.x = get(0); q.y = get(1); }
        else if constexpr (D == 2) { q.x = F::make(get(0), get(1)); q.y = F::make(get(2), get(3)); }
        else { q.x = F::make(get(0), get(1), get(2)); q.y = F::make(get(3), get(4), get(5)); }
        q.inf = *reinterpret_cast<const bool *>(w + 2 * D * L::N);
        return q;
    } else if constexpr (G::LPP == 2) {
        using Fq = typename G::Fq;
        constexpr size_t STRIDE = 4 * sizeof(Fq) + sizeof(uint32_t);   // sizeof(Affine<Fp2<Fq>>): 4 coefficients + flag word
        static_assert(sizeof(Affine<Fp2<Fq, 1>>) == STRIDE, "standard affine layout changed");
        const uint8_t *base = reinterpret_cast<const uint8_t *>(aff) + idx * STRIDE;
        const Fq *c = reinterpret_cast<const Fq *>(base);
        typename G::A q;
        q.x.c = c[lane_parity()];
        q.y.c = c[2 + lane_parity()];
        q.inf = *reinterpret_cast<const bool *>(base + 4 * sizeof(Fq));
        return q;
    } else {
        return aff[idx];
    }
}

// Partial-sum slots of bucket b under runs of run_len sorted entries ALIGNED TO THE ENTRY INDEX (run r = entries
// [r run_len, (r + 1) run_len) whatever bucket they fall in): the runs floor(o_b / L) .. floor((o_{b+1} - 1) / L) touch it.
PTAU_HD uint32_t msm_bucket_parts(uint32_t off, uint32_t cnt, uint32_t run_len) {
    return cnt ? (off + cnt - 1) / run_len - off / run_len + 1 : 0;
}

// One run: run_len consecutive sorted entries, cut at bucket boundaries — every thread does the same number of additions
// (the host sizes run_len so that the runs of a launch fill the resident lanes a whole number of times), and a bucket's
// partial sums land in the contiguous slots run_offsets[b] ... (msm_combine adds them up).
template <class G>
PTAU_HD void msm_accumulate_thread(const typename G::A *aff, const uint32_t *sorted, const uint32_t *offsets,
                                   const uint32_t *run_offsets, typename G::J *partials, uint32_t shift,
                                   uint32_t total_buckets, uint32_t v, uint32_t run_len = MSM_RUN) {
    using J = typename G::J;
    using F = typename G::F;
    const uint32_t total = offsets[total_buckets];
    const uint64_t e0w = (uint64_t)v * run_len;
    if (e0w >= total) return;
    const uint32_t e0 = (uint32_t)e0w, e1 = e0w + run_len < total ? e0 + run_len : total;
    uint32_t lo = 0, hi = total_buckets;   // largest b with offsets[b] <= e0: the (non-empty) bucket holding entry e0
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= e0) lo = mid; else hi = mid;
    }
    uint32_t b = lo, bend = offsets[b + 1];
    auto slot = [&](uint32_t bb) { return run_offsets[bb] + (v - offsets[bb] / run_len); };
    if constexpr (G::LAZY_LOOP && F::DEG == 2) {
        // Fq2 over a lane pair: the run is summed in the carry-free field (fpz.hpp), every point converted on the way in (one
        // base product per coordinate and lane against the ~40 % the additions save; for the prime-field groups the two
        // converting products would cancel the gain, they keep the reduced field here)
        using L = LazyOf<G>;
        using K = typename L::K;
        using Acc = typename L::Acc;
        Acc acc = Acc::infinity();
        for (uint32_t e = e0; e < e1; e++) {
            if (e == bend) {   // the next bucket starts here
                store_jac<G>(partials, slot(b), J{L::from_lazy(acc.X), L::from_lazy(acc.Y), L::from_lazy(acc.Zc)});
                acc = Acc::infinity();
                do { b++; bend = offsets[b + 1]; } while (bend <= e);
            }
            const uint32_t s = sorted[e];
            const typename G::A q = load_affine_std<G>(aff, (uint64_t)(s & 0x7fffffffu) + shift);
            if (q.inf) continue;
            const auto x = L::to_lazy(q.x), y = L::to_lazy(q.y);
            typename L::AY ys = y;
            if (s >> 31) ys = K::neg(y);
            acc = Acc::add_mixed(acc, x, ys);
        }
        store_jac<G>(partials, slot(b), J{L::from_lazy(acc.X), L::from_lazy(acc.Y), L::from_lazy(acc.Zc)});
        return;
    }
    if constexpr (msm_aff_lazy<G>()) {
        // one-lane prime field with a carry-free twin: the array already holds that field's form (msm_aff_lazy)
        using L = LazyOf<G>;
        using K = typename L::K;
        using Acc = typename L::Acc;
        Acc acc = Acc::infinity();
        for (uint32_t e = e0; e < e1; e++) {
            if (e == bend) {
                store_jac<G>(partials, slot(b), J{L::from_lazy(acc.X), L::from_lazy(acc.Y), L::from_lazy(acc.Zc)});
                acc = Acc::infinity();
                do { b++; bend = offsets[b + 1]; } while (bend <= e);
            }
            const uint32_t s = sorted[e];
            const uint32_t *w = reinterpret_cast<const uint32_t *>(aff) + ((uint64_t)(s & 0x7fffffffu) + shift) * (2 * L::N + 1);
            if (w[2 * L::N]) continue;
            const auto x = L::template load<typename L::C>(w), y = L::template load<typename L::C>(w + L::N);
            typename L::AY ys = y;
            if (s >> 31) ys = K::neg(y);
            acc = Acc::add_mixed(acc, x, ys);
        }
        store_jac<G>(partials, slot(b), J{L::from_lazy(acc.X), L::from_lazy(acc.Y), L::from_lazy(acc.Zc)});
        return;
    }
    J acc = J::infinity();
    for (uint32_t e = e0; e < e1; e++) {
        if (e == bend) {
            store_jac<G>(partials, slot(b), acc);
            acc = J::infinity();
            do { b++; bend = offsets[b + 1]; } while (bend <= e);
        }
        const uint32_t s = sorted[e];
        typename G::A q = load_affine_std<G>(aff, (uint64_t)(s & 0x7fffffffu) + shift);
        if (s >> 31) q.y = F::neg(q.y);
        acc = J::add_mixed(acc, q);
    }
    store_jac<G>(partials, slot(b), acc);
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
msm_accumulate_kernel(const typename G::A *__restrict__ aff, const uint32_t *__restrict__ sorted,
                      const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ run_offsets,
                      typename G::J *__restrict__ partials, uint32_t shift, uint32_t total_buckets, uint32_t run_len) {
    const uint64_t v = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G::LPP;
    if (v * run_len >= offsets[total_buckets]) return;   // the grid is sized for the worst case, the entry count lives on the device
    msm_accumulate_thread<G>(aff, sorted, offsets, run_offsets, partials, shift, total_buckets, (uint32_t)v, run_len);
}

template <class G>
PTAU_HD void msm_combine_thread(const typename G::J *partials, const uint32_t *run_offsets, typename G::J *buckets, uint64_t t) {
    using J = typename G::J;
    J acc = J::infinity();
    const uint32_t v1 = run_offsets[t + 1];
    for (uint32_t v = run_offsets[t]; v < v1; v++) acc = J::add(acc, partials[v]);
    buckets[t] = acc;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
msm_combine_kernel(const typename G::J *__restrict__ partials, const uint32_t *__restrict__ run_offsets,
                   typename G::J *__restrict__ buckets, uint64_t total_buckets) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_buckets) return;
    msm_combine_thread<G>(partials, run_offsets, buckets, t);
}

// thread (w, s): sum_{b in segment s} (b + 1) * bucket[w][b]  =  sum of running sums + lo * (segment sum)
template <class G>
PTAU_HD void msm_bucket_reduce_thread(const typename G::J *buckets, typename G::J *parts, const MsmPlan &pl, uint64_t t) {
    using J = typename G::J;
    const uint32_t nseg = pl.B / pl.seg;
    const uint32_t w = (uint32_t)(t / nseg), sg = (uint32_t)(t % nseg), lo = sg * pl.seg;
    const J *bw = buckets + (uint64_t)w * pl.B;
    J run = J::infinity(), tot = J::infinity();
    for (uint32_t b = lo + pl.seg; b-- > lo;) {
        run = J::add(run, bw[b]);
        tot = J::add(tot, run);
    }
    if (lo) {   // + lo * run, MSB-first double-and-add over the (<= 16) bits of lo
        J acc = J::infinity();
        for (int bit = 31 - __builtin_clz(lo); bit >= 0; bit--) {
            acc = J::dbl(acc);
            if ((lo >> bit) & 1u) acc = J::add(acc, run);
        }
        tot = J::add(tot, acc);
    }
    parts[t] = tot;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
msm_bucket_reduce_kernel(const typename G::J *__restrict__ buckets, typename G::J *__restrict__ parts, MsmPlan pl,
                         uint64_t total) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    msm_bucket_reduce_thread<G>(buckets, parts, pl, t);
}

// rows x cols -> rows x T: thread (r, t) sums in[r][t], in[r][t + T], ...
template <class G>
PTAU_HD void reduce_rows_thread(const typename G::J *in, typename G::J *out, uint32_t cols, uint32_t T, uint64_t id) {
    using J = typename G::J;
    const uint64_t r = id / T, t = id % T;
    J acc = J::infinity();
    for (uint64_t j = t; j < cols; j += T) acc = J::add(acc, in[r * cols + j]);
    out[id] = acc;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
reduce_rows_kernel(const typename G::J *__restrict__ in, typename G::J *__restrict__ out, uint32_t cols, uint32_t T,
                   uint64_t total) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    reduce_rows_thread<G>(in, out, cols, T, id);
}

// sum_w 2^(c w) win[w]  (one thread: W*c doublings in sequence is inherent to the weighting)
template <class G>
PTAU_HD void msm_horner_thread(const typename G::J *win, typename G::J *out, const MsmPlan &pl) {
    using J = typename G::J;
    J acc = J::infinity();
    for (int w = pl.W - 1; w >= 0; w--) {
        if (!acc.is_inf())
            for (int d = 0; d < pl.c; d++) acc = J::dbl(acc);
        acc = J::add(acc, win[w]);
    }
    *out = acc;
}
// one block per sum (the plain and the shifted sum of power_pairs run side by side): block b reads win + b*W, writes out + b*out_stride
template <class G>
__global__ void __launch_bounds__(64)
msm_horner_kernel(const typename G::J *__restrict__ win, typename G::J *__restrict__ out, MsmPlan pl, uint64_t out_stride) {
    if (threadIdx.x == 0) msm_horner_thread<G>(win + (uint64_t)blockIdx.x * pl.W, out + (uint64_t)blockIdx.x * out_stride, pl);
}

// y^2 == x^3 + a x + b for uncompressed inputs (compressed reads are on the curve by construction)
template <class G>
PTAU_HD int on_curve_thread(const uint8_t *in, uint64_t i) {
    using F = typename G::F;
    typename G::A p;
    if (!read_uncompressed<F>(in + i * G::UNCOMP, p)) return ERRBIT_NONCANONICAL;
    if (p.inf) return 0;
    return F::sqr(p.y) == curve_rhs<F, typename G::J::CfgT>(p.x) ? 0 : ERRBIT_NOT_ON_CURVE;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
on_curve_kernel(const uint8_t *__restrict__ in, uint64_t n, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = on_curve_thread<G>(in, i);
    if (bad) atomicOr(err, bad);
}

// ---- K3': subgroup membership through an efficient endomorphism (BLS12-377) -------------------------------
// Same accept/reject set as [r]P == O (setup-utils check_subgroup, SubgroupCheckMode::Direct), at 1/2 (G1) and
// 1/4 (G2) of the doublings — eprint 2021/1130 (the test ark-bls12-377 0.4's is_in_correct_subgroup_assuming_on_curve
// applies):   G1: (C0 x, y) == -[X^2] P        G2: psi(P) = (C0 conj(x), C1 conj(y)) == [X] P.
// X = 0x8508c00000000001 has 7 set bits: plain MSB-first double-and-add, wave-uniform control flow, no table.
// The endomorphism membership test in the carry-free field, on (x, y) given in the reduced field.  For G1 (ENDO_KIND 1) the pair
// may also be the X, Y of a JACOBIAN point (X, Y, Z), Z != 0, with Z simply ignored: (X, Y) is an affine point of the isomorphic
// curve y^2 = x^3 + b Z^6, the isomorphism (x, y) -> (x / Z^2, y / Z^3) is defined over Fq and commutes with the endomorphism
// (beta x, y) and with scalar multiplication, and neither the doubling nor the mixed addition involves b — the verdict is the
// one of the point itself (endo_subgroup_jac_kernel: the bucket sums of the membership-fused MSM need no normalisation).
template <class G>
PTAU_HD int endo_membership_lazy(const typename G::F &x29, const typename G::F &y29) {
    using Cfg = typename G::J::CfgT;
    constexpr uint64_t X = Cfg::ENDO_X;
    const auto px = LazyOf<G>::to_lazy(x29), py = LazyOf<G>::to_lazy(y29);
    // the same test in the carry-free field (fpz.hpp).  G1's second multiplication by X starts from a Jacobian point
    // (Xb, Yb, Zb): that is the affine point (Xb, Yb) of the isomorphic curve y^2 = x^3 + b Zb^6, neither the doubling nor
    // the mixed addition involves b, so the chain runs there with mixed additions and Z maps back as Z Zb.
    using L = LazyOf<G>;
    using K = typename L::K;
    using Acc = typename L::Acc;
    const typename L::AY pya = py;
    Acc t = {px, py, K::one()};
    for (int b = 62; b >= 0; b--) {   // bit 63 of X is set
        t = Acc::dbl(t);
        if ((X >> b) & 1ull) t = Acc::add_mixed(t, px, pya);
    }
    if (t.is_inf()) return ERRBIT_NOT_IN_SUBGROUP;
    typename L::LZ::template E<1, 1> c0, c1;
#pragma unroll
    for (int j = 0; j < L::N; j++) { c0.l[j] = Cfg::ENDO_C0Z[j]; c1.l[j] = Cfg::ENDO_C1Z[j]; }
    if constexpr (Cfg::ENDO_KIND == 1) {
        const auto zb = t.Zc;
        const auto bx = K::vreduce(t.X), by = K::vreduce(t.Y);   // below 3p, as add_mixed takes them
        const typename L::AX bxa = bx;
        const typename L::AY bya = by;
        Acc u = {bx, by, K::one()};
        for (int b = 62; b >= 0; b--) {
            u = Acc::dbl(u);
            if ((X >> b) & 1ull) u = Acc::add_mixed(u, bxa, bya);
        }
        if (u.is_inf()) return ERRBIT_NOT_IN_SUBGROUP;
        const auto zf = K::mul(u.Zc, zb);
        const auto zz = K::sqr(zf);
        const auto ex = K::mul(px, c0);
        const bool okx = K::is_zero(K::norm(K::sub(u.X, K::mul(ex, zz))));
        const bool oky = K::is_zero(K::norm(K::add(u.Y, K::mul(py, K::mul(zz, zf)))));   // Y == -y Z^3
        return okx && oky ? 0 : ERRBIT_NOT_IN_SUBGROUP;
    } else {
        const auto ex = K::mul_base(K::conj(px), c0), ey = K::mul_base(K::conj(py), c1);
        const auto zz = K::sqr(t.Zc);
        const bool okx = K::is_zero(K::norm(K::sub(t.X, K::mul(ex, zz))));
        const bool oky = K::is_zero(K::norm(K::sub(t.Y, K::mul(ey, K::mul(zz, t.Zc)))));
        return okx && oky ? 0 : ERRBIT_NOT_IN_SUBGROUP;
    }
}

template <class G>
PTAU_HD int endo_subgroup_thread(const uint8_t *in, int compressed_in, uint64_t i) {
    using F = typename G::F;
    using B = typename F::Base;
    using A = typename G::A;
    using J = typename G::J;
    using Cfg = typename J::CfgT;
    static_assert(Cfg::ENDO_KIND == 1 || Cfg::ENDO_KIND == 2, "no endomorphism test for this group");
    A p;
    int bad = read_point_any<G>(in, compressed_in, i, p);
    if (bad) return bad;
    if (p.inf) return 0;
    if (!compressed_in && !(F::sqr(p.y) == curve_rhs<F, Cfg>(p.x))) return ERRBIT_NOT_ON_CURVE;
    constexpr uint64_t X = Cfg::ENDO_X;
    if constexpr (G::LAZY_LOOP) return endo_membership_lazy<G>(p.x, p.y);
    J t = J::from_affine(p);
    for (int b = 62; b >= 0; b--) {   // bit 63 of X is set
        t = J::dbl(t);
        if ((X >> b) & 1ull) t = J::add_mixed(t, p);
    }
    B c0, c1;
#pragma unroll
    for (int j = 0; j < B::N; j++) { c0.l[j] = Cfg::ENDO_C0[j]; c1.l[j] = Cfg::ENDO_C1[j]; }
    F ex, ey;   // the endomorphism image (affine) the Jacobian multiple must equal
    if constexpr (Cfg::ENDO_KIND == 1) {
        const J base = t;
        for (int b = 62; b >= 0; b--) {
            t = J::dbl(t);
            if ((X >> b) & 1ull) t = J::add(t, base);
        }
        ex = F::mul(p.x, c0);
        ey = F::neg(p.y);
    } else {
        ex = F::mul_base(F::conj(p.x), c0);
        ey = F::mul_base(F::conj(p.y), c1);
    }
    if (t.is_inf()) return ERRBIT_NOT_IN_SUBGROUP;
    const F zz = F::sqr(t.Z);
    const bool ok = (t.X == F::mul(ex, zz)) && (t.Y == F::mul(ey, F::mul(zz, t.Z)));
    return ok ? 0 : ERRBIT_NOT_IN_SUBGROUP;
}
template <class G, int CIN>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
endo_subgroup_kernel(const uint8_t *__restrict__ in, uint64_t n, int *__restrict__ err) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G::LPP;
    if (i >= n) return;
    int bad = endo_subgroup_thread<G>(in, CIN, i);
    if (bad) atomicOr(err, bad);
}

// The same test on JACOBIAN points (X, Y, Z) with Z ignored (endo_membership_lazy's comment): the bucket sums of the
// membership-fused MSM, straight from the combine pass.  G1 with a carry-free twin only (on G2 the Frobenius inside psi does not
// commute with a scaling by Z in Fq2).
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
endo_subgroup_jac_kernel(const typename G::J *__restrict__ pts, uint64_t n, int *__restrict__ err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename G::J p = pts[i];
    if (p.Z.is_zero()) return;   // an empty bucket
    const int bad = endo_membership_lazy<G>(p.X, p.Y);
    if (bad) atomicOr(err, bad);
}

// ---- K3'' : membership of an MNT twist point through psi (gen_constants.py PSI_*): scratch[i] = [|t - 1|] P_i was left by
// K1 (fixed scalar); accept iff psi(P_i) == +-scratch[i].  Exact for any curve point: the kernel of psi - [t - 1] has r points.
template <class G>
PTAU_HD int psi_check_thread(const uint8_t *in, int compressed_in, const typename G::J *scratch, uint64_t i) {
    using F = typename G::F;
    using B = typename F::Base;
    using Cfg = typename G::J::CfgT;
    static_assert(Cfg::PSI_KIND == 1 && F::DEG >= 2, "no psi membership test for this group");
    typename G::A p;
    int bad = read_point_any<G>(in, compressed_in, i, p);
    if (bad) return bad;
    if (p.inf) return 0;
    const typename G::J R = scratch[i];
    if (R.is_inf()) return ERRBIT_NOT_IN_SUBGROUP;
    const B cx = B::load_const(Cfg::PSI_CX), cy = B::load_const(Cfg::PSI_CY);
    F fx, fy;
    if constexpr (F::DEG == 2) { fx = F::conj(p.x); fy = F::conj(p.y); }
    else { fx = F::frobenius(p.x); fy = F::frobenius(p.y); }
    const F ex = F::mul_base(fx, cx);
    F ey = F::mul_base(fy, cy);
    if (Cfg::PSI_NEG) ey = F::neg(ey);   // psi(P) = [t - 1]P = -[|t - 1|]P
    const F zz = F::sqr(R.Z);
    const bool ok = (R.X == F::mul(ex, zz)) && (R.Y == F::mul(ey, F::mul(zz, R.Z)));
    return ok ? 0 : ERRBIT_NOT_IN_SUBGROUP;
}
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
psi_check_kernel(const uint8_t *__restrict__ in, int compressed_in, const typename G::J *__restrict__ scratch, uint64_t n,
                 int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = psi_check_thread<G>(in, compressed_in, scratch, i);
    if (bad) atomicOr(err, bad);
}

// Batched normalisation + serialisation.  Thread t owns elements t, t+T, t+2T, ... (< n).
template <class G>
PTAU_HD int normalize_thread(const typename G::J *scratch, typename G::F *pre, uint8_t *out, int compressed_out,
                             uint64_t n, uint64_t T, uint64_t t, int check_nonzero) {
    using F = typename G::F;
    using A = typename G::A;
    int bad = 0;
    F acc = F::one();
    for (uint64_t i = t; i < n; i += T) {
        pre[i] = acc;
        F z = scratch[i].Z;
        if (!z.is_zero()) acc = F::mul(acc, z);
    }
    F inv = F::inverse(acc);
    uint64_t cnt = (n - t + T - 1) / T;
    const int osz = compressed_out ? G::COMP : G::UNCOMP;
    for (uint64_t j = cnt; j-- > 0;) {
        uint64_t i = t + j * T;
        typename G::J q = scratch[i];
        A a;
        if (q.Z.is_zero()) {
            a.inf = true;
            a.x = F::zero();
            a.y = F::zero();
            if (check_nonzero) bad |= ERRBIT_INFINITY;
        } else {
            F zi = F::mul(inv, pre[i]);
            inv = F::mul(inv, q.Z);
            F zi2 = F::sqr(zi);
            a.x = F::mul(q.X, zi2);
            a.y = F::mul(q.Y, F::mul(zi2, zi));
            a.inf = false;
        }
        write_point<F>(out + i * osz, a, compressed_out != 0);
    }
    return bad;
}

template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
normalize_kernel(const typename G::J *__restrict__ scratch, typename G::F *__restrict__ pre,
                 uint8_t *__restrict__ out, int compressed_out, uint64_t n, uint64_t T,
                 int check_nonzero, int *__restrict__ err) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int bad = normalize_thread<G>(scratch, pre, out, compressed_out, n, T, t, check_nonzero);
    if (bad) atomicOr(err, bad);
}

// Flag every scratch point that is NOT the point at infinity (subgroup check: r*P must be O).
template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
expect_infinity_kernel(const typename G::J *__restrict__ scratch, uint64_t n, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!scratch[i].Z.is_zero()) atomicOr(err, ERRBIT_NOT_IN_SUBGROUP);
}

// K2: compressed -> uncompressed (Phase1::decompress); no scalar multiplication.
template <class G>
PTAU_HD int decompress_thread(const uint8_t *in, uint8_t *out, uint64_t i, int check_nonzero) {
    using A = typename G::A;
    using J = typename G::J;
    A p;
    int rc = read_compressed<typename G::F, typename J::CfgT>(in + i * G::COMP, p);
    int bad = 0;
    if (rc == 1) bad |= ERRBIT_NONCANONICAL;
    if (rc == 2) bad |= ERRBIT_NOT_ON_CURVE;
    if (check_nonzero && p.inf) bad |= ERRBIT_INFINITY;
    if (rc) { p.inf = true; p.x = G::F::zero(); p.y = G::F::zero(); }
    write_point<typename G::F>(out + i * G::UNCOMP, p, false);
    return bad;
}

template <class G>
__global__ void __launch_bounds__(256, G::MIN_WAVES)
decompress_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t n, int check_nonzero,
                  int *__restrict__ err) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G::LPP;
    if (i >= n) return;
    int bad = decompress_thread<G>(in, out, i, check_nonzero);
    if (bad) atomicOr(err, bad);
}

// ---- A/B variant (north_star wording: "window tables staged in LDS"): K1 of a one-lane carry-free G1 group with the per-point
// window table in LDS.  160 KB of LDS per CU hold at most a FOUR-entry table (1P..4P, common Z: 2 x 14 limbs per entry) for
// the 256 resident lanes of ONE wave per SIMD, so the windows are 3 bits (85 windows: 252 doublings + ~74 additions instead of
// 252 + 60).  Selected by PTAU_AB_LDS_TABLE=1 for BatchExpMode exact-path launches; measured against the HBM table in
// profiles/r03_ab_lds_table.txt — it is NOT the product path.
template <class G>
__global__ void __launch_bounds__(256, 1)
window_scalar_mul_lds_table_kernel(const uint8_t *__restrict__ in, typename G::J *__restrict__ scratch, uint64_t n,
                                   ScalarSource src, int check_input, int *__restrict__ err) {
    using Fr = typename G::Fr;
    using A = typename G::A;
    using J = typename G::J;
    using L = LazyOf<G>;
    using K = typename L::K;
    using AccT = typename L::Acc;
    using KM = typename LoopAcc<G, true>::zc_type;
    constexpr int NL = Fr::N, E3 = 4, NW3 = 85;
    static_assert(G::LANE_MODE == LANES_ONE && G::F::DEG == 1 && NL == 8 && L::N == 14, "written for BLS12-377 G1");
    __shared__ uint32_t tab[E3 * 2 * L::N * 256];   // [entry][coordinate word][lane]: conflict-free
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    A p;
    int bad = read_point_any<G>(in, 0, i, p);
    if ((check_input & 1) && p.inf) bad |= ERRBIT_INFINITY;
    if (bad || p.inf) {
        store_jac<G>(scratch, i, J::infinity());
        if (bad) atomicOr(err, bad);
        return;
    }
    auto slot = [&](int e, int w) -> uint32_t & { return tab[((e * 2 * L::N) + w) * 256 + threadIdx.x]; };
    auto st = [&](int e, int c, const auto &v) {
#pragma unroll
        for (int j = 0; j < L::N; j++) slot(e, c * L::N + j) = v.l[j];
    };
    auto ld = [&](int e, int c) {
        KM v;
#pragma unroll
        for (int j = 0; j < L::N; j++) v.l[j] = slot(e, c * L::N + j);
        return v;
    };
    const typename L::C px = L::to_lazy(p.x), py = L::to_lazy(p.y);
    const typename L::AX pxa = px;
    const typename L::AY pya = py;
    AccT cur = {px, py, K::one()};
    decltype(cur.Zc) zs[E3 - 1];
    KM pre[E3 - 1];
    KM zl = K::one();
#pragma unroll
    for (int e = 1; e < E3; e++) {
        cur = AccT::add_mixed(cur, pxa, pya);
        st(e, 0, cur.X);
        st(e, 1, cur.Y);
        zs[e - 1] = cur.Zc;
        pre[e - 1] = zl;
        if (!K::is_zero(cur.Zc)) zl = K::mul(zl, cur.Zc);
    }
    KM suf = K::one();
#pragma unroll
    for (int e = E3 - 1; e >= 1; e--) {
        const auto ze = zs[e - 1];
        if (K::is_zero(ze)) {
            st(e, 0, K::zero());
            st(e, 1, K::zero());
        } else {
            const KM c = K::mul(pre[e - 1], suf);
            suf = K::mul(suf, ze);
            const auto c2 = K::sqr(c);
            st(e, 0, KM(K::mul(decltype(cur.X)(ld(e, 0)), c2)));
            st(e, 1, KM(K::mul(decltype(cur.Y)(ld(e, 1)), K::mul(c2, c))));
        }
    }
    const KM zcz = zl;
    {
        const auto c2 = K::sqr(zcz);
        st(0, 0, KM(K::mul(px, c2)));
        st(0, 1, KM(K::mul(py, K::mul(c2, zcz))));
    }
    // signed 3-bit digits: k + sum_w 4 8^w, digit_w = ((k' >> 3w) & 7) - 4 in [-4, 3]
    uint32_t k[NL];
    thread_scalar<Fr>(src, i, k);
    {
        uint32_t cy = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            uint32_t c = 0;
            for (int w = 0; w < NW3; w++) {
                const int b = 3 * w + 2;
                if (b >= 32 * j && b < 32 * j + 32) c |= 1u << (b - 32 * j);
            }
            k[j] = addc(k[j], c, cy);
        }
    }
    AccT acc = AccT::infinity();
    for (int w = NW3 - 1; w >= 0; w--) {
        if (!acc.is_inf()) {
#pragma nounroll
            for (int d = 0; d < 3; d++) acc = AccT::dbl(acc);
        }
        const int bit = 3 * w, li = bit >> 5, sh = bit & 31;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            lo = (j == li) ? k[j] : lo;
            hi = (j == li + 1) ? k[j] : hi;
        }
        const int dgt = (int)((uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & 7u) - 4;
        if (dgt != 0) {
            const int mag = dgt < 0 ? -dgt : dgt;
            typename L::AX x = ld(mag - 1, 0);
            const KM y = ld(mag - 1, 1);
            if (!(K::limbs_all_zero(x) && K::limbs_all_zero(y))) {
                typename L::AY ys = y;
                if (dgt < 0) ys = K::neg(y);
                acc = AccT::add_mixed(acc, x, ys);
            }
        }
    }
    const auto z = K::mul(acc.Zc, zcz);
    store_jac<G>(scratch, i, J{L::from_lazy(acc.X), L::from_lazy(acc.Y), L::from_lazy(z)});
}

inline bool ab_lds_table_enabled() {
    static const bool v = [] {
        const char *e = getenv("PTAU_AB_LDS_TABLE");
        return e && e[0] == '1';
    }();
    return v;
}

}  // namespace ptau
#include "k1_lazy_a.hpp"
namespace ptau {

// ---------------------------------------------------------------------------------------------
// Host-side launchers, one table per (curve, group); capi.hip dispatches through these.
struct GroupOps {
    int comp, uncomp;          // serialized sizes
    size_t jac_bytes, fe_bytes;
    size_t wtab_entry_bytes;   // one window-table entry of one point in K1 (>= jac_bytes: the widest kernel's layout)
    int fr_limbs32, fr_bits;
    // builds the device scalar table on the host: [base, tau^(2^k)...] Montgomery 32-bit limbs
    void (*build_table)(const uint64_t *tau, const uint64_t *coeff, uint64_t start, uint32_t *out);
    void (*scalar_mul)(hipStream_t, const uint8_t *in, int cin, void *scratch, uint64_t n, const uint32_t *table,
                       int check_input, int *err);
    void (*window_scalar_mul)(hipStream_t, const uint8_t *in, int cin, void *scratch, void *wtab, uint64_t n,
                              ScalarSource src, int check_input, int use_glv, int *err);
    void (*reduce_sum)(hipStream_t, const void *in, void *out, uint64_t n, uint64_t T);
    // bucket MSM (K4)
    size_t affine_bytes;
    void (*msm_prepare)(hipStream_t, const uint8_t *in, int cin, void *aff, uint32_t *dig, uint32_t *counts, uint64_t n_points,
                        uint64_t n_scalars, ScalarSource src, MsmPlan pl, int *err);
    void (*msm_accumulate)(hipStream_t, const void *aff, const uint32_t *sorted, const uint32_t *offsets,
                           const uint32_t *run_offsets, void *partials, uint32_t shift, uint32_t total_buckets, uint64_t max_runs,
                           uint32_t run_len);
    int msm_lanes_per_run;   // threads the accumulate pass spends on one run (2 where the lane-pair twin serves it)
    void (*msm_combine)(hipStream_t, const void *partials, const uint32_t *run_offsets, void *buckets, uint64_t total_buckets);
    void (*msm_bucket_reduce)(hipStream_t, const void *buckets, void *parts, MsmPlan pl, uint64_t total);
    void (*reduce_rows)(hipStream_t, const void *in, void *out, uint32_t rows, uint32_t cols, uint32_t T);
    void (*msm_horner)(hipStream_t, const void *win, void *out, MsmPlan pl, int nsum, uint64_t out_stride);
    void (*on_curve)(hipStream_t, const uint8_t *in, uint64_t n, int *err);
    void (*normalize)(hipStream_t, const void *scratch, void *pre, uint8_t *out, int cout, uint64_t n, uint64_t T,
                      int check_nonzero, int *err);
    void (*expect_infinity)(hipStream_t, const void *scratch, uint64_t n, int *err);
    // endomorphism membership test incl. the on-curve test of uncompressed input; NULL where the group has none
    void (*endo_subgroup)(hipStream_t, const uint8_t *in, int cin, uint64_t n, int *err);
    // ... on Jacobian points in the group's standard layout, Z ignored (BLS12-377 G1: bucket sums need no normalisation); or NULL
    void (*endo_subgroup_jac)(hipStream_t, const void *jac, uint64_t n, int *err);
    // membership through K1's GLV loop with the fixed pair (a, b), [a]P + [b]phi(P) == O (SCALAR_SUBGROUP_ENDO): BW6-761
    bool subgroup_via_glv;
    // the membership test costs a small fraction of what the endomorphism multiplication saves over the plain one (BLS12-377
    // G2: 64 doublings against 188): BatchExpMode::Auto probes the vector and takes the GLS path when every element passes
    bool membership_cheap;
    // the group has an endomorphism-accelerated multiplication at all (GLV / GLS; use_glv of window_scalar_mul is honoured)
    bool has_endo_mul;
    // ... and the batched membership probe of BatchExpMode::Auto costs less than the endomorphism saves: BLS12-377 17 % / 49 %,
    // BW6-761 8 %, MNT6-753 G2 18 %; on the MNT4-753 twist since round 4 (bucket sums and their tails in the lane-pair-per-
    // coefficient field, the psi comparison inside K1: probe 8.9 ms + GLS 92 ms beside its tests against 112 ms plain, +7 %;
    // in round 3 the 16 ms probe and the 13 ms saving cancelled)
    bool auto_probe_pays;
    // membership through psi (MNT twists): K1 with the fixed scalar psi_scalar (fr_limbs32 words), then psi_check on its result
    const uint32_t *psi_scalar;
    void (*psi_check)(hipStream_t, const uint8_t *in, int cin, const void *scratch, uint64_t n, int *err);
    void (*decompress)(hipStream_t, const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero, int *err);
    const uint32_t *group_order;   // r, canonical 32-bit limbs (fr_limbs32)
    // rough cost of ONE exact membership test in mixed additions of the group (msm_plan's bucket weight in the membership-fused
    // MSM): doublings x 7/11 + additions of the test the library runs for this group
    int subgroup_cost_adds;
    // O(1) host-side decompression for the pairing tail (a handful of public-key / head elements)
    int (*decompress_host)(const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero);
};

// PTAU_LANE_PAIR=0 keeps Fq2 groups on the single-lane kernels (A/B runs).
inline bool lane_pair_enabled() {
    static const bool v = [] {
        const char *e = getenv("PTAU_LANE_PAIR");
        return !(e && e[0] == '0');
    }();
    return v;
}

// GL: the two-threads-per-point twin of G (same curve and byte formats) or void: an Fq2-pair twin serves the tau-update, the
// membership test and the bucket sums of uncompressed points; a limb-split twin serves the tau-update (K1).
template <class GL, int MODE>
constexpr bool twin_is() {
    if constexpr (std::is_void_v<GL>) return false;
    else return GL::LANE_MODE == MODE;
}
inline bool probe_pays_on_mnt4_twist() {
    static const bool v = [] {
        const char *e = getenv("PTAU_AUTO_PROBE_MNT4");
        return !(e && e[0] == '0');
    }();
    return v;
}
// PTAU_LAZY_A=0 keeps the a != 0 groups on the round-2 kernels (reduced field, pair layout): A/B runs.
inline bool lazy_a_enabled() {
    static const bool v = [] {
        const char *e = getenv("PTAU_LAZY_A");
        return !(e && e[0] == '0');
    }();
    return v;
}

// GA: the carry-free one-pair-per-coefficient K1 of an a != 0 group (LazyAGroup, k1_lazy_a.hpp) or void.
template <class G, class GL = void, class GA = void>
struct Launch {
    using Fr = typename G::Fr;
    static dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

    static void build_table(const uint64_t *tau, const uint64_t *coeff, uint64_t start, uint32_t *out) {
        constexpr int N = Fr::N;
        uint32_t c[N];
        auto load = [&](const uint64_t *src) {
            for (int i = 0; i < N; i++) c[i] = (uint32_t)(src[i / 2] >> (32 * (i & 1)));
            return Fr::from_canonical(c);
        };
        Fr t = load(tau);
        Fr base = coeff ? load(coeff) : Fr::one();
        // base *= tau^start
        Fr pw = t;
        for (uint64_t s = start; s; s >>= 1) {
            if (s & 1) base = Fr::mul(base, pw);
            pw = Fr::sqr(pw);
        }
        for (int i = 0; i < N; i++) out[i] = base.l[i];
        pw = t;
        for (int k = 0; k < SCALAR_TABLE_ENTRIES; k++) {
            for (int i = 0; i < N; i++) out[(1 + k) * N + i] = pw.l[i];
            pw = Fr::sqr(pw);
        }
    }
    static void scalar_mul(hipStream_t s, const uint8_t *in, int cin, void *scratch, uint64_t n, const uint32_t *table,
                           int check_input, int *err) {
        if (!n) return;
        hipLaunchKernelGGL(scalar_mul_kernel<G>, grid(n), dim3(256), 0, s, in, cin, (typename G::J *)scratch, n, table,
                           check_input, err);
    }
    static void window_scalar_mul(hipStream_t s, const uint8_t *in, int cin, void *scratch, void *wtab, uint64_t n,
                                  ScalarSource src, int check_input, int use_glv, int *err) {
        if (!n) return;
        if constexpr (!std::is_void_v<GA>) {
            static_assert(GA::UNCOMP == G::UNCOMP && 3 * GA::DEG * GA::N29 * 4 == sizeof(typename G::J), "standard layouts");
            if (!cin && lazy_a_enabled()) {   // 2 / 4 / 6 lanes per point; results land in G's Jacobian layout
                const uint64_t blocks = (n + GA::POINTS_PER_BLOCK - 1) / GA::POINTS_PER_BLOCK;
                hipLaunchKernelGGL(window_scalar_mul_lazy_a_kernel<GA>, dim3((unsigned)blocks), dim3(256), 0, s, in, (uint32_t *)scratch,
                                   (uint32_t *)wtab, n, src, check_input, src.mode == SCALAR_TAU_TABLE ? use_glv : 0, err);
                return;
            }
        }
        if (src.mode == SCALAR_FIXED_PSI) src.mode = SCALAR_FIXED;   // the kernels below leave the comparison to psi_check_kernel
        if constexpr (!std::is_void_v<GL>) {
            if (!cin && lane_pair_enabled()) {   // two threads per point; results land in G's Jacobian layout
                hipLaunchKernelGGL((window_scalar_mul_kernel<GL, 0>), grid(n * 2), dim3(256), 0, s, in, (typename GL::J *)scratch,
                                   (typename GL::J *)wtab, n, src, check_input, use_glv, err);
                return;
            }
        }
        if constexpr (G::LAZY_LOOP && G::LANE_MODE == LANES_ONE && G::F::DEG == 1 && Fr::N == 8) {
            if (!cin && !use_glv && src.mode != SCALAR_SUBGROUP_ENDO && ab_lds_table_enabled()) {   // A/B only (see the kernel)
                hipLaunchKernelGGL(window_scalar_mul_lds_table_kernel<G>, grid(n), dim3(256), 0, s, in, (typename G::J *)scratch, n, src,
                                   check_input, err);
                return;
            }
        }
        if (cin)
            hipLaunchKernelGGL((window_scalar_mul_kernel<G, 1>), grid(n), dim3(256), 0, s, in, (typename G::J *)scratch,
                               (typename G::J *)wtab, n, src, check_input, use_glv, err);
        else
            hipLaunchKernelGGL((window_scalar_mul_kernel<G, 0>), grid(n), dim3(256), 0, s, in, (typename G::J *)scratch,
                               (typename G::J *)wtab, n, src, check_input, use_glv, err);
    }
    static void reduce_sum(hipStream_t s, const void *in, void *out, uint64_t n, uint64_t T) {
        if (!T) return;
        if constexpr (lazy_a_tails()) {
            if (lazy_a_enabled()) {
                hipLaunchKernelGGL(reduce_sum_lazy_a_kernel<GA>, ga_grid(T), dim3(256), 0, s, (const uint32_t *)in, (uint32_t *)out, n, T);
                return;
            }
        }
        hipLaunchKernelGGL(reduce_sum_kernel<G>, grid(T), dim3(256), 0, s, (const typename G::J *)in, (typename G::J *)out, n, T);
    }
    static void msm_prepare(hipStream_t s, const uint8_t *in, int cin, void *aff, uint32_t *dig, uint32_t *counts,
                            uint64_t n_points, uint64_t n_scalars, ScalarSource src, MsmPlan pl, int *err) {
        if (!n_points) return;
        hipLaunchKernelGGL(msm_prepare_kernel<G>, grid(n_points), dim3(256), 0, s, in, cin, (typename G::A *)aff, dig, counts,
                           n_points, n_scalars, src, pl, err);
    }
    static void msm_accumulate(hipStream_t s, const void *aff, const uint32_t *sorted, const uint32_t *offsets,
                               const uint32_t *run_offsets, void *partials, uint32_t shift, uint32_t total_buckets, uint64_t max_runs,
                               uint32_t run_len) {
        if constexpr (!std::is_void_v<GA>) {
            if constexpr (GA::DEG >= 2) {   // the twists of the a != 0 curves: one lane pair per tower coefficient (k1_lazy_a.hpp)
                static_assert(sizeof(typename G::A) == (2 * GA::DEG * GA::N29 + 1) * 4 && sizeof(typename G::J) == 3 * GA::DEG * GA::N29 * 4,
                              "standard affine / Jacobian layouts");
                if (lazy_a_enabled()) {
                    const uint64_t blocks = (max_runs + GA::POINTS_PER_BLOCK - 1) / GA::POINTS_PER_BLOCK;
                    hipLaunchKernelGGL(msm_accumulate_lazy_a_kernel<GA>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t *)aff, sorted,
                                       offsets, run_offsets, (uint32_t *)partials, shift, total_buckets, max_runs, run_len);
                    return;
                }
            }
        }
        if constexpr (!std::is_void_v<GL>) {
            static_assert(sizeof(typename G::A) == (2 * G::F::DEG * G::Fq::N + 1) * 4 || G::Fq::LB != 29, "standard affine layout");
            if (lane_pair_enabled()) {   // two threads per run; affine input and Jacobian partials in G's layouts
                hipLaunchKernelGGL(msm_accumulate_kernel<GL>, grid(max_runs * 2), dim3(256), 0, s, (const typename GL::A *)aff,
                                   sorted, offsets, run_offsets, (typename GL::J *)partials, shift, total_buckets, run_len);
                return;
            }
        }
        hipLaunchKernelGGL(msm_accumulate_kernel<G>, grid(max_runs), dim3(256), 0, s, (const typename G::A *)aff, sorted,
                           offsets, run_offsets, (typename G::J *)partials, shift, total_buckets, run_len);
    }
    // the small kernels behind the bucket sums on the twists of the a != 0 curves: one lane pair per tower coefficient
    static constexpr bool lazy_a_tails() {
        if constexpr (!std::is_void_v<GA>) return GA::DEG >= 2;
        else return false;
    }
    template <class T = GA>
    static dim3 ga_grid(uint64_t threads) { return dim3((unsigned)((threads + T::POINTS_PER_BLOCK - 1) / T::POINTS_PER_BLOCK)); }
    static void msm_combine(hipStream_t s, const void *partials, const uint32_t *run_offsets, void *buckets, uint64_t total_buckets) {
        if constexpr (lazy_a_tails()) {
            if (lazy_a_enabled()) {
                hipLaunchKernelGGL(msm_combine_lazy_a_kernel<GA>, ga_grid(total_buckets), dim3(256), 0, s, (const uint32_t *)partials, run_offsets,
                                   (uint32_t *)buckets, total_buckets);
                return;
            }
        }
        hipLaunchKernelGGL(msm_combine_kernel<G>, grid(total_buckets), dim3(256), 0, s, (const typename G::J *)partials,
                           run_offsets, (typename G::J *)buckets, total_buckets);
    }
    static void msm_bucket_reduce(hipStream_t s, const void *buckets, void *parts, MsmPlan pl, uint64_t total) {
        if constexpr (lazy_a_tails()) {
            if (lazy_a_enabled()) {
                hipLaunchKernelGGL(msm_bucket_reduce_lazy_a_kernel<GA>, ga_grid(total), dim3(256), 0, s, (const uint32_t *)buckets, (uint32_t *)parts, pl, total);
                return;
            }
        }
        hipLaunchKernelGGL(msm_bucket_reduce_kernel<G>, grid(total), dim3(256), 0, s, (const typename G::J *)buckets,
                           (typename G::J *)parts, pl, total);
    }
    static void reduce_rows(hipStream_t s, const void *in, void *out, uint32_t rows, uint32_t cols, uint32_t T) {
        const uint64_t total = (uint64_t)rows * T;
        if constexpr (lazy_a_tails()) {
            if (lazy_a_enabled()) {
                hipLaunchKernelGGL(reduce_rows_lazy_a_kernel<GA>, ga_grid(total), dim3(256), 0, s, (const uint32_t *)in, (uint32_t *)out, cols, T, total);
                return;
            }
        }
        hipLaunchKernelGGL(reduce_rows_kernel<G>, grid(total), dim3(256), 0, s, (const typename G::J *)in, (typename G::J *)out,
                           cols, T, total);
    }
    static void msm_horner(hipStream_t s, const void *win, void *out, MsmPlan pl, int nsum, uint64_t out_stride) {
        if constexpr (lazy_a_tails()) {
            if (lazy_a_enabled()) {
                hipLaunchKernelGGL(msm_horner_lazy_a_kernel<GA>, dim3(nsum), dim3(64), 0, s, (const uint32_t *)win, (uint32_t *)out, pl, out_stride);
                return;
            }
        }
        hipLaunchKernelGGL(msm_horner_kernel<G>, dim3(nsum), dim3(64), 0, s, (const typename G::J *)win, (typename G::J *)out, pl,
                           out_stride);
    }
    static void on_curve(hipStream_t s, const uint8_t *in, uint64_t n, int *err) {
        if (!n) return;
        hipLaunchKernelGGL(on_curve_kernel<G>, grid(n), dim3(256), 0, s, in, n, err);
    }
    static void normalize(hipStream_t s, const void *scratch, void *pre, uint8_t *out, int cout, uint64_t n, uint64_t T,
                          int check_nonzero, int *err) {
        if (!n) return;
        hipLaunchKernelGGL(normalize_kernel<G>, grid(T), dim3(256), 0, s, (const typename G::J *)scratch,
                           (typename G::F *)pre, out, cout, n, T, check_nonzero, err);
    }
    static void expect_infinity(hipStream_t s, const void *scratch, uint64_t n, int *err) {
        if (!n) return;
        hipLaunchKernelGGL(expect_infinity_kernel<G>, grid(n), dim3(256), 0, s, (const typename G::J *)scratch, n, err);
    }
    static void endo_subgroup_jac(hipStream_t s, const void *jac, uint64_t n, int *err) {
        if constexpr (G::J::CfgT::ENDO_KIND == 1 && G::LAZY_LOOP && G::LANE_MODE == LANES_ONE) {
            if (!n) return;
            hipLaunchKernelGGL(endo_subgroup_jac_kernel<G>, grid(n), dim3(256), 0, s, (const typename G::J *)jac, n, err);
        }
    }
    static void endo_subgroup(hipStream_t s, const uint8_t *in, int cin, uint64_t n, int *err) {
        if constexpr (G::J::CfgT::ENDO_KIND != 0) {
            if (!n) return;
            if constexpr (twin_is<GL, LANES_FQ2_PAIR>()) {
                if (!cin && lane_pair_enabled()) {
                    hipLaunchKernelGGL((endo_subgroup_kernel<GL, 0>), grid(n * 2), dim3(256), 0, s, in, n, err);
                    return;
                }
            }
            if (cin) hipLaunchKernelGGL((endo_subgroup_kernel<G, 1>), grid(n), dim3(256), 0, s, in, n, err);
            else hipLaunchKernelGGL((endo_subgroup_kernel<G, 0>), grid(n), dim3(256), 0, s, in, n, err);
        }
    }
    static void psi_check(hipStream_t s, const uint8_t *in, int cin, const void *scratch, uint64_t n, int *err) {
        if constexpr (G::J::CfgT::PSI_KIND == 1) {
            if (!n) return;
            if constexpr (!std::is_void_v<GA>) {
                if (!cin && lazy_a_enabled()) return;   // window_scalar_mul (SCALAR_FIXED_PSI) compared in the kernel
            }
            hipLaunchKernelGGL(psi_check_kernel<G>, grid(n), dim3(256), 0, s, in, cin, (const typename G::J *)scratch, n, err);
        }
    }
    static void decompress(hipStream_t s, const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero, int *err) {
        if (!n) return;
        if constexpr (twin_is<GL, LANES_LIMB_SPLIT>()) {
            if (lane_pair_enabled()) {   // square root on lane pairs (fpl.hpp); same bytes in, same bytes out
                hipLaunchKernelGGL(decompress_kernel<GL>, grid(n * 2), dim3(256), 0, s, in, out, n, check_nonzero, err);
                return;
            }
        }
        hipLaunchKernelGGL(decompress_kernel<G>, grid(n), dim3(256), 0, s, in, out, n, check_nonzero, err);
    }
    static int decompress_host(const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero) {
        int bad = 0;
        for (uint64_t i = 0; i < n; i++) bad |= decompress_thread<G>(in, out, i, check_nonzero);
        return bad;
    }
    static GroupOps ops() {
        GroupOps o;
        o.comp = G::COMP;
        o.uncomp = G::UNCOMP;
        o.jac_bytes = sizeof(typename G::J);
        o.fe_bytes = sizeof(typename G::F);
        o.wtab_entry_bytes = sizeof(typename G::J);
        if constexpr (!std::is_void_v<GA>) o.wtab_entry_bytes = std::max(o.wtab_entry_bytes, (size_t)GA::WTAB_ENTRY_BYTES);
        o.fr_limbs32 = Fr::N;
        o.fr_bits = Fr::Params::BITS;
        o.build_table = build_table;
        o.scalar_mul = scalar_mul;
        o.window_scalar_mul = window_scalar_mul;
        o.on_curve = on_curve;
        o.reduce_sum = reduce_sum;
        o.affine_bytes = msm_aff_bytes<G>();
        o.msm_prepare = msm_prepare;
        o.msm_accumulate = msm_accumulate;
        o.msm_lanes_per_run = 1;
        if constexpr (!std::is_void_v<GL>) o.msm_lanes_per_run = GL::LPP;
        if constexpr (!std::is_void_v<GA>) {
            if constexpr (GA::DEG >= 2) {
                if (lazy_a_enabled()) o.msm_lanes_per_run = GA::LPP;
            }
        }
        o.msm_combine = msm_combine;
        o.msm_bucket_reduce = msm_bucket_reduce;
        o.reduce_rows = reduce_rows;
        o.msm_horner = msm_horner;
        o.normalize = normalize;
        o.expect_infinity = expect_infinity;
        o.endo_subgroup = G::J::CfgT::ENDO_KIND != 0 ? endo_subgroup : nullptr;
        o.endo_subgroup_jac = nullptr;
        if constexpr (G::J::CfgT::ENDO_KIND == 1 && G::LAZY_LOOP && G::LANE_MODE == LANES_ONE) o.endo_subgroup_jac = endo_subgroup_jac;
        o.subgroup_via_glv = G::J::CfgT::GLV_KIND == 2;
        o.membership_cheap = G::J::CfgT::ENDO_KIND == 2 && Fr::N == 8;
        o.has_endo_mul = G::J::CfgT::GLV_KIND != 0 || (G::J::CfgT::ENDO_KIND == 2 && Fr::N == 8);
        if constexpr (!std::is_void_v<GA>) o.has_endo_mul = o.has_endo_mul || GA::HAS_GLS;
        // PTAU_AUTO_PROBE_MNT4=0 restores the round-3 choice (no probe on the MNT4-753 twist) for A/B runs
        o.auto_probe_pays = o.has_endo_mul && (!(G::J::CfgT::PSI_KIND == 1 && G::F::DEG == 2) || probe_pays_on_mnt4_twist());
        if constexpr (G::J::CfgT::PSI_KIND == 1) {
            static_assert(sizeof(G::J::CfgT::PSI_N) / 4 == Fr::N, "psi scalar must have the scalar field's word count");
            o.psi_scalar = G::J::CfgT::PSI_N;
            o.psi_check = psi_check;
        } else {
            o.psi_scalar = nullptr;
            o.psi_check = nullptr;
        }
        o.decompress = decompress;
        o.group_order = Fr::Params::MOD;
        {
            using Cfg = typename G::J::CfgT;
            constexpr int RB = Fr::Params::BITS;
            o.subgroup_cost_adds = Cfg::ENDO_KIND == 1 ? 93          // two chains of 63 doublings + 6 additions
                                 : Cfg::ENDO_KIND == 2 ? 46          // one chain
                                 : Cfg::GLV_KIND == 2 ? RB / 2 * 7 / 11 + RB / 8     // [a]P + [b]phi(P), halves of r
                                 : Cfg::PSI_KIND == 1 ? RB / 2 * 8 / 14 + RB / 8     // [|t - 1|]P, a != 0 doublings
                                 : RB * 8 / 12 + RB / 4;                              // [r]P (cofactor-1 groups never get here)
        }
        o.decompress_host = decompress_host;
        return o;
    }
};

// one accessor per curve TU
const GroupOps *ops_bls12_377(int group);
const GroupOps *ops_bw6_761(int group);
const GroupOps *ops_mnt4_753(int group);
const GroupOps *ops_mnt6_753(int group);

}  // namespace ptau
