// HIP kernels of the chunk tau-update (K1) and point decompression (K2), templated over one
// group of one curve; each curve's translation unit instantiates them (k_<curve>.hip).
//
// Replaces setup-utils' apply_powers / batch_exp / read_batch / write_batch and
// Phase1::decompress — reference call sites /root/reference/src/bin/contribute.rs:811-823, 968-986.
//
// Pipeline per vector of n points (all device-resident):
//   scalar_mul_kernel : thread i derives e_i = c * tau^(start+i) from a tau^(2^k) table, reads
//                       point i (uncompressed, or compressed + sqrt), runs the scalar multiplication
//                       and stores the Jacobian result to scratch.
//   normalize_kernel  : Montgomery-trick batched inversion (one inversion per K points), affine
//                       conversion, arkworks serialisation (sign flag from the canonical y).
#pragma once
#include "ec.hpp"

namespace ptau {

// Error bits accumulated in a device word by the kernels.
enum : int { ERRBIT_NONCANONICAL = 1, ERRBIT_NOT_ON_CURVE = 2, ERRBIT_INFINITY = 4, ERRBIT_NOT_IN_SUBGROUP = 8 };

constexpr int SCALAR_TABLE_ENTRIES = 64;   // tau^(2^k), k < 64, preceded by the base coefficient

// Host+device view of one group's types.
template <class FqP, class FrP, class Cfg>
struct GroupTypes {
    using Fq = Fp<FqP>;
    using Fr = Fp<FrP>;
    using F = std::conditional_t<Cfg::DEG == 1, Fq,
              std::conditional_t<Cfg::DEG == 2, Fp2<Fq, Cfg::NR>, Fp3<Fq, Cfg::NR>>>;
    using J = Jac<F, Cfg>;
    using A = Affine<F>;
    static constexpr int COMP = F::BYTES;
    static constexpr int UNCOMP = 2 * F::BYTES;
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
__global__ void __launch_bounds__(256)
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
enum : int { SCALAR_TAU_TABLE = 0, SCALAR_FIXED = 1, SCALAR_ARRAY = 2, SCALAR_PRF = 3 };
struct ScalarSource {
    int mode;
    const uint32_t *ptr;      // TAU_TABLE: the tau^(2^k) table; FIXED: NL canonical limbs; ARRAY: n x NL canonical limbs
    uint64_t seed0, seed1;    // PRF: 128-bit seed
    uint64_t index_base;      // PRF: global index of element 0 (slices of one vector agree across GPUs)
};

PTAU_HD uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// word j of the random coefficient rho_idx of merge_pairs / power_pairs (setup-utils draws Fr::rand from
// thread_rng; here a counter-based generator keyed by a per-verification 128-bit secret, so that any slice
// of a vector, on any GPU, sees the same rho for the same global index).  rho < 2^(BITS-1) < r.
PTAU_HD uint32_t prf_word(uint64_t seed0, uint64_t seed1, uint64_t idx, int nl, int j) {
    uint64_t z = splitmix64(seed0 + (idx * (uint64_t)nl + (uint64_t)j + 1) * 0x9E3779B97F4A7C15ull) ^ seed1;
    return (uint32_t)(splitmix64(z) >> 32);
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

constexpr int WINDOW_BITS = 4;
constexpr int WINDOW_ENTRIES = 8;

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
        scratch[i] = J::infinity();
        return bad;
    }
    // table: e -> (e+1) * P
    J cur = J::from_affine(p);
    wtab[i] = cur;
    for (int e = 1; e < WINDOW_ENTRIES; e++) {
        cur = J::add_mixed(cur, p);
        wtab[(uint64_t)e * n + i] = cur;
    }
    uint32_t k[NL];
    if (src.mode == SCALAR_FIXED) {          // one public scalar for every point (subgroup check: the group order)
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = src.ptr[j];
    } else if (src.mode == SCALAR_ARRAY) {   // explicit per-point scalars (MSM with caller-supplied coefficients)
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = src.ptr[i * NL + j];
    } else if (src.mode == SCALAR_PRF) {     // random linear combination coefficients
        constexpr int TOPBITS = (Fr::Params::BITS - 1) - 32 * (NL - 1);   // bits kept in the top limb
#pragma unroll
        for (int j = 0; j < NL; j++) k[j] = prf_word(src.seed0, src.seed1, src.index_base + i, NL, j);
        k[NL - 1] &= (TOPBITS >= 32) ? 0xffffffffu : ((1u << TOPBITS) - 1u);
    } else {
        Fr ek = scalar_for_index<Fr>(src.ptr, i);
        ek.to_canonical(k);
    }
    using Cfg = typename J::CfgT;
    if constexpr (Cfg::HAS_GLV && NL == 8) {
        if (use_glv) {
            // k = k1 + k2*lambda, k*P = k1*P + k2*phi(P), phi(X, Y, Z) = (beta X, Y, Z): 128-bit interleaved windows
            uint32_t kk[2][4];
            glv_decompose<Cfg>(k, kk[0], kk[1]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t c2 = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) kk[h][j] = addc(kk[h][j], 0x88888888u, c2);   // k1, k2 < 2^126.2: no carry out
            }
            F beta;
#pragma unroll
            for (int j = 0; j < F::N; j++) beta.l[j] = Cfg::GLV_BETA[j];
            J acc = J::infinity();
            for (int w = 31; w >= 0; w--) {
                if (!acc.is_inf()) {
#pragma nounroll
                    for (int d = 0; d < WINDOW_BITS; d++) acc = J::dbl(acc);
                }
#pragma nounroll
                for (int h = 0; h < 2; h++) {
                    uint32_t limb = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) limb = (j == (w >> 3)) ? (h ? kk[1][j] : kk[0][j]) : limb;
                    int dgt = (int)((limb >> ((w & 7) * 4)) & 15u) - 8;
                    if (dgt != 0) {
                        int mag = dgt < 0 ? -dgt : dgt;
                        J ent = wtab[(uint64_t)(mag - 1) * n + i];
                        if (h) ent.X = F::mul(ent.X, beta);
                        if (dgt < 0) ent.Y = F::neg(ent.Y);
                        acc = J::add(acc, ent);
                    }
                }
            }
            scratch[i] = acc;
            return 0;
        }
    }
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) k[j] = addc(k[j], 0x88888888u, c);
    J acc = J::infinity();
    for (int w = NW - 1; w >= 0; w--) {
        if (!acc.is_inf()) {
#pragma nounroll
            for (int d = 0; d < WINDOW_BITS; d++) acc = J::dbl(acc);
        }
        uint32_t limb = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) limb = (j == (w >> 3)) ? k[j] : limb;   // register select, no scratch
        int dgt = (int)((limb >> ((w & 7) * 4)) & 15u) - 8;
        if (dgt != 0) {
            int mag = dgt < 0 ? -dgt : dgt;
            J ent = wtab[(uint64_t)(mag - 1) * n + i];
            if (dgt < 0) ent.Y = F::neg(ent.Y);
            acc = J::add(acc, ent);
        }
    }
    scratch[i] = acc;
    return 0;
}

template <class G>
__global__ void __launch_bounds__(256)
window_scalar_mul_kernel(const uint8_t *__restrict__ in, int compressed_in, typename G::J *__restrict__ scratch,
                         typename G::J *__restrict__ wtab, uint64_t n, ScalarSource src, int check_input,
                         int use_glv, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = window_scalar_mul_thread<G>(in, compressed_in, scratch, wtab, n, i, src, check_input, use_glv);
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
__global__ void __launch_bounds__(256)
reduce_sum_kernel(const typename G::J *__restrict__ in, typename G::J *__restrict__ out, uint64_t n, uint64_t T) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    reduce_sum_thread<G>(in, out, n, T, t);
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
__global__ void __launch_bounds__(256)
on_curve_kernel(const uint8_t *__restrict__ in, uint64_t n, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = on_curve_thread<G>(in, i);
    if (bad) atomicOr(err, bad);
}

// Scalar multiplication of every point by ONE public scalar of `nbits` bits (subgroup check with
// the group order, cofactor clearing): result Jacobian to scratch.
template <class G>
PTAU_HD int fixed_scalar_mul_thread(const uint8_t *in, int compressed_in, typename G::J *scratch, uint64_t i,
                                    const uint32_t *scalar, int nbits) {
    using A = typename G::A;
    using J = typename G::J;
    A p;
    int bad = read_point_any<G>(in, compressed_in, i, p);
    if (bad) {
        scratch[i] = J::infinity();
        return bad;
    }
    J acc = J::infinity();
    for (int b = nbits - 1; b >= 0; b--) {
        acc = J::dbl(acc);
        if ((scalar[b >> 5] >> (b & 31)) & 1u) acc = J::add_mixed(acc, p);   // wave-uniform branch
    }
    scratch[i] = acc;
    return 0;
}

template <class G>
__global__ void __launch_bounds__(256)
fixed_scalar_mul_kernel(const uint8_t *__restrict__ in, int compressed_in, typename G::J *__restrict__ scratch,
                        uint64_t n, const uint32_t *__restrict__ scalar, int nbits, int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = fixed_scalar_mul_thread<G>(in, compressed_in, scratch, i, scalar, nbits);
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
__global__ void __launch_bounds__(256)
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
__global__ void __launch_bounds__(256)
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
__global__ void __launch_bounds__(256)
decompress_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t n, int check_nonzero,
                  int *__restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = decompress_thread<G>(in, out, i, check_nonzero);
    if (bad) atomicOr(err, bad);
}

// ---------------------------------------------------------------------------------------------
// Host-side launchers, one table per (curve, group); capi.hip dispatches through these.
struct GroupOps {
    int comp, uncomp;          // serialized sizes
    size_t jac_bytes, fe_bytes;
    int fr_limbs32, fr_bits;
    // builds the device scalar table on the host: [base, tau^(2^k)...] Montgomery 32-bit limbs
    void (*build_table)(const uint64_t *tau, const uint64_t *coeff, uint64_t start, uint32_t *out);
    void (*scalar_mul)(hipStream_t, const uint8_t *in, int cin, void *scratch, uint64_t n, const uint32_t *table,
                       int check_input, int *err);
    void (*window_scalar_mul)(hipStream_t, const uint8_t *in, int cin, void *scratch, void *wtab, uint64_t n,
                              ScalarSource src, int check_input, int use_glv, int *err);
    void (*reduce_sum)(hipStream_t, const void *in, void *out, uint64_t n, uint64_t T);
    void (*on_curve)(hipStream_t, const uint8_t *in, uint64_t n, int *err);
    void (*fixed_scalar_mul)(hipStream_t, const uint8_t *in, int cin, void *scratch, uint64_t n,
                             const uint32_t *scalar, int nbits, int *err);
    void (*normalize)(hipStream_t, const void *scratch, void *pre, uint8_t *out, int cout, uint64_t n, uint64_t T,
                      int check_nonzero, int *err);
    void (*expect_infinity)(hipStream_t, const void *scratch, uint64_t n, int *err);
    void (*decompress)(hipStream_t, const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero, int *err);
    const uint32_t *group_order;   // r, canonical 32-bit limbs (fr_limbs32)
    // O(1) host-side decompression for the pairing tail (a handful of public-key / head elements)
    int (*decompress_host)(const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero);
};

template <class G>
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
        hipLaunchKernelGGL(window_scalar_mul_kernel<G>, grid(n), dim3(256), 0, s, in, cin, (typename G::J *)scratch,
                           (typename G::J *)wtab, n, src, check_input, use_glv, err);
    }
    static void reduce_sum(hipStream_t s, const void *in, void *out, uint64_t n, uint64_t T) {
        if (!T) return;
        hipLaunchKernelGGL(reduce_sum_kernel<G>, grid(T), dim3(256), 0, s, (const typename G::J *)in, (typename G::J *)out, n, T);
    }
    static void on_curve(hipStream_t s, const uint8_t *in, uint64_t n, int *err) {
        if (!n) return;
        hipLaunchKernelGGL(on_curve_kernel<G>, grid(n), dim3(256), 0, s, in, n, err);
    }
    static void fixed_scalar_mul(hipStream_t s, const uint8_t *in, int cin, void *scratch, uint64_t n,
                                 const uint32_t *scalar, int nbits, int *err) {
        if (!n) return;
        hipLaunchKernelGGL(fixed_scalar_mul_kernel<G>, grid(n), dim3(256), 0, s, in, cin, (typename G::J *)scratch, n,
                           scalar, nbits, err);
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
    static void decompress(hipStream_t s, const uint8_t *in, uint8_t *out, uint64_t n, int check_nonzero, int *err) {
        if (!n) return;
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
        o.fr_limbs32 = Fr::N;
        o.fr_bits = Fr::Params::BITS;
        o.build_table = build_table;
        o.scalar_mul = scalar_mul;
        o.window_scalar_mul = window_scalar_mul;
        o.on_curve = on_curve;
        o.reduce_sum = reduce_sum;
        o.fixed_scalar_mul = fixed_scalar_mul;
        o.normalize = normalize;
        o.expect_infinity = expect_infinity;
        o.decompress = decompress;
        o.group_order = Fr::Params::MOD;
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
