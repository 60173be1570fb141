// Reduced-radix Montgomery prime field for gfx950: limbs of 29 bits in 32-bit VGPRs.
//
// Same interface as Fp<P> (fp.hpp; MI355X-native equivalent of ark-ff 0.4.2 `Fp<MontBackend<_, N>>`, reached by the
// reference through setup-utils' batch_exp, /root/reference/src/bin/contribute.rs:811-823).  Why a second radix:
// tools/ubench_issue.hip measures v_mad_u64_u32 at 4.8 cycles per wave64 instruction and the v_addc_co_u32 that
// carries a full-width 32x32 product into a third accumulator word at another 4.4 — the carry is half of every
// multiply-accumulate of the 32-bit form.  With 29-bit limbs a product is below 2^58 and the 2N products of a
// product-scanning column (26 for 377 bits, 54 for 761) fit ONE 64-bit accumulator: the carry instruction disappears.
// N grows from 12 to 13 limbs (351 MACs instead of 300) and additions need explicit carry chains, but the multiplier
// goes from 60 to 77 G products/s and the square from 69 to 92 (tools/ubench_mul29.hip, profiles/).
//   value = sum l[i] 2^(29 i), l[i] < 2^29, value < p (always fully reduced: == and is_zero stay limb-wise);
//   Montgomery form with R = 2^(29 N).
#pragma once
#include "fp.hpp"

namespace ptau {

// acc += x * y, 64-bit accumulator, no carry out by construction.  Plain C++: the compiler selects v_mad_u64_u32 itself
// (no carry flag to thread through, unlike the 32-bit form), and — unlike one-instruction asm statements, each of which
// was followed by a hazard s_nop: 314 per product — schedules the chain freely.
PTAU_HD void mac64_vv(uint64_t &acc, uint32_t x, uint32_t y) { acc += (uint64_t)x * y; }
PTAU_HD void mac64_vs(uint64_t &acc, uint32_t x, uint32_t c) { acc += (uint64_t)x * c; }   // c: compile-time constant

template <class P>
struct Fp29 {
    static constexpr int N = P::N;       // 29-bit limbs
    static constexpr int N32 = P::N32;   // packed 32-bit words of a canonical value
    static constexpr int LB = 29;
    static constexpr uint32_t MASK = (1u << LB) - 1u;
    static constexpr int DEG = 1;
    static constexpr int LANES = 1;
    static constexpr int BYTES = P::BYTES;
    using Base = Fp29<P>;
    using Params = P;
    using Twin = Fp<typename P::Twin32>;   // the same modulus on packed 32-bit limbs (inversion runs there)
    static_assert(P::LB == 29, "parameter struct generated for another radix");
    uint32_t l[N];

    PTAU_HD static Fp29 load_const(const uint32_t *c) {   // a constant given as N limbs
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = c[i];
        return r;
    }
    PTAU_HD static Fp29 zero() {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    PTAU_HD static Fp29 one() {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R1[i];
        return r;
    }
    PTAU_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    PTAU_HD bool operator==(const Fp29 &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    PTAU_HD bool operator!=(const Fp29 &b) const { return !(*this == b); }

    // a + b mod p: the chain of a + b - p (signed carries) and the chain of a + b, selected by the final borrow
    PTAU_HD static Fp29 add(const Fp29 &a, const Fp29 &b) {
        uint32_t d[N], s[N];
        int32_t cy = 0;
        uint32_t c2 = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t ab = a.l[i] + b.l[i];                       // < 2^30
            const int32_t v = (int32_t)ab - (int32_t)P::MOD[i] + cy;   // arithmetic >> = floor division
            cy = v >> LB;
            d[i] = (uint32_t)v & MASK;
            const uint32_t w = ab + c2;
            c2 = w >> LB;
            s[i] = w & MASK;
        }
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = cy < 0 ? s[i] : d[i];
        return r;
    }
    PTAU_HD static Fp29 sub(const Fp29 &a, const Fp29 &b) {
        uint32_t d[N];
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t v = (int32_t)a.l[i] - (int32_t)b.l[i] + cy;
            cy = v >> LB;
            d[i] = (uint32_t)v & MASK;
        }
        const uint32_t mask = cy < 0 ? 0xffffffffu : 0u;   // negative: add p (mod 2^(29 N))
        uint32_t c2 = 0;
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t w = d[i] + (P::MOD[i] & mask) + c2;
            c2 = w >> LB;
            r.l[i] = w & MASK;
        }
        return r;
    }
    PTAU_HD static Fp29 neg(const Fp29 &a) {
        const uint32_t nz = a.is_zero() ? 0u : 0xffffffffu;
        Fp29 r;
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t v = (int32_t)(P::MOD[i] & nz) - (int32_t)a.l[i] + cy;
            cy = v >> LB;
            r.l[i] = (uint32_t)v & MASK;
        }
        return r;
    }
    PTAU_HD static Fp29 dbl(const Fp29 &a) { return add(a, a); }

    // Montgomery product / square, product scanning.  Column i sums at most N products a_j b_k and N products m_j p_k,
    // each below 2^58 (2^59 for the pre-doubled cross terms of a square, of which there are at most N/2): < 2^64.
    template <bool SQR>
    PTAU_HD static void ps_mul(const uint32_t (&a)[N], const uint32_t (&b)[N], uint32_t (&out)[N]) {
        static_assert(2 * N <= 60, "column sum would overflow 64 bits");
        uint32_t m[N], r[N], d[N];
        if constexpr (SQR) {
#pragma unroll
            for (int i = 0; i < N; i++) d[i] = a[i] << 1;
        }
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if constexpr (SQR) {
#pragma unroll
                for (int j = 0; 2 * j < i; j++) mac64_vv(acc, d[j], a[i - j]);
                if ((i & 1) == 0) mac64_vv(acc, a[i / 2], a[i / 2]);
            } else {
#pragma unroll
                for (int j = 0; j <= i; j++) mac64_vv(acc, a[j], b[i - j]);
            }
#pragma unroll
            for (int j = 0; j < i; j++) mac64_vs(acc, m[j], P::MOD[i - j]);
            m[i] = ((uint32_t)acc * P::INV) & MASK;
            mac64_vs(acc, m[i], P::MOD[0]);
            acc >>= LB;
        }
#pragma unroll
        for (int i = N; i < 2 * N; i++) {
            if constexpr (SQR) {
#pragma unroll
                for (int j = i - N + 1; 2 * j < i; j++) mac64_vv(acc, d[j], a[i - j]);
                if ((i & 1) == 0 && i / 2 < N) mac64_vv(acc, a[i / 2], a[i / 2]);
            } else {
#pragma unroll
                for (int j = i - N + 1; j < N; j++) mac64_vv(acc, a[j], b[i - j]);
            }
#pragma unroll
            for (int j = i - N + 1; j < N; j++) mac64_vs(acc, m[j], P::MOD[i - j]);
            r[i - N] = i == 2 * N - 1 ? (uint32_t)acc : ((uint32_t)acc & MASK);
            acc >>= LB;
        }
        // result < 2p: one conditional subtraction
        uint32_t u[N];
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t v = (int32_t)r[i] - (int32_t)P::MOD[i] + cy;
            cy = v >> LB;
            u[i] = (uint32_t)v & MASK;
        }
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = cy < 0 ? r[i] : u[i];
    }
    PTAU_HD static Fp29 mul_inl(const Fp29 &a, const Fp29 &b) {
        Fp29 r;
        ps_mul<false>(a.l, b.l, r.l);
        return r;
    }
    PTAU_HD static Fp29 sqr_inl(const Fp29 &a) {
        Fp29 r;
        ps_mul<true>(a.l, a.l, r.l);
        return r;
    }
    // one out-of-line copy per field: register ABI (vector arguments in VGPRs) up to 13 limbs, by reference beyond
    static constexpr bool MUL_OUT_OF_LINE = (N > PTAU_VEC_ABI_MAX_LIMBS);
    static constexpr bool POINT_OPS_INLINE = (N <= 13);
    typedef uint32_t VecN __attribute__((ext_vector_type(N)));
    __host__ __device__ __noinline__ static VecN mul_vec(VecN a, VecN b) {
        uint32_t x[N], y[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) { x[i] = a[i]; y[i] = b[i]; }
        ps_mul<false>(x, y, r);
        VecN o;
#pragma unroll
        for (int i = 0; i < N; i++) o[i] = r[i];
        return o;
    }
    __host__ __device__ __noinline__ static VecN sqr_vec(VecN a) {
        uint32_t x[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = a[i];
        ps_mul<true>(x, x, r);
        VecN o;
#pragma unroll
        for (int i = 0; i < N; i++) o[i] = r[i];
        return o;
    }
    // k squarings in a row in ONE call (square roots: runs of squarings between the window products of the exponentiation, the
    // 37-step chain of the torsion stage): the limbs stay in registers between squarings — no argument marshalling, call,
    // entry wait and return per squaring (~40 of a squaring's ~450 instructions)
    __host__ __device__ __noinline__ static VecN sqr_n_vec(VecN a, int k) {
        uint32_t x[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = a[i];
#pragma nounroll
        for (int t = 0; t < k; t++) {
            ps_mul<true>(x, x, r);
#pragma unroll
            for (int i = 0; i < N; i++) x[i] = r[i];
        }
        VecN o;
#pragma unroll
        for (int i = 0; i < N; i++) o[i] = x[i];
        return o;
    }
    __host__ __device__ __noinline__ static Fp29 mul_out(const Fp29 &a, const Fp29 &b) { return mul_inl(a, b); }
    __host__ __device__ __noinline__ static Fp29 sqr_out(const Fp29 &a) { return sqr_inl(a); }
    // -DPTAU_INLINE_MUL13 inlines the 13-limb product at every call site: measured +3.5 % on K1 (32.3 vs 31.2 M updates/s)
    // for a 10-minute compile of the BLS12-377 translation unit and a 2.5x larger code object — off.
    PTAU_HD static Fp29 mul(const Fp29 &a, const Fp29 &b) {
#ifdef PTAU_INLINE_MUL13
        if constexpr (!MUL_OUT_OF_LINE) return mul_inl(a, b);
#endif
        if constexpr (MUL_OUT_OF_LINE) {
            return mul_out(a, b);
        } else {
            VecN x, y;
#pragma unroll
            for (int i = 0; i < N; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
            VecN o = mul_vec(x, y);
            Fp29 r;
#pragma unroll
            for (int i = 0; i < N; i++) r.l[i] = o[i];
            return r;
        }
    }
    PTAU_HD static Fp29 sqr(const Fp29 &a) {
#ifdef PTAU_INLINE_MUL13
        if constexpr (!MUL_OUT_OF_LINE) return sqr_inl(a);
#endif
        if constexpr (MUL_OUT_OF_LINE) {
            return sqr_out(a);
        } else {
            VecN x;
#pragma unroll
            for (int i = 0; i < N; i++) x[i] = a.l[i];
            VecN o = sqr_vec(x);
            Fp29 r;
#pragma unroll
            for (int i = 0; i < N; i++) r.l[i] = o[i];
            return r;
        }
    }

    template <bool ENABLE = !MUL_OUT_OF_LINE>
    PTAU_HD static std::enable_if_t<ENABLE, Fp29> sqr_n(const Fp29 &a, int k) {   // a^(2^k); only where the register ABI serves the field
        VecN x;
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = a.l[i];
        const VecN o = sqr_n_vec(x, k);
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = o[i];
        return r;
    }

    // k * a for a small positive constant (4 <= k <= 7 when p fills its top limb: k p's top limb must fit 32 bits; up to 15
    // otherwise): the integer k a, a quotient estimate from the top limb, k a - q p, conditional subtractions — instead of a
    // double-and-add chain of modular additions (~117 instructions each).  The Fq2 non-residue multiples (x 5, x 4 on
    // BLS12-377) sit in every Fq2 product and square.
    PTAU_HD static Fp29 mul_small_fast(const Fp29 &a, unsigned k) {
        constexpr int TL = (P::BITS - 1) / LB;
        uint32_t v[N];
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const uint64_t acc = (uint64_t)a.l[j] * k + c;
            v[j] = (uint32_t)acc & MASK;
            c = acc >> LB;
        }
        const uint64_t tv = (uint64_t)v[TL] | ((uint64_t)(TL + 1 < N ? v[TL + 1 < N ? TL + 1 : N - 1] : (uint32_t)c) << LB);
        constexpr uint64_t step = (uint64_t)P::MOD[TL] + 1u;
        uint32_t q = 0;
#pragma unroll
        for (unsigned i = 1; i < 16; i++)
            if (i < k) q += tv >= i * step ? 1u : 0u;
        uint32_t w[N];
        int64_t cy = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int64_t acc = (int64_t)v[j] - (int64_t)((uint64_t)q * P::MOD[j]) + cy;
            w[j] = (uint32_t)acc & MASK;
            cy = acc >> LB;
        }
        // below 3p (2p when p's top limb has more than a few bits): two conditional subtractions.  When p fills its limbs
        // (BLS12-377: 13 x 29 = 377 bits) 2p does not fit the digits: `extra` is the bit above them, cleared by the first pass.
        int32_t extra = (int32_t)c + (int32_t)cy;
        Fp29 r;
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            uint32_t d[N];
            int32_t b = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                const int32_t t = (int32_t)w[j] - (int32_t)P::MOD[j] + b;
                b = t >> LB;
                d[j] = (uint32_t)t & MASK;
            }
            const bool keep = extra + b < 0;
#pragma unroll
            for (int j = 0; j < N; j++) w[j] = keep ? w[j] : d[j];
            extra = keep ? extra : extra + b;
        }
#pragma unroll
        for (int j = 0; j < N; j++) r.l[j] = w[j];
        return r;
    }
    PTAU_HD static Fp29 mul_small(const Fp29 &a, int k) {
        if (k == 0) return zero();
        bool ng = k < 0;
        unsigned uk = ng ? (unsigned)(-k) : (unsigned)k;
        if (uk >= 16 && uk < 32) {
            Fp29 t = dbl(mul_small_fast(a, uk / 2));
            if (uk & 1u) t = add(t, a);
            return ng ? neg(t) : t;
        }
        if (uk >= 4 && uk < 16) {
            const Fp29 t = mul_small_fast(a, uk);
            return ng ? neg(t) : t;
        }
        Fp29 acc = zero(), base = a;
        bool first = true;
        while (uk) {
            if (uk & 1u) { acc = first ? base : add(acc, base); first = false; }
            uk >>= 1;
            if (uk) base = dbl(base);
        }
        return ng ? neg(acc) : acc;
    }

    // canonical value as N limbs of 29 bits
    PTAU_HD static Fp29 from_canonical(const uint32_t *c) {
        Fp29 t, r2;
#pragma unroll
        for (int i = 0; i < N; i++) { t.l[i] = c[i]; r2.l[i] = P::R2[i]; }
        return mul(t, r2);
    }
    PTAU_HD void to_canonical(uint32_t *c) const {
        Fp29 o;
#pragma unroll
        for (int i = 0; i < N; i++) o.l[i] = i == 0 ? 1u : 0u;
        Fp29 r = mul(*this, o);
#pragma unroll
        for (int i = 0; i < N; i++) c[i] = r.l[i];
    }
    // 29-bit limbs <-> packed 32-bit words of the same integer (compile-time bit positions after unrolling)
    PTAU_HD static void pack32(const uint32_t (&c)[N], uint32_t (&w)[N32]) {
        uint64_t acc = 0;
        int bits = 0, j = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            acc |= (uint64_t)c[i] << bits;
            bits += LB;
            if (bits >= 32) {
                if (j < N32) w[j] = (uint32_t)acc;
                j++;
                acc >>= 32;
                bits -= 32;
            }
        }
        if (j < N32) w[j] = (uint32_t)acc;
    }
    PTAU_HD static void unpack32(const uint32_t (&w)[N32], uint32_t (&c)[N]) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = LB * i, li = bit >> 5, sh = bit & 31;
            const uint64_t lo = li < N32 ? w[li] : 0u, hi = li + 1 < N32 ? w[li + 1] : 0u;
            c[i] = (uint32_t)(((hi << 32) | lo) >> sh) & MASK;
        }
    }

    __host__ __device__ __noinline__ static Fp29 pow_limbs(const Fp29 &a, const uint32_t *e, int top_bit) {
        return pow_sliding(a, e, top_bit);
    }
    // Inversion runs on the packed twin (Kaliski's binary almost-inverse, fp.hpp): for the raw integer v it returns
    // v^-1 2^(64 N32); with v = a R that is (a^-1 R) 2^(64 N32 - 58 N), and the Montgomery product with the constant
    // INV_FIX = 2^(58 N - 64 N32) R removes the excess power of two.
    __host__ __device__ __noinline__ static Fp29 inverse(const Fp29 &a) {
        if (a.is_zero()) return zero();
        Twin t;
        pack32(a.l, t.l);
        const Twin ti = Twin::inverse(t);
        Fp29 r, fix;
        unpack32(ti.l, r.l);
#pragma unroll
        for (int i = 0; i < N; i++) fix.l[i] = P::INV_FIX[i];
        return mul(r, fix);
    }

    PTAU_HD static bool canonical_gt_half(const uint32_t *c) {   // c: 29-bit limbs
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) cy = ((int32_t)P::HALF[i] - (int32_t)c[i] + cy) >> LB;
        return cy < 0;
    }

    // Pohlig-Hellman in the 2-Sylow subgroup, shared by the square roots below.  b = h^d (h = g^2, d < 2^E, E = S - 1) when the
    // root exists; d is found W bits at a time from the bottom and gd = g^(-d) returned.  Window j needs
    // (b h^(-y_j))^(2^sh_j), y_j = the digits found so far, sh_j = E - W j - w_j.  The textbook form strips every digit from b
    // and squares the result sh_j times again: sum sh_j = 105 squarings for BLS12-377's S = 46.  Here ONE chain of squarings
    // of the ORIGINAL b keeps the L powers c_j = b^(2^sh_j) (sh_0 = 37 squarings in all), and the correction
    // h^(-y_j 2^sh_j) = g^(-(y_j << (sh_j + 1))) is a power of g with a KNOWN exponent: the product of the byte-aligned table
    // entries SQRT_GNEG[q][byte q of the exponent] — j + 1 lookups for window j.  37 squarings + 20 products instead of
    // 105 + 18 (round 3: 123 of a square root's ~575 products were this stage).
    // Returns 0, or 1 when the FIRST window already fails (b has full order 2^S: the operand is a non-residue), 2 when a later
    // one does (not a power of h at all: cannot happen for field elements, kept as a guard).
    PTAU_HD static int torsion_log(const Fp29 &b, Fp29 &gd) {
        constexpr int E = P::TS_S - 1, W = P::SQRT_W, L = P::SQRT_L;
        static_assert(E + 1 <= 63, "the exponent arithmetic below is 64-bit");
        if constexpr (W == 0) {
            gd = one();
            return 0;
        } else {
            auto wj_of = [](int j) { return (E - W * j) < W ? (E - W * j) : W; };
            auto table = [](int q, uint32_t k) {
                Fp29 f;
#pragma unroll
                for (int i = 0; i < N; i++) f.l[i] = P::SQRT_GNEG[((uint32_t)q << W) + k][i];
                return f;
            };
            Fp29 c[L];
            {
                Fp29 t = b;
                int s = 0;
#pragma unroll
                for (int j = L - 1; j >= 0; j--) {
                    const int sh = E - W * j - wj_of(j);
                    t = sqr_times(t, sh - s);
                    s = sh;
                    c[j] = t;
                }
            }
            uint64_t y = 0;
#pragma unroll
            for (int j = 0; j < L; j++) {
                const int wj = wj_of(j), sh = E - W * j - wj;
                Fp29 t = c[j];
                if (j > 0) {
                    const uint64_t e = y << (sh + 1);              // < 2^E: bits [sh + 1, sh + 1 + W j)
                    const int q0 = (sh + 1) / W, q1 = (sh + W * j) / W;
                    Fp29 f = table(q0, (uint32_t)(e >> (W * q0)) & ((1u << W) - 1u));
#pragma unroll
                    for (int q = q0 + 1; q <= q1; q++) f = mul(f, table(q, (uint32_t)(e >> (W * q)) & ((1u << W) - 1u)));
                    t = mul(t, f);
                }
                const uint32_t lo = t.l[0];
                int K = -1;
                for (int k = 0; k < (1 << W); k++) K = (P::SQRT_ZLOW[k] == lo) ? k : K;
                if (K < 0) return j == 0 ? 1 : 2;
                y |= (uint64_t)(K >> (W - wj)) << (W * j);
            }
            gd = table(0, (uint32_t)y & ((1u << W) - 1u));
#pragma unroll
            for (int q = 1; q < L; q++) gd = mul(gd, table(q, (uint32_t)(y >> (W * q)) & ((1u << W) - 1u)));
            return 0;
        }
    }

    // Square root: see Fp::sqrt_ex (one exponentiation + Pohlig-Hellman in the 2-Sylow subgroup, tables in P).
    __host__ __device__ __noinline__ static bool sqrt_ex(const Fp29 &a, Fp29 &out, Fp29 *inv_out) {
        if (a.is_zero()) {
            out = zero();
            if (inv_out) *inv_out = zero();
            return true;
        }
        const Fp29 w = pow_limbs(a, P::TS_TM1H, P::TS_TM1H_BITS - 1);
        const Fp29 x = mul(a, w);
        Fp29 gd;
        if (torsion_log(mul(x, w), gd)) return false;
        out = mul(x, gd);
        if (sqr(out) != a) return false;
        if (inv_out) *inv_out = mul(out, sqr(mul(w, gd)));
        return true;
    }
    // For the square roots of Fq[u]/(u^2 - nr) (ext.hpp): out = sqrt(a), residue = true — or, when a is NOT a square,
    // out = sqrt(nr a), residue = false (nr = P::SQRT_NR, the tower's non-residue: nr a is a square then).  inv_out = 1/out.
    // One exponentiation serves both cases: (nr a)^((T-1)/2) = SQRT_NR_C a^((T-1)/2); a non-residue shows at the FIRST window of
    // the Pohlig-Hellman stage (a^T has full order 2^S, its 2^(E-W)-th power is no 2^W-th root of unity).
    static constexpr bool HAS_SQRT_OR_NR = P::SQRT_NR != 0 && P::SQRT_W > 0;
    __host__ __device__ __noinline__ static bool sqrt_or_nr(const Fp29 &a, Fp29 &out, Fp29 &inv_out, bool &residue) {
        residue = true;
        if (a.is_zero()) {
            out = zero();
            inv_out = zero();
            return true;
        }
        Fp29 w = pow_limbs(a, P::TS_TM1H, P::TS_TM1H_BITS - 1);
        Fp29 aa = a;
        for (int attempt = 0; attempt < 2; attempt++) {
            const Fp29 x = mul(aa, w);
            Fp29 gd;
            if (torsion_log(mul(x, w), gd) == 0) {
                out = mul(x, gd);
                if (sqr(out) != aa) return false;
                inv_out = mul(out, sqr(mul(w, gd)));
                return true;
            }
            if (attempt) return false;
            residue = false;
            aa = mul_small(a, P::SQRT_NR);
            w = mul(w, load_const(P::SQRT_NR_C));
        }
        return false;
    }
    PTAU_HD static bool sqrt(const Fp29 &a, Fp29 &out) { return sqrt_ex(a, out, nullptr); }

    // ---- canonical little-endian bytes (arkworks serialize) ----
    PTAU_HD static bool read_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, Fp29 &out) {
        uint32_t w[N32];
#pragma unroll
        for (int i = 0; i < N32; i++) w[i] = 0;
        for (int i = 0; i < BYTES; i++) {
            uint32_t byte = in[i];
            if (i == BYTES - 1 && with_flags) {
                *flags = (uint8_t)(byte & 0xC0u);
                byte &= 0x3Fu;
            }
            w[i >> 2] |= byte << (8 * (i & 3));
        }
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N32; i++) (void)subb(w[i], P::MOD32[i], br);
        uint32_t c[N];
        unpack32(w, c);
        out = from_canonical(c);
        return br != 0;   // value < p
    }
    PTAU_HD void write_bytes(uint8_t *out, uint8_t flags) const {
        uint32_t c[N], w[N32];
        to_canonical(c);
        pack32(c, w);
#pragma unroll
        for (int i = 0; i < BYTES; i++) {
            uint8_t byte = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
            if (i == BYTES - 1) byte |= flags;
            out[i] = byte;
        }
    }
    PTAU_HD int sign_class() const {
        if (is_zero()) return 0;
        uint32_t c[N];
        to_canonical(c);
        return canonical_gt_half(c) ? 1 : -1;
    }
};

// The field class a parameter struct was generated for.
template <class P, int LBITS = P::LB>
struct FieldOfT { using type = Fp<P>; };
template <class P>
struct FieldOfT<P, 29> { using type = Fp29<P>; };
template <class P>
using FieldOf = typename FieldOfT<P>::type;

}  // namespace ptau
