// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (CDNA4) VALU.
//
// MI355X-native equivalent of ark-ff 0.4.2 `Fp<MontBackend<_, N>>`, which the reference reaches
// through setup-utils' batch_exp (call site /root/reference/src/bin/contribute.rs:811-823).
// The hot instruction is v_mad_u64_u32 (32x32+64 -> 64); limbs live in VGPRs, every loop is
// unrolled at compile time so no limb array is ever runtime-indexed (no scratch).
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include <cstdint>

#include "curve_constants.inc"

#define PTAU_HD __host__ __device__ __forceinline__

namespace ptau {

PTAU_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t &carry) {
    uint64_t t = (uint64_t)a + b + carry;
    carry = (uint32_t)(t >> 32);
    return (uint32_t)t;
}
PTAU_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t &borrow) {
    uint64_t t = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(t >> 32) & 1u;
    return (uint32_t)t;
}

// 96-bit column accumulator of the product-scanning multiplier
struct Acc96 {
    uint64_t lo;
    uint32_t hi;
};
// acc += x * y   (both in VGPRs)
PTAU_HD void mac_vv(Acc96 &a, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(a.lo), "+v"(a.hi) : "v"(x), "v"(y) : "vcc");
#else
    unsigned __int128 t = (unsigned __int128)x * y + a.lo;
    a.lo = (uint64_t)t;
    a.hi += (uint32_t)(t >> 64);
#endif
}
// acc += x * c   (c a compile-time constant: lives in an SGPR, one constant-bus read)
PTAU_HD void mac_vs(Acc96 &a, uint32_t x, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(a.lo), "+v"(a.hi) : "v"(x), "s"(c) : "vcc");
#else
    mac_vv(a, x, c);
#endif
}
// acc += x*y + z*c in ONE asm statement (halves the compiler's post-asm s_nop padding)
PTAU_HD void mac_pair(Acc96 &a, uint32_t x, uint32_t y, uint32_t z, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(a.lo), "+v"(a.hi) : "v"(x), "v"(y), "v"(z), "s"(c) : "vcc");
#else
    mac_vv(a, x, y);
    mac_vv(a, z, c);
#endif
}
PTAU_HD void acc_add32(Acc96 &a, uint32_t x) {
    uint64_t t = a.lo + x;
    a.hi += t < a.lo ? 1u : 0u;
    a.lo = t;
}
PTAU_HD void acc_shift(Acc96 &a) {
    a.lo = (a.lo >> 32) | ((uint64_t)a.hi << 32);
    a.hi = 0;
}

// a^e for a PUBLIC exponent (wave-uniform control flow): sliding 4-bit windows over the odd powers a, a^3, ..., a^15, kept
// in a local array (private memory; the index is the same in every lane) — about top_bit / 5 + 8 products instead of one
// per set bit (the square roots' exponents are dense: 377- and 753-bit exponents cost ~85 / ~160 products instead of
// ~165 / ~376).
// k squarings in a row: ONE call where the field has a chained form (Fp29::sqr_n: the limbs stay in the callee's registers
// between squarings, no marshalling / call / wait per squaring), a loop of squarings elsewhere
template <class F, class = void>
struct HasSqrN : std::false_type {};
template <class F>
struct HasSqrN<F, std::void_t<decltype(F::sqr_n(std::declval<const F &>(), 0))>> : std::true_type {};
template <class F>
PTAU_HD F sqr_times(const F &a, int k) {
    if constexpr (HasSqrN<F>::value) {
        return k > 0 ? F::sqr_n(a, k) : a;
    } else {
        F r = a;
        for (int i = 0; i < k; i++) r = F::sqr(r);
        return r;
    }
}

template <class F>
PTAU_HD F pow_sliding(const F &a, const uint32_t *e, int top_bit) {
    F tab[8];
    tab[0] = a;
    const F a2 = F::sqr(a);
    for (int i = 1; i < 8; i++) tab[i] = F::mul(tab[i - 1], a2);
    auto bit = [&](int i) -> uint32_t { return (e[i >> 5] >> (i & 31)) & 1u; };
    F r = F::one();
    bool started = false;
    int i = top_bit, pend = 0;   // pend: squarings owed to r (zero bits and the window's width), paid in one chained call
    while (i >= 0) {
        if (!bit(i)) {
            if (started) pend++;
            i--;
            continue;
        }
        int j = i >= 3 ? i - 3 : 0;
        while (!bit(j)) j++;   // the window ends on its lowest set bit
        uint32_t val = 0;
        for (int k = i; k >= j; k--) val = (val << 1) | bit(k);
        if (started) {
            r = sqr_times(r, pend + (i - j + 1));
            pend = 0;
            r = F::mul(r, tab[val >> 1]);
        } else {
            r = tab[val >> 1];
            started = true;
        }
        i = j - 1;
    }
    return sqr_times(r, pend);
}

template <class P>
struct Fp {
    static constexpr int N = P::N;
    static constexpr int DEG = 1;
    static constexpr int LANES = 1;          // lanes of a wavefront that hold one element
    static constexpr int BYTES = P::BYTES;   // serialized size of one element
    using Base = Fp<P>;
    using Params = P;
    uint32_t l[N];

    PTAU_HD static Fp load_const(const uint32_t *c) {   // a constant given as N limbs
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = c[i];
        return r;
    }
    PTAU_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    PTAU_HD static Fp one() {
        Fp r;
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
    PTAU_HD bool operator==(const Fp &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    PTAU_HD bool operator!=(const Fp &b) const { return !(*this == b); }

    // r = a + b mod p   (a, b < p < 2^(32N-1): the sum never carries out of N limbs)
    PTAU_HD static Fp add(const Fp &a, const Fp &b) {
        Fp t, u;
        uint32_t c = 0, br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t.l[i] = addc(a.l[i], b.l[i], c);
#pragma unroll
        for (int i = 0; i < N; i++) u.l[i] = subb(t.l[i], P::MOD[i], br);
#pragma unroll
        for (int i = 0; i < N; i++) t.l[i] = br ? t.l[i] : u.l[i];
        return t;
    }
    PTAU_HD static Fp sub(const Fp &a, const Fp &b) {
        Fp t;
        uint32_t br = 0, c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t.l[i] = subb(a.l[i], b.l[i], br);
        uint32_t mask = 0u - br;
#pragma unroll
        for (int i = 0; i < N; i++) t.l[i] = addc(t.l[i], P::MOD[i] & mask, c);
        return t;
    }
    PTAU_HD static Fp neg(const Fp &a) {
        Fp t;
        uint32_t br = 0;
        uint32_t nz = a.is_zero() ? 0u : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < N; i++) t.l[i] = subb(P::MOD[i] & nz, a.l[i], br);
        return t;
    }
    PTAU_HD static Fp dbl(const Fp &a) { return add(a, a); }

    // Montgomery product a*b/R mod p, product-scanning (FIPS) form: one 96-bit column accumulator,
    // every limb product is ONE v_mad_u64_u32 (carry-out in VCC) + ONE v_addc_co_u32.  Measured on
    // MI355X (tools/ubench_mul.hip) this runs at the v_mad_u64_u32 pipe peak (~17 T MAC/s chip-wide,
    // 60 G 377-bit products/s), 1.58x the operand-scanning CIOS form it replaced.
    // The triangular index pattern needs full unrolling (4.6 KB of MAC/ADDC ISA for N = 12), so the
    // body lives in ONE out-of-line device function per field:
    //   N <= 12 : operands/results are ext_vector_type(N) values -> passed in VGPRs (v0..v23 / v0..v11)
    //   N  = 24 : by reference through private memory (48 dwords exceed the 32 argument VGPRs)
    PTAU_HD static void fips_mul(const uint32_t (&a)[N], const uint32_t (&b)[N], uint32_t (&out)[N]) {
        uint32_t m[N], r[N];
        Acc96 acc{0, 0};
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < i; j++) mac_pair(acc, a[j], b[i - j], m[j], P::MOD[i - j]);
            mac_vv(acc, a[i], b[0]);
            m[i] = (uint32_t)acc.lo * P::INV;
            mac_vs(acc, m[i], P::MOD[0]);
            acc_shift(acc);
        }
#pragma unroll
        for (int i = N; i < 2 * N; i++) {
#pragma unroll
            for (int j = i - N + 1; j < N; j++) mac_pair(acc, a[j], b[i - j], m[j], P::MOD[i - j]);
            r[i - N] = (uint32_t)acc.lo;
            acc_shift(acc);
        }
        // p < 2^(32N-1) keeps the result below 2p: one conditional subtraction
        uint32_t u[N];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = subb(r[i], P::MOD[i], br);
        const bool keep = br && (uint32_t)acc.lo == 0;
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = keep ? r[i] : u[i];
    }
    // Montgomery square: the cross products a_j a_k (j < k) are taken once against d = 2a (2a < 2^(32N)
    // for every supported modulus), so a square issues N(N+1)/2 + N^2 MACs instead of 2 N^2.
    // d is 2a as ONE integer (limb j carries bit 31 of limb j-1 = c_{j-1}), hence
    //   2 sum_{j<k} a_j a_k B^(j+k) = sum_{j<k} d_j a_k B^(j+k) + sum_k c_{k-1} a_k B^(2k):
    // the diagonal column 2k takes a_k^2 plus a_k when bit 31 of a_{k-1} is set.
    PTAU_HD static void fips_sqr(const uint32_t (&a)[N], uint32_t (&out)[N]) {
        static_assert(P::BITS + 1 <= 32 * N, "2a must fit N limbs");
        uint32_t d[N], m[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = (a[i] << 1) | (i ? (a[i - 1] >> 31) : 0u);
        Acc96 acc{0, 0};
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < i; j++) {
                if (2 * j < i) mac_pair(acc, d[j], a[i - j], m[j], P::MOD[i - j]);
                else mac_vs(acc, m[j], P::MOD[i - j]);
            }
            if ((i & 1) == 0) {
                mac_vv(acc, a[i / 2], a[i / 2]);
                if (i) acc_add32(acc, (0u - (a[i / 2 - 1] >> 31)) & a[i / 2]);
            }
            m[i] = (uint32_t)acc.lo * P::INV;
            mac_vs(acc, m[i], P::MOD[0]);
            acc_shift(acc);
        }
#pragma unroll
        for (int i = N; i < 2 * N; i++) {
#pragma unroll
            for (int j = i - N + 1; j < N; j++) {
                if (2 * j < i) mac_pair(acc, d[j], a[i - j], m[j], P::MOD[i - j]);
                else mac_vs(acc, m[j], P::MOD[i - j]);
            }
            if ((i & 1) == 0 && i / 2 < N) {
                mac_vv(acc, a[i / 2], a[i / 2]);
                acc_add32(acc, (0u - (a[i / 2 - 1] >> 31)) & a[i / 2]);
            }
            r[i - N] = (uint32_t)acc.lo;
            acc_shift(acc);
        }
        uint32_t u[N];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = subb(r[i], P::MOD[i], br);
        const bool keep = br && (uint32_t)acc.lo == 0;
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = keep ? r[i] : u[i];
    }
    PTAU_HD static Fp sqr_inl(const Fp &a) {
        Fp r;
        fips_sqr(a.l, r.l);
        return r;
    }
    PTAU_HD static Fp mul_inl(const Fp &a, const Fp &b) {
        Fp r;
        fips_mul(a.l, b.l, r.l);
        return r;
    }
// Products of up to this many limbs use the register ABI (vector arguments in VGPRs), wider ones go by reference.
// 27 (everything in registers: 32 argument VGPRs, the rest of the second operand on the stack) was measured on MI355X:
// it cuts the point operations' scratch from 1.4 KB to 88 B but their callers then take AGPRs and run one wave per SIMD
// — BW6-761 3.2 -> 2.9 M updates/s, MNT4 G2 218 k -> 120 k at 2^18; MNT6 +14 % at 2^16.  13 stays.
#ifndef PTAU_VEC_ABI_MAX_LIMBS
#define PTAU_VEC_ABI_MAX_LIMBS 13
#endif
    static constexpr bool MUL_OUT_OF_LINE = (N > PTAU_VEC_ABI_MAX_LIMBS);          // by-reference (memory) convention
    static constexpr bool POINT_OPS_INLINE = (N <= 13);   // a Jacobian point operation over this field inlines at <= 256 registers
    typedef uint32_t VecN __attribute__((ext_vector_type(N)));
    __host__ __device__ __noinline__ static VecN mul_vec(VecN a, VecN b) {
        uint32_t x[N], y[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) { x[i] = a[i]; y[i] = b[i]; }
        fips_mul(x, y, r);
        VecN o;
#pragma unroll
        for (int i = 0; i < N; i++) o[i] = r[i];
        return o;
    }
    __host__ __device__ __noinline__ static Fp mul_out(const Fp &a, const Fp &b) { return mul_inl(a, b); }
    PTAU_HD static Fp mul(const Fp &a, const Fp &b) {
        if constexpr (MUL_OUT_OF_LINE) {
            return mul_out(a, b);
        } else {
            VecN x, y;
#pragma unroll
            for (int i = 0; i < N; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
            VecN o = mul_vec(x, y);
            Fp r;
#pragma unroll
            for (int i = 0; i < N; i++) r.l[i] = o[i];
            return r;
        }
    }
    __host__ __device__ __noinline__ static VecN sqr_vec(VecN a) {
        uint32_t x[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = a[i];
        fips_sqr(x, r);
        VecN o;
#pragma unroll
        for (int i = 0; i < N; i++) o[i] = r[i];
        return o;
    }
    __host__ __device__ __noinline__ static Fp sqr_out(const Fp &a) { return sqr_inl(a); }
    PTAU_HD static Fp sqr(const Fp &a) {
        if constexpr (MUL_OUT_OF_LINE) {
            return sqr_out(a);
        } else {
            VecN x;
#pragma unroll
            for (int i = 0; i < N; i++) x[i] = a.l[i];
            VecN o = sqr_vec(x);
            Fp r;
#pragma unroll
            for (int i = 0; i < N; i++) r.l[i] = o[i];
            return r;
        }
    }

    // small signed constant multiple via double-and-add (used for extension-field non-residues)
    PTAU_HD static Fp mul_small(const Fp &a, int k) {
        if (k == 0) return zero();
        bool ng = k < 0;
        unsigned uk = ng ? (unsigned)(-k) : (unsigned)k;
        Fp acc = zero(), base = a;
        bool first = true;
        while (uk) {
            if (uk & 1u) { acc = first ? base : add(acc, base); first = false; }
            uk >>= 1;
            if (uk) base = dbl(base);
        }
        return ng ? neg(acc) : acc;
    }

    PTAU_HD static Fp from_canonical(const uint32_t *c) {
        Fp t, r2;
#pragma unroll
        for (int i = 0; i < N; i++) { t.l[i] = c[i]; r2.l[i] = P::R2[i]; }
        return mul(t, r2);
    }
    PTAU_HD void to_canonical(uint32_t *c) const {
        Fp o;
#pragma unroll
        for (int i = 0; i < N; i++) o.l[i] = i == 0 ? 1u : 0u;
        Fp r = mul(*this, o);
#pragma unroll
        for (int i = 0; i < N; i++) c[i] = r.l[i];
    }

    // a^e for a public exponent given as N 32-bit limbs (MSB-first square-and-multiply)
    __host__ __device__ __noinline__ static Fp pow_limbs(const Fp &a, const uint32_t *e, int top_bit) {
        return pow_sliding(a, e, top_bit);
    }
    PTAU_HD static Fp inverse_fermat(const Fp &a) { return pow_limbs(a, P::PM2, P::BITS - 1); }
    // Montgomery inverse by the binary extended Euclid (Kaliski): phase 1 yields a^-1 2^k mod p for the k <= 2 BITS
    // halving steps it took; 64 N - k modular doublings then land on a^-1 2^(64 N) R^-1 = a^-1 R for a Montgomery
    // input a R.  64 N cheap shift/add steps (~40 k VALU instructions for 12 limbs, ~150 k for 24) against a
    // (p - 2)-exponentiation of ~570 (1140) field products — what batch normalisation pays once per group of points.
    __host__ __device__ __noinline__ static Fp inverse(const Fp &a) {
        if (a.is_zero()) return zero();
        uint32_t u[N], v[N], r[N], s[N];
#pragma unroll
        for (int i = 0; i < N; i++) { u[i] = P::MOD[i]; v[i] = a.l[i]; r[i] = 0; s[i] = i == 0 ? 1u : 0u; }
        auto shr1 = [](uint32_t (&x)[N]) {
#pragma unroll
            for (int i = 0; i < N; i++) x[i] = (x[i] >> 1) | (i + 1 < N ? (x[i + 1] << 31) : 0u);
        };
        auto shl1 = [](uint32_t (&x)[N]) {
#pragma unroll
            for (int i = N - 1; i >= 0; i--) x[i] = (x[i] << 1) | (i ? (x[i - 1] >> 31) : 0u);
        };
        int k = 0;
        for (;;) {
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < N; i++) nz |= v[i];
            if (!nz) break;
            if (!(u[0] & 1u)) {
                shr1(u);
                shl1(s);
            } else if (!(v[0] & 1u)) {
                shr1(v);
                shl1(r);
            } else {
                uint32_t d[N], br = 0;
#pragma unroll
                for (int i = 0; i < N; i++) d[i] = subb(v[i], u[i], br);
                if (br) {    // u > v: u = (u - v)/2, r += s, s *= 2
                    uint32_t b2 = 0;
#pragma unroll
                    for (int i = 0; i < N; i++) u[i] = subb(u[i], v[i], b2);
                    shr1(u);
                    uint32_t c = 0;
#pragma unroll
                    for (int i = 0; i < N; i++) r[i] = addc(r[i], s[i], c);
                    shl1(s);
                } else {     // v >= u: v = (v - u)/2, s += r, r *= 2   (v == u ends the loop with v = 0)
#pragma unroll
                    for (int i = 0; i < N; i++) v[i] = d[i];
                    shr1(v);
                    uint32_t c = 0;
#pragma unroll
                    for (int i = 0; i < N; i++) s[i] = addc(s[i], r[i], c);
                    shl1(r);
                }
            }
            k++;
        }
        // r < 2p: reduce, negate (the invariant gives -a^-1 2^k), then scale by 2^(64N - k)
        Fp t;
        {
            uint32_t w[N], br = 0;
#pragma unroll
            for (int i = 0; i < N; i++) w[i] = subb(r[i], P::MOD[i], br);
#pragma unroll
            for (int i = 0; i < N; i++) t.l[i] = br ? r[i] : w[i];
        }
        t = neg(t);
        for (; k < 64 * N; k++) t = dbl(t);
        return t;
    }

    // canonical value > (p-1)/2 ?   (arkworks: y "negative" iff y > -y)
    PTAU_HD static bool canonical_gt_half(const uint32_t *c) {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) (void)subb(P::HALF[i], c[i], br);
        return br != 0;
    }
    PTAU_HD static bool canonical_lt_mod(const uint32_t *c) {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) (void)subb(c[i], P::MOD[i], br);
        return br != 0;
    }

    // Square root; returns false when a is a non-residue.  p - 1 = 2^S T:
    //   w = a^((T-1)/2) (the one exponentiation), x = a w = a^((T+1)/2), b = x w = a^T, so x^2 = a b and b lies in
    //   the 2^S-torsion <g>.  For a residue b = (g^2)^d, d < 2^(S-1); d is recovered W bits at a time from the
    //   bottom (Pohlig-Hellman: raise to 2^(E-Wj-w_j), look the order-2^W element up, strip the window with a
    //   tabulated power of g) and sqrt(a) = x g^(-d).  About S^2/(2W) squarings with wave-uniform control flow,
    //   against ~S^2/4 and heavy divergence for the Tonelli-Shanks loop (S = 46 for BLS12-377 Fq).
    // inv_out (optional) receives 1/sqrt(a) = sqrt(a) (w g^(-d))^2 for three more products: a^(-1) = a^(T-1) b^(-1).
    __host__ __device__ __noinline__ static bool sqrt_ex(const Fp &a, Fp &out, Fp *inv_out) {
        if (a.is_zero()) {
            out = zero();
            if (inv_out) *inv_out = zero();
            return true;
        }
        const Fp w = pow_limbs(a, P::TS_TM1H, P::TS_TM1H_BITS - 1);
        const Fp x = mul(a, w);
        Fp b = mul(x, w);
        Fp gd = one();
        constexpr int E = P::TS_S - 1, W = P::SQRT_W;
        for (int j = 0; j < P::SQRT_L; j++) {
            const int wj = (E - W * j) < W ? (E - W * j) : W;
            const int sh = E - W * j - wj;
            Fp t = b;
            for (int k = 0; k < sh; k++) t = sqr(t);
            const uint32_t lo = t.l[0];
            int K = -1;
            for (int k = 0; k < (1 << W); k++) K = (P::SQRT_ZLOW[k] == lo) ? k : K;   // uniform table, scalar loads
            if (K < 0) return false;   // a^T outside <g^2>: non-residue
            const int d = K >> (W - wj);
            Fp f;
#pragma unroll
            for (int i = 0; i < N; i++) f.l[i] = P::SQRT_GNEG[(j << W) + d][i];
            gd = mul(gd, f);
            b = mul(b, sqr(f));
        }
        out = mul(x, gd);
        if (sqr(out) != a) return false;
        if (inv_out) *inv_out = mul(out, sqr(mul(w, gd)));
        return true;
    }
    PTAU_HD static bool sqrt(const Fp &a, Fp &out) { return sqrt_ex(a, out, nullptr); }

    // ---- canonical little-endian bytes (arkworks serialize) ----
    // Reads BYTES bytes; *flags receives the top two bits of the last byte when with_flags.
    PTAU_HD static bool read_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, Fp &out) {
        uint32_t c[N];
#pragma unroll
        for (int i = 0; i < N; i++) c[i] = 0;
        for (int i = 0; i < BYTES; i++) {
            uint32_t byte = in[i];
            if (i == BYTES - 1 && with_flags) {
                *flags = (uint8_t)(byte & 0xC0u);
                byte &= 0x3Fu;
            }
            c[i >> 2] |= byte << (8 * (i & 3));
        }
        bool ok = canonical_lt_mod(c);
        out = from_canonical(c);
        return ok;
    }
    PTAU_HD void write_bytes(uint8_t *out, uint8_t flags) const {
        uint32_t c[N];
        to_canonical(c);
#pragma unroll
        for (int i = 0; i < BYTES; i++) {
            uint8_t byte = (uint8_t)(c[i >> 2] >> (8 * (i & 3)));
            if (i == BYTES - 1) byte |= flags;
            out[i] = byte;
        }
    }
    // -1 / 0 / +1: sign class used for the y flag: +1 iff this > -this
    PTAU_HD int sign_class() const {
        if (is_zero()) return 0;
        uint32_t c[N];
        to_canonical(c);
        return canonical_gt_half(c) ? 1 : -1;
    }
};

}  // namespace ptau
