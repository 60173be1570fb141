// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (CDNA4) VALU.
//
// MI355X-native equivalent of ark-ff 0.4.2 `Fp<MontBackend<_, N>>`, which the reference reaches
// through setup-utils' batch_exp (call site /root/reference/src/bin/contribute.rs:811-823).
// The hot instruction is v_mad_u64_u32 (32x32+64 -> 64); limbs live in VGPRs, every loop is
// unrolled at compile time so no limb array is ever runtime-indexed (no scratch).
#pragma once
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

template <class P>
struct Fp {
    static constexpr int N = P::N;
    static constexpr int DEG = 1;
    static constexpr int BYTES = P::BYTES;   // serialized size of one element
    using Base = Fp<P>;
    using Params = P;
    uint32_t l[N];

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

    // Montgomery product a*b/R mod p: CIOS over 32-bit limbs, one v_mad_u64_u32 per limb product.
    // The OUTER loop stays rolled (code size / compile time: an unrolled 12x12 body is ~7 KB of
    // ISA per call site); b is shifted down one limb per iteration so that no register array is
    // ever indexed by the loop counter (no scratch, no v_movrel).
    PTAU_HD static Fp mul_inl(const Fp &a, const Fp &b) {
        uint32_t t[N + 1];
        uint32_t bb[N];
#pragma unroll
        for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) bb[i] = b.l[i];
#pragma nounroll
        for (int i = 0; i < N; i++) {
            const uint32_t bi = bb[0];
#pragma unroll
            for (int j = 0; j + 1 < N; j++) bb[j] = bb[j + 1];
            uint32_t carry = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                uint64_t v = (uint64_t)a.l[j] * bi + t[j] + carry;
                t[j] = (uint32_t)v;
                carry = (uint32_t)(v >> 32);
            }
            // p < 2^(32N-1) keeps the running value below 2p: one extra limb suffices
            uint64_t top = (uint64_t)t[N] + carry;
            uint32_t m = t[0] * P::INV;
            uint64_t v = (uint64_t)m * P::MOD[0] + t[0];
            carry = (uint32_t)(v >> 32);
#pragma unroll
            for (int j = 1; j < N; j++) {
                v = (uint64_t)m * P::MOD[j] + t[j] + carry;
                t[j - 1] = (uint32_t)v;
                carry = (uint32_t)(v >> 32);
            }
            top += carry;
            t[N - 1] = (uint32_t)top;
            t[N] = (uint32_t)(top >> 32);
        }
        Fp r, u;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) u.l[i] = subb(t[i], P::MOD[i], br);
        bool keep = br && t[N] == 0;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = keep ? t[i] : u.l[i];
        return r;
    }
    // Wide fields (24 limbs) call ONE out-of-line copy: operands then travel through private
    // memory (a few dwordx4 accesses per ~1150-MAC product) instead of inlining ~10 KB of ISA at
    // every call site, which made the kernels uncompilable in reasonable time.
    static constexpr bool MUL_OUT_OF_LINE = (N > 12);
    __host__ __device__ __noinline__ static Fp mul_out(const Fp &a, const Fp &b) { return mul_inl(a, b); }
    PTAU_HD static Fp mul(const Fp &a, const Fp &b) {
        if constexpr (MUL_OUT_OF_LINE) return mul_out(a, b);
        else return mul_inl(a, b);
    }
    PTAU_HD static Fp sqr(const Fp &a) { return mul(a, a); }

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
        Fp r = one();
        for (int i = top_bit; i >= 0; i--) {
            r = sqr(r);
            if ((e[i >> 5] >> (i & 31)) & 1u) r = mul(r, a);
        }
        return r;
    }
    PTAU_HD static Fp inverse(const Fp &a) { return pow_limbs(a, P::PM2, P::BITS - 1); }

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

    // Tonelli-Shanks square root; returns false when a is a non-residue.
    __host__ __device__ __noinline__ static bool sqrt(const Fp &a, Fp &out) {
        if (a.is_zero()) { out = zero(); return true; }
        Fp x = pow_limbs(a, P::TS_T1H, P::BITS - 1);
        Fp b = pow_limbs(a, P::TS_T, P::BITS - 1);
        Fp c;
#pragma unroll
        for (int i = 0; i < N; i++) c.l[i] = P::TS_C[i];
        const Fp o = one();
        int m = P::TS_S;
        while (b != o) {
            int i = 0;
            Fp b2 = b;
            while (b2 != o && i < m) { b2 = sqr(b2); i++; }
            if (i >= m) return false;   // non-residue
            Fp e = c;
            for (int k = 0; k < m - i - 1; k++) e = sqr(e);
            x = mul(x, e);
            c = sqr(e);
            b = mul(b, c);
            m = i;
        }
        out = x;
        return true;
    }

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
