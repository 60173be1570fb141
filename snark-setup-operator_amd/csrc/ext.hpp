// Quadratic / cubic extension fields Fq[u]/(u^D - NR) over a Montgomery base field.
//
// MI355X-native equivalent of ark-ff 0.4.2 QuadExtField / CubicExtField (the G2 base fields of
// BLS12-377, MNT4-753 (D = 2) and MNT6-753 (D = 3)); NR is a small signed integer so the
// non-residue product is a short addition chain instead of a field multiplication.
#pragma once
#include <type_traits>

#include "fpl.hpp"

namespace ptau {

// base fields that can tell "square / non-square" and produce the root of a or of NR a from one exponentiation (fp29.hpp)
template <class F, class = void>
struct HasSqrtOrNr : std::false_type {};
template <class F>
struct HasSqrtOrNr<F, std::void_t<decltype(F::HAS_SQRT_OR_NR)>> : std::bool_constant<F::HAS_SQRT_OR_NR> {};
template <class F>
constexpr bool has_sqrt_or_nr() { return HasSqrtOrNr<F>::value; }

template <class F, int NR>
struct Fp2 {
    static constexpr int DEG = 2;
    static constexpr int BYTES = 2 * F::BYTES;
    using Base = F;
    F c0, c1;

    PTAU_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
    PTAU_HD static Fp2 one() { return {F::one(), F::zero()}; }
    PTAU_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    PTAU_HD bool operator==(const Fp2 &b) const { return c0 == b.c0 && c1 == b.c1; }
    PTAU_HD bool operator!=(const Fp2 &b) const { return !(*this == b); }
    PTAU_HD static Fp2 add(const Fp2 &a, const Fp2 &b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
    PTAU_HD static Fp2 sub(const Fp2 &a, const Fp2 &b) { return {F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
    PTAU_HD static Fp2 neg(const Fp2 &a) { return {F::neg(a.c0), F::neg(a.c1)}; }
    PTAU_HD static Fp2 dbl(const Fp2 &a) { return {F::dbl(a.c0), F::dbl(a.c1)}; }
    // Over a 12-limb base the three base products are register-ABI calls (Fp::mul_vec), so the Karatsuba glue is
    // inlined and operands never leave the VGPRs; over 24 limbs the whole product is one by-reference function.
    static constexpr bool MUL_OUT_OF_LINE = F::MUL_OUT_OF_LINE;
    // measured on the lane-split MNT4-753 G2 (fpl.hpp): point operations inlined around register-ABI calls 307 k updates/s,
    // out of line by reference 320 k — they stay out of line
    static constexpr bool POINT_OPS_INLINE = false;
    __host__ __device__ __noinline__ static Fp2 mul_out(const Fp2 &a, const Fp2 &b) { return mul_inl(a, b); }
    __host__ __device__ __noinline__ static Fp2 sqr_out(const Fp2 &a) { return sqr_inl(a); }
    PTAU_HD static Fp2 mul(const Fp2 &a, const Fp2 &b) {
        if constexpr (MUL_OUT_OF_LINE) return mul_out(a, b);
        else return mul_inl(a, b);
    }
    PTAU_HD static Fp2 sqr(const Fp2 &a) {
        if constexpr (MUL_OUT_OF_LINE) return sqr_out(a);
        else return sqr_inl(a);
    }
    // Karatsuba: 3 base multiplications
    PTAU_HD static Fp2 mul_inl(const Fp2 &a, const Fp2 &b) {
        F v0 = F::mul(a.c0, b.c0);
        F v1 = F::mul(a.c1, b.c1);
        F s = F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1));
        Fp2 r;
        r.c1 = F::sub(F::sub(s, v0), v1);
        r.c0 = F::add(v0, F::mul_small(v1, NR));
        return r;
    }
    // complex squaring: 2 base multiplications
    PTAU_HD static Fp2 sqr_inl(const Fp2 &a) {
        F v0 = F::mul(a.c0, a.c1);
        F t = F::mul(F::add(a.c0, a.c1), F::add(a.c0, F::mul_small(a.c1, NR)));
        Fp2 r;
        r.c0 = F::sub(F::sub(t, v0), F::mul_small(v0, NR));
        r.c1 = F::dbl(v0);
        return r;
    }
    PTAU_HD static Fp2 mul_base(const Fp2 &a, const F &k) { return {F::mul(a.c0, k), F::mul(a.c1, k)}; }
    PTAU_HD static Fp2 make(const F &c0, const F &c1) { return {c0, c1}; }
    PTAU_HD static Fp2 make_pair_scaled(const Fp2 &a, int k) { return {F::mul_small(a.c0, k), F::mul_small(a.c1, k)}; }
    PTAU_HD static Fp2 conj(const Fp2 &a) { return {a.c0, F::neg(a.c1)}; }
    static constexpr int LANES = F::LANES;   // lanes of a wavefront that hold one element (2 over a lane-split base field)
    __host__ __device__ __noinline__ static Fp2 inverse(const Fp2 &a) {
        F n = F::sub(F::sqr(a.c0), F::mul_small(F::sqr(a.c1), NR));
        F ni = F::inverse(n);
        return {F::mul(a.c0, ni), F::neg(F::mul(a.c1, ni))};
    }
    // sqrt via the norm (complex method): 2 base sqrts + 1 inversion
    __host__ __device__ __noinline__ static bool sqrt(const Fp2 &a, Fp2 &out) {
        if (a.c1.is_zero()) {
            F r;
            if (F::sqrt(a.c0, r)) { out = {r, F::zero()}; return true; }
            // a0 = NR * y^2  =>  sqrt = y*u
            F t = F::mul(a.c0, F::inverse(F::mul_small(F::one(), NR)));
            if (!F::sqrt(t, r)) return false;
            out = {F::zero(), r};
            return true;
        }
        F norm = F::sub(F::sqr(a.c0), F::mul_small(F::sqr(a.c1), NR));
        F alpha;
        if (!F::sqrt(norm, alpha)) return false;
        F two_inv = F::from_canonical(F::Params::QP1H);   // (p+1)/2 = 1/2
        F delta = F::mul(F::add(a.c0, alpha), two_inv);
        if constexpr (has_sqrt_or_nr<F>()) {
            // Of delta+ = (a0 + alpha)/2 and delta- = NR (a1/2)^2 / delta+ exactly one is a square.  ONE exponentiation decides
            // and serves both: if delta+ is a square, x0 = sqrt(delta+), x1 = a1 / (2 x0); if not, with s = sqrt(NR delta+)
            // (sqrt_or_nr) x0 = sqrt(delta-) = NR (a1/2) / s and x1 = a1 / (2 x0) = s / NR — no second exponentiation (every
            // wave used to pay it: one lane in two needs the other candidate).
            static_assert(F::Params::SQRT_NR == NR, "the base field's tower constants are for another non-residue");
            F s, s_inv;
            bool residue;
            if (!F::sqrt_or_nr(delta, s, s_inv, residue)) return false;
            if (s.is_zero()) return false;
            const F half_a1 = F::mul(a.c1, two_inv);
            if (residue) out = {s, F::mul(half_a1, s_inv)};
            else out = {F::mul_small(F::mul(half_a1, s_inv), NR), F::mul(s, F::load_const(F::Params::SQRT_NR_INV))};
            return sqr(out) == a;
        } else {
            F x0, x0_inv;
            if (!F::sqrt_ex(delta, x0, &x0_inv)) {
                delta = F::mul(F::sub(a.c0, alpha), two_inv);
                if (!F::sqrt_ex(delta, x0, &x0_inv)) return false;
            }
            if (x0.is_zero()) return false;
            F x1 = F::mul(F::mul(a.c1, two_inv), x0_inv);
            out = {x0, x1};
            return sqr(out) == a;
        }
    }
    PTAU_HD static bool read_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, Fp2 &out) {
        uint8_t dummy = 0;
        bool ok = F::read_bytes(in, false, &dummy, out.c0);
        ok &= F::read_bytes(in + F::BYTES, with_flags, flags, out.c1);
        return ok;
    }
    PTAU_HD void write_bytes(uint8_t *out, uint8_t flags) const {
        c0.write_bytes(out, 0);
        c1.write_bytes(out + F::BYTES, flags);
    }
    // arkworks Ord: highest coefficient first
    PTAU_HD int sign_class() const {
        int s = c1.sign_class();
        return s ? s : c0.sign_class();
    }
};

// ---- Fq2 spread over a LANE PAIR --------------------------------------------------------------------------
// One element a0 + a1 u lives in two adjacent lanes: the even lane holds a0, the odd lane a1 (N VGPRs each instead of
// 2N in one lane).  A Jacobian G2 point then costs what a G1 point costs in registers (36 VGPRs per lane), point
// operations stay inlined at <= 256 VGPRs without spills and the kernel runs 2+ waves per SIMD — a single-lane Fq2
// point operation takes all 512 unified registers, spills, and runs at one third of the multiplier peak.
// Products need the partner's coefficient: one DPP quad_perm [1,0,3,2] move per limb.  Schoolbook instead of
// Karatsuba (4 base products on 2 lanes = 2 per lane), complex squaring (1 per lane).  Every lane of a pair executes
// the same instruction stream; all data-dependent branches in the callers are pair-uniform by construction
// (is_zero / == combine both lanes).  Device only: the CPU emulation of the kernel bodies uses Fp2.
// lane_parity() / lane_xchg(): fpl.hpp

template <class F, int NR>
struct Fp2L {
    static constexpr int DEG = 2;
    static constexpr int BYTES = 2 * F::BYTES;
    static constexpr int LANES = 2;
    static constexpr bool MUL_OUT_OF_LINE = false;
    static constexpr bool POINT_OPS_INLINE = F::POINT_OPS_INLINE;
    using Base = F;
    F c;   // a0 on even lanes, a1 on odd lanes

    PTAU_HD static F other(const F &v) {
        F r;
#pragma unroll
        for (int i = 0; i < F::N; i++) r.l[i] = lane_xchg(v.l[i]);
        return r;
    }
    PTAU_HD static F sel(bool odd, const F &if_odd, const F &if_even) {
        F r;
#pragma unroll
        for (int i = 0; i < F::N; i++) r.l[i] = odd ? if_odd.l[i] : if_even.l[i];
        return r;
    }
    // NB every exchange is executed unconditionally by both lanes of a pair: a DPP move issued under a lane-divergent
    // condition would read its partner while that lane is masked off.
    PTAU_HD static bool both(bool mine) {
        const uint32_t theirs = lane_xchg(mine ? 1u : 0u);
        return mine && theirs != 0u;
    }

    PTAU_HD static Fp2L make(const F &c0, const F &c1) { return {lane_parity() ? c1 : c0}; }
    PTAU_HD static Fp2L zero() { return {F::zero()}; }
    PTAU_HD static Fp2L one() { return {lane_parity() ? F::zero() : F::one()}; }
    PTAU_HD bool is_zero() const { return both(c.is_zero()); }
    PTAU_HD bool operator==(const Fp2L &b) const { return both(c == b.c); }
    PTAU_HD bool operator!=(const Fp2L &b) const { return !(*this == b); }
    PTAU_HD static Fp2L add(const Fp2L &a, const Fp2L &b) { return {F::add(a.c, b.c)}; }
    PTAU_HD static Fp2L sub(const Fp2L &a, const Fp2L &b) { return {F::sub(a.c, b.c)}; }
    PTAU_HD static Fp2L neg(const Fp2L &a) { return {F::neg(a.c)}; }
    PTAU_HD static Fp2L dbl(const Fp2L &a) { return {F::dbl(a.c)}; }
    PTAU_HD static Fp2L conj(const Fp2L &a) { return {lane_parity() ? F::neg(a.c) : a.c}; }
    PTAU_HD static Fp2L mul_base(const Fp2L &a, const F &k) { return {F::mul(a.c, k)}; }
    PTAU_HD static Fp2L make_pair_scaled(const Fp2L &a, int k) { return {F::mul_small(a.c, k)}; }
    // even: a0 b0 + NR a1 b1      odd: a1 b0 + a0 b1
    PTAU_HD static Fp2L mul(const Fp2L &a, const Fp2L &b) {
        const bool odd = lane_parity() != 0;
        const F ao = other(a.c), bo = other(b.c);
        const F t1 = F::mul(a.c, sel(odd, bo, b.c));    // even: a0 b0   odd: a1 b0
        const F t2 = F::mul(ao, sel(odd, b.c, bo));     // even: a1 b1   odd: a0 b1
        return {F::add(t1, sel(odd, t2, F::mul_small(t2, NR)))};
    }
    // complex squaring, one base product per lane:  even: (a0 + a1)(a0 + NR a1)   odd: a0 a1
    PTAU_HD static Fp2L sqr(const Fp2L &a) {
        const bool odd = lane_parity() != 0;
        const F ao = other(a.c);
        const F l = sel(odd, ao, F::add(a.c, ao));
        // a0 + NR a1 and t - (1 + NR) v as the shortest chains: for NR = -5 that is a0 - 5 a1 and t + 4 v
        const F nra = NR < 0 ? F::sub(a.c, F::mul_small(ao, -NR)) : F::add(a.c, F::mul_small(ao, NR));
        const F r = sel(odd, a.c, nra);
        const F t = F::mul(l, r);
        const F v = other(t);                            // even lanes receive v = a0 a1
        const F even = (1 + NR) <= 0 ? F::add(t, F::mul_small(v, -(1 + NR))) : F::sub(t, F::mul_small(v, 1 + NR));
        return {sel(odd, F::dbl(t), even)};
    }
    __device__ __noinline__ static Fp2L inverse(const Fp2L &a) {
        const bool odd = lane_parity() != 0;
        const F sq = F::sqr(a.c), sqo = other(sq);
        const F n = F::sub(sel(odd, sqo, sq), F::mul_small(sel(odd, sq, sqo), NR));   // a0^2 - NR a1^2 on both lanes
        const F ni = F::inverse(n);
        const F r = F::mul(a.c, ni);
        return {sel(odd, F::neg(r), r)};
    }
    // compressed input never reaches the lane-pair instances (capi.hip dispatches them for uncompressed points only)
    PTAU_HD static bool sqrt(const Fp2L &, Fp2L &out) { out = zero(); return false; }
    PTAU_HD static bool read_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, Fp2L &out) {
        const bool odd = lane_parity() != 0;
        uint8_t fl = 0;
        bool ok = F::read_bytes(in + (odd ? F::BYTES : 0), with_flags && odd, &fl, out.c);
        const uint32_t fl_other = lane_xchg((uint32_t)fl);
        if (with_flags) *flags = (uint8_t)(odd ? (uint32_t)fl : fl_other);   // the flag bits sit in the last byte of a1
        return both(ok);
    }
    PTAU_HD void write_bytes(uint8_t *out, uint8_t flags) const {
        const bool odd = lane_parity() != 0;
        c.write_bytes(out + (odd ? F::BYTES : 0), odd ? flags : (uint8_t)0);
    }
    // arkworks Ord: highest coefficient first
    PTAU_HD int sign_class() const {
        const bool odd = lane_parity() != 0;
        const int mine = c.sign_class();
        const int theirs = (int)lane_xchg((uint32_t)mine);
        const int s1 = odd ? mine : theirs, s0 = odd ? theirs : mine;
        return s1 ? s1 : s0;
    }
};

template <class F, int NR>
struct Fp3 {
    static constexpr int DEG = 3;
    static constexpr int BYTES = 3 * F::BYTES;
    using Base = F;
    F c0, c1, c2;

    PTAU_HD static Fp3 zero() { return {F::zero(), F::zero(), F::zero()}; }
    PTAU_HD static Fp3 one() { return {F::one(), F::zero(), F::zero()}; }
    PTAU_HD bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    PTAU_HD bool operator==(const Fp3 &b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
    PTAU_HD bool operator!=(const Fp3 &b) const { return !(*this == b); }
    PTAU_HD static Fp3 add(const Fp3 &a, const Fp3 &b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1), F::add(a.c2, b.c2)}; }
    PTAU_HD static Fp3 sub(const Fp3 &a, const Fp3 &b) { return {F::sub(a.c0, b.c0), F::sub(a.c1, b.c1), F::sub(a.c2, b.c2)}; }
    PTAU_HD static Fp3 neg(const Fp3 &a) { return {F::neg(a.c0), F::neg(a.c1), F::neg(a.c2)}; }
    PTAU_HD static Fp3 dbl(const Fp3 &a) { return {F::dbl(a.c0), F::dbl(a.c1), F::dbl(a.c2)}; }
    // (inlining the Fq3 point operations over a lane-split base field sent the compiler into a > 35-minute compile of one
    // translation unit: they stay out of line, by reference, for every base field)
    static constexpr bool MUL_OUT_OF_LINE = true;
    static constexpr bool POINT_OPS_INLINE = false;
    __host__ __device__ __noinline__ static Fp3 mul_out(const Fp3 &a, const Fp3 &b) { return mul_inl(a, b); }
    PTAU_HD static Fp3 mul(const Fp3 &a, const Fp3 &b) {
        if constexpr (MUL_OUT_OF_LINE) return mul_out(a, b);
        else return mul_inl(a, b);
    }
    // Karatsuba-style: 6 base multiplications
    PTAU_HD static Fp3 mul_inl(const Fp3 &a, const Fp3 &b) {
        F v0 = F::mul(a.c0, b.c0), v1 = F::mul(a.c1, b.c1), v2 = F::mul(a.c2, b.c2);
        F t12 = F::sub(F::sub(F::mul(F::add(a.c1, a.c2), F::add(b.c1, b.c2)), v1), v2);   // a1b2 + a2b1
        F t01 = F::sub(F::sub(F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1)), v0), v1);   // a0b1 + a1b0
        F t02 = F::sub(F::sub(F::mul(F::add(a.c0, a.c2), F::add(b.c0, b.c2)), v0), v2);   // a0b2 + a2b0
        Fp3 r;
        r.c0 = F::add(v0, F::mul_small(t12, NR));
        r.c1 = F::add(t01, F::mul_small(v2, NR));
        r.c2 = F::add(t02, v1);
        return r;
    }
    PTAU_HD static Fp3 sqr(const Fp3 &a) { return mul(a, a); }
    PTAU_HD static Fp3 mul_base(const Fp3 &a, const F &k) { return {F::mul(a.c0, k), F::mul(a.c1, k), F::mul(a.c2, k)}; }
    PTAU_HD static Fp3 make(const F &c0, const F &c1, const F &c2) { return {c0, c1, c2}; }
    static constexpr int LANES = F::LANES;
    __host__ __device__ __noinline__ static Fp3 inverse(const Fp3 &a) {
        F t0 = F::sub(F::sqr(a.c0), F::mul_small(F::mul(a.c1, a.c2), NR));
        F t1 = F::sub(F::mul_small(F::sqr(a.c2), NR), F::mul(a.c0, a.c1));
        F t2 = F::sub(F::sqr(a.c1), F::mul(a.c0, a.c2));
        F n = F::add(F::mul(a.c0, t0), F::mul_small(F::add(F::mul(a.c2, t1), F::mul(a.c1, t2)), NR));
        F ni = F::inverse(n);
        return {F::mul(t0, ni), F::mul(t1, ni), F::mul(t2, ni)};
    }
    // Frobenius a -> a^q
    PTAU_HD static Fp3 frobenius(const Fp3 &a) {
        const F c1 = F::load_const(F::Params::FROB3_C1), c2 = F::load_const(F::Params::FROB3_C2);
        return {a.c0, F::mul(a.c1, c1), F::mul(a.c2, c2)};
    }
    __host__ __device__ __noinline__ static Fp3 pow_limbs(const Fp3 &a, const uint32_t *e, int top_bit) {
        return pow_sliding(a, e, top_bit);
    }
    // sqrt through the norm: b = (a^((q+1)/2))^q = a^((q^2+q)/2), so b^2 * a = a^(1+q+q^2) = N(a) in Fq
    // and sqrt(a) = sqrt_Fq(N(a)) / b.   One 753-bit Fq3 exponentiation + one base-field sqrt.
    __host__ __device__ __noinline__ static bool sqrt(const Fp3 &a, Fp3 &out) {
        if (a.is_zero()) { out = zero(); return true; }
        Fp3 b = frobenius(pow_limbs(a, F::Params::QP1H_E, F::Params::BITS - 1));
        Fp3 nrm = mul(sqr(b), a);
        if (!nrm.c1.is_zero() || !nrm.c2.is_zero()) return false;   // cannot happen for valid field data
        F s;
        if (!F::sqrt(nrm.c0, s)) return false;
        Fp3 bi = inverse(b);
        out = {F::mul(bi.c0, s), F::mul(bi.c1, s), F::mul(bi.c2, s)};
        return sqr(out) == a;
    }
    PTAU_HD static bool read_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, Fp3 &out) {
        uint8_t dummy = 0;
        bool ok = F::read_bytes(in, false, &dummy, out.c0);
        ok &= F::read_bytes(in + F::BYTES, false, &dummy, out.c1);
        ok &= F::read_bytes(in + 2 * F::BYTES, with_flags, flags, out.c2);
        return ok;
    }
    PTAU_HD void write_bytes(uint8_t *out, uint8_t flags) const {
        c0.write_bytes(out, 0);
        c1.write_bytes(out + F::BYTES, 0);
        c2.write_bytes(out + 2 * F::BYTES, flags);
    }
    PTAU_HD int sign_class() const {
        int s = c2.sign_class();
        if (s) return s;
        s = c1.sign_class();
        return s ? s : c0.sign_class();
    }
};

}  // namespace ptau
