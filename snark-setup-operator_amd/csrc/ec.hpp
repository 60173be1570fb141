// Short-Weierstrass Jacobian arithmetic, generic over the base field (Fq, Fq2, Fq3).
//
// MI355X-native equivalent of ark-ec 0.4.2 short_weierstrass::{Affine, Projective} as used by
// setup-utils' batch_exp (reference call site /root/reference/src/bin/contribute.rs:811-823).
// Outputs are normalised to affine before serialisation, so results are canonical and
// bit-identical to any other correct scalar multiplication.
#pragma once
#include "ext.hpp"

namespace ptau {

// Build a field element of type F from the per-coefficient constant tables of a curve config.
template <class F, class Cfg>
struct CurveConsts {
    using B = typename F::Base;
    PTAU_HD static B base_from(const uint32_t *p) { return B::load_const(p); }
    PTAU_HD static F coeff_a() {
        if constexpr (F::DEG == 1) return base_from(Cfg::A0);
        else if constexpr (F::DEG == 2) return F::make(base_from(Cfg::A0), base_from(Cfg::A1));
        else return F::make(base_from(Cfg::A0), base_from(Cfg::A1), base_from(Cfg::A2));
    }
    PTAU_HD static F coeff_b() {
        if constexpr (F::DEG == 1) return base_from(Cfg::B0);
        else if constexpr (F::DEG == 2) return F::make(base_from(Cfg::B0), base_from(Cfg::B1));
        else return F::make(base_from(Cfg::B0), base_from(Cfg::B1), base_from(Cfg::B2));
    }
};

template <class F>
struct Affine {
    F x, y;
    bool inf;
};

template <class F, class Cfg>
struct Jac {
    using CfgT = Cfg;
    F X, Y, Z;   // Z == 0  <=>  point at infinity

    PTAU_HD static Jac infinity() { return {F::one(), F::one(), F::zero()}; }
    PTAU_HD bool is_inf() const { return Z.is_zero(); }
    PTAU_HD static Jac from_affine(const Affine<F> &p) {
        if (p.inf) return infinity();
        return {p.x, p.y, F::one()};
    }

    // Point operations on wide fields are single out-of-line copies (code size / compile time);
    // the 12-limb G1 path keeps them inlined so coordinates stay in VGPRs.
    static constexpr bool OPS_OUT_OF_LINE = !F::POINT_OPS_INLINE;
    __host__ __device__ __noinline__ static Jac dbl_out(const Jac &p) { return dbl_inl(p); }
    __host__ __device__ __noinline__ static Jac add_mixed_out(const Jac &p, const Affine<F> &q) { return add_mixed_inl(p, q); }
    __host__ __device__ __noinline__ static Jac add_out(const Jac &p, const Jac &q) { return add_inl(p, q); }
    PTAU_HD static Jac dbl(const Jac &p) {
        if constexpr (OPS_OUT_OF_LINE) return dbl_out(p);
        else return dbl_inl(p);
    }
    PTAU_HD static Jac add_mixed(const Jac &p, const Affine<F> &q) {
        if constexpr (OPS_OUT_OF_LINE) return add_mixed_out(p, q);
        else return add_mixed_inl(p, q);
    }
    PTAU_HD static Jac add(const Jac &p, const Jac &q) {
        if constexpr (OPS_OUT_OF_LINE) return add_out(p, q);
        else return add_inl(p, q);
    }

    // a * v for the curve coefficient a.  The MNT coefficients are small — a = 2 (MNT4 G1), 11 (MNT6 G1), 26 (MNT4 twist,
    // in Fq), 11 u^2 (MNT6 twist) — so the product is a short addition chain (A_SMALL, A_SHIFT: a = A_SMALL * u^A_SHIFT)
    // instead of a field multiplication in every doubling.
    PTAU_HD static F mul_by_a(const F &v) {
        if constexpr (Cfg::A_SMALL != 0) {
            if constexpr (F::DEG == 1) {
                return F::mul_small(v, Cfg::A_SMALL);
            } else if constexpr (F::DEG == 2) {
                static_assert(Cfg::A_SHIFT == 0, "Fq2 twist coefficient expected in the base field");
                return F::make_pair_scaled(v, Cfg::A_SMALL);
            } else {
                static_assert(Cfg::A_SHIFT == 2, "Fq3 twist coefficient expected as A_SMALL u^2");
                using B = typename F::Base;   // (c0, c1, c2) u^2 = (NR c1, NR c2, c0)
                return F::make(B::mul_small(B::mul_small(v.c1, Cfg::NR), Cfg::A_SMALL),
                               B::mul_small(B::mul_small(v.c2, Cfg::NR), Cfg::A_SMALL), B::mul_small(v.c0, Cfg::A_SMALL));
            }
        } else {
            return F::mul(CurveConsts<F, Cfg>::coeff_a(), v);
        }
    }

    // dbl-2009-l (a = 0) / dbl-2007-bl (general a)
    PTAU_HD static Jac dbl_inl(const Jac &p) {
        if (p.is_inf()) return p;
        F XX = F::sqr(p.X);
        F YY = F::sqr(p.Y);
        F YYYY = F::sqr(YY);
        F S = F::sub(F::sub(F::sqr(F::add(p.X, YY)), XX), YYYY);
        S = F::dbl(S);
        F M = F::add(F::dbl(XX), XX);
        if constexpr (!Cfg::A_IS_ZERO) {
            F ZZ = F::sqr(p.Z);
            M = F::add(M, mul_by_a(F::sqr(ZZ)));
        }
        Jac r;
        r.Z = F::dbl(F::mul(p.Y, p.Z));
        r.X = F::sub(F::sqr(M), F::dbl(S));
        F y8 = F::dbl(F::dbl(F::dbl(YYYY)));
        r.Y = F::sub(F::mul(M, F::sub(S, r.X)), y8);
        return r;
    }

    // madd-2007-bl: p + q with q affine
    PTAU_HD static Jac add_mixed_inl(const Jac &p, const Affine<F> &q) {
        if (q.inf) return p;
        if (p.is_inf()) return from_affine(q);
        F Z1Z1 = F::sqr(p.Z);
        F U2 = F::mul(q.x, Z1Z1);
        F S2 = F::mul(F::mul(q.y, p.Z), Z1Z1);
        F H = F::sub(U2, p.X);
        F rr = F::sub(S2, p.Y);
        if (H.is_zero()) {
            if (rr.is_zero()) return dbl_inl(p);
            return infinity();
        }
        rr = F::dbl(rr);
        F HH = F::sqr(H);
        F I = F::dbl(F::dbl(HH));
        F J = F::mul(H, I);
        F V = F::mul(p.X, I);
        Jac r;
        r.X = F::sub(F::sub(F::sqr(rr), J), F::dbl(V));
        r.Y = F::sub(F::mul(rr, F::sub(V, r.X)), F::dbl(F::mul(p.Y, J)));
        r.Z = F::sub(F::sub(F::sqr(F::add(p.Z, H)), Z1Z1), HH);
        return r;
    }

    // add-2007-bl: full Jacobian addition
    PTAU_HD static Jac add_inl(const Jac &p, const Jac &q) {
        if (p.is_inf()) return q;
        if (q.is_inf()) return p;
        F Z1Z1 = F::sqr(p.Z);
        F Z2Z2 = F::sqr(q.Z);
        F U1 = F::mul(p.X, Z2Z2);
        F U2 = F::mul(q.X, Z1Z1);
        F S1 = F::mul(F::mul(p.Y, q.Z), Z2Z2);
        F S2 = F::mul(F::mul(q.Y, p.Z), Z1Z1);
        F H = F::sub(U2, U1);
        F rr = F::sub(S2, S1);
        if (H.is_zero()) {
            if (rr.is_zero()) return dbl_inl(p);
            return infinity();
        }
        rr = F::dbl(rr);
        F I = F::sqr(F::dbl(H));
        F J = F::mul(H, I);
        F V = F::mul(U1, I);
        Jac r;
        r.X = F::sub(F::sub(F::sqr(rr), J), F::dbl(V));
        r.Y = F::sub(F::mul(rr, F::sub(V, r.X)), F::dbl(F::mul(S1, J)));
        r.Z = F::mul(F::sub(F::sub(F::sqr(F::add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
        return r;
    }

    PTAU_HD static Jac neg(const Jac &p) { return {p.X, F::neg(p.Y), p.Z}; }
};

// y^2 = x^3 + a x + b right-hand side
template <class F, class Cfg>
PTAU_HD F curve_rhs(const F &x) {
    F r = F::mul(F::sqr(x), x);
    if constexpr (!Cfg::A_IS_ZERO) r = F::add(r, F::mul(CurveConsts<F, Cfg>::coeff_a(), x));
    return F::add(r, CurveConsts<F, Cfg>::coeff_b());
}

// arkworks SWFlags
constexpr uint8_t FLAG_NEG = 0x80, FLAG_INF = 0x40;

// Uncompressed read (Validate::No): x | y+flags.  Returns false on a non-canonical coordinate.
template <class F>
PTAU_HD bool read_uncompressed(const uint8_t *in, Affine<F> &p) {
    uint8_t fl = 0, dummy = 0;
    bool ok = F::read_bytes(in, false, &dummy, p.x);
    ok &= F::read_bytes(in + F::BYTES, true, &fl, p.y);
    p.inf = (fl & FLAG_INF) != 0;
    return ok;
}

// Compressed read: x+flags, y recovered by sqrt and the sign flag.
// rc: 0 ok, 1 non-canonical x, 2 x not on curve.
template <class F, class Cfg>
PTAU_HD int read_compressed(const uint8_t *in, Affine<F> &p) {
    uint8_t fl = 0;
    bool ok = F::read_bytes(in, true, &fl, p.x);
    if (!ok) return 1;
    p.inf = (fl & FLAG_INF) != 0;
    if (p.inf) { p.y = F::zero(); return 0; }
    F y;
    if (!F::sqrt(curve_rhs<F, Cfg>(p.x), y)) return 2;
    bool neg = y.sign_class() > 0;
    bool want_neg = (fl & FLAG_NEG) != 0;
    p.y = (neg != want_neg) ? F::neg(y) : y;
    return 0;
}

template <class F>
PTAU_HD void write_point(uint8_t *out, const Affine<F> &p, bool compressed) {
    if (p.inf) {
        const int n = compressed ? F::BYTES : 2 * F::BYTES;
        for (int i = 0; i < n; i++) out[i] = (i == n - 1) ? FLAG_INF : 0;
        return;
    }
    uint8_t fl = p.y.sign_class() > 0 ? FLAG_NEG : 0;
    if (compressed) {
        p.x.write_bytes(out, fl);
    } else {
        p.x.write_bytes(out, 0);
        p.y.write_bytes(out + F::BYTES, fl);
    }
}

}  // namespace ptau
