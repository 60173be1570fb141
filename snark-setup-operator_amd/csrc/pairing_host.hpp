// Host-side pairing tail of the verify path: setup_utils::check_same_ratio
// (`E::pairing(a, d) == E::pairing(b, c)`; reference call sites
// /root/reference/src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591).
//
// This is the O(1)-per-vector tail (about a dozen pairings per chunk), kept on the host CPU by design
// (SURVEY K5).  One generic reduced-Tate construction serves the four curves: Fq^K = Fq[w]/(w^K - NU),
// the twist field embedded by u -> w^(K/e), untwist psi(x', y') = (x' w^(2S), y' w^(3S)), Miller loop over
// the group order with denominator elimination, final exponentiation = Frobenius part * Phi_K(q)/r.
// Any non-degenerate bilinear pairing yields the same boolean as arkworks' optimal ate pairing.
#pragma once
#include <array>
#include <vector>

#include "ec.hpp"

namespace ptau {

template <class FqP, class PC>
struct ExtK {
    using Fq = FieldOf<FqP>;
    static constexpr int K = PC::K;
    std::array<Fq, K> c;

    static Fq load(const uint32_t *p) {
        Fq r;
        for (int i = 0; i < Fq::N; i++) r.l[i] = p[i];
        return r;
    }
    static ExtK zero() {
        ExtK r;
        for (auto &x : r.c) x = Fq::zero();
        return r;
    }
    static ExtK one() {
        ExtK r = zero();
        r.c[0] = Fq::one();
        return r;
    }
    bool operator==(const ExtK &o) const {
        for (int i = 0; i < K; i++)
            if (!(c[i] == o.c[i])) return false;
        return true;
    }
    static ExtK mul(const ExtK &a, const ExtK &b) {
        Fq t[2 * K - 1];
        for (auto &x : t) x = Fq::zero();
        for (int i = 0; i < K; i++) {
            if (a.c[i].is_zero()) continue;
            for (int j = 0; j < K; j++) {
                if (b.c[j].is_zero()) continue;
                t[i + j] = Fq::add(t[i + j], Fq::mul(a.c[i], b.c[j]));
            }
        }
        ExtK r;
        for (int i = 0; i < K; i++) {
            r.c[i] = t[i];
            if (i + K < 2 * K - 1) r.c[i] = Fq::add(r.c[i], Fq::mul_small(t[i + K], PC::NU));
        }
        return r;
    }
    static ExtK sqr(const ExtK &a) { return mul(a, a); }
    // a^(q^j): coefficient i times GAMMA1^(i*j)
    static ExtK frobenius(const ExtK &a, int j) {
        Fq g1 = load(PC::GAMMA1), g = Fq::one();
        for (int t = 0; t < j; t++) g = Fq::mul(g, g1);
        ExtK r;
        Fq cur = Fq::one();
        for (int i = 0; i < K; i++) {
            r.c[i] = Fq::mul(a.c[i], cur);
            cur = Fq::mul(cur, g);
        }
        return r;
    }
    // inverse by Gauss-Jordan elimination on the multiplication-by-a matrix
    static ExtK inverse(const ExtK &a) {
        Fq M[K][K + 1];
        ExtK col = a;
        for (int j = 0; j < K; j++) {
            for (int i = 0; i < K; i++) M[i][j] = col.c[i];
            ExtK nx;
            nx.c[0] = Fq::mul_small(col.c[K - 1], PC::NU);
            for (int i = 1; i < K; i++) nx.c[i] = col.c[i - 1];
            col = nx;
        }
        for (int i = 0; i < K; i++) M[i][K] = i == 0 ? Fq::one() : Fq::zero();
        for (int cidx = 0; cidx < K; cidx++) {
            int piv = cidx;
            while (piv < K && M[piv][cidx].is_zero()) piv++;
            if (piv == K) return zero();   // not invertible (a == 0)
            if (piv != cidx)
                for (int j = 0; j <= K; j++) std::swap(M[cidx][j], M[piv][j]);
            Fq iv = Fq::inverse(M[cidx][cidx]);
            for (int j = 0; j <= K; j++) M[cidx][j] = Fq::mul(M[cidx][j], iv);
            for (int r = 0; r < K; r++) {
                if (r == cidx || M[r][cidx].is_zero()) continue;
                Fq f = M[r][cidx];
                for (int j = 0; j <= K; j++) M[r][j] = Fq::sub(M[r][j], Fq::mul(f, M[cidx][j]));
            }
        }
        ExtK r;
        for (int i = 0; i < K; i++) r.c[i] = M[i][K];
        return r;
    }
    static ExtK pow_limbs(const ExtK &a, const uint32_t *e, int bits) {
        ExtK r = one();
        for (int i = bits - 1; i >= 0; i--) {
            r = sqr(r);
            if ((e[i >> 5] >> (i & 31)) & 1u) r = mul(r, a);
        }
        return r;
    }
};

template <class FqP, class FrP, class G1Cfg, class G2Cfg, class PC>
struct TatePairing {
    using Fq = FieldOf<FqP>;
    using E = ExtK<FqP, PC>;
    static constexpr int K = PC::K;
    static constexpr int DEG2 = G2Cfg::DEG;
    using F2 = std::conditional_t<DEG2 == 1, Fq, std::conditional_t<DEG2 == 2, Fp2<Fq, G2Cfg::NR>, Fp3<Fq, G2Cfg::NR>>>;

    static const Fq &coeff(const F2 &v, int i) {
        if constexpr (DEG2 == 1) return v;
        else if constexpr (DEG2 == 2) return i == 0 ? v.c0 : v.c1;
        else return i == 0 ? v.c0 : (i == 1 ? v.c1 : v.c2);
    }
    // twist-field element times w^shift, as an element of Fq^K
    static E embed(const F2 &v, int shift) {
        E out = E::zero();
        const int step = K / DEG2;
        const Fq nu_inv = E::load(PC::NU_INV);
        for (int i = 0; i < DEG2; i++) {
            int idx = i * step + shift;
            Fq cf = coeff(v, i);
            while (idx < 0) { idx += K; cf = Fq::mul(cf, nu_inv); }
            while (idx >= K) { idx -= K; cf = Fq::mul_small(cf, PC::NU); }
            out.c[idx] = Fq::add(out.c[idx], cf);
        }
        return out;
    }
    // f_{r,P}(psi(Q)) up to factors in Fq (killed by the final exponentiation), P affine in G1, Q affine on
    // the twist, neither at infinity.  T runs in Jacobian coordinates, so no field inversion is needed:
    //   doubling  (T=(X,Y,Z)):  N = 3X^2 + aZ^4,  Z3 = 2YZ,   l * Z3 Z^2 = (Z3 Z^2) y_Q - (N Z^2) x_Q + (N X - 2Y^2)
    //   addition  (T + P):      H = x_P Z^2 - X,  R = y_P Z^3 - Y,  Z3 = Z H,   l * Z3 = Z3 y_Q - R x_Q + (R x_P - Z3 y_P)
    static E miller(const Affine<Fq> &P, const Affine<F2> &Q) {
        using J = Jac<Fq, G1Cfg>;
        const E xq = embed(Q.x, 2 * PC::SIGN), yq = embed(Q.y, 3 * PC::SIGN);
        const Fq a = CurveConsts<Fq, G1Cfg>::coeff_a();
        J T = J::from_affine(P);
        E f = E::one();
        auto line = [&](const Fq &cy, const Fq &cx, const Fq &c0) {   // cy*y_Q - cx*x_Q + c0
            E l;
            for (int i = 0; i < K; i++) l.c[i] = Fq::sub(Fq::mul(cy, yq.c[i]), Fq::mul(cx, xq.c[i]));
            l.c[0] = Fq::add(l.c[0], c0);
            return l;
        };
        for (int i = FrP::BITS - 2; i >= 0; i--) {
            {
                Fq ZZ = Fq::sqr(T.Z), XX = Fq::sqr(T.X);
                Fq N = Fq::add(Fq::dbl(XX), XX);
                if constexpr (!G1Cfg::A_IS_ZERO) N = Fq::add(N, Fq::mul(a, Fq::sqr(ZZ)));
                Fq Z3 = Fq::dbl(Fq::mul(T.Y, T.Z));
                Fq c0 = Fq::sub(Fq::mul(N, T.X), Fq::dbl(Fq::sqr(T.Y)));
                f = E::mul(E::sqr(f), line(Fq::mul(Z3, ZZ), Fq::mul(N, ZZ), c0));
                T = J::dbl_inl(T);
            }
            if ((FrP::MOD[i >> 5] >> (i & 31)) & 1u) {
                Fq ZZ = Fq::sqr(T.Z);
                Fq H = Fq::sub(Fq::mul(P.x, ZZ), T.X);
                if (H.is_zero()) {   // T = -P: vertical line (eliminated); only at the last step, r*P = O
                    T = J::infinity();
                    continue;
                }
                Fq R = Fq::sub(Fq::mul(P.y, Fq::mul(ZZ, T.Z)), T.Y);
                Fq Z3 = Fq::mul(T.Z, H);
                Fq c0 = Fq::sub(Fq::mul(R, P.x), Fq::mul(Z3, P.y));
                f = E::mul(f, line(Z3, R, c0));
                T = J::add_mixed_inl(T, P);
            }
        }
        return f;
    }
    static E final_exp(const E &f) {
        E finv = E::inverse(f), t;
        if constexpr (K == 12) {
            t = E::mul(E::frobenius(f, 6), finv);
            t = E::mul(E::frobenius(t, 2), t);
        } else if constexpr (K == 6) {
            t = E::mul(E::frobenius(f, 3), finv);
            t = E::mul(E::frobenius(t, 1), t);
        } else {
            t = E::mul(E::frobenius(f, 2), finv);
        }
        return E::pow_limbs(t, PC::HARD, PC::HARD_BITS);
    }
    // e(a, d) == e(b, c)  <=>  e(a, d) * e(-b, c) == 1     (one shared final exponentiation)
    // rc: 1 same ratio, 0 different, -1 malformed input (non-canonical / infinity)
    static int check_same_ratio(const uint8_t *a_b, const uint8_t *b_b, const uint8_t *c_b, const uint8_t *d_b) {
        Affine<Fq> a, b;
        Affine<F2> c, d;
        if (!read_uncompressed<Fq>(a_b, a) || !read_uncompressed<Fq>(b_b, b) || !read_uncompressed<F2>(c_b, c) ||
            !read_uncompressed<F2>(d_b, d))
            return -1;
        if (a.inf || b.inf || c.inf || d.inf) return 0;
        b.y = Fq::neg(b.y);
        E f = E::mul(miller(a, d), miller(b, c));
        return final_exp(f) == E::one() ? 1 : 0;
    }
};

}  // namespace ptau
