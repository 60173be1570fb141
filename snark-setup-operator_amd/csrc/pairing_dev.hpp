// K5 on the GPU: setup_utils::check_same_ratio — e(a, d) == e(b, c) — as a HIP kernel (reference call sites
// /root/reference/src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591; per chunk: src/bin/contribute.rs:968-986).
//
// Same construction as the host tail (pairing_host.hpp): reduced Tate pairing over Fq^K = Fq[w]/(w^K - NU), Miller loop
// over the group order with denominator elimination, final exponentiation = Frobenius part * Phi_K(q)/r; one generic body
// serves the four curves (K = 12, 6, 4, 6).  What is GPU-shaped about it:
//   * an Fq^K element lives in K LANES, one Fq coefficient per lane (a group of G = 16 / 8 / 4 lanes); a product is K base
//     products per lane instead of K^2 in one thread, with the partner coefficients read from LDS (the group's scratch row);
//   * one wavefront per check: lane group 0 runs the Miller loop of (a, d), group 1 that of (-b, c), both at once; the two
//     values meet through LDS, ONE final exponentiation decides the check;
//   * the checks of a verification (3 proofs of knowledge, 5 chunk-0 before/after checks, 4 power ratios) are one launch,
//     one workgroup each, on as many CUs.
// The point T of the Miller loop is carried redundantly by the lanes of a group (cheap next to the Fq^K products) so that
// the line coefficients are wave-local values; every branch is uniform over a group.
#pragma once
#include "ec.hpp"

namespace ptau {

constexpr int pairing_group_lanes(int K) { return K <= 4 ? 4 : (K <= 8 ? 8 : 16); }

template <class FqP, class FrP, class G1Cfg, class G2Cfg, class PC>
struct TatePairingDev {
    using Fq = FieldOf<FqP>;
    static constexpr int K = PC::K;
    static constexpr int G = pairing_group_lanes(K);   // lanes per Fq^K element (lanes >= K of a group hold zero)
    static constexpr int NGROUPS = 64 / G;
    static constexpr int DEG2 = G2Cfg::DEG;
    static constexpr int NL = Fq::N;
    using F2 = std::conditional_t<DEG2 == 1, Fq, std::conditional_t<DEG2 == 2, Fp2<Fq, G2Cfg::NR>, Fp3<Fq, G2Cfg::NR>>>;

    // LDS: per lane group three rows of K coefficients (a, b and NU * b of the product in flight)
    struct Shared {
        uint32_t row[NGROUPS][3][K][NL];
    };

    struct Ctx {
        Shared *sh;
        int lane;    // index of this lane's coefficient (0 .. G-1)
        int grp;     // lane group within the wavefront
        __device__ bool live() const { return lane < K; }
    };

    __device__ static void put(const Ctx &c, int which, const Fq &v) {
        if (c.live()) {
#pragma unroll
            for (int i = 0; i < NL; i++) c.sh->row[c.grp][which][c.lane][i] = v.l[i];
        }
    }
    __device__ static Fq get(const Ctx &c, int which, int idx) {
        Fq r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = c.sh->row[c.grp][which][idx][i];
        return r;
    }
    __device__ static Fq load(const uint32_t *p) { return Fq::load_const(p); }

    // One coefficient of a * b: sum_j a_j * b'_(lane - j), b' = b below the wrap and NU * b above it (w^K = NU).
    // AMASK: bit j set when a_j may be non-zero (the sparse line functions skip the rest; a compile-time constant).
    // Every lane of the workgroup must call it (two barriers inside).
    template <uint32_t AMASK = 0xffffffffu>
    __device__ static Fq mul(const Ctx &c, const Fq &a, const Fq &b) {
        __syncthreads();   // the previous product's readers are done with the rows
        put(c, 0, a);
        put(c, 1, b);
        put(c, 2, Fq::mul_small(b, PC::NU));
        __syncthreads();
        Fq acc = Fq::zero();
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (!((AMASK >> j) & 1u)) continue;
            int idx = c.lane - j;
            const bool wrap = idx < 0;
            idx += wrap ? K : 0;
            idx = c.live() ? idx : 0;
            const Fq aj = get(c, 0, j);
            const Fq lo = get(c, 1, idx), hi = get(c, 2, idx);
            Fq bj;
#pragma unroll
            for (int i = 0; i < NL; i++) bj.l[i] = wrap ? hi.l[i] : lo.l[i];
            acc = Fq::add(acc, Fq::mul(aj, bj));
        }
        return c.live() ? acc : Fq::zero();
    }
    __device__ static Fq sqr(const Ctx &c, const Fq &a) { return mul(c, a, a); }

    // gamma^(lane * j): the Frobenius constant of this lane's coefficient
    __device__ static Fq frob_const(const Ctx &c, int j) {
        Fq g = Fq::one();
        const Fq g1 = load(PC::GAMMA1);
        const int e = (c.live() ? c.lane : 0) * j;
        for (int bit = 7; bit >= 0; bit--) {   // e <= 11 * 11 < 256
            g = Fq::sqr(g);
            if ((e >> bit) & 1) g = Fq::mul(g, g1);
        }
        return g;
    }
    __device__ static Fq frobenius(const Ctx &c, const Fq &a, int j) { return Fq::mul(a, frob_const(c, j)); }

    // a^-1 = A / N(a) with A = prod_{e=1}^{K-1} a^(q^e) and N(a) = a * A in Fq (coefficient 0)
    __device__ static Fq inverse(const Ctx &c, const Fq &a) {
        Fq A = frobenius(c, a, 1);
        for (int e = 2; e < K; e++) A = mul(c, A, frobenius(c, a, e));
        const Fq n = mul(c, a, A);
        __syncthreads();
        put(c, 0, n);
        __syncthreads();
        const Fq n0 = get(c, 0, 0);
        return Fq::mul(A, Fq::inverse(n0));
    }
    __device__ static Fq pow_limbs(const Ctx &c, const Fq &a, const uint32_t *e, int bits) {
        Fq r = c.lane == 0 ? Fq::one() : Fq::zero();
        for (int i = bits - 1; i >= 0; i--) {
            r = sqr(c, r);
            if ((e[i >> 5] >> (i & 31)) & 1u) r = mul(c, r, a);
        }
        return r;
    }

    __device__ static const Fq &coeff(const F2 &v, int i) {
        if constexpr (DEG2 == 1) return v;
        else if constexpr (DEG2 == 2) return i == 0 ? v.c0 : v.c1;
        else return i == 0 ? v.c0 : (i == 1 ? v.c1 : v.c2);
    }
    // this lane's coefficient of (twist-field element) * w^shift
    __device__ static Fq embed(const Ctx &c, const F2 &v, int shift) {
        Fq out = Fq::zero();
        constexpr int step = K / DEG2;
        const Fq nu_inv = load(PC::NU_INV);
        for (int i = 0; i < DEG2; i++) {
            int idx = i * step + shift;
            Fq cf = coeff(v, i);
            while (idx < 0) { idx += K; cf = Fq::mul(cf, nu_inv); }
            while (idx >= K) { idx -= K; cf = Fq::mul_small(cf, PC::NU); }
            if (idx == c.lane) out = Fq::add(out, cf);
        }
        return out;
    }
    // positions of Fq^K at which psi(Q)'s coordinates (and the constant term) can be non-zero
    static constexpr uint32_t line_mask() {
        uint32_t m = 1u;
        constexpr int step = K / DEG2;
        for (int i = 0; i < DEG2; i++)
            for (int sh : {2 * PC::SIGN, 3 * PC::SIGN}) {
                int idx = ((i * step + sh) % K + K) % K;
                m |= 1u << idx;
            }
        return m;
    }

    // f_{r,P}(psi(Q)) up to factors in Fq: see pairing_host.hpp for the line functions
    __device__ static Fq miller(const Ctx &c, const Affine<Fq> &P, const Affine<F2> &Q) {
        using J = Jac<Fq, G1Cfg>;
        const Fq xq = embed(c, Q.x, 2 * PC::SIGN), yq = embed(c, Q.y, 3 * PC::SIGN);
        const Fq a = CurveConsts<Fq, G1Cfg>::coeff_a();
        J T = J::from_affine(P);
        Fq f = c.lane == 0 ? Fq::one() : Fq::zero();
        auto line = [&](const Fq &cy, const Fq &cx, const Fq &c0) {   // cy*y_Q - cx*x_Q + c0
            Fq l = Fq::sub(Fq::mul(cy, yq), Fq::mul(cx, xq));
            if (c.lane == 0) l = Fq::add(l, c0);
            return c.live() ? l : Fq::zero();
        };
        constexpr uint32_t LM = line_mask();
        for (int i = FrP::BITS - 2; i >= 0; i--) {
            {
                Fq ZZ = Fq::sqr(T.Z), XX = Fq::sqr(T.X);
                Fq Nn = Fq::add(Fq::dbl(XX), XX);
                if constexpr (!G1Cfg::A_IS_ZERO) Nn = Fq::add(Nn, Fq::mul(a, Fq::sqr(ZZ)));
                Fq Z3 = Fq::dbl(Fq::mul(T.Y, T.Z));
                Fq c0 = Fq::sub(Fq::mul(Nn, T.X), Fq::dbl(Fq::sqr(T.Y)));
                const Fq l = line(Fq::mul(Z3, ZZ), Fq::mul(Nn, ZZ), c0);
                f = mul<LM>(c, l, sqr(c, f));
                T = J::dbl_inl(T);
            }
            if ((FrP::MOD[i >> 5] >> (i & 31)) & 1u) {
                Fq ZZ = Fq::sqr(T.Z);
                Fq Hh = Fq::sub(Fq::mul(P.x, ZZ), T.X);
                // T = -P: vertical line (eliminated); for points of order r only at the last step.  The two lane groups
                // carry different points, so the product below runs in any case (barriers inside) — with the line 1.
                const bool vertical = Hh.is_zero();
                Fq R = Fq::sub(Fq::mul(P.y, Fq::mul(ZZ, T.Z)), T.Y);
                Fq Z3 = Fq::mul(T.Z, Hh);
                Fq c0 = Fq::sub(Fq::mul(R, P.x), Fq::mul(Z3, P.y));
                Fq l = line(Z3, R, c0);
                const Fq unit = c.lane == 0 ? Fq::one() : Fq::zero();
#pragma unroll
                for (int t = 0; t < NL; t++) l.l[t] = vertical ? unit.l[t] : l.l[t];
                f = mul<LM>(c, l, f);
                T = vertical ? J::infinity() : J::add_mixed_inl(T, P);
            }
        }
        return f;
    }
    __device__ static Fq final_exp(const Ctx &c, const Fq &f) {
        const Fq finv = inverse(c, f);
        Fq t;
        if constexpr (K == 12) {
            t = mul(c, frobenius(c, f, 6), finv);
            t = mul(c, frobenius(c, t, 2), t);
        } else if constexpr (K == 6) {
            t = mul(c, frobenius(c, f, 3), finv);
            t = mul(c, frobenius(c, t, 1), t);
        } else {
            t = mul(c, frobenius(c, f, 2), finv);
        }
        if constexpr (PC::HARD_KIND == 1) {
            // BLS12: Phi_12(q)/r = 1 + c (x + q)(x^2 + q^2 - 1), c = (x - 1)^2 / 3 (gen_constants.py checks the identity), so
            //   t^hard = t * (a^(x^2) * a^(q^2) * a^-1),  a = g^x * g^q,  g = t^c
            // with a^-1 = a^(q^6) (t is in the cyclotomic subgroup after the easy part: t^(q^6 + 1) = 1).  One 125-bit and three
            // 64-bit powers (x has seven bits set) instead of the 1255-bit one: ~400 Fq^12 products instead of ~1900.
            const Fq g = pow_limbs(c, t, PC::HARD_C, PC::HARD_C_BITS);
            const Fq a = mul(c, pow_limbs(c, g, PC::HARD_X, PC::HARD_X_BITS), frobenius(c, g, 1));
            const Fq ax2 = pow_limbs(c, pow_limbs(c, a, PC::HARD_X, PC::HARD_X_BITS), PC::HARD_X, PC::HARD_X_BITS);
            const Fq b = mul(c, mul(c, ax2, frobenius(c, a, 2)), frobenius(c, a, 6));
            return mul(c, t, b);
        } else {
            return pow_limbs(c, t, PC::HARD, PC::HARD_BITS);
        }
    }

    // One workgroup (one wavefront) per check.  pts: per check the four UNCOMPRESSED points a | b | c | d back to back.
    // result[k]: 1 same ratio, 0 different, -1 malformed (non-canonical coordinate).  gt (optional): K * N32 canonical
    // 32-bit words of the reduced pairing product e(a, d) * e(-b, c) per check (test hook).
    __device__ static void check_body(const uint8_t *pts, int *result, uint32_t *gt) {
        __shared__ Shared sh;
        __shared__ int bad_flag;
        const int tid = threadIdx.x, k = blockIdx.x;
        Ctx c{&sh, tid % G, tid / G};
        constexpr int U1 = 2 * Fq::BYTES, U2 = 2 * F2::BYTES;
        const uint8_t *base = pts + (size_t)k * (2 * U1 + 2 * U2);
        if (tid == 0) bad_flag = 0;
        __syncthreads();
        // even groups: (a, d); odd groups: (-b, c)   (groups beyond the first two repeat them)
        const bool second = (c.grp & 1) != 0;
        Affine<Fq> P;
        Affine<F2> Q;
        bool ok = read_uncompressed<Fq>(base + (second ? U1 : 0), P);
        ok &= read_uncompressed<F2>(base + 2 * U1 + (second ? 0 : U2), Q);
        Affine<Fq> Po;   // the other G1 point and G2 point, for the infinity / canonical tests of the whole check
        Affine<F2> Qo;
        ok &= read_uncompressed<Fq>(base + (second ? 0 : U1), Po);
        ok &= read_uncompressed<F2>(base + 2 * U1 + (second ? U2 : 0), Qo);
        if (!ok) bad_flag = 1;
        __syncthreads();
        if (bad_flag) {
            if (tid == 0) result[k] = -1;
            return;
        }
        if (P.inf || Q.inf || Po.inf || Qo.inf) {   // uniform over the workgroup
            if (tid == 0) result[k] = 0;
            return;
        }
        if (second) P.y = Fq::neg(P.y);
        Fq f = miller(c, P, Q);
        // product of the two Miller values: group g takes its partner's coefficients through LDS
        __syncthreads();
        put(c, 0, f);
        __syncthreads();
        Fq other = Fq::zero();
        if (c.live()) {
#pragma unroll
            for (int i = 0; i < NL; i++) other.l[i] = sh.row[c.grp ^ 1][0][c.lane][i];
        }
        __syncthreads();
        f = mul(c, f, other);
        const Fq e = final_exp(c, f);
        const Fq expect = c.lane == 0 ? Fq::one() : Fq::zero();
        const bool mine = !c.live() || e == expect;
        const unsigned long long ballot = __ballot(mine);
        if (tid == 0) result[k] = (ballot & ((1ull << G) - 1ull)) == ((1ull << G) - 1ull) ? 1 : 0;
        if (gt && c.grp == 0 && c.live()) {
            uint32_t cw[Fq::N32 > NL ? Fq::N32 : NL];
            canonical_words(e, cw);
            for (int i = 0; i < Fq::N32; i++) gt[((size_t)k * K + c.lane) * Fq::N32 + i] = cw[i];
        }
    }
    // canonical value as packed 32-bit words, whatever the radix of the field class
    template <class T = Fq>
    __device__ static void canonical_words(const T &v, uint32_t *w) {
        if constexpr (FqP::LB == 29) {
            uint32_t c29[NL], w32[Fq::N32];
            v.to_canonical(c29);
            Fq::pack32(c29, w32);
            for (int i = 0; i < Fq::N32; i++) w[i] = w32[i];
        } else {
            v.to_canonical(w);
        }
    }
};

template <class TP>
__global__ void __launch_bounds__(64) pairing_check_kernel(const uint8_t *__restrict__ pts, int *__restrict__ result, uint32_t *__restrict__ gt) {
    TP::check_body(pts, result, gt);
}

// host-side launcher table, one per curve (k_<curve>.hip)
struct PairingOps {
    int K, n32;               // Fq^K as K x n32 canonical words per GT element
    size_t check_bytes;       // a | b | c | d, uncompressed
    void (*launch)(hipStream_t, const uint8_t *d_pts, int *d_result, uint32_t *d_gt, int n);
};
template <class TP>
struct PairingLaunch {
    static void launch(hipStream_t s, const uint8_t *d_pts, int *d_result, uint32_t *d_gt, int n) {
        if (n > 0) hipLaunchKernelGGL(pairing_check_kernel<TP>, dim3((unsigned)n), dim3(64), 0, s, d_pts, d_result, d_gt);
    }
    static PairingOps ops() {
        PairingOps o;
        o.K = TP::K;
        o.n32 = TP::Fq::N32;
        o.check_bytes = 2 * (size_t)(2 * TP::Fq::BYTES) + 2 * (size_t)(2 * TP::F2::BYTES);
        o.launch = launch;
        return o;
    }
};
const PairingOps *pairing_ops_bls12_377();
const PairingOps *pairing_ops_bw6_761();
const PairingOps *pairing_ops_mnt4_753();
const PairingOps *pairing_ops_mnt6_753();

}  // namespace ptau
