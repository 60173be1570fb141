// Carry-free arithmetic for the a != 0 curves (MNT4-753 / MNT6-753) with EVERY tower coefficient on its own lane pair:
//   G1          : Fq      = FpLZ<P28>              2 lanes per point (the element's limbs split over the pair, fplz.hpp)
//   G2 (MNT4)   : Fq2     = Fp2QZ<FpLZ, 13>        4 lanes per point — pair 0 holds c0, pair 1 holds c1
//   G2 (MNT6)   : Fq3     = Fp3HZ<FpLZ, 11>        6 lanes per point — pair j holds c_j, ten points per wavefront
// A Jacobian point is then 3 x 14 VGPRs per lane whatever the extension degree, every point operation is inlined around
// register-ABI base products (FpLZ::mul_vec / sqr_vec) and NOTHING of the window loop goes through memory — the round-2
// kernels ran Fq2 / Fq3 point operations out of line by reference (3.4-5.6 KB of scratch per lane, 11 000-20 000 x the
// algorithmic bytes through the L2, 35-39 % of wave cycles waiting: profiles/r02_pmc_cfg4.json).
// Spreading costs multiply-accumulates (schoolbook 4 / 9 base products where Karatsuba needs 3 / 6) but a square stays at
// 2 / 6 and squares dominate the doubling (1M + 8S), and the sums, differences and non-residue multiples of the tower —
// ~230-instruction cross-lane carry chains each in the reduced field — become 14 limb-wise instructions.
//
// Field: 28 limbs of 28 bits, R = 2^784 >= 2^31 p (two limbs more than the reduced 26 x 29-bit form): values grow by
// small factors through sums and differences and never need reducing before they meet a product again (V1 V2 <= 2^30);
// only LIMBS need their carries propagated (`norm`) when a bound would overflow — all of it checked at compile time through
// the E<U, V> types, as in fpz.hpp / fplz.hpp.
//
// Cross-lane moves: partner coefficients of a quad by DPP quad_perm; of a 6-lane group by ds_bpermute (no LDS memory is
// touched: the crossbar only).  Every lane of a point executes the same instruction stream; data-dependent branches are
// uniform over the point's lanes by construction (is_zero combines them).
// Serves setup-utils' batch_exp on the MNT curves — /root/reference/src/bin/contribute.rs:811-823.  Device only.
#pragma once
#include "fplz.hpp"

namespace ptau {

#if defined(__HIP_DEVICE_COMPILE__)
#define PTAU_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (ctrl), 0xF, 0xF, true))
#else
#define PTAU_DPP(v, ctrl) (v)
#endif

// ---- Fq2 = Fq[u] / (u^2 - NR) over a QUAD: lanes 0-1 hold c0 (low / high limbs), lanes 2-3 hold c1 ----
template <class Z, int NR>
struct Fp2QZ {
    static_assert(NR >= 2 && NR <= 14, "non-residue multiples are formed limb-wise");
    static constexpr int H = Z::H;
    static constexpr int DEG = 2;
    static constexpr int LANES = 4;
    using Base = Z;
    template <int U, int V>
    struct E {
        typename Z::template E<U, V> c;   // this lane pair's coefficient (this lane's half of its limbs)
        template <int U2, int V2>
        PTAU_HD operator E<U2, V2>() const { return {c}; }
    };
    static constexpr int PV = 2;                 // a product comes out of ONE reduction: below 2p
    static constexpr int SV = 2 + 2 * (Z::subc_k(14, 8));   // a square's: see sqr()
    using Prod = E<1, PV>;

    template <class T>
    PTAU_HD static auto nrm1(const T &v) {   // limbs below 2^28 (a carry pass only where the bound is looser)
        if constexpr (T::LIMB_BOUND == 1) return v;
        else return Z::norm(v);
    }

    PTAU_HD static uint32_t coef() {   // which coefficient this lane's pair holds
#if defined(__HIP_DEVICE_COMPILE__)
        return (threadIdx.x >> 1) & 1u;
#else
        return 0;
#endif
    }
    PTAU_HD static uint32_t lane_in_point() {
#if defined(__HIP_DEVICE_COMPILE__)
        return threadIdx.x & 3u;
#else
        return 0;
#endif
    }
    template <class T>
    PTAU_HD static T other(const T &v) {   // the other pair's coefficient: quad_perm [2,3,0,1]
        T r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = PTAU_DPP(v.l[i], 0x4E);
        return r;
    }
    template <class T>
    PTAU_HD static T from_c0(const T &v) {   // quad_perm [0,1,0,1]
        T r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = PTAU_DPP(v.l[i], 0x44);
        return r;
    }
    template <class T>
    PTAU_HD static T from_c1(const T &v) {   // quad_perm [2,3,2,3]
        T r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = PTAU_DPP(v.l[i], 0xEE);
        return r;
    }
    PTAU_HD static bool all4(bool mine) {   // `mine` is already uniform over the pair
        const uint32_t theirs = PTAU_DPP(mine ? 1u : 0u, 0x4E);
        return mine && theirs != 0u;
    }

    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2, V1 + V2> add(const E<U1, V1> &a, const E<U2, V2> &b) { return {Z::add(a.c, b.c)}; }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static auto sub(const E<U1, V1> &a, const E<U2, V2> &b) {
        const auto r = Z::sub(a.c, b.c);
        return E<decltype(r)::LIMB_BOUND, decltype(r)::VALUE_BOUND>{r};
    }
    template <int U, int V>
    PTAU_HD static auto neg(const E<U, V> &a) {
        const auto r = Z::neg(a.c);
        return E<decltype(r)::LIMB_BOUND, decltype(r)::VALUE_BOUND>{r};
    }
    template <int K, int U, int V>
    PTAU_HD static E<K * U, K * V> times(const E<U, V> &a) { return {Z::template times<K>(a.c)}; }
    template <int U, int V>
    PTAU_HD static E<1, V> norm(const E<U, V> &a) { return {Z::norm(a.c)}; }

    // pair 0: a0 b0 + NR a1 b1      pair 1: a1 b0 + a0 b1 — ONE fused two-term product per pair (FpLZ::row_mul_multi): the
    // pair scans its own coefficient of a against b0 and the other pair's (times NR in pair 0) against b1.  Register ABI:
    // 2 x H VGPRs in, H out; the partner's coefficients arrive by DPP inside.
    typedef typename Z::VecH VecH;
    __device__ __noinline__ static VecH mul2_vec(VecH a, VecH b) {
        uint32_t x[2][H], y[2][H], r[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            x[0][i] = a[i];
            x[1][i] = PTAU_DPP(a[i], 0x4E);
            y[0][i] = PTAU_DPP(b[i], 0x44);
            y[1][i] = PTAU_DPP(b[i], 0xEE);
        }
        const uint32_t f[2] = {1u, coef() ? 1u : (uint32_t)NR};
        Z::template row_mul_multi<2>(x, y, f, r);
        VecH o;
#pragma unroll
        for (int i = 0; i < H; i++) o[i] = r[i];
        return o;
    }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static Prod mul(const E<U1, V1> &a, const E<U2, V2> &b) {
        static_assert((long long)(1 + NR) * V1 * V2 <= Z::VPROD, "the fused product would not come back below 2p");
        Prod r;
#if defined(__HIP_DEVICE_COMPILE__)
        const auto an = nrm1(a.c);                   // units per iteration: (1 + NR) U1 U2 + 1 <= 17 needs both normalised
        const auto bn = nrm1(b.c);
        VecH x, y;
#pragma unroll
        for (int i = 0; i < H; i++) { x[i] = an.l[i]; y[i] = bn.l[i]; }
        const VecH o = mul2_vec(x, y);
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = o[i];
#else
        (void)b;
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = a.c.l[i];
#endif
        return r;
    }
    // complex squaring, ONE base product per pair:  pair 0: (a0 + a1)(a0 + NR a1) - (1 + NR) a0 a1     pair 1: 2 a0 a1.
    // Out of line with a register ABI (H VGPRs in, H out): the sums, the product and the correction live in the callee, the
    // window loop's code stays small (eight squares per doubling) and the glue's temporaries are not alive across calls.
    // The operand: normalised, below SQ_VIN p.
    static constexpr int SQ_VIN = 6144;
    __device__ __noinline__ static VecH sqr2_vec(VecH av) {
        static_assert((1 + NR) % 2 == 0 && (1 + NR) / 2 <= 7, "the correction (1 + NR) a0 a1 is subtracted in two halves");
        constexpr int V = SQ_VIN;
        typename Z::template E<1, V> a;
#pragma unroll
        for (int i = 0; i < H; i++) a.l[i] = av[i];
        const uint32_t m0 = coef() ? 0u : 0xffffffffu;      // all ones in the pair that holds c0
        const auto ot = other(a);
        const auto nr = Z::template times<NR>(ot);
        typename Z::template E<2, 2 * V> l;
        typename Z::template E<1 + NR, (1 + NR) * V> r;
#pragma unroll
        for (int i = 0; i < H; i++) {
            l.l[i] = ot.l[i] + (a.l[i] & m0);                // pair 0: a0 + a1        pair 1: a0
            r.l[i] = a.l[i] + (nr.l[i] & m0);                // pair 0: a0 + NR a1     pair 1: a1
        }
        const auto rn = Z::norm(r);
        uint32_t lx[H], rx[H], tx[H];
#pragma unroll
        for (int i = 0; i < H; i++) { lx[i] = l.l[i]; rx[i] = rn.l[i]; }
        static_assert((long long)2 * V * (1 + NR) * V <= Z::VPROD, "the square's product would not come back below 2p");
        Z::row_mul(lx, rx, tx);                               // pair 0: (a0 + a1)(a0 + NR a1)     pair 1: a0 a1
        typename Z::Nrm t;
#pragma unroll
        for (int i = 0; i < H; i++) t.l[i] = tx[i];
        const auto v = from_c1(t);
        constexpr int HK = (1 + NR) / 2;
        const auto hv = Z::template times<HK>(v);
        const auto d1 = Z::norm(Z::sub(t, hv));
        const auto d2 = Z::sub(d1, hv);                       // t - (1 + NR) a0 a1 + a multiple of p
        static_assert(decltype(d2)::VALUE_BOUND <= SV, "square bound");
        typename Z::template E<decltype(d2)::LIMB_BOUND, decltype(d2)::VALUE_BOUND> o;
#pragma unroll
        for (int i = 0; i < H; i++) o.l[i] = m0 ? d2.l[i] : t.l[i] + t.l[i];
        const auto on = Z::norm(o);
        VecH out;
#pragma unroll
        for (int i = 0; i < H; i++) out[i] = on.l[i];
        return out;
    }
    template <int U, int V>
    PTAU_HD static E<1, SV> sqr(const E<U, V> &a_in) {
        static_assert(V <= SQ_VIN, "operand of the out-of-line Fq2 square out of range");
        E<1, SV> r;
#if defined(__HIP_DEVICE_COMPILE__)
        const auto a = nrm1(a_in.c);
        VecH x;
#pragma unroll
        for (int i = 0; i < H; i++) x[i] = a.l[i];
        const VecH o = sqr2_vec(x);
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = o[i];
#else
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = a_in.c.l[i];
#endif
        return r;
    }
    // a * v for the twist coefficient a = A_SMALL in the base field
    template <int A, int V>
    PTAU_HD static auto mul_by_a(const E<1, V> &x) {
        static_assert(A == 26, "MNT4-753 twist: a = 2 * 13");
        return times<2>(norm(times<13>(x)));
    }
    PTAU_HD static E<1, 2> zero() { return {Z::zero()}; }
    PTAU_HD static E<1, 2> one() {
        const auto o = Z::one();
        E<1, 2> r;
        const uint32_t m0 = coef() ? 0u : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = o.l[i] & m0;
        return r;
    }
    template <int V>
    PTAU_HD static bool is_zero(const E<1, V> &a) { return all4(Z::is_zero(a.c)); }
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero(const E<U, V> &a) { return all4(Z::limbs_all_zero(a.c)); }
    // plumbing for the kernels: this lane's limbs of an element, a base-field value as this pair's coefficient, a word of
    // the LAST coefficient's pair in every lane, a predicate over the point's lanes
    template <class T>
    PTAU_HD static auto &raw(T &v) { return v.c; }
    template <class T>
    PTAU_HD static E<T::LIMB_BOUND, T::VALUE_BOUND> wrap(const T &z) { return {z}; }
    PTAU_HD static uint32_t from_last(uint32_t v) { return PTAU_DPP(v, 0xEE); }
    PTAU_HD static bool all_lanes(bool mine) { return all4(mine); }
};

// ---- Fq3 = Fq[u] / (u^3 - NR) over SIX lanes: pair j (lanes 2j, 2j + 1 of the group) holds c_j; groups of six lanes
// from lane 0 of the wavefront, ten per wave64 (lanes 60-63 idle) ----
template <class Z, int NR>
struct Fp3HZ {
    static_assert(NR >= 2 && NR <= 13, "non-residue multiples are formed limb-wise");
    static constexpr int H = Z::H;
    static constexpr int DEG = 3;
    static constexpr int LANES = 6;
    static constexpr int POINTS_PER_WAVE = 10;
    using Base = Z;
    template <int U, int V>
    struct E {
        typename Z::template E<U, V> c;
        template <int U2, int V2>
        PTAU_HD operator E<U2, V2>() const { return {c}; }
    };
    static constexpr int PV = 2, SV = 2;    // products and squares come out of ONE reduction: below 2p
    using Prod = E<1, PV>;

    template <class T>
    PTAU_HD static auto nrm1(const T &v) {   // limbs below 2^28 (a carry pass only where the bound is looser)
        if constexpr (T::LIMB_BOUND == 1) return v;
        else return Z::norm(v);
    }

    struct Pos { uint32_t lane, grp, j, base; };   // wave-local lane, group, coefficient, first lane of the group
    PTAU_HD static Pos pos() {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t lane = threadIdx.x & 63u, g = (lane * 43u) >> 8;   // lane / 6 for lane < 64
        return {lane, g, (lane - 6u * g) >> 1, 6u * g};
#else
        return {0, 0, 0, 0};
#endif
    }
    PTAU_HD static uint32_t coef() { return pos().j; }
    PTAU_HD static uint32_t lane_in_point() {
        const Pos p = pos();
        return p.lane - p.base;
    }
    PTAU_HD static uint32_t bperm(uint32_t src_lane, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
#else
        (void)src_lane;
        return v;
#endif
    }
    // the coefficient held by pair (j + k) mod 3
    template <class T>
    PTAU_HD static T rot(const T &v, uint32_t k) {
        const Pos p = pos();
        uint32_t jj = p.j + k;
        jj = jj >= 3u ? jj - 3u : jj;
        const uint32_t src = p.base + 2u * jj + (p.lane & 1u);
        T r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = bperm(src, v.l[i]);
        return r;
    }
    // coefficient k in every pair
    template <class T>
    PTAU_HD static T from_c(const T &v, uint32_t k) {
        const Pos p = pos();
        const uint32_t src = p.base + 2u * k + (p.lane & 1u);
        T r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = bperm(src, v.l[i]);
        return r;
    }
    PTAU_HD static bool all6(bool mine) {
#if defined(__HIP_DEVICE_COMPILE__)
        const Pos p = pos();
        const unsigned long long b = __ballot(mine ? 1 : 0);
        return ((b >> p.base) & 0x3Full) == 0x3Full;
#else
        return mine;
#endif
    }

    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2, V1 + V2> add(const E<U1, V1> &a, const E<U2, V2> &b) { return {Z::add(a.c, b.c)}; }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static auto sub(const E<U1, V1> &a, const E<U2, V2> &b) {
        const auto r = Z::sub(a.c, b.c);
        return E<decltype(r)::LIMB_BOUND, decltype(r)::VALUE_BOUND>{r};
    }
    template <int U, int V>
    PTAU_HD static auto neg(const E<U, V> &a) {
        const auto r = Z::neg(a.c);
        return E<decltype(r)::LIMB_BOUND, decltype(r)::VALUE_BOUND>{r};
    }
    template <int K, int U, int V>
    PTAU_HD static E<K * U, K * V> times(const E<U, V> &a) { return {Z::template times<K>(a.c)}; }
    template <int U, int V>
    PTAU_HD static E<1, V> norm(const E<U, V> &a) { return {Z::norm(a.c)}; }

    // r_j = sum_k a_{(j-k) mod 3} b_k, the terms that wrap below u^0 times NR — ONE fused three-term product per pair
    // (FpLZ::row_mul_multi): the row operands are b0, b1, b2, the scanned operands this pair's a_j, a_{j-1}, a_{j-2}, taken
    // from NR a (normalised once, at the source pair) where the term wraps.  Register ABI: 2 x H VGPRs in, H out; every
    // cross-pair move is a ds_bpermute inside.
    typedef typename Z::VecH VecH;
    __device__ __noinline__ static VecH mul3_vec(VecH a, VecH b) {
        const Pos p = pos();
        typename Z::template E<1, 1> av;
#pragma unroll
        for (int i = 0; i < H; i++) av.l[i] = a[i];
        const auto an = Z::norm(Z::template times<NR>(av));   // NR a, limbs below 2^28 again
        const uint32_t h = p.lane & 1u;
        const uint32_t jm1 = p.j == 0u ? 2u : p.j - 1u, jm2 = p.j == 2u ? 0u : p.j + 1u;   // (j - 1) mod 3, (j - 2) mod 3
        uint32_t y[3][H], r[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            y[0][i] = bperm(p.base + h, b[i]);
            y[1][i] = bperm(p.base + 2u + h, b[i]);
            y[2][i] = bperm(p.base + 4u + h, b[i]);
        }
        // The scanned limbs are fetched where they are consumed (limb i of a_{j-k} sits in register i mod H of the low / high
        // lane of pair j - k; what that pair SENDS is NR a where the term wraps: to pair j + 1 iff it is pair 2, to pair
        // j + 2 iff it is pair 1 or 2), so only a and NR a are held: the callee stays within the caller-saved registers.
        Z::template row_mul_scan<3>([&](int k, int i) {
            const int t = i < H ? i : i - H;
            const uint32_t top = i < H ? 0u : 1u;
            if (k == 0) return i < H ? lane_bcast_even(a[t]) : lane_bcast_odd(a[t]);
            if (k == 1) return bperm(p.base + 2u * jm1 + top, p.j == 2u ? an.l[t] : a[t]);
            return bperm(p.base + 2u * jm2 + top, p.j >= 1u ? an.l[t] : a[t]);
        }, y, r);
        VecH o;
#pragma unroll
        for (int i = 0; i < H; i++) o[i] = r[i];
        return o;
    }
    // r0 = a0 a0 + (2 NR a1) a2,  r1 = (2 a0) a1 + (NR a2) a2,  r2 = (2 a0) a2 + a1 a1: one fused two-term product per pair
    __device__ __noinline__ static VecH sqr3_vec(VecH a) {
        const Pos p = pos();
        typename Z::template E<1, 1> av;
#pragma unroll
        for (int i = 0; i < H; i++) av.l[i] = a[i];
        const auto an = Z::norm(Z::template times<NR>(av));
        const uint32_t h = p.lane & 1u;
        const uint32_t j1 = p.j == 2u ? 0u : p.j + 1u, j2 = p.j == 0u ? 2u : p.j - 1u;      // (j + 1) mod 3, (j + 2) mod 3
        uint32_t y[2][H], r[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            // rows: a_j | then a2, a2, a1
            const uint32_t r1p = bperm(p.base + 2u * j1 + h, a[i]);   // a_{j+1}: a1 | a2 | a0
            const uint32_t r2 = bperm(p.base + 2u * j2 + h, a[i]);    // a_{j+2}: a2 | a0 | a1
            y[0][i] = a[i];
            y[1][i] = p.j == 1u ? r1p : r2;
        }
        const uint32_t f0 = p.j == 0u ? 1u : 2u, f1 = p.j == 0u ? 2u : 1u;
        // scanned: a0 (times 2 off pair 0) | then NR a1, NR a2, a1 (times 2 in pair 0)
        Z::template row_mul_scan<2>([&](int k, int i) {
            const int t = i < H ? i : i - H;
            const uint32_t top = i < H ? 0u : 1u;
            if (k == 0) return bperm(p.base + top, a[t]) * f0;
            const uint32_t r1n = bperm(p.base + 2u * j1 + top, an.l[t]);   // NR a_{j+1}
            const uint32_t r2s = bperm(p.base + 2u * j2 + top, a[t]);      //    a_{j+2}
            return (p.j == 2u ? r2s : r1n) * f1;
        }, y, r);
        VecH o;
#pragma unroll
        for (int i = 0; i < H; i++) o[i] = r[i];
        return o;
    }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static Prod mul(const E<U1, V1> &a, const E<U2, V2> &b) {
        static_assert((long long)(1 + 2 * NR) * V1 * V2 <= Z::VPROD, "the fused product would not come back below 2p");
        Prod r;
#if defined(__HIP_DEVICE_COMPILE__)
        // the scanned operand must be normalised (its NR multiple is formed inside); the row operand may carry limbs up to
        // 5 x 2^28 (3 U + 1 <= 17 units per iteration).  The product commutes: scan the normalised one.
        VecH x, y;
        if constexpr (U1 == 1 && U2 <= 5) {
#pragma unroll
            for (int i = 0; i < H; i++) { x[i] = a.c.l[i]; y[i] = b.c.l[i]; }
        } else if constexpr (U2 == 1 && U1 <= 5) {
#pragma unroll
            for (int i = 0; i < H; i++) { x[i] = b.c.l[i]; y[i] = a.c.l[i]; }
        } else {
            const auto an = nrm1(a.c);
            const auto bn = Z::norm(b.c);
#pragma unroll
            for (int i = 0; i < H; i++) { x[i] = an.l[i]; y[i] = bn.l[i]; }
        }
        const VecH o = mul3_vec(x, y);
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = o[i];
#else
        (void)b;
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = a.c.l[i];
#endif
        return r;
    }
    template <int U, int V>
    PTAU_HD static Prod sqr(const E<U, V> &a) {
        static_assert((long long)(2 + 2 * NR) * V * V <= Z::VPROD, "the fused square would not come back below 2p");
        Prod r;
#if defined(__HIP_DEVICE_COMPILE__)
        const auto an = nrm1(a.c);
        VecH x;
#pragma unroll
        for (int i = 0; i < H; i++) x[i] = an.l[i];
        const VecH o = sqr3_vec(x);
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = o[i];
#else
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = a.c.l[i];
#endif
        return r;
    }
    // a * v for the twist coefficient a = A u^2:  (c0, c1, c2) u^2 = (NR c1, NR c2, c0)
    template <int A, int V>
    PTAU_HD static E<1, A * NR * V> mul_by_a(const E<1, V> &x) {
        static_assert(A <= 15, "small twist coefficient");
        const uint32_t f = coef() < 2u ? (uint32_t)NR : 1u;
        const auto y = Z::norm(Z::template times<A>(rot(x.c, 1)));
        typename Z::template E<NR, A * NR * V> s;
#pragma unroll
        for (int i = 0; i < H; i++) s.l[i] = y.l[i] * f;
        return {Z::norm(s)};
    }
    PTAU_HD static E<1, 2> zero() { return {Z::zero()}; }
    PTAU_HD static E<1, 2> one() {
        const auto o = Z::one();
        E<1, 2> r;
        const uint32_t m0 = coef() ? 0u : 0xffffffffu;
#pragma unroll
        for (int i = 0; i < H; i++) r.c.l[i] = o.l[i] & m0;
        return r;
    }
    template <int V>
    PTAU_HD static bool is_zero(const E<1, V> &a) { return all6(Z::is_zero(a.c)); }
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero(const E<U, V> &a) { return all6(Z::limbs_all_zero(a.c)); }
    template <class T>
    PTAU_HD static auto &raw(T &v) { return v.c; }
    template <class T>
    PTAU_HD static E<T::LIMB_BOUND, T::VALUE_BOUND> wrap(const T &z) { return {z}; }
    PTAU_HD static uint32_t from_last(uint32_t v) {
        const Pos p = pos();
        return bperm(p.base + 4u + (p.lane & 1u), v);
    }
    PTAU_HD static bool all_lanes(bool mine) { return all6(mine); }
};

// The prime field seen through the same interface (G1): FpLZ itself plus the tower hooks.
template <class Z>
struct Fp1PZ : Z {
    static constexpr int DEG = 1;
    static constexpr int LANES = 2;
    using Base = Z;
    using Prod = typename Z::Nrm;
    PTAU_HD static uint32_t coef() { return 0; }
    PTAU_HD static uint32_t lane_in_point() { return lane_parity(); }
    template <class T>
    PTAU_HD static auto &raw(T &v) { return v; }
    template <class T>
    PTAU_HD static const T &wrap(const T &z) { return z; }
    PTAU_HD static uint32_t from_last(uint32_t v) { return v; }
    PTAU_HD static bool all_lanes(bool mine) { return Z::both(mine); }
};

// ---- Jacobian arithmetic on y^2 = x^3 + a x + b, a != 0, over one of the fields above -----------------------------------
// Coordinates between operations: limbs normalised, values below VC p (the formulas below are closed under that bound —
// the assignments to CT fail to compile otherwise).  Z = 0 mod p  <=>  the point at infinity.
template <class K, class Cfg, int VC_>
struct LazyJacA {
    static constexpr int VC = VC_;
    using CT = typename K::template E<1, VC>;
    CT X, Y, Zc;

    PTAU_HD static LazyJacA infinity() { return {K::one(), K::one(), K::zero()}; }
    PTAU_HD bool is_inf() const { return K::is_zero(Zc); }

    template <class T>
    PTAU_HD static auto a_times(const T &zz2) {
        if constexpr (K::DEG == 1) return K::template times<Cfg::A_SMALL>(zz2);
        else return K::template mul_by_a<Cfg::A_SMALL>(zz2);
    }

    // dbl-2007-bl: 1M + 8S.  The point at infinity maps to itself (Z3 = 2 Y1 Z1).  Ordered so that at most six field
    // elements are alive at any product call: the register-ABI callee leaves the caller 112 VGPRs (the callee-saved half of
    // v32-v255), and every value alive across a call beyond that is a scratch store and reload — the whole HBM traffic of
    // this kernel (profiles/r03_pmc_cfg4*.json).
    PTAU_HD static LazyJacA dbl(const LazyJacA &p) {
        const auto YY = K::sqr(p.Y);
        const auto ZZ = K::sqr(p.Zc);
        LazyJacA r;
        r.Zc = K::norm(K::sub(K::sqr(K::add(p.Y, p.Zc)), K::add(YY, ZZ)));
        const auto ZZ2 = K::sqr(ZZ);
        const auto XX = K::sqr(p.X);
        const auto M = K::norm(K::add(K::template times<3>(XX), a_times(ZZ2)));
        const auto YYYY = K::sqr(YY);
        const auto t = K::sqr(K::add(p.X, YY));
        const auto S = K::norm(K::template times<2>(K::sub(t, K::add(XX, YYYY))));
        r.X = K::norm(K::sub(K::sqr(M), K::template times<2>(S)));
        r.Y = K::norm(K::sub(K::mul(M, K::sub(S, r.X)), K::template times<8>(YYYY)));
        return r;
    }

    // add-2007-bl with Z3 = 2 Z1 Z2 H: 12M + 4S.  The second operand is a TABLE ENTRY that is read from memory where it is
    // used (src.x(), src.y(), src.z(): one use each but z, which is read first and consumed by three products), never held
    // across calls; the exceptional cases (p = O, p = +-q) are rebuilt from the entry as well, so p's coordinates die as soon
    // as the formula has consumed them: at most seven field elements are alive at any product call.
    // src.y() returns the entry's y already negated when the digit is negative; the entry is never the point at infinity.
    template <class Src>
    PTAU_HD static LazyJacA add(const LazyJacA &p, const Src &src) {
        const bool pinf = p.is_inf();
        const auto Z12 = [&] {
            const auto z2 = src.z();
            return K::mul(p.Zc, z2);
        }();
        const auto Z2Z2 = K::sqr(src.z());
        const auto U1 = K::mul(p.X, Z2Z2);
        const auto S1 = K::mul(K::mul(p.Y, src.z()), Z2Z2);
        const auto Z1Z1 = K::sqr(p.Zc);
        const auto U2 = K::mul(src.x(), Z1Z1);
        const auto S2 = K::mul(K::mul(src.y(), p.Zc), Z1Z1);
        const auto Hh = K::norm(K::sub(U2, U1));
        const auto rh = K::norm(K::sub(S2, S1));
        const auto I = K::sqr(K::template times<2>(Hh));
        const auto J = K::mul(Hh, I);
        const auto Vv = K::mul(U1, I);
        LazyJacA r;
        r.Zc = K::norm(K::template times<2>(K::mul(Z12, Hh)));
        const auto rr = K::template times<2>(rh);
        r.X = K::norm(K::sub(K::sqr(rr), K::add(J, K::template times<2>(Vv))));
        r.Y = K::norm(K::sub(K::mul(rr, K::sub(Vv, r.X)), K::template times<2>(K::mul(S1, J))));
        // exceptional cases overwrite the generic result: one join, rare paths, everything rebuilt from the entry
        if (pinf || K::is_zero(Hh)) {
            const LazyJacA q = {src.x(), K::mul(K::norm(src.y()), K::one()), src.z()};   // y may carry a negation's bound
            if (pinf) r = q;
            else if (K::is_zero(rh)) r = dbl(q);
            else r = infinity();
        }
        return r;
    }

    // madd-2007-bl with Z3 = 2 Z1 H (7M + 4S): the second operand AFFINE (a point of the MSM's input vector), read ONCE by the
    // caller at the top (src.x(), src.y()) and re-read only on the rare exceptional join (src.x_again(), src.y_again()), so
    // neither its nor p's coordinates stay alive for that join; at most seven field elements are alive at any product call
    // (the same ordering as LazyJac2::add_src, fpz.hpp).  The formula does not involve the curve's a.
    template <class Src>
    PTAU_HD static LazyJacA add_affine(const LazyJacA &p, const Src &src) {
        const auto x2 = src.x();
        const auto y2 = src.y();
        const bool pinf = p.is_inf();
        const auto Z1Z1 = K::sqr(p.Zc);
        const auto Hh = K::norm(K::sub(K::mul(x2, Z1Z1), p.X));
        const auto rh = K::norm(K::sub(K::mul(K::mul(y2, p.Zc), Z1Z1), p.Y));
        const bool hz = K::is_zero(Hh), rz = K::is_zero(rh);
        LazyJacA r;
        r.Zc = K::norm(K::template times<2>(K::mul(p.Zc, Hh)));
        const auto I = K::template times<4>(K::sqr(Hh));   // (2 H)^2 as 4 H^2: H carries p.X's bound, 2 H would exceed the square's
        const auto J = K::mul(Hh, I);
        const auto Vv = K::mul(p.X, I);
        const auto R2 = K::template times<4>(K::sqr(rh));  // (2 rh)^2
        r.X = K::norm(K::sub(R2, K::add(J, K::template times<2>(Vv))));
        r.Y = K::norm(K::sub(K::mul(K::template times<2>(rh), K::sub(Vv, r.X)), K::template times<2>(K::mul(p.Y, J))));
        if (pinf || hz) {   // one join, rare paths, everything rebuilt from the entry
            const LazyJacA q = {src.x_again(), K::mul(K::norm(src.y_again()), K::one()), K::one()};
            if (pinf) r = q;
            else if (rz) r = dbl(q);
            else r = infinity();
        }
        return r;
    }
};

}  // namespace ptau
