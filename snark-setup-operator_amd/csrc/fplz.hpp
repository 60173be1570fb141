// The carry-free field (fpz.hpp) for a WIDE modulus split over a lane pair, as FpL (fpl.hpp): the low half of the limbs in
// the even lane, the high half in the odd lane — BW6-761: 28 limbs of 28 bits, R = 2^784 >= 2^23 p, the same limb count as
// the reduced 28 x 29-bit form, so the products cost the same multiply-accumulates and lose their normalising tail, and
// every sum / difference of the point formulas becomes H limb-wise instructions per lane (they were ~230-instruction
// chains across the pair: carry folds through DPP, conditional subtraction).
//   * E<U, V>: this lane's H limbs, "every limb below U 2^28 (+ a carry's worth), value below V p" — bounds checked at
//     compile time exactly as in FpZ (a slot of the distributed accumulator row sums 2H products: U1 U2 <= 16).
//   * the product is FpL's row-distributed operand scanning; its tail only propagates carries (the even lane's carry-out
//     enters the odd lane's two lowest limbs, no ripple: limb 1 may exceed 2^28 by < 2^9, inside the column bound's slack).
//   * residue tests: the value's lowest limb decides all but a 2^-28-rare case (j = l_0 / p_0 mod 2^28 must be below V);
//     the rest is gathered, normalised exactly and compared with j p.
// Same interface as FpZ towards LazyJac.  Device only (DPP); serves batch_exp, /root/reference/src/bin/contribute.rs:811-823.
#pragma once
#include "fpl.hpp"
#include "fpz.hpp"

namespace ptau {

template <class P>
struct FpLZ {
    using Params = P;
    static constexpr int N = P::N;
    static constexpr int H = N / 2;
    static constexpr int LB = P::LB;
    static constexpr uint32_t MASK = (1u << LB) - 1u;
    using L29 = FpL<typename P::P29>;
    using F29 = typename L29::Full;
    static_assert(LB == 28 && N % 2 == 0, "28-bit limbs, an even number of them (half per lane)");
    static constexpr int SLACK = LB * N - F29::Params::BITS;   // R = 2^(28 N) >= 2^SLACK p
    static_assert(SLACK >= 15, "R must exceed 2^15 p");
    // A product (a b + m p) / R < a b / R + p comes back below 2p when a b < R p, i.e. V1 V2 <= 2^SLACK.  BW6-761 (28 limbs,
    // 761 bits): 2^23; MNT4/MNT6-753 (28 limbs for 753 bits — the reduced form has 26 of 29 bits): 2^31, so values there
    // practically never need reducing, only limbs need their carries propagated.
    static constexpr long long VPROD = 1ll << (SLACK > 31 ? 30 : SLACK - 1);
    static constexpr int VMAX = SLACK - 1 > 29 ? (1 << 29) : (1 << (SLACK - 1));
    static_assert(2 * H * 17 < 512, "a slot of the accumulator row would overflow 64 bits");

    PTAU_HD static bool hi_lane() { return lane_parity() != 0; }

    template <int U, int V>
    struct E {
        static_assert(U >= 1 && U <= 15, "a limb would not fit 32 bits");
        static_assert(V >= 1 && V <= VMAX, "the value would not stay below R");
        static constexpr int LIMB_BOUND = U, VALUE_BOUND = V;
        uint32_t l[H];
        template <int U2, int V2>
        PTAU_HD operator E<U2, V2>() const {
            static_assert(U2 >= U && V2 >= V, "bounds can only be loosened");
            E<U2, V2> r;
#pragma unroll
            for (int i = 0; i < H; i++) r.l[i] = l[i];
            return r;
        }
    };
    using Nrm = E<1, 2>;

    struct Limbs { uint32_t l[N]; };
    // floor(K p / 2^(28 (N - 1))): the top limb of K p
    static constexpr long long top_of(long long K) {
        unsigned long long c = 0;
        for (int i = 0; i < N - 1; i++) c = ((unsigned long long)K * P::MOD[i] + c) >> LB;
        return (long long)((unsigned long long)K * P::MOD[N - 1] + c);
    }
    // The multiple K p that sub() adds before subtracting a value below V2 p whose limbs are below (UB - 1) 2^28 (+ a
    // carry's worth): every limb of the biased form must dominate the subtrahend's.  Below the top that is the bias; the top
    // limb of K p, less the UB borrowed from it, must reach the top limb of any value below V2 p.  When p's own top limb is
    // >= 16 (BW6-761) K = V2 + 1 does; a modulus that leaves the top limb empty (753 bits in 28 limbs) needs K p to reach
    // 2^(28 (N - 1)) (UB + 1) more: K = V2 + ~9 (UB + 1).
    static constexpr int subc_k(int V2, int UB) {
        if (P::MOD[N - 1] >= 16) return V2 + 1;
        long long K = V2 + 1;
        while (top_of(K) - UB < top_of(V2) + 1) K++;
        return (int)K;
    }
    // K p with every limb below the top raised by UB 2^28, borrowed from the next limb up (see FpZ::make_subc)
    static constexpr Limbs make_subc(int K, int UB) {
        Limbs r{};
        uint64_t c = 0;
        for (int i = 0; i < N; i++) {
            const uint64_t t = (uint64_t)K * P::MOD[i] + c;
            if (i < N - 1) {
                r.l[i] = (uint32_t)(t & MASK) + ((uint32_t)UB << LB) - (i ? (uint32_t)UB : 0u);
                c = t >> LB;
            } else {
                r.l[i] = (uint32_t)t - (uint32_t)UB;
            }
        }
        return r;
    }
    template <int K, int UB>
    struct SubC {
        static_assert(UB <= 15, "bias does not fit a limb");
        static constexpr Limbs v = make_subc(K, UB);
    };
    // this lane's half of a constant given as N limbs
    PTAU_HD static uint32_t mine(const uint32_t *c, int j) { return lane_const(c[j], c[H + j], lane_hi_mask()); }

    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2, V1 + V2> add(const E<U1, V1> &a, const E<U2, V2> &b) {
        E<U1 + U2, V1 + V2> r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2 + 2, V1 + subc_k(V2, U2 + 1)> sub(const E<U1, V1> &a, const E<U2, V2> &b) {
        const uint32_t hm = lane_hi_mask_here();
        constexpr int K = subc_k(V2, U2 + 1);
        E<U1 + U2 + 2, V1 + K> r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = a.l[i] + lane_const(SubC<K, U2 + 1>::v.l[i], SubC<K, U2 + 1>::v.l[H + i], hm) - b.l[i];
        return r;
    }
    template <int U, int V>
    PTAU_HD static E<U + 2, subc_k(V, U + 1)> neg(const E<U, V> &a) {
        const uint32_t hm = lane_hi_mask_here();
        constexpr int K = subc_k(V, U + 1);
        E<U + 2, K> r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = lane_const(SubC<K, U + 1>::v.l[i], SubC<K, U + 1>::v.l[H + i], hm) - a.l[i];
        return r;
    }
    template <int K, int U, int V>
    PTAU_HD static E<K * U, K * V> times(const E<U, V> &a) {
        static_assert(K >= 2 && K <= 15, "small multiples used by the point formulas");
        E<K * U, K * V> r;
#pragma unroll
        for (int i = 0; i < H; i++) {
            if constexpr (K == 2) r.l[i] = a.l[i] + a.l[i];
            else if constexpr (K == 3) r.l[i] = a.l[i] + a.l[i] + a.l[i];
            else if constexpr (K == 4) r.l[i] = a.l[i] << 2;
            else if constexpr (K == 8) r.l[i] = a.l[i] << 3;
            else r.l[i] = a.l[i] * (uint32_t)K;   // 11, 13, ...: shift-and-add chains chosen by the compiler
        }
        return r;
    }
    // carries propagated inside each lane; the even lane's carry-out (< 16) is added to the odd lane's lowest limb as it is
    template <int U, int V>
    PTAU_HD static E<1, V> norm(const E<U, V> &a) {
        const bool hi = hi_lane();
        E<1, V> r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < H - 1; i++) {
            const uint32_t w = a.l[i] + c;
            r.l[i] = w & MASK;
            c = w >> LB;
        }
        const uint32_t w = a.l[H - 1] + c;
        r.l[H - 1] = hi ? w : (w & MASK);
        const uint32_t from_even = lane_xchg(w >> LB);
        r.l[0] += hi ? from_even : 0u;
        return r;
    }

    // ---- Montgomery product / square, accumulator row distributed over the pair (FpL::row_mul), carry-propagating tail ----
    PTAU_HD static void row_finish(const uint64_t (&acc)[H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < H - 1; j++) {
            const uint64_t v = acc[j] + c;
            out[j] = (uint32_t)v & MASK;
            c = v >> LB;
        }
        const uint64_t vt = acc[H - 1] + c;
        out[H - 1] = hi ? (uint32_t)vt : ((uint32_t)vt & MASK);   // odd lane: the value's top, below 2p / 2^(28 (N - 1))
        const uint64_t co = vt >> LB;                            // even lane: carry into the odd lane's digits, < 2^36
        const uint32_t clo = lane_xchg((uint32_t)co & MASK), chi = lane_xchg((uint32_t)(co >> LB));
        const uint32_t w0 = out[0] + (hi ? clo : 0u);
        out[0] = w0 & MASK;
        out[1] += (hi ? chi : 0u) + (w0 >> LB);
    }
    PTAU_HD static void row_mul(const uint32_t (&a)[H], const uint32_t (&b)[H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        uint32_t pm[H];
#pragma unroll
        for (int j = 0; j < H; j++) pm[j] = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
        uint64_t acc[H];
#pragma unroll
        for (int j = 0; j < H; j++) acc[j] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t ai = i < H ? lane_bcast_even(a[i]) : lane_bcast_odd(a[i - H]);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], ai, b[j]);
            const uint32_t m = lane_bcast_even(((uint32_t)acc[0] * P::INV) & MASK);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], m, pm[j]);
            const uint64_t x = acc[0];
            const uint32_t recv = lane_xchg((uint32_t)x & MASK);
#pragma unroll
            for (int j = 0; j + 1 < H; j++) acc[j] = acc[j + 1];
            acc[0] += x >> LB;
            acc[H - 1] = recv;
        }
        row_finish(acc, out);
    }
    PTAU_HD static void row_sqr(const uint32_t (&a)[H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        uint32_t pm[H], az[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            pm[j] = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
            az[j] = hi ? a[j] : 0u;
        }
        const uint32_t dbl_odd = hi ? 1u : 0u;
        uint64_t acc[H];
#pragma unroll
        for (int j = 0; j < H; j++) acc[j] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (i < H) {
                const uint32_t ai = lane_bcast_even(a[i]) << dbl_odd;
#pragma unroll
                for (int j = 0; j < H; j++) mac64_vv(acc[j], ai, a[j]);
            } else {
                const int d = i - H;
                const uint32_t ai = lane_bcast_odd(a[d]);
                mac64_vv(acc[d], ai, az[d]);
                const uint32_t ai2 = ai << 1;
#pragma unroll
                for (int j = d + 1; j < H; j++) mac64_vv(acc[j], ai2, az[j]);
            }
            const uint32_t m = lane_bcast_even(((uint32_t)acc[0] * P::INV) & MASK);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], m, pm[j]);
            const uint64_t x = acc[0];
            const uint32_t recv = lane_xchg((uint32_t)x & MASK);
#pragma unroll
            for (int j = 0; j + 1 < H; j++) acc[j] = acc[j + 1];
            acc[0] += x >> LB;
            acc[H - 1] = recv;
        }
        row_finish(acc, out);
    }
    // (sum_k f_k x_k y_k) / R mod p with ONE reduction: the NP products are accumulated in the same pass of the distributed
    // row (NP H multiply-accumulates per lane and iteration for the products, H for m p) — a tower product a0 b0 + NR a1 b1
    // costs 3 H per iteration instead of the 4 H of two reduced products, a three-term one 4 H instead of 6 H, and the sums,
    // non-residue multiples and carry passes between the partial products disappear.  x_k is scanned (its limb i broadcast to
    // the pair in iteration i, times the small per-lane factor f_k), y_k is the row operand.  Limb bounds: the units a slot
    // collects per iteration, sum_k f_k U(x_k) U(y_k) + 1, must stay within 17 (fourteen iterations per lane: < 2^64).
    // General form: scan(k, i) delivers limb i (0 <= i < N) of the k-th scanned operand, already broadcast to the pair and
    // scaled — the caller decides where it comes from (a DPP broadcast of a register, a ds_bpermute from another pair's lane
    // issued just in time, so that the scanned operands need not be held in registers).
    template <int NP, class Scan>
    PTAU_HD static void row_mul_scan(const Scan &scan, const uint32_t (&y)[NP][H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        uint32_t pm[H];
#pragma unroll
        for (int j = 0; j < H; j++) pm[j] = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
        uint64_t acc[H];
#pragma unroll
        for (int j = 0; j < H; j++) acc[j] = 0;
        // one iteration: the scanned limbs against the rows, then m p and the shift of the row
        auto step = [&](const uint32_t (&xi)[NP]) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
#pragma unroll
                for (int j = 0; j < H; j++) mac64_vv(acc[j], xi[k], y[k][j]);
            }
            const uint32_t m = lane_bcast_even(((uint32_t)acc[0] * P::INV) & MASK);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], m, pm[j]);
            const uint64_t x0 = acc[0];
            const uint32_t recv = lane_xchg((uint32_t)x0 & MASK);
#pragma unroll
            for (int j = 0; j + 1 < H; j++) acc[j] = acc[j + 1];
            acc[0] += x0 >> LB;
            acc[H - 1] = recv;
        };
        // (two loops of H: one loop of N iterations exceeds the full-unroll budget and is left rolled, its operands in scratch)
#pragma unroll
        for (int i = 0; i < H; i++) {
            uint32_t xi[NP];
#pragma unroll
            for (int k = 0; k < NP; k++) xi[k] = scan(k, i);
            step(xi);
        }
#pragma unroll
        for (int i = H; i < N; i++) {
            uint32_t xi[NP];
#pragma unroll
            for (int k = 0; k < NP; k++) xi[k] = scan(k, i);
            step(xi);
        }
        row_finish(acc, out);
    }
    template <int NP>
    PTAU_HD static void row_mul_multi(const uint32_t (&x)[NP][H], const uint32_t (&y)[NP][H], const uint32_t (&f)[NP], uint32_t (&out)[H]) {
        row_mul_scan<NP>([&](int k, int i) { return (i < H ? lane_bcast_even(x[k][i < H ? i : 0]) : lane_bcast_odd(x[k][i < H ? 0 : i - H])) * f[k]; },
                         y, out);
    }
    typedef uint32_t VecH __attribute__((ext_vector_type(H)));
    __device__ __noinline__ static VecH mul_vec(VecH a, VecH b) {
        uint32_t x[H], y[H], r[H];
#pragma unroll
        for (int j = 0; j < H; j++) { x[j] = a[j]; y[j] = b[j]; }
        row_mul(x, y, r);
        VecH o;
#pragma unroll
        for (int j = 0; j < H; j++) o[j] = r[j];
        return o;
    }
    __device__ __noinline__ static VecH sqr_vec(VecH a) {
        uint32_t x[H], r[H];
#pragma unroll
        for (int j = 0; j < H; j++) x[j] = a[j];
        row_sqr(x, r);
        VecH o;
#pragma unroll
        for (int j = 0; j < H; j++) o[j] = r[j];
        return o;
    }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static Nrm mul(const E<U1, V1> &a, const E<U2, V2> &b) {
        static_assert(U1 * U2 <= 16, "a slot of the product's row would overflow 64 bits: normalise an operand first");
        static_assert((long long)V1 * V2 <= VPROD, "the product would not come back below 2p");
        Nrm r;
#if defined(__HIP_DEVICE_COMPILE__)
        VecH x, y;
#pragma unroll
        for (int j = 0; j < H; j++) { x[j] = a.l[j]; y[j] = b.l[j]; }
        const VecH o = mul_vec(x, y);
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
#else
        (void)b;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = a.l[j];   // lane-split fields exist on the device only
#endif
        return r;
    }
    template <int U, int V>
    PTAU_HD static Nrm sqr(const E<U, V> &a) {
        static_assert(U * U <= 16 && U <= 7, "a slot of the square's row would overflow 64 bits: normalise the operand first");
        static_assert((long long)V * V <= VPROD, "the square would not come back below 2p");
        Nrm r;
#if defined(__HIP_DEVICE_COMPILE__)
        VecH x;
#pragma unroll
        for (int j = 0; j < H; j++) x[j] = a.l[j];
        const VecH o = sqr_vec(x);
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
#else
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = a.l[j];
#endif
        return r;
    }

    PTAU_HD static Nrm zero() {
        Nrm r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = 0;
        return r;
    }
    PTAU_HD static Nrm one() {
        Nrm r;
#pragma unroll
        for (int i = 0; i < H; i++) r.l[i] = mine(P::R1, i);
        return r;
    }
    PTAU_HD static bool both(bool m) {
        const uint32_t theirs = lane_xchg(m ? 1u : 0u);
        return m && theirs != 0u;
    }
    // all N limbs of a value in both lanes
    template <int U, int V>
    PTAU_HD static void gather(const E<U, V> &a, uint32_t *f) {   // f: at least N entries
        const bool hi = hi_lane();
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint32_t o = lane_xchg(a.l[j]);
            f[j] = hi ? o : a.l[j];
            f[H + j] = hi ? a.l[j] : o;
        }
    }
    template <int V>
    PTAU_HD static bool is_zero(const E<1, V> &a) {
        const uint32_t l0 = lane_bcast_even(a.l[0]);
        const uint32_t j = (0u - l0 * P::INV) & MASK;   // l_0 / p_0 mod 2^28: the only multiple of p it can be is j p
        if (j >= (uint32_t)V) return false;
        uint32_t f[N];
        gather(a, f);
        uint32_t c = 0, diff = 0;
        uint64_t pc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t w = f[i] + c;
            const uint64_t t = (uint64_t)j * P::MOD[i] + pc;
            diff |= (i < N - 1 ? (w & MASK) : w) ^ (i < N - 1 ? (uint32_t)(t & MASK) : (uint32_t)t);
            c = w >> LB;
            pc = t >> LB;
        }
        return diff == 0;
    }
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero(const E<U, V> &a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < H; i++) o |= a.l[i];
        return both(o == 0);
    }
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero_local(const E<U, V> &a) {   // this lane's half only
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < H; i++) o |= a.l[i];
        return o == 0;
    }

    // FpL (29-bit limbs, fully reduced) <-> this representation; off the hot path: through the whole element in each lane
    PTAU_HD static Nrm from29(const L29 &a) {
        const bool hi = hi_lane();
        const F29 f = a.gather();
        E<1, 1> t, c;
#pragma unroll
        for (int k = 0; k < H; k++) {
            uint32_t half[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int bit = LB * (h * H + k), s = bit / 29, o = bit % 29;
                uint32_t v = s < F29::N ? f.l[s] >> o : 0u;
                if (o + LB > 29 && s + 1 < F29::N) v |= f.l[s + 1] << (29 - o);
                half[h] = v & MASK;
            }
            t.l[k] = hi ? half[1] : half[0];
            c.l[k] = mine(P::FROM29, k);
        }
        return mul(t, c);
    }
    template <int U, int V>
    PTAU_HD static L29 to29(const E<U, V> &a) {
        E<1, 1> c;
#pragma unroll
        for (int k = 0; k < H; k++) c.l[k] = mine(P::TO29, k);
        const Nrm v = mul(a, c);
        uint32_t f[N + 1], u[N];
        gather(v, f);
        // exact digits, then one conditional subtraction (the value is below 2p)
        uint32_t cc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t w = f[i] + cc;
            f[i] = i < N - 1 ? (w & MASK) : w;
            cc = i < N - 1 ? (w >> LB) : 0u;
        }
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t w = (int32_t)f[i] - (int32_t)P::MOD[i] + cy;
            cy = w >> LB;
            u[i] = i < N - 1 ? ((uint32_t)w & MASK) : (uint32_t)w;
        }
#pragma unroll
        for (int i = 0; i < N; i++) f[i] = cy < 0 ? f[i] : u[i];
        f[N] = 0;
        F29 r;
#pragma unroll
        for (int k = 0; k < F29::N; k++) {
            const int bit = 29 * k, s = bit / LB, o = bit % LB;
            uint32_t w = s < N ? f[s] >> o : 0u;
            if (s + 1 <= N) w |= f[s + 1] << (LB - o);
            r.l[k] = w & F29::MASK;
        }
        return L29::scatter(r);
    }
};

}  // namespace ptau
