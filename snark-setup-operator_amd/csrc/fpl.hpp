// Prime-field element SPLIT OVER A LANE PAIR: the wide base fields (MNT4/MNT6-753: 26 limbs of 29 bits, BW6-761: 28) with
// the low half of the limbs in the even lane and the high half in the odd lane.
//
// Why (profiles/r02_start/): with one 24-27-limb element per lane a Jacobian point is 72-80 VGPRs before any temporary, the
// point operations live in private memory (3-10 KB of scratch per lane), the tau-update of a 2^18-point BW6-761 vector
// moves 427 GB through the L2 for 75 MB of algorithmic traffic and runs at 30 % of the integer-multiplier peak; a
// 2^16-point MNT chunk is 1024 waves, ONE per SIMD, with nothing to hide those loads behind.  Spread over two lanes an
// element is 13-14 VGPRs per lane (a G1 point 39-42), the whole point arithmetic stays in registers with the product as a
// register-ABI call, and the same chunk is twice as many waves.
//
// Same representation as Fp29<P> (fp29.hpp): Montgomery form with R = 2^(29 N), limbs of 29 bits, always fully reduced —
// the even lane holds limbs [0, H), the odd lane limbs [H, N), H = N / 2 — so a lane-split kernel and a single-lane
// kernel exchange values limb for limb (store_jac, kernels_impl.hpp).
//
// Product: operand scanning with the accumulator row distributed.  Iteration i adds a_i * b and m_i * p to N 64-bit
// accumulators (H per lane: 2H bare v_mad_u64_u32 per lane, no carry instruction — 2N products below 2^58 fit 64 bits),
// m_i comes from the even lane's position 0 and is broadcast by one DPP move, then the row shifts down one limb: the even
// lane's position 0 leaves its carry to position 1, the odd lane's lowest accumulator crosses to the even lane's top.
// MAC count per pair = 2 N^2, the same as the single-lane product scanning; the cross-lane price is ~12 plain
// instructions per iteration.  Carry chains (add, sub, the final reduction) run in both lanes at once with a zero
// carry-in; the even lane's carry-out is then folded into the odd lane's lowest limb(s) and only the (2^-29-rare) case
// where that ripples further takes a lane-local slow path — no second pass.
//
// Device only (DPP quad_perm [1,0,3,2]); every cross-lane move is executed unconditionally by both lanes of a pair, and
// every data-dependent branch in the callers is pair-uniform (is_zero / == combine both lanes).  Functions with no
// bearing on throughput (inverse, square root, byte I/O) gather the element into each lane and run the Fp29 code
// redundantly.
#pragma once
#include "fp29.hpp"

namespace ptau {

PTAU_HD uint32_t lane_parity() {
#if defined(__HIP_DEVICE_COMPILE__)
    return threadIdx.x & 1u;
#else
    return 0;
#endif
}
// A per-lane constant — c_lo in the even lane of a pair, c_hi in the odd lane — as ARITHMETIC on two instruction literals.
// Written as a select of two table reads (hi ? C[H + j] : C[j]) the compiler turns it into ONE read through a selected
// address: a vector memory load per limb per use, with its latency at the head of every product (five loads and a wait
// per call, 14 per subtraction: 400 loads per window of the MNT kernels, profiles/r03_*).  mask = all ones in the odd lane.
PTAU_HD uint32_t lane_hi_mask() { return 0u - lane_parity(); }
// The same mask as a value the optimiser cannot see through: constants formed from it are rebuilt where they are used (two
// cheap instructions per limb) instead of being hoisted out of the window loop and kept — i.e. spilled — across it.
PTAU_HD uint32_t lane_hi_mask_here() {
    uint32_t m = lane_hi_mask();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m));
#endif
    return m;
}
PTAU_HD uint32_t lane_const(uint32_t c_lo, uint32_t c_hi, uint32_t hi_mask) { return c_lo + ((c_hi - c_lo) & hi_mask); }
PTAU_HD uint32_t lane_xchg(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
#else
    return v;
#endif
}
// the even (odd) lane's value in both lanes of every pair: one DPP move, no select
PTAU_HD uint32_t lane_bcast_even(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0 /* quad_perm [0,0,2,2] */, 0xF, 0xF, true);
#else
    return v;
#endif
}
PTAU_HD uint32_t lane_bcast_odd(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5 /* quad_perm [1,1,3,3] */, 0xF, 0xF, true);
#else
    return v;
#endif
}

template <class P>
struct FpL {
    static constexpr int N = P::N;       // limbs of the whole element
    static constexpr int H = N / 2;      // limbs per lane
    static constexpr int N32 = P::N32;
    static constexpr int LB = 29;
    static constexpr uint32_t MASK = (1u << LB) - 1u;
    static constexpr int DEG = 1;
    static constexpr int LANES = 2;
    static constexpr int BYTES = P::BYTES;
    static constexpr bool MUL_OUT_OF_LINE = false;
    static constexpr bool POINT_OPS_INLINE = true;   // point operations inline around register-ABI product calls
    using Base = FpL<P>;
    using Params = P;
    using Full = Fp29<P>;                // the same element in one lane
    static_assert(P::LB == 29 && N % 2 == 0, "lane-split fields need an even number of 29-bit limbs");
    static_assert(2 * N <= 60, "accumulator row would overflow 64 bits");
    uint32_t l[H];

    PTAU_HD static bool hi_lane() { return lane_parity() != 0; }
    // this lane's half of a compile-time constant given as N limbs
    PTAU_HD static FpL load_const(const uint32_t *c) {
        const uint32_t hm = lane_hi_mask();
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = lane_const(c[j], c[H + j], hm);
        return r;
    }
    PTAU_HD static FpL zero() {
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = 0;
        return r;
    }
    PTAU_HD static FpL one() { return load_const(P::R1); }
    PTAU_HD static bool both(bool mine) {
        const uint32_t theirs = lane_xchg(mine ? 1u : 0u);
        return mine && theirs != 0u;
    }
    PTAU_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < H; j++) o |= l[j];
        return both(o == 0);
    }
    PTAU_HD bool operator==(const FpL &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < H; j++) o |= l[j] ^ b.l[j];
        return both(o == 0);
    }
    PTAU_HD bool operator!=(const FpL &b) const { return !(*this == b); }

    // the element in one lane (both lanes get all N limbs) and back
    PTAU_HD Full gather() const {
        const bool hi = hi_lane();
        Full f;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint32_t o = lane_xchg(l[j]);
            f.l[j] = hi ? o : l[j];
            f.l[H + j] = hi ? l[j] : o;
        }
        return f;
    }
    PTAU_HD static FpL scatter(const Full &f) {
        const bool hi = hi_lane();
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = hi ? f.l[H + j] : f.l[j];
        return r;
    }

    // ---- carry plumbing between the two lanes of a pair ----
    // Fold a carry (or borrow: signed) that arrived from the even lane into the odd lane's digits d[0..H): d[0] += cin, with
    // the ripple beyond d[0] on a lane-local slow path.  Returns the carry out of the top digit (0 almost always).
    PTAU_HD static int32_t fold_carry(uint32_t (&d)[H], int32_t cin) {
        int32_t v = (int32_t)d[0] + cin;
        int32_t c = v >> LB;
        d[0] = (uint32_t)v & MASK;
        if (c != 0) {   // |cin| <= 1 moved d[0] across a limb boundary: probability 2^-29 per operation, no cross-lane move inside
#pragma unroll
            for (int j = 1; j < H; j++) {
                v = (int32_t)d[j] + c;
                c = v >> LB;
                d[j] = (uint32_t)v & MASK;
            }
        }
        return c;
    }

    // a + b mod p
    PTAU_HD static FpL add_inl(const FpL &a, const FpL &b) {
        const bool hi = hi_lane();
        uint32_t d[H], s[H];
        int32_t cy = 0;
        uint32_t c2 = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint32_t ab = a.l[j] + b.l[j];                                   // < 2^30
            const uint32_t pj = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
            const int32_t v = (int32_t)ab - (int32_t)pj + cy;
            cy = v >> LB;
            d[j] = (uint32_t)v & MASK;
            const uint32_t w = ab + c2;
            c2 = w >> LB;
            s[j] = w & MASK;
        }
        // the even lane's carry-outs enter the odd lane's chains
        const int32_t cy_in = (int32_t)lane_xchg((uint32_t)cy);
        const int32_t c2_in = (int32_t)lane_xchg(c2);
        const int32_t top = cy + fold_carry(d, hi ? cy_in : 0);    // odd lane: the borrow of the whole a + b - p
        (void)fold_carry(s, hi ? c2_in : 0);
        const int32_t top_o = (int32_t)lane_xchg((uint32_t)top);
        const bool borrow = (hi ? top : top_o) < 0;                 // a + b < p: keep the sum
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = borrow ? s[j] : d[j];
        return r;
    }
    PTAU_HD static FpL sub_inl(const FpL &a, const FpL &b) {
        const bool hi = hi_lane();
        uint32_t d[H];
        int32_t cy = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const int32_t v = (int32_t)a.l[j] - (int32_t)b.l[j] + cy;
            cy = v >> LB;
            d[j] = (uint32_t)v & MASK;
        }
        const int32_t cy_in = (int32_t)lane_xchg((uint32_t)cy);
        const int32_t top = cy + fold_carry(d, hi ? cy_in : 0);
        const int32_t top_o = (int32_t)lane_xchg((uint32_t)top);
        const uint32_t mask = (hi ? top : top_o) < 0 ? 0xffffffffu : 0u;   // negative: add p (mod 2^(29 N))
        uint32_t c2 = 0;
        uint32_t r[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint32_t pj = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
            const uint32_t w = d[j] + (pj & mask) + c2;
            c2 = w >> LB;
            r[j] = w & MASK;
        }
        const int32_t c2_in = (int32_t)lane_xchg(c2);
        (void)fold_carry(r, hi ? c2_in : 0);
        FpL o;
#pragma unroll
        for (int j = 0; j < H; j++) o.l[j] = r[j];
        return o;
    }
    PTAU_HD static FpL neg(const FpL &a) {
        const bool hi = hi_lane();
        const uint32_t nz = a.is_zero() ? 0u : 0xffffffffu;
        uint32_t d[H];
        int32_t cy = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint32_t pj = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
            const int32_t v = (int32_t)(pj & nz) - (int32_t)a.l[j] + cy;
            cy = v >> LB;
            d[j] = (uint32_t)v & MASK;
        }
        const int32_t cy_in = (int32_t)lane_xchg((uint32_t)cy);
        (void)fold_carry(d, hi ? cy_in : 0);
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = d[j];
        return r;
    }
    PTAU_HD static FpL dbl(const FpL &a) { return add(a, a); }
#ifndef PTAU_FPL_ADD_CALLS
#define PTAU_FPL_ADD_CALLS 0   // 1: sums and differences as register-ABI calls instead of ~230 inlined instructions each — measured 2-4 % slower (BW6-761, MNT G1)
#endif
    typedef uint32_t VecA __attribute__((ext_vector_type(H)));
    __device__ __noinline__ static VecA add_vec(VecA a, VecA b) {
        FpL x, y;
#pragma unroll
        for (int j = 0; j < H; j++) { x.l[j] = a[j]; y.l[j] = b[j]; }
        const FpL r = add_inl(x, y);
        VecA o;
#pragma unroll
        for (int j = 0; j < H; j++) o[j] = r.l[j];
        return o;
    }
    __device__ __noinline__ static VecA sub_vec(VecA a, VecA b) {
        FpL x, y;
#pragma unroll
        for (int j = 0; j < H; j++) { x.l[j] = a[j]; y.l[j] = b[j]; }
        const FpL r = sub_inl(x, y);
        VecA o;
#pragma unroll
        for (int j = 0; j < H; j++) o[j] = r.l[j];
        return o;
    }
    PTAU_HD static FpL add(const FpL &a, const FpL &b) {
#if defined(__HIP_DEVICE_COMPILE__) && PTAU_FPL_ADD_CALLS
        VecA x, y;
#pragma unroll
        for (int j = 0; j < H; j++) { x[j] = a.l[j]; y[j] = b.l[j]; }
        const VecA o = add_vec(x, y);
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
        return r;
#else
        return add_inl(a, b);
#endif
    }
    PTAU_HD static FpL sub(const FpL &a, const FpL &b) {
#if defined(__HIP_DEVICE_COMPILE__) && PTAU_FPL_ADD_CALLS
        VecA x, y;
#pragma unroll
        for (int j = 0; j < H; j++) { x[j] = a.l[j]; y[j] = b.l[j]; }
        const VecA o = sub_vec(x, y);
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
        return r;
#else
        return sub_inl(a, b);
#endif
    }

    // ---- Montgomery product, accumulator row distributed over the pair (see the header comment) ----
    typedef uint32_t VecH __attribute__((ext_vector_type(H)));
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
            // limb i of a in both lanes: one DPP broadcast from the lane that holds it
            const uint32_t ai = i < H ? lane_bcast_even(a[i]) : lane_bcast_odd(a[i - H]);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], ai, b[j]);
            // m from the even lane's position 0, broadcast to the pair
            const uint32_t m = lane_bcast_even(((uint32_t)acc[0] * P::INV) & MASK);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], m, pm[j]);
            // Shift the row down one limb, the same code in both lanes: the lowest accumulator leaves its carry to the next
            // one and its low 29 bits to the partner, whose top slot they become.  Even lane: those bits are zero by the
            // choice of m, so the odd lane's top slot restarts at zero; odd lane: they are limb H of the row and arrive as
            // the even lane's new top slot.
            const uint64_t x = acc[0];
            const uint32_t recv = lane_xchg((uint32_t)x & MASK);
#pragma unroll
            for (int j = 0; j + 1 < H; j++) acc[j] = acc[j + 1];
            acc[0] += x >> LB;
            acc[H - 1] = recv;
        }
        row_finish(acc, pm, out);
    }
    // second half of a product: the N-position row -> digits, fully reduced
    PTAU_HD static void row_finish(const uint64_t (&acc)[H], const uint32_t (&pm)[H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        // carry-propagate both halves at once (carry-in 0), then fold the even lane's carry-out (< 2^36) into the odd
        // lane's two lowest digits
        uint32_t r[H];
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint64_t v = acc[j] + c;
            r[j] = (uint32_t)v & MASK;
            c = v >> LB;
        }
        {
            const uint32_t clo = lane_xchg((uint32_t)c & MASK), chi = lane_xchg((uint32_t)(c >> LB));
            const uint32_t w0 = r[0] + (hi ? clo : 0u);
            r[0] = w0 & MASK;
            const uint32_t w1 = r[1] + (hi ? chi : 0u) + (w0 >> LB);
            r[1] = w1 & MASK;
            uint32_t cc = w1 >> LB;
            if (cc) {   // rare: lane-local ripple
#pragma unroll
                for (int j = 2; j < H; j++) {
                    const uint32_t w = r[j] + cc;
                    r[j] = w & MASK;
                    cc = w >> LB;
                }
            }
        }
        // result < 2p: one conditional subtraction
        uint32_t d[H];
        int32_t cy = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const int32_t v = (int32_t)r[j] - (int32_t)pm[j] + cy;
            cy = v >> LB;
            d[j] = (uint32_t)v & MASK;
        }
        const int32_t cy_in = (int32_t)lane_xchg((uint32_t)cy);
        const int32_t top = cy + fold_carry(d, hi ? cy_in : 0);
        const int32_t top_o = (int32_t)lane_xchg((uint32_t)top);
        const bool borrow = (hi ? top : top_o) < 0;
#pragma unroll
        for (int j = 0; j < H; j++) out[j] = borrow ? r[j] : d[j];
    }
    // Square: the cross terms between the two halves are computed once, doubled (odd lane, iterations i < H), and the
    // high half's own square skips the products below its diagonal (iterations i >= H, where the even lane adds nothing):
    // H (H - 1) / 2 fewer MAC slots per lane than the product, the same column values.
    PTAU_HD static void row_sqr(const uint32_t (&a)[H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        uint32_t pm[H], az[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            pm[j] = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
            az[j] = hi ? a[j] : 0u;        // the row of iterations i >= H: the even lane contributes nothing there
        }
        const uint32_t dbl_odd = hi ? 1u : 0u;
        uint64_t acc[H];
#pragma unroll
        for (int j = 0; j < H; j++) acc[j] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (i < H) {
                // even lane: a_i * (low half), every j (both orders of a pair: no doubling); odd lane: 2 a_i * (high half)
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
            // m from the even lane's position 0, broadcast to the pair
            const uint32_t m = lane_bcast_even(((uint32_t)acc[0] * P::INV) & MASK);
#pragma unroll
            for (int j = 0; j < H; j++) mac64_vv(acc[j], m, pm[j]);
            // Shift the row down one limb, the same code in both lanes: the lowest accumulator leaves its carry to the next
            // one and its low 29 bits to the partner, whose top slot they become.  Even lane: those bits are zero by the
            // choice of m, so the odd lane's top slot restarts at zero; odd lane: they are limb H of the row and arrive as
            // the even lane's new top slot.
            const uint64_t x = acc[0];
            const uint32_t recv = lane_xchg((uint32_t)x & MASK);
#pragma unroll
            for (int j = 0; j + 1 < H; j++) acc[j] = acc[j + 1];
            acc[0] += x >> LB;
            acc[H - 1] = recv;
        }
        row_finish(acc, pm, out);
    }
    // one out-of-line copy per field, register ABI (H limbs per operand in VGPRs)
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
    PTAU_HD static FpL mul(const FpL &a, const FpL &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        VecH x, y;
#pragma unroll
        for (int j = 0; j < H; j++) { x[j] = a.l[j]; y[j] = b.l[j]; }
        const VecH o = mul_vec(x, y);
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
        return r;
#else
        (void)b;
        return a;   // lane-split fields exist on the device only
#endif
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
    // k squarings in a row in ONE call (the square roots' exponentiations: fp.hpp sqr_times)
    __device__ __noinline__ static VecH sqr_n_vec(VecH a, int k) {
        uint32_t x[H], r[H];
#pragma unroll
        for (int j = 0; j < H; j++) x[j] = a[j];
#pragma nounroll
        for (int t = 0; t < k; t++) {
            row_sqr(x, r);
#pragma unroll
            for (int j = 0; j < H; j++) x[j] = r[j];
        }
        VecH o;
#pragma unroll
        for (int j = 0; j < H; j++) o[j] = x[j];
        return o;
    }
    PTAU_HD static FpL sqr_n(const FpL &a, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
        VecH x;
#pragma unroll
        for (int j = 0; j < H; j++) x[j] = a.l[j];
        const VecH o = sqr_n_vec(x, k);
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
        return r;
#else
        (void)k;
        return a;
#endif
    }
    PTAU_HD static FpL sqr(const FpL &a) {
#if defined(__HIP_DEVICE_COMPILE__)
        VecH x;
#pragma unroll
        for (int j = 0; j < H; j++) x[j] = a.l[j];
        const VecH o = sqr_vec(x);
        FpL r;
#pragma unroll
        for (int j = 0; j < H; j++) r.l[j] = o[j];
        return r;
#else
        return a;
#endif
    }

    // r - p when r >= p, for r < 2p given as digits (both lanes' chains at once, the borrow folded across)
    PTAU_HD static void reduce_once(const uint32_t (&r)[H], const uint32_t (&pm)[H], uint32_t (&out)[H]) {
        const bool hi = hi_lane();
        uint32_t d[H];
        int32_t cy = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const int32_t v = (int32_t)r[j] - (int32_t)pm[j] + cy;
            cy = v >> LB;
            d[j] = (uint32_t)v & MASK;
        }
        const int32_t cy_in = (int32_t)lane_xchg((uint32_t)cy);
        const int32_t top = cy + fold_carry(d, hi ? cy_in : 0);
        const int32_t top_o = (int32_t)lane_xchg((uint32_t)top);
        const bool borrow = (hi ? top : top_o) < 0;
#pragma unroll
        for (int j = 0; j < H; j++) out[j] = borrow ? r[j] : d[j];
    }
    // k * a for a small positive constant k (4 <= k <= 15), without the double-and-add chain of modular additions (each ~230
    // instructions on a lane pair): the integer k a (one MAC per limb), a quotient estimate q from its top limbs against
    // p's (q <= floor(k a / p) <= q + 2), k a - q p (one MAC per limb), two conditional subtractions.  The non-residue and
    // curve-coefficient multiples of the extension fields (5, 11, 13, ...) are every third instruction of an Fq2 / Fq3 product.
    PTAU_HD static FpL mul_small_fast(const FpL &a, unsigned k) {
        constexpr int TL = (P::BITS - 1) / LB;   // the limb that holds p's top bits
        static_assert(TL >= H && TL < N, "p's top limb must sit in the odd lane");
        const bool hi = hi_lane();
        uint32_t pm[H], v[H];
#pragma unroll
        for (int j = 0; j < H; j++) pm[j] = lane_const(P::MOD[j], P::MOD[H + j], lane_hi_mask());
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const uint64_t acc = (uint64_t)a.l[j] * k + c;
            v[j] = (uint32_t)acc & MASK;
            c = acc >> LB;
        }
        const uint32_t c_even = lane_xchg((uint32_t)c);            // the even lane's carry-out enters the odd lane's digits
        uint32_t ctop = (uint32_t)c;                                 // odd lane: bits above 29 N
        ctop += (uint32_t)fold_carry(v, hi ? (int32_t)c_even : 0);
        // quotient estimate in the odd lane from the top ~29-60 bits, broadcast to the pair
        const uint64_t tv = (uint64_t)v[TL - H] | ((uint64_t)(TL + 1 < N ? v[TL + 1 - H < H ? TL + 1 - H : H - 1] : ctop) << LB);
        constexpr uint64_t step = (uint64_t)P::MOD[TL] + 1u;
        uint32_t q = 0;
#pragma unroll
        for (unsigned i = 1; i < 16; i++)
            if (i < k) q += tv >= i * step ? 1u : 0u;
        q = lane_bcast_odd(q);
        // w = k a - q p  (signed chain; the bits above 29 N cancel: the result is below 3p)
        uint32_t w[H];
        int64_t cy = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            const int64_t acc = (int64_t)v[j] - (int64_t)((uint64_t)q * pm[j]) + cy;
            w[j] = (uint32_t)acc & MASK;
            cy = acc >> LB;
        }
        const int32_t cy_even = (int32_t)lane_xchg((uint32_t)(int32_t)cy);
        (void)fold_carry(w, hi ? cy_even : 0);
        uint32_t r1[H];
        FpL r;
        reduce_once(w, pm, r1);
        reduce_once(r1, pm, r.l);
        return r;
    }
    PTAU_HD static FpL mul_small(const FpL &a, int k) {
        if (k == 0) return zero();
        bool ng = k < 0;
        unsigned uk = ng ? (unsigned)(-k) : (unsigned)k;
        if (uk >= 16 && uk < 32) {   // e.g. the MNT4 twist's a = 26 = 2 * 13
            FpL t = dbl(mul_small_fast(a, uk / 2));
            if (uk & 1u) t = add(t, a);
            return ng ? neg(t) : t;
        }
        if (uk >= 4 && uk < 16) {
            const FpL t = mul_small_fast(a, uk);
            return ng ? neg(t) : t;
        }
        FpL acc = zero(), base = a;
        bool first = true;
        while (uk) {
            if (uk & 1u) { acc = first ? base : add(acc, base); first = false; }
            uk >>= 1;
            if (uk) base = dbl(base);
        }
        return ng ? neg(acc) : acc;
    }

    // canonical value as N limbs of 29 bits (the same pointer in both lanes)
    PTAU_HD static FpL from_canonical(const uint32_t *c) { return mul(load_const(c), load_const(P::R2)); }
    PTAU_HD void to_canonical(uint32_t *c) const {
        uint32_t one_raw[N];
#pragma unroll
        for (int i = 0; i < N; i++) one_raw[i] = i == 0 ? 1u : 0u;
        const Full f = mul(*this, load_const(one_raw)).gather();
#pragma unroll
        for (int i = 0; i < N; i++) c[i] = f.l[i];
    }
    // off the hot path: the single-lane code, run redundantly by both lanes
    PTAU_HD static FpL inverse(const FpL &a) { return scatter(Full::inverse(a.gather())); }
    __device__ __noinline__ static FpL pow_limbs(const FpL &a, const uint32_t *e, int top_bit) {
        return pow_sliding(a, e, top_bit);
    }
    // Square root on the pair: Fp::sqrt_ex's algorithm (one exponentiation + Pohlig-Hellman in the 2-Sylow subgroup) with the
    // lane-split product / square; the table key is the element's lowest limb (even lane), broadcast to the pair.
    __device__ __noinline__ static bool sqrt_ex(const FpL &a, FpL &out, FpL *inv_out) {
        if (a.is_zero()) {
            out = zero();
            if (inv_out) *inv_out = zero();
            return true;
        }
        const FpL w = pow_limbs(a, P::TS_TM1H, P::TS_TM1H_BITS - 1);
        const FpL x = mul(a, w);
        FpL b = mul(x, w);
        FpL gd = one();
        constexpr int E = P::TS_S - 1, W = P::SQRT_W;
        bool ok = true;
        for (int j = 0; j < P::SQRT_L; j++) {
            const int wj = (E - W * j) < W ? (E - W * j) : W;
            const int sh = E - W * j - wj;
            FpL t = b;
            t = sqr_times(t, sh);
            const uint32_t lo = lane_bcast_even(t.l[0]);
            int K = -1;
            for (int k = 0; k < (1 << W); k++) K = (P::SQRT_ZLOW[k] == lo) ? k : K;
            if (K < 0) { ok = false; K = 0; }   // a^T outside <g^2>: non-residue (keep going: every lane of the pair runs the same calls)
            const int d = K >> (W - wj);
            const FpL f = load_const(P::SQRT_GNEG[(j << W) + d]);
            gd = mul(gd, f);
            b = mul(b, sqr(f));
        }
        out = mul(x, gd);
        if (sqr(out) != a) ok = false;
        if (inv_out) *inv_out = mul(out, sqr(mul(w, gd)));
        return ok;
    }
    // Fp29::sqrt_or_nr on the pair (the square roots of Fq[u]/(u^2 - nr), ext.hpp): out = sqrt(a), residue = true, or
    // out = sqrt(nr a), residue = false; inv_out = 1/out.  The lookup key is broadcast from the even lane, so both lanes of a
    // pair take the same branches.
    static constexpr bool HAS_SQRT_OR_NR = P::SQRT_NR != 0 && P::SQRT_W > 0;
    __device__ __noinline__ static bool sqrt_or_nr(const FpL &a, FpL &out, FpL &inv_out, bool &residue) {
        residue = true;
        if (a.is_zero()) {
            out = zero();
            inv_out = zero();
            return true;
        }
        FpL w = pow_limbs(a, P::TS_TM1H, P::TS_TM1H_BITS - 1);
        FpL aa = a;
        constexpr int E = P::TS_S - 1, W = P::SQRT_W;
        for (int attempt = 0; attempt < 2; attempt++) {
            const FpL x = mul(aa, w);
            FpL b = mul(x, w);
            FpL gd = one();
            bool fail = false;
            for (int j = 0; j < P::SQRT_L; j++) {
                const int wj = (E - W * j) < W ? (E - W * j) : W;
                const int sh = E - W * j - wj;
                FpL t = b;
                t = sqr_times(t, sh);
                const uint32_t lo = lane_bcast_even(t.l[0]);
                int K = -1;
                for (int k = 0; k < (1 << W); k++) K = (P::SQRT_ZLOW[k] == lo) ? k : K;
                if (K < 0) { fail = true; break; }
                const int d = K >> (W - wj);
                const FpL f = load_const(P::SQRT_GNEG[(j << W) + d]);
                gd = mul(gd, f);
                b = mul(b, sqr(f));
            }
            if (!fail) {
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
    PTAU_HD static bool sqrt(const FpL &a, FpL &out) { return sqrt_ex(a, out, nullptr); }
    PTAU_HD static bool read_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, FpL &out) {
        Full f;
        const bool ok = Full::read_bytes(in, with_flags, flags, f);   // both lanes read the same bytes
        out = scatter(f);
        return ok;
    }
    PTAU_HD void write_bytes(uint8_t *out, uint8_t flags) const {
        const Full f = gather();
        if (!hi_lane()) f.write_bytes(out, flags);
    }
    PTAU_HD int sign_class() const { return gather().sign_class(); }
};

}  // namespace ptau
