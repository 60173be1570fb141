// Carry-free ("lazy") prime-field arithmetic for the inner loop of K1 on gfx950: 28-bit limbs, ONE LIMB MORE than the
// modulus needs.
//
// Why (tools/ubench_valu.hip, profiles/r02_ubench_valu.txt): next to a v_mad_u64_u32 (5.2 cycles per wave64) a plain VALU
// instruction is not free — 2.4-3.1 cycles for add / sub / and / right shift, 4.3-4.6 for every three-operand integer
// form, 64-bit shift, carry through VCC, select — and a modular addition of the fully reduced 13 x 29-bit field is a ~100
// instruction carry chain plus a conditional subtraction (~350 cycles, a fifth of a square).  A doubling has fourteen of
// them: 26 % of its time.  Here R = 2^(28 N) >= 2^15 p, so
//   * a sum is N limb-wise additions, a difference N limb-wise a_i + (K p)_i - b_i against a multiple of p whose limbs are
//     pre-biased (no borrow can occur), a small multiple is a shift — no carry propagates, nothing is compared with p;
//   * the product accepts such operands as they are (limbs up to 2^32, values up to 2^14 p) and returns limbs < 2^28 and a
//     value below 2p WITHOUT a final conditional subtraction: (a b + m p) / R < a b / R + p;
//   * carries are propagated (`norm`, 3 cheap instructions per limb) only where a limb bound would otherwise overflow.
// The price is one more limb in the product: 378 multiply-accumulates instead of 325.
//
// Overflow safety is a COMPILE-TIME property: an element carries its bounds in its type, E<U, V> = "every limb below
// U 2^28, value below V p"; each operation computes the bounds of its result and static_asserts what it needs
// (product: U1 U2 <= 16 keeps a column sum below 2^64; V1 V2 <= 2^15 keeps the result below 2p).
// Same Montgomery residue as Fp29 up to the constant factor R / R29: from29 / to29 convert by one product.
// Used between the point read and the result store of window_scalar_mul_kernel (table and window loop), by the phi / psi
// membership kernels and by the G2 bucket accumulation (kernels_impl.hpp, LAZY_LOOP); the reference paths they serve are
// batch_exp, check_subgroup and dense_multiexp — /root/reference/src/bin/contribute.rs:811-823, 140-145, 968-986.
#pragma once
#include "fp29.hpp"
#include "fpl.hpp"   // lane_parity / lane_xchg / lane_bcast_*

namespace ptau {

template <class P>
struct FpZ {
    static constexpr int N = P::N;
    static constexpr int LB = P::LB;
    static constexpr uint32_t MASK = (1u << LB) - 1u;
    using F29 = Fp29<typename P::P29>;
    static_assert(LB == 28 && P::MOD[0] == 1u && P::INV == MASK, "written for p = 1 mod 2^28");
    static_assert(LB * N >= F29::Params::BITS + 15, "R must exceed 2^15 p");
    static_assert(P::MOD[N - 1] >= 16, "the biased multiples of p (SubC) borrow up to 15 from p's top limb");

    template <int U, int V>
    struct E {
        static_assert(U >= 1 && U <= 15, "a limb would not fit 32 bits");
        static_assert(V >= 1 && V <= (1 << 14), "the value would not stay below R");
        static constexpr int LIMB_BOUND = U, VALUE_BOUND = V;
        uint32_t l[N];
        // widening only: a tighter bound is also a looser one
        template <int U2, int V2>
        PTAU_HD operator E<U2, V2>() const {
            static_assert(U2 >= U && V2 >= V, "bounds can only be loosened");
            E<U2, V2> r;
#pragma unroll
            for (int i = 0; i < N; i++) r.l[i] = l[i];
            return r;
        }
    };
    using Nrm = E<1, 2>;   // what a product returns

    struct Limbs { uint32_t l[N]; };
    // K p with every limb below the top raised by UB 2^28, each raise borrowed from the next limb up: the same integer,
    // and limb i >= UB 2^28 - UB >= any limb below (UB - 1) 2^28; the top limb (K p >> 364) - UB still exceeds the top limb
    // of any value below (K - 1) p.
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

    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2, V1 + V2> add(const E<U1, V1> &a, const E<U2, V2> &b) {
        E<U1 + U2, V1 + V2> r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    // a - b + (V2 + 1) p: non-negative limb by limb
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2 + 2, V1 + V2 + 1> sub(const E<U1, V1> &a, const E<U2, V2> &b) {
        E<U1 + U2 + 2, V1 + V2 + 1> r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = a.l[i] + SubC<V2 + 1, U2 + 1>::v.l[i] - b.l[i];
        return r;
    }
    template <int U, int V>
    PTAU_HD static E<U + 2, V + 1> neg(const E<U, V> &a) {
        E<U + 2, V + 1> r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = SubC<V + 1, U + 1>::v.l[i] - a.l[i];
        return r;
    }
    template <int K, int U, int V>
    PTAU_HD static E<K * U, K * V> times(const E<U, V> &a) {
        static_assert(K == 2 || K == 3 || K == 4 || K == 5 || K == 8, "small multiples used by the point formulas");
        E<K * U, K * V> r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if constexpr (K == 2) r.l[i] = a.l[i] + a.l[i];
            else if constexpr (K == 3) r.l[i] = a.l[i] + a.l[i] + a.l[i];
            else if constexpr (K == 4) r.l[i] = a.l[i] << 2;
            else if constexpr (K == 5) r.l[i] = (a.l[i] << 2) + a.l[i];
            else r.l[i] = a.l[i] << 3;
        }
        return r;
    }
    // propagate carries: limbs < 2^28 again (the top limb keeps the value's high part: < V 2^13)
    template <int U, int V>
    PTAU_HD static E<1, V> norm(const E<U, V> &a) {
        E<1, V> r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
            const uint32_t w = a.l[i] + c;   // < U 2^28 + 16 <= 2^32
            r.l[i] = w & MASK;
            c = w >> LB;
        }
        r.l[N - 1] = a.l[N - 1] + c;
        return r;
    }

    // Value reduction without a product: q = floor(top limb / (p's top limb + 1)) never exceeds value / p and misses it by
    // less than 1 + V / 6884, so a - q p lies in [0, 3p) for V <= 256; limbs come out normalised (one borrow chain).
    template <int V>
    PTAU_HD static E<1, 3> vreduce(const E<1, V> &a) {
        static_assert(V <= 256, "quotient estimate analysed for values below 256 p");
        constexpr uint32_t D = P::MOD[N - 1] + 1u;
        constexpr uint32_t MAGIC = (uint32_t)((1ull << 32) / D);           // floor: the estimate can only fall short
        static_assert((uint64_t)256 * D < (1ull << 32), "top limb of the operand must fit the 32-bit estimate");
        const uint32_t q = (uint32_t)(((uint64_t)a.l[N - 1] * MAGIC) >> 32);   // in {floor(top / D) - 1, floor(top / D)}
        E<1, 3> r;
        uint64_t c = 0;
        int32_t b = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint64_t t = (uint64_t)q * P::MOD[i] + c;               // limb i of q p
            c = t >> LB;
            const int32_t v = (int32_t)a.l[i] - (int32_t)(i < N - 1 ? ((uint32_t)t & MASK) : (uint32_t)t) + b;
            b = v >> LB;
            r.l[i] = i < N - 1 ? ((uint32_t)v & MASK) : (uint32_t)v;
        }
        return r;
    }

    // Montgomery product / square, product scanning, no final subtraction.  Column sums: at most N products of operand
    // limbs (< U1 U2 2^56 each, U1 U2 <= 16) and N products m_j p_k (< 2^56): below 238 2^56 < 2^64.
    template <bool SQR>
    PTAU_HD static void ps_mul(const uint32_t (&a)[N], const uint32_t (&b)[N], uint32_t (&out)[N]) {
        uint32_t m[N], d[N];
        if constexpr (SQR) {
#pragma unroll
            for (int i = 0; i < N; i++) d[i] = a[i] + a[i];
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
            m[i] = (0u - (uint32_t)acc) & MASK;   // INV = -1
            acc += m[i];                           // MOD[0] = 1
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
            out[i - N] = i == 2 * N - 1 ? (uint32_t)acc : ((uint32_t)acc & MASK);
            acc >>= LB;
        }
    }
    // (a b + c d) / R mod p with ONE reduction (product scanning, both products in every column): 2 N^2 + N (N - 1) multiply-
    // accumulates instead of the 2 (N^2 + N (N - 1)) of two reduced products.  Column sums: N products a_j b_k (< U1 U2 2^56),
    // N products c_j d_k (< U3 U4 2^56) and N products m_j p_k: U1 U2 + U3 U4 <= 16 keeps them below 2^64.
    PTAU_HD static void ps_mul2(const uint32_t (&a)[N], const uint32_t (&b)[N], const uint32_t (&c)[N], const uint32_t (&d)[N],
                                uint32_t (&out)[N]) {
        uint32_t m[N];
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) {
                mac64_vv(acc, a[j], b[i - j]);
                mac64_vv(acc, c[j], d[i - j]);
            }
#pragma unroll
            for (int j = 0; j < i; j++) mac64_vs(acc, m[j], P::MOD[i - j]);
            m[i] = (0u - (uint32_t)acc) & MASK;   // INV = -1
            acc += m[i];                           // MOD[0] = 1
            acc >>= LB;
        }
#pragma unroll
        for (int i = N; i < 2 * N; i++) {
#pragma unroll
            for (int j = i - N + 1; j < N; j++) {
                mac64_vv(acc, a[j], b[i - j]);
                mac64_vv(acc, c[j], d[i - j]);
            }
#pragma unroll
            for (int j = i - N + 1; j < N; j++) mac64_vs(acc, m[j], P::MOD[i - j]);
            out[i - N] = i == 2 * N - 1 ? (uint32_t)acc : ((uint32_t)acc & MASK);
            acc >>= LB;
        }
    }
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
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static Nrm mul(const E<U1, V1> &a, const E<U2, V2> &b) {
        static_assert(U1 * U2 <= 16, "a column of the product would overflow 64 bits: normalise an operand first");
        static_assert(V1 * V2 <= (1 << 15), "the product would not come back below 2p");
        VecN x, y;
#pragma unroll
        for (int i = 0; i < N; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
        const VecN o = mul_vec(x, y);
        Nrm r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = o[i];
        return r;
    }
    template <int U, int V>
    PTAU_HD static Nrm sqr(const E<U, V> &a) {
        static_assert(U * U <= 16, "a column of the square would overflow 64 bits: normalise the operand first");
        static_assert(V * V <= (1 << 15), "the square would not come back below 2p");
        VecN x;
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = a.l[i];
        const VecN o = sqr_vec(x);
        Nrm r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = o[i];
        return r;
    }

    PTAU_HD static Nrm zero() {
        Nrm r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    PTAU_HD static Nrm one() {
        Nrm r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R1[i];
        return r;
    }
    // value = 0 mod p?  A normalised value below V p is a multiple j p only if its lowest limb is j (p = 1 mod 2^28):
    // one comparison settles all but a 2^-28-rare case, which is compared limb by limb with j p.
    template <int V>
    PTAU_HD static bool is_zero(const E<1, V> &a) {
        const uint32_t j = a.l[0];
        if (j >= (uint32_t)V) return false;
        uint64_t c = 0;
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint64_t t = (uint64_t)j * P::MOD[i] + c;
            diff |= a.l[i] ^ (i < N - 1 ? (uint32_t)(t & MASK) : (uint32_t)t);
            c = t >> LB;
        }
        return diff == 0;
    }
    // literally all limbs zero (the table's marker for the point at infinity; from29(0) is exactly that)
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero(const E<U, V> &a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= a.l[i];
        return o == 0;
    }

    template <int U, int V>
    PTAU_HD static bool limbs_all_zero_local(const E<U, V> &a) { return limbs_all_zero(a); }   // (one lane holds the element)

    // Fp29 (fully reduced, R29 = 2^(29 N29)) <-> this representation
    PTAU_HD static Nrm from29(const F29 &a) {
        E<1, 1> t;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int bit = LB * k, s = bit / 29, o = bit % 29;
            uint32_t v = s < F29::N ? a.l[s] >> o : 0u;
            if (o + LB > 29 && s + 1 < F29::N) v |= a.l[s + 1] << (29 - o);
            t.l[k] = v & MASK;
        }
        E<1, 1> c;
#pragma unroll
        for (int i = 0; i < N; i++) c.l[i] = P::FROM29[i];
        return mul(t, c);
    }
    template <int U, int V>
    PTAU_HD static F29 to29(const E<U, V> &a) {
        static_assert(U <= 16 && V <= (1 << 15), "operand of the converting product out of range");
        E<1, 1> c;
#pragma unroll
        for (int i = 0; i < N; i++) c.l[i] = P::TO29[i];
        const Nrm v = mul(a, c);
        // below 2p: one conditional subtraction makes it canonical
        uint32_t u[N];
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t w = (int32_t)v.l[i] - (int32_t)P::MOD[i] + cy;
            cy = w >> LB;
            u[i] = i < N - 1 ? ((uint32_t)w & MASK) : (uint32_t)w;
        }
        uint32_t z[N + 1];
#pragma unroll
        for (int i = 0; i < N; i++) z[i] = cy < 0 ? v.l[i] : u[i];
        z[N] = 0;
        F29 r;
#pragma unroll
        for (int k = 0; k < F29::N; k++) {
            const int bit = 29 * k, s = bit / LB, o = bit % LB;
            uint32_t w = z[s] >> o;
            if (s + 1 <= N) w |= z[s + 1] << (LB - o);
            r.l[k] = w & F29::MASK;
        }
        return r;
    }
};

// Jacobian arithmetic on y^2 = x^3 + b (a = 0) over FpZ: the window loop's doubling and mixed addition with every sum and
// difference carry-free.  The bounds in the member types are closed under both operations (checked by the compiler through
// E<>'s static_asserts): X below 31p, Y below 19p, Z below 2p, limbs normalised.
template <class Z>
struct LazyJac {
    using Nrm = typename Z::Nrm;
    using XT = typename Z::template E<1, 31>;
    using YT = typename Z::template E<1, 19>;
    using AX = typename Z::template E<1, 3>;   // affine x of the point added (a table entry, or a reduced coordinate)
    using AY = typename Z::template E<3, 4>;   // its y, possibly negated on the fly
    XT X;
    YT Y;
    Nrm Zc;   // = 0 mod p  <=>  point at infinity

    PTAU_HD static LazyJac infinity() { return {Z::one(), Z::one(), Z::zero()}; }
    PTAU_HD bool is_inf() const { return Z::is_zero(Zc); }

    // dbl-2009-l
    PTAU_HD static LazyJac dbl(const LazyJac &p) {
        const auto A = Z::sqr(p.X);
        const auto B = Z::sqr(p.Y);
        const auto C = Z::sqr(B);
        const auto T2 = Z::sqr(Z::add(p.X, B));
        const auto D = Z::norm(Z::template times<2>(Z::sub(T2, Z::add(A, C))));   // 2((X+B)^2 - A - C)
        const auto E3 = Z::template times<3>(A);
        const auto F = Z::sqr(E3);
        LazyJac r;
        r.Zc = Z::mul(Z::template times<2>(p.Y), p.Zc);
        r.X = Z::norm(Z::sub(F, Z::template times<2>(D)));
        r.Y = Z::norm(Z::sub(Z::mul(E3, Z::sub(D, r.X)), Z::template times<8>(C)));
        return r;
    }

    // The same addition ORDERED FOR LIVENESS, as LazyJac2::add_src: the entry's coordinates come from `src` — BOTH read at the
    // top (one memory latency for the pair; a read per use exposed a latency each: measured 10 % slower on BLS12-377 G1) and
    // dead after the second and fourth product; Z3 = 2 Z1 H is taken as soon as H exists (Z1 and Z1Z1 retire); the exceptional
    // cases are decided by two early flags and REBUILT FROM THE ENTRY (src.x_again(), src.y_again(): a re-read on the rare
    // path) at the single join, so neither p's nor the entry's coordinates stay alive for them.  At most seven elements are
    // alive at any product call — what the lane-split field (BW6-761: 14 VGPRs per element and lane, register-ABI product
    // calls, eight elements of callee-saved registers) needs to keep the window loop out of scratch.
    template <class Src>
    PTAU_HD static LazyJac add_src(const LazyJac &p, const Src &src) {
        const AX x2 = src.x();
        const AY y2 = src.y();
        const bool pinf = p.is_inf();
        const auto Z1Z1 = Z::sqr(p.Zc);
        const auto H = Z::norm(Z::sub(Z::mul(x2, Z1Z1), p.X));
        const auto rh = Z::norm(Z::sub(Z::mul(Z::mul(y2, p.Zc), Z1Z1), p.Y));
        const bool hz = Z::is_zero(H), rz = Z::is_zero(rh);
        LazyJac r;
        r.Zc = Z::mul(Z::template times<2>(p.Zc), H);
        const auto I = Z::template times<4>(Z::sqr(H));
        const auto J = Z::mul(H, I);
        const auto Vv = Z::mul(p.X, I);
        const auto rr = Z::template times<2>(rh);
        r.X = Z::norm(Z::sub(Z::sqr(rr), Z::add(J, Z::template times<2>(Vv))));
        r.Y = Z::norm(Z::sub(Z::mul(rr, Z::sub(Vv, r.X)), Z::template times<2>(Z::mul(p.Y, J))));
        if (pinf || hz) {   // one join, rare paths, everything rebuilt from the entry
            const LazyJac q = {src.x_again(), Z::norm(src.y_again()), Z::one()};
            if (pinf) r = q;
            else if (rz) r = dbl(q);
            else r = infinity();
        }
        return r;
    }

    // madd-2007-bl with Z3 = 2 Z1 H; q = (x2, y2) affine and not the point at infinity
    PTAU_HD static LazyJac add_mixed(const LazyJac &p, const AX &x2, const AY &y2) {
        const auto Z1Z1 = Z::sqr(p.Zc);
        const auto U2 = Z::mul(x2, Z1Z1);
        const auto S2 = Z::mul(Z::mul(y2, p.Zc), Z1Z1);
        const auto H = Z::norm(Z::sub(U2, p.X));
        const auto rh = Z::norm(Z::sub(S2, p.Y));
        const auto rr = Z::template times<2>(rh);
        const auto HH = Z::sqr(H);
        const auto I = Z::template times<4>(HH);
        const auto J = Z::mul(H, I);
        const auto Vv = Z::mul(p.X, I);
        LazyJac r;
        r.X = Z::norm(Z::sub(Z::sqr(rr), Z::add(J, Z::template times<2>(Vv))));
        r.Y = Z::norm(Z::sub(Z::mul(rr, Z::sub(Vv, r.X)), Z::template times<2>(Z::mul(p.Y, J))));
        r.Zc = Z::mul(Z::template times<2>(p.Zc), H);
        // the exceptional cases (p = O, q = +-p) overwrite the generic result: one join, rare path (with early returns the
        // accumulator went through scratch memory in the one-lane kernel; on the lane-split field the two forms time the same)
        const bool pinf = p.is_inf();
        if (pinf || Z::is_zero(H)) {
            if (pinf) r = {x2, Z::norm(y2), Z::one()};
            else if (Z::is_zero(rh)) r = dbl(p);
            else r = infinity();
        }
        return r;
    }
};

// Fq2 = Fq[u] / (u^2 + 5) over a LANE PAIR — a0 in the even lane, a1 in the odd lane, as Fp2L (ext.hpp) — with the
// coefficients in FpZ.  Both lanes run the same instructions; what differs per lane is data: the partner's coefficient
// arrives by one DPP move per limb (quad_perm exchange / broadcast), and the per-lane choice between the two combination
// rules is a lane mask or a per-lane shift amount, not a branch.
//   product:  even  a0 b0 - 5 a1 b1     odd  a1 b0 + a0 b1         (two base products per lane)
//   square :  even  (a0 + a1)(a0 - 5 a1) + 4 a0 a1     odd  2 a0 a1   (ONE base product per lane)
// Results are normalised, values below 13p / 10p; the base product's bounds (U1 U2 <= 16, V1 V2 <= 2^15) are checked at
// compile time on the operands actually formed.
template <class Z, int NR>
struct Fp2Z {
    static_assert(NR == -5, "combination rules written for the non-residue -5 (BLS12-377)");
    static constexpr int N = Z::N;
    template <int U, int V>
    struct E {
        typename Z::template E<U, V> c;
        template <int U2, int V2>
        PTAU_HD operator E<U2, V2>() const { return {c}; }
    };
    using M = E<1, 2>;    // a product (one fused two-term product per lane, one reduction: below 2p)
    using S = E<1, 10>;   // a square
    using R = E<1, 3>;    // after reduce()
    using Nrm = typename Z::Nrm;

    PTAU_HD static bool both(bool mine) {   // executed by both lanes of a pair, unconditionally
        const uint32_t theirs = lane_xchg(mine ? 1u : 0u);
        return mine && theirs != 0u;
    }
    template <class T>
    PTAU_HD static T other(const T &v) {   // the partner lane's coefficient
        T r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = lane_xchg(v.l[i]);
        return r;
    }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2, V1 + V2> add(const E<U1, V1> &a, const E<U2, V2> &b) { return {Z::add(a.c, b.c)}; }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static E<U1 + U2 + 2, V1 + V2 + 1> sub(const E<U1, V1> &a, const E<U2, V2> &b) { return {Z::sub(a.c, b.c)}; }
    template <int U, int V>
    PTAU_HD static E<U + 2, V + 1> neg(const E<U, V> &a) { return {Z::neg(a.c)}; }
    template <int K, int U, int V>
    PTAU_HD static E<K * U, K * V> times(const E<U, V> &a) { return {Z::template times<K>(a.c)}; }
    template <int U, int V>
    PTAU_HD static E<1, V> norm(const E<U, V> &a) { return {Z::norm(a.c)}; }
    template <int U, int V>
    PTAU_HD static R reduce(const E<U, V> &a) { return {Z::vreduce(Z::norm(a.c))}; }
    // (a0, -a1)
    template <int U, int V>
    PTAU_HD static E<U + 2, V + 1> conj(const E<U, V> &a) {
        const bool odd = lane_parity() != 0;
        const auto n = Z::neg(a.c);
        E<U + 2, V + 1> r;
#pragma unroll
        for (int i = 0; i < N; i++) r.c.l[i] = odd ? n.l[i] : a.c.l[i];
        return r;
    }
    // by an element of Fq (the same value in both lanes)
    template <int U, int V, class KT>
    PTAU_HD static E<1, 2> mul_base(const E<U, V> &a, const KT &k) { return {Z::mul(a.c, k)}; }

    // even lane: a0 b0 - 5 a1 b1 = a0 b0 + (5 (KN p - a1)) b1      odd lane: a1 b0 + a0 b1 — ONE fused two-term product per lane
    // (FpZ::ps_mul2, one reduction).  The negation uses a FIXED multiple KN p (one out-of-line copy serves every call site):
    // the operand passed as `a` must be normalised with a value below (KN - 1) p.  Register ABI: 2 x N VGPRs in, N out; the
    // partner's coefficients arrive by DPP inside.
    static constexpr int KN = 32;
    typedef typename Z::VecN VecN;
    __device__ __noinline__ static VecN mul2_vec(VecN a, VecN b) {
        const bool odd = lane_parity() != 0;
        uint32_t x1[N], y1[N], x2[N], y2[N], r[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t ao = lane_xchg(a[i]);
            const uint32_t ng = Z::template SubC<KN, 2>::v.l[i] - ao;        // limb of KN p - a1 (biased: no borrow)
            x1[i] = a[i];
            y1[i] = lane_bcast_even(b[i]);
            x2[i] = odd ? ao : (ng << 2) + ng;                                 // odd: a0     even: 5 (KN p - a1)
            y2[i] = lane_bcast_odd(b[i]);
        }
        Z::ps_mul2(x1, y1, x2, y2, r);
        VecN o;
#pragma unroll
        for (int i = 0; i < N; i++) o[i] = r[i];
        return o;
    }
    template <class T>
    PTAU_HD static auto nrm1(const T &v) {
        if constexpr (T::LIMB_BOUND == 1) return v;
        else return Z::norm(v);
    }
    template <int U1, int V1, int U2, int V2>
    PTAU_HD static M mul(const E<U1, V1> &a, const E<U2, V2> &b) {
        // scan-side choice: the operand whose value bound fits the fixed negation constant goes first (the product commutes)
        if constexpr (V1 > KN - 1 && V2 <= KN - 1) {
            return mul(b, a);
        } else {
            static_assert(V1 <= KN - 1, "fused Fq2 product: one operand must stay below (KN - 1) p");
            // column units: 1 (a0 b0) + 15 (5 x the biased negation, limbs below 3 x 2^28) = 16: both operands normalised
            static_assert((long long)(V1 + 5 * (KN + 1)) * V2 <= (1 << 15), "the fused product would not come back below 2p");
            M r;
#if defined(__HIP_DEVICE_COMPILE__)
            const auto an = nrm1(a.c);
            const auto bn = nrm1(b.c);
            VecN x, y;
#pragma unroll
            for (int i = 0; i < N; i++) { x[i] = an.l[i]; y[i] = bn.l[i]; }
            const VecN o = mul2_vec(x, y);
#pragma unroll
            for (int i = 0; i < N; i++) r.c.l[i] = o[i];
#else
            (void)b;
#pragma unroll
            for (int i = 0; i < N; i++) r.c.l[i] = a.c.l[i];
#endif
            return r;
        }
    }
    template <int V>
    PTAU_HD static S sqr(const E<1, V> &a) {
        const uint32_t em = lane_parity() ? 0u : 0xffffffffu;   // all ones in the even lane
        const auto ot = other(a.c);
        const auto m5 = Z::sub(a.c, Z::template times<5>(ot));  // even: a0 - 5 a1 (+ a multiple of p)   <8, 6V + 1>
        typename Z::template E<2, 2 * V> l;
        typename Z::template E<8, 6 * V + 1> r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            l.l[i] = ot.l[i] + (a.c.l[i] & em);                 // even a0 + a1     odd a0
            r.l[i] = (m5.l[i] & em) | (a.c.l[i] & ~em);         // even a0 - 5 a1   odd a1
        }
        const Nrm t = Z::mul(l, r);                             // even (a0 + a1)(a0 - 5 a1)   odd a0 a1
        // even: t + 4 v with v = the odd lane's t; odd: t + t.  One broadcast and one shift-add with a per-lane shift amount.
        const uint32_t sh = em & 2u;
        typename Z::template E<5, 10> o;
#pragma unroll
        for (int i = 0; i < N; i++) o.l[i] = t.l[i] + (lane_bcast_odd(t.l[i]) << sh);
        return {Z::norm(o)};
    }
    PTAU_HD static E<1, 2> zero() { return {Z::zero()}; }
    PTAU_HD static E<1, 2> one() {
        const bool odd = lane_parity() != 0;
        const Nrm o = Z::one();
        E<1, 2> r;
#pragma unroll
        for (int i = 0; i < N; i++) r.c.l[i] = odd ? 0u : o.l[i];
        return r;
    }
    template <int V>
    PTAU_HD static bool is_zero(const E<1, V> &a) { return both(Z::is_zero(a.c)); }
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero(const E<U, V> &a) { return both(Z::limbs_all_zero(a.c)); }
    template <int U, int V>
    PTAU_HD static bool limbs_all_zero_local(const E<U, V> &a) { return Z::limbs_all_zero(a.c); }   // this lane's coefficient only
};

// Jacobian arithmetic on y^2 = x^3 + b' over Fp2Z (a = 0).  Coordinates between operations: X and Y reduced below 3p, Z
// below 31p, all normalised; every intermediate bound is in the types.
template <class K>
struct LazyJac2 {
    using XT = typename K::R;
    using YT = typename K::R;
    using ZT = typename K::template E<1, 31>;   // a product (doubling) or Z1 + H squared minus two squares (addition)
    using AX = typename K::M;                   // affine x of a table entry (a product; possibly after psi^j)
    using AY = typename K::template E<3, 14>;   // affine y, possibly negated on the fly
    XT X;
    YT Y;
    ZT Zc;

    PTAU_HD static LazyJac2 infinity() { return {K::one(), K::one(), K::zero()}; }
    PTAU_HD bool is_inf() const { return K::is_zero(Zc); }

    PTAU_HD static LazyJac2 dbl(const LazyJac2 &p) {
        const auto A = K::sqr(p.X);
        const auto B = K::sqr(p.Y);
        const auto C = K::sqr(B);
        const auto T2 = K::sqr(K::norm(K::add(p.X, B)));
        const auto D = K::norm(K::template times<2>(K::sub(T2, K::add(A, C))));   // <1, 62>
        const auto E3 = K::norm(K::template times<3>(A));                          // <1, 30>
        const auto F = K::sqr(E3);
        LazyJac2 r;
        r.Zc = K::mul(K::template times<2>(p.Y), p.Zc);
        r.X = K::reduce(K::sub(F, K::template times<2>(D)));
        r.Y = K::reduce(K::sub(K::mul(E3, K::sub(D, r.X)), K::template times<8>(C)));
        return r;
    }
    // The second operand of an addition held in registers.
    struct FromRegs {
        const AX &x_;
        const AY &y_;
        PTAU_HD AX x() const { return x_; }
        PTAU_HD AY y() const { return y_; }
        PTAU_HD AX x_again() const { return x_; }
        PTAU_HD AY y_again() const { return y_; }
    };
    // q = (x2, y2) affine, not the point at infinity
    PTAU_HD static LazyJac2 add_mixed(const LazyJac2 &p, const AX &x2, const AY &y2) { return add_src(p, FromRegs{x2, y2}); }

    // madd-2007-bl, ORDERED FOR LIVENESS.  Every product is a register-ABI call that leaves the caller the callee-saved half of
    // v32-v255 (112 VGPRs = eight 14-limb elements); whatever is alive across a call beyond that is a scratch store and a
    // reload (the round-3 form kept up to ten elements alive: 915 spilled VGPRs, 66 scratch dwords per point operation).  Here
    //   * the entry's coordinates come from `src`, BOTH AT THE TOP (one memory latency; for the GLS loop a table read plus
    //     psi^h) and are dead after the second and the fourth product;
    //   * Z3 = (Z1 + H)^2 - Z1Z1 - HH is taken as soon as HH exists, which retires Z1 and Z1Z1 before J, V and the results;
    //   * the exceptional cases (p = O, q = +-p) are decided from two flags taken when H and rh exist and REBUILT FROM THE ENTRY
    //     (src.x_again(), src.y_again(): re-read on the rare path) at the single join (q = p means 2p = 2q, q being affine), so
    //     neither p's nor the entry's coordinates stay alive for them.
    // At most seven elements are alive at any call (X Y Zc Z1Z1 H rh + HH), six from there on.
    template <class Src>
    PTAU_HD static LazyJac2 add_src(const LazyJac2 &p, const Src &src) {
        const AX x2 = src.x();
        const AY y2 = src.y();
        const bool pinf = p.is_inf();
        const auto Z1Z1 = K::sqr(p.Zc);
        const auto H = K::norm(K::sub(K::mul(x2, Z1Z1), p.X));                      // <1, 6>
        const auto rh = K::norm(K::sub(K::mul(K::mul(y2, p.Zc), Z1Z1), p.Y));       // <1, 6>
        const bool hz = K::is_zero(H), rz = K::is_zero(rh);   // pair-uniform
        const auto HH = K::sqr(H);
        LazyJac2 r;
        // Z3 = (Z1 + H)^2 - Z1^2 - H^2: a square is ONE base product per lane where the product 2 Z1 H is two
        r.Zc = K::norm(K::sub(K::sqr(K::norm(K::add(p.Zc, H))), K::add(Z1Z1, HH)));
        const auto I = K::template times<4>(HH);
        const auto J = K::mul(H, I);
        const auto Vv = K::mul(p.X, I);
        const auto R2 = K::template times<4>(K::sqr(rh));   // (2 rh)^2
        r.X = K::reduce(K::sub(R2, K::add(J, K::template times<2>(Vv))));
        r.Y = K::reduce(K::sub(K::mul(K::template times<2>(rh), K::sub(Vv, r.X)), K::template times<2>(K::mul(p.Y, J))));
        if (pinf || hz) {   // one join, rare paths, everything rebuilt from the entry
            const LazyJac2 q = {K::reduce(src.x_again()), K::reduce(src.y_again()), K::one()};
            if (pinf) r = q;
            else if (rz) r = dbl(q);
            else r = infinity();
        }
        return r;
    }
};

}  // namespace ptau
