// K1 for the a != 0 curves (MNT4-753 / MNT6-753, both groups) in the carry-free field with one lane pair per tower
// coefficient (lazy_a.hpp): reads uncompressed points, builds the per-point window table 1P..8P (Jacobian entries, 3 x 14
// limbs per lane), runs the signed 4-bit window multiplication and leaves the Jacobian result in the STANDARD layout that
// normalize_kernel of the single-lane group type reads (X.c0 X.c1 .. Y.c0 .. Z.c0 .., 26 limbs of 29 bits each, Montgomery).
// Replaces setup-utils' batch_exp for these groups — /root/reference/src/bin/contribute.rs:811-823 — and, with a fixed
// scalar, the multiplications of the membership tests (check_subgroup, contribute.rs:140-145).
// Included by kernels_impl.hpp (ScalarSource, thread_scalar, WINDOW_*, ERRBIT_* live there).
#pragma once
#include "lazy_a.hpp"

namespace ptau {

template <class FqP, class FrP, class Cfg>
struct LazyAGroup {
    using P28 = typename LazyTwin<FqP>::type;
    using Z = FpLZ<P28>;
    using CfgT = Cfg;
    static constexpr int DEG = Cfg::DEG;
    using K = std::conditional_t<DEG == 1, Fp1PZ<Z>, std::conditional_t<DEG == 2, Fp2QZ<Z, Cfg::NR>, Fp3HZ<Z, Cfg::NR>>>;
    static constexpr int LPP = 2 * DEG;                      // lanes per point
    // value bound of the coordinates between point operations (the formulas are closed under it: checked by the compiler)
    static constexpr int VC = DEG == 1 ? 512 : (DEG == 2 ? 3072 : 1024);
    using Acc = LazyJacA<K, Cfg, VC>;
    using CT = typename Acc::CT;
    using Fr = Fp<FrP>;
    using L29 = FpL<FqP>;                                    // a reduced coefficient over the pair: byte I/O, result store
    static constexpr int N29 = FqP::N, H29 = N29 / 2;
    static constexpr int H = Z::H;
    static constexpr int COEF_BYTES = FqP::BYTES;
    static constexpr int COMP = DEG * COEF_BYTES, UNCOMP = 2 * COMP;
    static constexpr int ENTRY_WORDS = 3 * H;                // one table entry of one lane: X | Y | Z
    static constexpr size_t WTAB_ENTRY_BYTES = (size_t)LPP * ENTRY_WORDS * 4;   // one table entry of one point
    static constexpr int POINTS_PER_BLOCK = DEG == 3 ? 4 * 10 : 256 / LPP;
    // GLS on the twist: psi = twist o Frobenius o untwist acts as [t - 1] on G2 (gen_constants.py PSI_*): two 377-bit halves
    static constexpr bool HAS_GLS = Cfg::PSI_KIND == 1 && DEG >= 2;
};

// k = k0 + k1 N for N = |t - 1| (12 words), k below the group order (24 words): k1 = floor(k / N) through the reciprocal
// floor(2^768 / N), at most three corrections; both halves below 2^377.  Valid use: points of the prime-order subgroup only.
template <class Cfg>
PTAU_HD inline void gls2_split(const uint32_t (&k)[24], uint32_t (&k0)[12], uint32_t (&k1)[12]) {
    uint32_t rc[13], nn[12];
#pragma unroll
    for (int t = 0; t < 13; t++) rc[t] = Cfg::PSI_RECIP[t];
#pragma unroll
    for (int t = 0; t < 12; t++) nn[t] = Cfg::PSI_N12[t];
    uint32_t prod[37];
    bn_mul<24, 13>(k, rc, prod);
    uint32_t q[13];
#pragma unroll
    for (int t = 0; t < 13; t++) q[t] = prod[24 + t];
    uint32_t qn[25];
    bn_mul<13, 12>(q, nn, qn);
    uint32_t rem[13];
    uint32_t br = 0;
#pragma unroll
    for (int t = 0; t < 13; t++) rem[t] = subb(k[t], qn[t], br);
    for (int it = 0; it < 4; it++) {
        uint32_t tt[13];
        uint32_t b2 = 0;
#pragma unroll
        for (int t = 0; t < 13; t++) tt[t] = subb(rem[t], t < 12 ? nn[t] : 0u, b2);
        if (b2) break;   // rem < N
#pragma unroll
        for (int t = 0; t < 13; t++) rem[t] = tt[t];
        uint32_t c = 1;
#pragma unroll
        for (int t = 0; t < 13; t++) q[t] = addc(q[t], 0u, c);
    }
#pragma unroll
    for (int t = 0; t < 12; t++) { k0[t] = rem[t]; k1[t] = q[t]; }
}

// One base-field coefficient from its canonical little-endian bytes straight into the carry-free Montgomery form of the
// lane pair: both lanes read the bytes, test value < p on the packed words, cut their own 14 limbs of 28 bits and multiply by
// Rz^2 (one lane-split product).  flags: the top two bits of the last byte when with_flags.
template <class LG>
__device__ inline bool lazy_a_read_coeff(const uint8_t *in, bool with_flags, uint8_t *flags, typename LG::Z::Nrm &out) {
    using Z = typename LG::Z;
    using P29 = typename LG::L29::Params;
    constexpr int N32 = P29::N32, BYTES = P29::BYTES, H = Z::H, LB = Z::LB;
    uint32_t w[N32 + 1];
#pragma unroll
    for (int t = 0; t <= N32; t++) w[t] = 0;
    for (int t = 0; t < BYTES; t++) {
        uint32_t byte = in[t];
        if (t == BYTES - 1 && with_flags) {
            *flags = (uint8_t)(byte & 0xC0u);
            byte &= 0x3Fu;
        }
        w[t >> 2] |= byte << (8 * (t & 3));
    }
    uint32_t br = 0;
#pragma unroll
    for (int t = 0; t < N32; t++) (void)subb(w[t], P29::MOD32[t], br);
    const bool hi = lane_parity() != 0;
    typename Z::template E<1, 1> c, r2;
#pragma unroll
    for (int t = 0; t < H; t++) {
        uint32_t half[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int bit = LB * (h * H + t), s = bit >> 5, o = bit & 31;
            uint32_t v = s < N32 ? w[s] >> o : 0u;
            if (o + LB > 32 && s + 1 < N32) v |= w[s + 1] << (32 - o);
            half[h] = v & Z::MASK;
        }
        c.l[t] = hi ? half[1] : half[0];
        r2.l[t] = Z::mine(Z::Params::R2, t);
    }
    out = Z::mul(c, r2);
    return br != 0;   // value < p
}

// point index of this thread; false for idle lanes (lanes 60-63 of a wave in the 6-lane layout) and past the end
template <class LG>
__device__ inline bool lazy_a_point(uint64_t n, uint64_t &i) {
    if constexpr (LG::DEG == 3) {
        const uint32_t lane = threadIdx.x & 63u, g = (lane * 43u) >> 8;
        if (g >= 10u) return false;
        i = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 10u + g;
    } else {
        i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG::LPP;
    }
    return i < n;
}

template <class LG>
__device__ inline int window_scalar_mul_lazy_a_thread(const uint8_t *__restrict__ in, uint32_t *__restrict__ scratch,
                                                      uint32_t *__restrict__ wtab, uint64_t n, uint64_t i, const ScalarSource &src,
                                                      int check_input, int use_gls) {
    using K = typename LG::K;
    using Z = typename LG::Z;
    using Acc = typename LG::Acc;
    using CT = typename LG::CT;
    using L29 = typename LG::L29;
    using Fr = typename LG::Fr;
    using Cfg_ = typename LG::CfgT;
    constexpr int D = LG::DEG, H = LG::H, NL = Fr::N, NW = NL * 8;
    static_assert(Fr::Params::BITS <= 32 * NL - 3, "scalar field leaves no headroom for the digit offset");
    const uint32_t j = K::coef(), hi = lane_parity();
    // result slot of this lane in the standard Jacobian layout: coordinate c, coefficient j, limbs [hi H29, hi H29 + H29)
    uint32_t *res = scratch + i * (uint64_t)(3 * D * LG::N29) + j * LG::N29 + hi * LG::H29;
    auto put29 = [&](const L29 &v, int coord) {
#pragma unroll
        for (int t = 0; t < LG::H29; t++) res[coord * D * LG::N29 + t] = v.l[t];
    };
    // ---- the point: this pair's coefficient of x and of y (flags sit in the last byte of y's LAST coefficient) ----
    const uint8_t *pp = in + i * LG::UNCOMP;
    uint8_t fl = 0, dummy = 0;
    typename Z::Nrm xz, yz;
    bool ok = lazy_a_read_coeff<LG>(pp + j * LG::COEF_BYTES, false, &dummy, xz);
    ok &= lazy_a_read_coeff<LG>(pp + LG::COMP + j * LG::COEF_BYTES, j == (uint32_t)(D - 1), &fl, yz);
    ok = K::all_lanes(ok);
    const uint32_t flags = K::from_last((uint32_t)fl);
    const bool inf = (flags & FLAG_INF) != 0;
    int bad = ok ? 0 : ERRBIT_NONCANONICAL;
    if ((check_input & 1) && inf) bad |= ERRBIT_INFINITY;
    if (bad || inf) {   // uniform over the point's lanes
        L29 one = L29::one(), zero = L29::zero();
        if (j != 0u) one = zero;
        put29(one, 0);
        put29(one, 1);
        put29(zero, 2);
        return bad;
    }
    const CT px = K::wrap(xz), py = K::wrap(yz), pz = K::one();
    // ---- table: entry e = (e + 1) P, this lane's 3 x H limbs contiguous; a multiple that is O is marked by Z = 0 limbs ----
    const uint64_t nslots = n * LG::LPP, slot = i * LG::LPP + K::lane_in_point();
    auto entry = [&](int e) { return wtab + ((uint64_t)e * nslots + slot) * LG::ENTRY_WORDS; };
    auto store_entry = [&](int e, const Acc &q) {
        uint32_t *w = entry(e);
        const bool qinf = q.is_inf();
#pragma unroll
        for (int t = 0; t < H; t++) {
            w[t] = K::raw(q.X).l[t];
            w[H + t] = K::raw(q.Y).l[t];
            w[2 * H + t] = qinf ? 0u : K::raw(q.Zc).l[t];
        }
    };
    // the second operand of an addition, read where the formula uses it (LazyJacA::add)
    struct FromRegs {   // P itself while the table is built
        const CT &x_, &y_, &z_;
        PTAU_HD CT x() const { return x_; }
        PTAU_HD CT y() const { return y_; }
        PTAU_HD CT z() const { return z_; }
    };
    struct FromTable {  // a table entry, its y negated on the way in when the digit is negative
        const uint32_t *e;
        bool negate;
        PTAU_HD CT load(int c) const {
            CT v;
            const volatile uint32_t *w = e + c * H;   // volatile: the loads stay where the formula consumes the value
#pragma unroll
            for (int t = 0; t < H; t++) K::raw(v).l[t] = w[t];
            return v;
        }
        PTAU_HD CT x() const { return load(0); }
        PTAU_HD auto y() const {
            const CT v = load(1);
            const auto nv = K::neg(v);
            decltype(K::neg(v)) r = v;
            if (negate) r = nv;
            return r;
        }
        PTAU_HD CT z() const { return load(2); }
    };
    // psi of a table entry, formed where the formula uses it: coefficient j of X, Y, Z times KX_j, KY_j, KZ_j (Frobenius and the
    // twist constants in one base product per coordinate; the sign of t - 1 sits in KY)
    struct FromTablePsi {
        const uint32_t *e;
        bool negate;
        PTAU_HD CT load(int c) const {
            CT v;
            const volatile uint32_t *w = e + c * H;
#pragma unroll
            for (int t = 0; t < H; t++) K::raw(v).l[t] = w[t];
            return v;
        }
        PTAU_HD static CT times_row(const CT &v, const uint32_t (*tbl)[2 * H]) {
            const uint32_t *row = &tbl[K::coef()][lane_parity() ? H : 0];
            typename Z::template E<1, 1> c;
#pragma unroll
            for (int t = 0; t < H; t++) c.l[t] = row[t];
            return K::wrap(Z::mul(K::raw(v), c));
        }
        PTAU_HD CT x() const {
            if constexpr (LG::HAS_GLS) return times_row(load(0), Cfg_::PSI_KXZ);
            else return load(0);
        }
        PTAU_HD auto y() const {
            CT v = load(1);
            if constexpr (LG::HAS_GLS) v = times_row(v, Cfg_::PSI_KYZ);
            const auto nv = K::neg(v);
            decltype(K::neg(v)) r = v;
            if (negate) r = nv;
            return r;
        }
        PTAU_HD CT z() const {
            if constexpr (LG::HAS_GLS) return times_row(load(2), Cfg_::PSI_KZZ);
            else return load(2);
        }
    };
    {
        Acc cur = {px, py, pz};
        store_entry(0, cur);
#pragma nounroll
        for (int e = 1; e < WINDOW_ENTRIES; e++) {
            cur = Acc::add(cur, FromRegs{px, py, pz});
            store_entry(e, cur);
        }
    }
    // ---- signed 4-bit windows, MSB first ----
    uint32_t k[NL];
    thread_scalar<Fr>(src, i, k);
    Acc acc = Acc::infinity();
    if constexpr (LG::HAS_GLS && NL == 24) {
        if (use_gls) {
            // k P = k0 P + k1 (+-psi(P)): two 377-bit halves over ONE table, interleaved signed 4-bit windows, psi applied to the
            // entry on the fly (three base products) — half the doublings.  Only for points of the prime-order subgroup.
            uint32_t kk[2][12];
            gls2_split<Cfg_>(k, kk[0], kk[1]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t c2 = 0;
#pragma unroll
                for (int t = 0; t < 12; t++) kk[h][t] = addc(kk[h][t], 0x88888888u, c2);   // halves < 2^377: no carry out of 384 bits
            }
            for (int w = 12 * 8 - 1; w >= 0; w--) {
                if (!acc.is_inf()) {
#pragma nounroll
                    for (int d = 0; d < WINDOW_BITS; d++) acc = Acc::dbl(acc);
                }
#pragma nounroll
                for (int h = 0; h < 2; h++) {
                    uint32_t limb = 0;
#pragma unroll
                    for (int t = 0; t < 12; t++) limb = (t == (w >> 3)) ? (h ? kk[1][t] : kk[0][t]) : limb;
                    const int dgt = (int)((limb >> ((w & 7) * 4)) & 15u) - 8;
                    if (dgt != 0) {
                        const int mag = dgt < 0 ? -dgt : dgt;
                        const FromTable ent{entry(mag - 1), dgt < 0};
                        if (!K::limbs_all_zero(ent.z())) {   // the multiple is not O
                            if (h == 0) acc = Acc::add(acc, ent);
                            else acc = Acc::add(acc, FromTablePsi{entry(mag - 1), dgt < 0});
                        }
                    }
                }
            }
            put29(Z::to29(K::raw(acc.X)), 0);
            put29(Z::to29(K::raw(acc.Y)), 1);
            put29(Z::to29(K::raw(acc.Zc)), 2);
            return 0;
        }
    }
    uint32_t cy = 0;
#pragma unroll
    for (int t = 0; t < NL; t++) k[t] = addc(k[t], 0x88888888u, cy);
    for (int w = NW - 1; w >= 0; w--) {
        if (!acc.is_inf()) {
#pragma nounroll
            for (int d = 0; d < WINDOW_BITS; d++) acc = Acc::dbl(acc);
        }
        uint32_t limb = 0;
#pragma unroll
        for (int t = 0; t < NL; t++) limb = (t == (w >> 3)) ? k[t] : limb;   // register select, no scratch
        const int dgt = (int)((limb >> ((w & 7) * 4)) & 15u) - 8;
        if (dgt != 0) {
            const int mag = dgt < 0 ? -dgt : dgt;
            const FromTable ent{entry(mag - 1), dgt < 0};
            if (!K::limbs_all_zero(ent.z())) acc = Acc::add(acc, ent);   // the multiple is not O
        }
    }
    int verdict = 0;
    if constexpr (LG::HAS_GLS) {
        // The psi membership test, compared HERE (SCALAR_FIXED_PSI: the scalar was |t - 1|): P is in the order-r subgroup iff
        // psi(P) == [t - 1]P, i.e. (+-psi(P)) == [|t - 1|]P = acc.  +-psi(P) is FromTablePsi of entry 0 (P itself, Z = 1: the sign of
        // t - 1 sits in the Y constants), affine; equality with the Jacobian acc by cross-multiplication — 2 squares' worth of
        // tower products per point instead of a second, single-lane kernel over Fq2 / Fq3 by reference (round 3:
        // psi_check_kernel<Mnt6G2>, 9.5 KB of scratch per lane, 21.6 ms for 68 waves).
        if (src.mode == SCALAR_FIXED_PSI) {
            const FromTablePsi q{entry(0), false};
            const bool ainf = acc.is_inf();
            const auto ZZ = K::sqr(acc.Zc);
            const auto dx = K::norm(K::sub(K::mul(q.x(), ZZ), acc.X));
            const auto dy = K::norm(K::sub(K::mul(K::mul(q.y(), acc.Zc), ZZ), acc.Y));
            const bool same = K::is_zero(dx) && K::is_zero(dy);
            if (ainf || !same) verdict = ERRBIT_NOT_IN_SUBGROUP;
        }
    }
    // ---- back to the reduced field, standard layout ----
    put29(Z::to29(K::raw(acc.X)), 0);
    put29(Z::to29(K::raw(acc.Y)), 1);
    put29(Z::to29(K::raw(acc.Zc)), 2);
    return verdict;
}

template <class LG>
__global__ void __launch_bounds__(256, 2)
window_scalar_mul_lazy_a_kernel(const uint8_t *__restrict__ in, uint32_t *__restrict__ scratch, uint32_t *__restrict__ wtab,
                                uint64_t n, ScalarSource src, int check_input, int use_gls, int *__restrict__ err) {
    uint64_t i;
    if (!lazy_a_point<LG>(n, i)) return;
    const int bad = window_scalar_mul_lazy_a_thread<LG>(in, scratch, wtab, n, i, src, check_input, use_gls);
    if (bad) atomicOr(err, bad);
}

// ---- K4 on the a != 0 twists: the accumulate pass of the bucket MSM (msm_accumulate_thread, kernels_impl.hpp) with one lane
// pair per tower coefficient.  A run = run_len consecutive sorted entries, cut at bucket boundaries; the run's lanes (4 on
// MNT4-753 G2, 6 on MNT6-753 G2) sum it by MIXED additions in the carry-free field.  Input: the standard affine array of the
// single-lane group type (x coefficients | y coefficients, 26 limbs of 29 bits each, | flag word) — each pair reads ITS
// coefficient's limbs and converts them with one base product per coordinate; output: partial sums in the standard Jacobian
// layout (msm_combine of the single-lane type reads them).  Round 3 ran this pass in the reduced lane-split field with the
// Fq2 / Fq3 point operations out of line by reference: msm_accumulate_kernel<Mnt6G2> 25 ms per vector at 40 % of wave cycles
// waiting, 4.3 KB of scratch per lane (profiles/r03_pmc_cfg4.json).  Serves setup-utils' dense_multiexp inside
// check_power_ratios — /root/reference/src/bin/verify_transcript.rs:812-822, contribute.rs:968-986.
template <class LG>
__device__ inline void msm_accumulate_lazy_a_thread(const uint32_t *__restrict__ aff, const uint32_t *__restrict__ sorted,
                                                    const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ run_offsets,
                                                    uint32_t *__restrict__ partials, uint32_t shift, uint32_t total_buckets, uint32_t v,
                                                    uint32_t run_len) {
    using K = typename LG::K;
    using Z = typename LG::Z;
    using Acc = typename LG::Acc;
    using CT = typename LG::CT;
    using L29 = typename LG::L29;
    constexpr int D = LG::DEG;
    constexpr uint32_t AFF_WORDS = 2 * D * LG::N29 + 1, JAC_WORDS = 3 * D * LG::N29;
    const uint32_t total = offsets[total_buckets];
    const uint64_t e0w = (uint64_t)v * run_len;
    if (e0w >= total) return;
    const uint32_t e0 = (uint32_t)e0w, e1 = e0w + run_len < total ? e0 + run_len : total;
    uint32_t lo = 0, hi_b = total_buckets;   // largest b with offsets[b] <= e0: the (non-empty) bucket holding entry e0
    while (hi_b - lo > 1) {
        const uint32_t mid = (lo + hi_b) >> 1;
        if (offsets[mid] <= e0) lo = mid; else hi_b = mid;
    }
    uint32_t b = lo, bend = offsets[b + 1];
    const uint32_t mine = K::coef() * LG::N29 + (lane_parity() ? LG::H29 : 0);   // this lane's limbs inside a coordinate
    auto store = [&](uint32_t bb, const Acc &acc) {
        uint32_t *res = partials + (uint64_t)(run_offsets[bb] + (v - offsets[bb] / run_len)) * JAC_WORDS + mine;
        const L29 x = Z::to29(K::raw(acc.X)), y = Z::to29(K::raw(acc.Y)), z = Z::to29(K::raw(acc.Zc));
#pragma unroll
        for (int t = 0; t < LG::H29; t++) {
            res[t] = x.l[t];
            res[D * LG::N29 + t] = y.l[t];
            res[2 * D * LG::N29 + t] = z.l[t];
        }
    };
    struct Entry {   // one affine point of the input vector, this pair's coefficient of x and y in the carry-free field
        const uint32_t *w;   // this lane's limbs of x
        bool negate;
        PTAU_HD static CT conv(const uint32_t *q) {
            L29 c;
#pragma unroll
            for (int t = 0; t < LG::H29; t++) c.l[t] = q[t];
            return K::wrap(Z::from29(c));
        }
        PTAU_HD static CT conv_again(const uint32_t *q0) {   // volatile: not merged with the read at the top
            const volatile uint32_t *q = q0;
            L29 c;
#pragma unroll
            for (int t = 0; t < LG::H29; t++) c.l[t] = q[t];
            return K::wrap(Z::from29(c));
        }
        PTAU_HD auto sign(const CT &yv) const {
            const auto nv = K::neg(yv);
            decltype(K::neg(yv)) r = yv;
            if (negate) r = nv;
            return r;
        }
        PTAU_HD CT x() const { return conv(w); }
        PTAU_HD auto y() const { return sign(conv(w + D * LG::N29)); }
        PTAU_HD CT x_again() const { return conv_again(w); }
        PTAU_HD auto y_again() const { return sign(conv_again(w + D * LG::N29)); }
    };
    Acc acc = Acc::infinity();
    for (uint32_t e = e0; e < e1; e++) {
        if (e == bend) {   // the next bucket starts here
            store(b, acc);
            acc = Acc::infinity();
            do { b++; bend = offsets[b + 1]; } while (bend <= e);
        }
        const uint32_t s = sorted[e];
        const uint32_t *w = aff + ((uint64_t)(s & 0x7fffffffu) + shift) * AFF_WORDS;
        if (w[2 * D * LG::N29] & 0xffu) continue;   // the point at infinity (bool flag in the word's low byte)
        acc = Acc::add_affine(acc, Entry{w + mine, (s >> 31) != 0});
    }
    store(b, acc);
}

template <class LG>
__global__ void __launch_bounds__(256, 2)
msm_accumulate_lazy_a_kernel(const uint32_t *__restrict__ aff, const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ offsets,
                             const uint32_t *__restrict__ run_offsets, uint32_t *__restrict__ partials, uint32_t shift,
                             uint32_t total_buckets, uint64_t max_runs, uint32_t run_len) {
    uint64_t v;
    if (!lazy_a_point<LG>(max_runs, v)) return;
    msm_accumulate_lazy_a_thread<LG>(aff, sorted, offsets, run_offsets, partials, shift, total_buckets, (uint32_t)v, run_len);
}

// ---- the small kernels behind the bucket sums (msm_combine / msm_bucket_reduce / reduce_rows / msm_horner / reduce_sum of
// kernels_impl.hpp), same algorithms, one lane pair per tower coefficient.  They run a few hundred to a few thousand threads
// of SEQUENTIAL Jacobian additions each — on a 2^16-point chunk of an MNT twist this latency chain (on the tail stream, with the
// bucket tests) outlasts the main stream's work; in the reduced field by reference an Fq3 addition is ~100 base products
// through 9 KB of scratch in one lane, here it is inlined register-ABI products over six.  Points travel in the standard
// Jacobian layout of the single-lane group type (26 limbs of 29 bits per coefficient): three converting products in, three out.
template <class LG>
struct LazyAStd {
    using K = typename LG::K;
    using Z = typename LG::Z;
    using Acc = typename LG::Acc;
    using CT = typename LG::CT;
    using L29 = typename LG::L29;
    static constexpr int D = LG::DEG;
    static constexpr uint32_t JAC_WORDS = 3 * D * LG::N29;
    __device__ static uint32_t mine() { return K::coef() * LG::N29 + (lane_parity() ? LG::H29 : 0); }
    __device__ static CT coord(const uint32_t *q) {
        L29 c;
#pragma unroll
        for (int t = 0; t < LG::H29; t++) c.l[t] = q[t];
        return K::wrap(Z::from29(c));
    }
    __device__ static Acc load(const uint32_t *base, uint64_t idx) {
        const uint32_t *w = base + idx * JAC_WORDS + mine();
        return {coord(w), coord(w + D * LG::N29), coord(w + 2 * D * LG::N29)};
    }
    __device__ static void store(uint32_t *base, uint64_t idx, const Acc &a) {
        uint32_t *res = base + idx * JAC_WORDS + mine();
        const L29 x = Z::to29(K::raw(a.X)), y = Z::to29(K::raw(a.Y)), z = Z::to29(K::raw(a.Zc));
#pragma unroll
        for (int t = 0; t < LG::H29; t++) {
            res[t] = x.l[t];
            res[D * LG::N29 + t] = y.l[t];
            res[2 * D * LG::N29 + t] = z.l[t];
        }
    }
    struct Regs {   // a Jacobian point held in registers as the second operand of LazyJacA::add
        const CT &x_, &y_, &z_;
        PTAU_HD CT x() const { return x_; }
        PTAU_HD CT y() const { return y_; }
        PTAU_HD CT z() const { return z_; }
    };
    __device__ static Acc add(const Acc &p, const Acc &q) {   // either operand may be the point at infinity
        if (q.is_inf()) return p;                               // (uniform over the point's lanes)
        return Acc::add(p, Regs{q.X, q.Y, q.Zc});
    }
};

template <class LG>
__global__ void __launch_bounds__(256, 2)
msm_combine_lazy_a_kernel(const uint32_t *__restrict__ partials, const uint32_t *__restrict__ run_offsets, uint32_t *__restrict__ buckets,
                          uint64_t total_buckets) {
    using S = LazyAStd<LG>;
    uint64_t t;
    if (!lazy_a_point<LG>(total_buckets, t)) return;
    typename S::Acc acc = S::Acc::infinity();
    const uint32_t v1 = run_offsets[t + 1];
    for (uint32_t v = run_offsets[t]; v < v1; v++) acc = S::add(acc, S::load(partials, v));
    S::store(buckets, t, acc);
}

// thread (w, s): sum_{b in segment s} (b + 1) * bucket[w][b]  =  sum of running sums + lo * (segment sum)
template <class LG>
__global__ void __launch_bounds__(256, 2)
msm_bucket_reduce_lazy_a_kernel(const uint32_t *__restrict__ buckets, uint32_t *__restrict__ parts, MsmPlan pl, uint64_t total) {
    using S = LazyAStd<LG>;
    using Acc = typename S::Acc;
    uint64_t t;
    if (!lazy_a_point<LG>(total, t)) return;
    const uint32_t nseg = pl.B / pl.seg;
    const uint32_t w = (uint32_t)(t / nseg), sg = (uint32_t)(t % nseg), lo = sg * pl.seg;
    const uint64_t base = (uint64_t)w * pl.B;
    Acc run = Acc::infinity(), tot = Acc::infinity();
    for (uint32_t b = lo + pl.seg; b-- > lo;) {
        run = S::add(run, S::load(buckets, base + b));
        tot = S::add(tot, run);
    }
    if (lo) {   // + lo * run, MSB-first double-and-add over the (<= 16) bits of lo
        Acc acc = Acc::infinity();
        for (int bit = 31 - __builtin_clz(lo); bit >= 0; bit--) {
            acc = Acc::dbl(acc);
            if ((lo >> bit) & 1u) acc = S::add(acc, run);
        }
        tot = S::add(tot, acc);
    }
    S::store(parts, t, tot);
}

// rows x cols -> rows x T: thread (r, t) sums in[r][t], in[r][t + T], ...
template <class LG>
__global__ void __launch_bounds__(256, 2)
reduce_rows_lazy_a_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t cols, uint32_t T, uint64_t total) {
    using S = LazyAStd<LG>;
    uint64_t id;
    if (!lazy_a_point<LG>(total, id)) return;
    const uint64_t r = id / T, t = id % T;
    typename S::Acc acc = S::Acc::infinity();
    for (uint64_t j = t; j < cols; j += T) acc = S::add(acc, S::load(in, r * cols + j));
    S::store(out, id, acc);
}

// sum_w 2^(c w) win[w]: one point's lanes per sum (block b = sum b)
// (256-thread bounds like every kernel of this file although one point's lanes work: a kernel compiled for one wave per SIMD
// would let the shared product routines use AGPRs, and a callee's register count is every caller's — K1 dropped to one wave)
template <class LG>
__global__ void __launch_bounds__(256, 2)
msm_horner_lazy_a_kernel(const uint32_t *__restrict__ win, uint32_t *__restrict__ out, MsmPlan pl, uint64_t out_stride) {
    using S = LazyAStd<LG>;
    using Acc = typename S::Acc;
    if (threadIdx.x >= (unsigned)LG::LPP) return;
    const uint64_t wbase = (uint64_t)blockIdx.x * pl.W;
    Acc acc = Acc::infinity();
    for (int w = pl.W - 1; w >= 0; w--) {
        if (!acc.is_inf())
            for (int d = 0; d < pl.c; d++) acc = Acc::dbl(acc);
        acc = S::add(acc, S::load(win, wbase + w));
    }
    S::store(out, (uint64_t)blockIdx.x * out_stride, acc);
}

// thread t adds elements t, t + T, t + 2T, ... (< n) into out[t]
template <class LG>
__global__ void __launch_bounds__(256, 2)
reduce_sum_lazy_a_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n, uint64_t T) {
    using S = LazyAStd<LG>;
    uint64_t t;
    if (!lazy_a_point<LG>(T, t)) return;
    typename S::Acc acc = S::Acc::infinity();
    for (uint64_t i = t; i < n; i += T) acc = S::add(acc, S::load(in, i));
    S::store(out, t, acc);
}

}  // namespace ptau
