// TEST INFRASTRUCTURE: runs the kernels' per-thread bodies (kernels_impl.hpp, host+device code) on
// the CPU so the field / curve / serde templates can be checked against the oracle without a GPU.
// Not part of the product: libptau_hip.so has no CPU path.
//
// usage: host_emul <op> <curve> <group> <n> <start> <cin> <cout> <tau_hex> <coeff_hex|-> < in > out
//   op = apply | decompress
//        field : base-field arithmetic of the group's Fq class.  stdin = n pairs (a, b) of canonical BYTES-long elements;
//               stdout per pair: a+b | a-b | a*b | a^2 | -a | 1/a (0 for a = 0) | sqrt flag byte + root (0 when none) | sign byte
//        fieldz: the same input; arithmetic of the carry-free twin field (fpz.hpp), see the op
//        gls2split: MNT twists only.  <n> = 0, <start> = the number of scalars; stdin = scalars x 24 u32 words; stdout per scalar: k0 (12 words) | k1 (12 words) of the
//               Euclidean split k = k0 + k1 |t - 1| K1's GLS loop consumes (k1_lazy_a.hpp)
//        msmplan: no input; <n> = points, <start> = bucket test cost (0 = the plain plan for 127-bit scalars), <cin> = resident runs
//               in thousands; stdout = text "c W B bits"
//        subgroup : stdin = n points; stdout = n bytes of error bits (0 = in the prime-order subgroup)
//        subgroupjac : BLS12-377 G1 only.  The same points, each scaled to a Jacobian representative (X, Y, Z) = (x z^2, y z^3, z)
//               with z = index + 2, through the membership test that IGNORES Z (endo_membership_lazy; the bucket sums of the
//               membership-fused MSM); stdout = n bytes of error bits
//        msm  : stdin = n points (uncompressed) then n x Fr::N u32 scalar limbs; <start> = window bits c (0 = the
//               library's own plan), <cout> = 1 adds the shifted sum (needs n + 1 points); stdout = 1 or 2 points
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels_impl.hpp"

using namespace ptau;

static std::vector<uint64_t> parse_hex(const char *h, int limbs) {
    std::vector<uint64_t> v(limbs, 0);
    size_t len = strlen(h);
    for (size_t i = 0; i < len; i++) {
        char ch = h[len - 1 - i];
        uint64_t d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        if (i / 16 < (size_t)limbs) v[i / 16] |= d << (4 * (i % 16));
    }
    return v;
}

template <class G>
static int run(const std::string &op, uint64_t n, uint64_t start, int cin, int cout, const char *tau_hex,
               const char *coeff_hex) {
    const size_t isz = cin ? G::COMP : G::UNCOMP, osz = cout ? G::COMP : G::UNCOMP;
    std::vector<uint8_t> in(n * isz), out;
    if (fread(in.data(), 1, in.size(), stdin) != in.size()) return 2;
    int bad = 0;
    if (op == "field") {
        using Fq = typename G::Fq;
        const int B = Fq::BYTES;
        in.resize(n * 2 * B);
        if (fread(in.data() + n * isz, 1, in.size() - n * isz, stdin) != in.size() - n * isz && in.size() > n * isz) return 2;
        out.assign(n * (8 * B + 2), 0);
        for (uint64_t i = 0; i < n; i++) {
            Fq a, b;
            uint8_t fl = 0;
            if (!Fq::read_bytes(in.data() + (2 * i) * B, false, &fl, a) || !Fq::read_bytes(in.data() + (2 * i + 1) * B, false, &fl, b)) return 5;
            uint8_t *o = out.data() + i * (8 * B + 2);
            Fq::add(a, b).write_bytes(o, 0);
            Fq::sub(a, b).write_bytes(o + B, 0);
            Fq::mul(a, b).write_bytes(o + 2 * B, 0);
            Fq::sqr(a).write_bytes(o + 3 * B, 0);
            Fq::neg(a).write_bytes(o + 4 * B, 0);
            Fq::inverse(a).write_bytes(o + 5 * B, 0);
            Fq r = Fq::zero();
            const bool ok = Fq::sqrt(a, r);
            o[6 * B] = ok ? 1 : 0;
            (ok ? r : Fq::zero()).write_bytes(o + 6 * B + 1, 0);
            o[7 * B + 1] = (uint8_t)(a.sign_class() + 1);
            // second serialisation slot keeps the layout fixed-size
        }
        fwrite(out.data(), 1, out.size(), stdout);
        return 0;
    }
    if (op == "fieldz") {
        // the carry-free twin of the group's base field (fpz.hpp): every result converted back and written canonically.
        // per pair (a, b): a+b | a-b | a*b | a^2 | -a | 5a - 8b (unnormalised operands into a product by 1) |
        //                  vreduce(31 a + 19 b ... as the point formulas stack them) | flags byte: bit0 is_zero(a - a), bit1
        //                  is_zero(a - b), bit2 is_zero(of 33 p built from limbs), bit3 is_zero(a) on the normalised value
        if constexpr (LazyOneLane<typename G::LazyP>::value && G::F::DEG == 1) {   // the lane-split twin (fplz.hpp) is device-only
            using Fq = typename G::Fq;
            using Z = FpZ<typename G::LazyP>;
            const int B = Fq::BYTES;
            in.resize(n * 2 * B);
            if (fread(in.data() + n * isz, 1, in.size() - n * isz, stdin) != in.size() - n * isz && in.size() > n * isz) return 2;
            out.assign(n * (8 * B + 1), 0);
            for (uint64_t i = 0; i < n; i++) {
                Fq a29, b29;
                uint8_t fl = 0;
                if (!Fq::read_bytes(in.data() + (2 * i) * B, false, &fl, a29) || !Fq::read_bytes(in.data() + (2 * i + 1) * B, false, &fl, b29)) return 5;
                const auto a = Z::from29(a29), b = Z::from29(b29);
                uint8_t *o = out.data() + i * (8 * B + 1);
                Z::to29(Z::add(a, b)).write_bytes(o, 0);
                Z::to29(Z::sub(a, b)).write_bytes(o + B, 0);
                Z::to29(Z::mul(a, b)).write_bytes(o + 2 * B, 0);
                Z::to29(Z::sqr(a)).write_bytes(o + 3 * B, 0);
                Z::to29(Z::neg(a)).write_bytes(o + 4 * B, 0);
                Z::to29(Z::sub(Z::template times<5>(a), Z::template times<8>(b))).write_bytes(o + 5 * B, 0);
                // a stack of sums and differences as deep as the point formulas make them, then the quotient-estimate reduction
                const auto s1 = Z::norm(Z::add(Z::template times<8>(a), Z::template times<4>(b)));          // < 24 p
                const auto big = Z::norm(Z::sub(Z::norm(Z::template times<8>(s1)), Z::template times<2>(a)));  // < 197 p
                Z::to29(Z::vreduce(big)).write_bytes(o + 6 * B, 0);
                {
                    // the fused Fq2 coefficient a0 b0 + 5 (KN p - a1) b1 with a0 = a1 = a, b0 = b1 = b: limbs of the second
                    // operand up to 15 x 2^28 (three units x 5), the column sums of both products share one reduction
                    uint32_t x2[Z::N];
                    typename Z::template E<1, 2> r2;
                    for (int j = 0; j < Z::N; j++) {
                        const uint32_t ng = Z::template SubC<32, 2>::v.l[j] - a.l[j];
                        x2[j] = (ng << 2) + ng;
                    }
                    Z::ps_mul2(a.l, b.l, x2, b.l, r2.l);
                    Z::to29(r2).write_bytes(o + 7 * B, 0);
                }
                uint8_t f = 0;
                f |= Z::is_zero(Z::norm(Z::sub(a, a))) ? 1 : 0;
                f |= Z::is_zero(Z::norm(Z::sub(a, b))) ? 2 : 0;
                typename Z::template E<1, 34> kp;
                for (int j = 0; j < Z::N; j++) kp.l[j] = Z::template SubC<33, 0>::v.l[j];   // 33 p, plain limbs
                f |= Z::is_zero(kp) ? 4 : 0;
                f |= Z::is_zero(a) ? 8 : 0;
                o[8 * B] = f;
            }
            fwrite(out.data(), 1, out.size(), stdout);
            return 0;
        } else {
            return 6;
        }
    }
    if (op == "gls2split") {
        if constexpr (G::J::CfgT::PSI_KIND == 1 && G::Fr::N == 24) {
            // <n> = 0 (no points to read), the scalar count travels in <start>
            const uint64_t ns = start;
            std::vector<uint32_t> k32(ns * 24), o32(ns * 24);
            if (ns && fread(k32.data(), 4, k32.size(), stdin) != k32.size()) return 2;
            for (uint64_t i = 0; i < ns; i++) {
                uint32_t k[24], k0[12], k1[12];
                for (int t = 0; t < 24; t++) k[t] = k32[i * 24 + t];
                gls2_split<typename G::J::CfgT>(k, k0, k1);
                for (int t = 0; t < 12; t++) { o32[i * 24 + t] = k0[t]; o32[i * 24 + 12 + t] = k1[t]; }
            }
            fwrite(o32.data(), 4, o32.size(), stdout);
            return 0;
        } else {
            return 6;
        }
    }
    if (op == "msm") {
        // the per-thread bodies of the bucket MSM (K4), serial on the CPU, same order of launches as capi.hip
        using J = typename G::J;
        const int nsum = cout ? 2 : 1;
        const uint64_t npts = n + (nsum - 1);
        in.resize(npts * G::UNCOMP);
        if (nsum == 2 && fread(in.data() + n * G::UNCOMP, 1, G::UNCOMP, stdin) != (size_t)G::UNCOMP) return 2;
        std::vector<uint32_t> k32(n * G::Fr::N);
        if (n && fread(k32.data(), 4, k32.size(), stdin) != k32.size()) return 2;
        const MsmPlan pl = msm_plan(std::max<uint64_t>(n, 1), G::Fr::Params::BITS, (int)start);
        const uint64_t wb = (uint64_t)pl.W * pl.B;
        std::vector<uint64_t> aff_store((npts * msm_aff_bytes<G>() + 7) / 8 + 1);   // the array's element is the group's own (msm_aff_bytes)
        typename G::A *const aff_p = reinterpret_cast<typename G::A *>(aff_store.data());
        std::vector<uint32_t> dig(pl.W * std::max<uint64_t>(n, 1)), sorted(dig.size()), counts(wb, 0), offsets(wb + 1), run_offsets(wb + 1);
        ScalarSource src{SCALAR_ARRAY, k32.data(), 0, 0, 0};
        for (uint64_t i = 0; i < npts; i++) bad |= msm_prepare_thread<G>(in.data(), 0, aff_p, dig.data(), counts.data(), npts, n, i, src, pl);
        // a short run length so that small test inputs still cross bucket boundaries inside a run (PTAU_MSM_RUN as in capi.hip)
        const uint32_t run_len = getenv("PTAU_MSM_RUN") ? (uint32_t)atoi(getenv("PTAU_MSM_RUN")) : 5u;
        uint32_t run = 0, rrun = 0;
        for (uint64_t j = 0; j < wb; j++) {
            offsets[j] = run; run_offsets[j] = rrun;
            rrun += msm_bucket_parts(run, counts[j], run_len); run += counts[j];
            counts[j] = offsets[j];
        }
        offsets[wb] = run; run_offsets[wb] = rrun;
        for (uint64_t i = 0; i < n; i++) msm_scatter_thread(dig.data(), counts.data(), sorted.data(), n, i, pl);
        std::vector<J> buckets(nsum * wb), parts(nsum * wb), runs(rrun + 1);
        for (int sidx = 0; sidx < nsum; sidx++) {
            for (uint32_t v = 0; (uint64_t)v * run_len < run; v++)
                msm_accumulate_thread<G>(aff_p, sorted.data(), offsets.data(), run_offsets.data(), runs.data(), sidx, (uint32_t)wb, v, run_len);
            for (uint64_t t = 0; t < wb; t++) msm_combine_thread<G>(runs.data(), run_offsets.data(), buckets.data() + sidx * wb, t);
        }
        const uint32_t rows = nsum * pl.W;
        uint32_t cols = pl.B / pl.seg;
        for (uint64_t t = 0; t < (uint64_t)rows * cols; t++) msm_bucket_reduce_thread<G>(buckets.data(), parts.data(), pl, t);
        J *cur = parts.data(), *nxt = buckets.data();
        while (cols > 1) {
            const uint32_t T = (cols + 15) / 16;
            for (uint64_t id = 0; id < (uint64_t)rows * T; id++) reduce_rows_thread<G>(cur, nxt, cols, T, id);
            std::swap(cur, nxt);
            cols = T;
        }
        std::vector<J> fin(nsum);
        for (int sidx = 0; sidx < nsum; sidx++) msm_horner_thread<G>(cur + sidx * pl.W, &fin[sidx], pl);
        std::vector<typename G::F> pre(nsum);
        out.resize(nsum * G::UNCOMP);
        for (int t = 0; t < nsum; t++) bad |= normalize_thread<G>(fin.data(), pre.data(), out.data(), 0, nsum, nsum, t, 0);
    } else if (op == "subgroupjac") {
        out.resize(n);
        if constexpr (G::J::CfgT::ENDO_KIND == 1 && G::LAZY_LOOP && G::LANE_MODE == LANES_ONE) {
            using F = typename G::F;
            for (uint64_t i = 0; i < n; i++) {
                typename G::A p;
                int b = read_point_any<G>(in.data(), cin, i, p);
                if (b || p.inf) { out[i] = (uint8_t)b; continue; }
                const F z = F::mul_small(F::one(), (int)(i % 1000) + 2), z2 = F::sqr(z);
                out[i] = (uint8_t)endo_membership_lazy<G>(F::mul(p.x, z2), F::mul(p.y, F::mul(z2, z)));
            }
        } else {
            return 4;
        }
        fwrite(out.data(), 1, out.size(), stdout);
        return 0;
    } else if (op == "subgroup") {
        // endomorphism membership test (K3'), one result byte (error bits) per point
        out.resize(n);
        if constexpr (G::J::CfgT::ENDO_KIND != 0) {
            for (uint64_t i = 0; i < n; i++) out[i] = (uint8_t)endo_subgroup_thread<G>(in.data(), cin, i);
        } else if constexpr (G::J::CfgT::GLV_KIND == 2) {
            // lattice-GLV curves (BW6-761): [a]P + [b]phi(P) == O through K1's GLV loop (SCALAR_SUBGROUP_ENDO), after the
            // on-curve test of an uncompressed encoding — as capi.hip's enqueue_subgroup_kernels does
            std::vector<typename G::J> scratch(n), wtab(8 * n);
            const ScalarSource src{SCALAR_SUBGROUP_ENDO, nullptr, 0, 0, 0};
            for (uint64_t i = 0; i < n; i++) {
                int b = cin ? 0 : on_curve_thread<G>(in.data(), i);
                b |= window_scalar_mul_thread<G>(in.data(), cin, scratch.data(), wtab.data(), n, i, src, 0, /*use_glv=*/1);
                if (!b && !scratch[i].Z.is_zero()) b |= ERRBIT_NOT_IN_SUBGROUP;
                out[i] = (uint8_t)b;
            }
        } else if constexpr (G::J::CfgT::PSI_KIND == 1) {
            // MNT twists: psi(P) == [t - 1]P — K1 with the fixed half-length scalar, then the psi comparison (capi.hip's order)
            std::vector<typename G::J> scratch(n), wtab(8 * n);
            const ScalarSource src{SCALAR_FIXED, G::J::CfgT::PSI_N, 0, 0, 0};
            for (uint64_t i = 0; i < n; i++) {
                int b = cin ? 0 : on_curve_thread<G>(in.data(), i);
                b |= window_scalar_mul_thread<G>(in.data(), cin, scratch.data(), wtab.data(), n, i, src, 0, /*use_glv=*/0);
                if (!b) b |= psi_check_thread<G>(in.data(), cin, scratch.data(), i);
                out[i] = (uint8_t)b;
            }
        } else {
            return 4;
        }
        fwrite(out.data(), 1, out.size(), stdout);
        return 0;
    } else if (op == "decompress") {
        out.resize(n * G::UNCOMP);
        for (uint64_t i = 0; i < n; i++) bad |= decompress_thread<G>(in.data(), out.data(), i, 0);
    } else {
        out.resize(n * osz);
        const int fl = (G::Fr::N + 1) / 2;
        auto tau = parse_hex(tau_hex, fl);
        std::vector<uint64_t> coeff;
        if (strcmp(coeff_hex, "-")) coeff = parse_hex(coeff_hex, fl);
        std::vector<uint32_t> table((1 + SCALAR_TABLE_ENTRIES) * G::Fr::N);
        Launch<G>::build_table(tau.data(), coeff.empty() ? nullptr : coeff.data(), start, table.data());
        std::vector<typename G::J> scratch(n);
        std::vector<typename G::F> pre(n);
        if (getenv("PTAU_SCALAR_MUL") && !strcmp(getenv("PTAU_SCALAR_MUL"), "bits")) {
            for (uint64_t i = 0; i < n; i++) bad |= scalar_mul_thread<G>(in.data(), cin, scratch.data(), i, table.data(), 0);
        } else {
            std::vector<typename G::J> wtab(n * WINDOW_ENTRIES);
            for (uint64_t i = 0; i < n; i++)
                bad |= window_scalar_mul_thread<G>(in.data(), cin, scratch.data(), wtab.data(), n, i, ScalarSource{SCALAR_TAU_TABLE, table.data(), 0, 0, 0}, 0,
                                                       getenv("PTAU_GLV") && !strcmp(getenv("PTAU_GLV"), "0") ? 0 : 1);
        }
        uint64_t T = (n + 3) / 4;
        for (uint64_t t = 0; t < T; t++) bad |= normalize_thread<G>(scratch.data(), pre.data(), out.data(), cout, n, T, t, 0);
    }
    fwrite(out.data(), 1, out.size(), stdout);
    return bad ? 3 : 0;
}

#define DISPATCH(FQ, FR, C1, C2)                                                                  \
    if (group == 1) return run<GroupTypes<FQ, FR, C1>>(op, n, start, cin, cout, argv[8], argv[9]); \
    return run<GroupTypes<FQ, FR, C2>>(op, n, start, cin, cout, argv[8], argv[9]);

int main(int argc, char **argv) {
    if (argc < 10) return 1;
    std::string op = argv[1];
    int curve = atoi(argv[2]), group = atoi(argv[3]);
    uint64_t n = strtoull(argv[4], 0, 10), start = strtoull(argv[5], 0, 10);
    int cin = atoi(argv[6]), cout = atoi(argv[7]);
    if (op == "msmplan") {
        const MsmPlanEx pl = msm_plan(std::max<uint64_t>(n, 1), PRF_BITS, 0, (int)start, (uint64_t)cin * 1000);
        printf("%d %d %u %d\n", pl.c, pl.W, pl.B, pl.bits);
        return 0;
    }
    switch (curve) {
    case 0: DISPATCH(BlsFqR29, BlsFr, BlsG1Cfg, BlsG2Cfg)
    case 1: DISPATCH(Bw6FqR29, Bw6Fr, Bw6G1Cfg, Bw6G2Cfg)
    case 2: DISPATCH(Mnt4FqR29, Mnt4Fr, Mnt4G1Cfg, Mnt4G2Cfg)
    case 3: DISPATCH(Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg, Mnt6G2Cfg)
    }
    return 1;
}
