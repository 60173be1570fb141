// Host-side key material of a contribution: setup_utils::derive_rng_from_seed, Phase1::key_generation,
// setup_utils::compute_g2_s / hash_to_g2 (reference call sites
// /root/reference/src/bin/contribute.rs:783-789 — same seed -> rng -> keys — and :811-823).
//
// O(1) work per contribution (three field samples, six G1 / six G2 scalar multiplications, three hash-to-G2), so it is
// host code by design; it reuses the kernels' own host+device field / curve templates.
//
// PARITY UNPINNED.  The bodies live in crates that are not in /root/reference (setup-utils / phase1 @ nimiq/snark-setup
// bd530da; rand 0.8 / rand_chacha 0.3.1, Cargo.lock:2927-2929; ark-ff / ark-ec 0.4.2) and there is no Rust toolchain
// here.  What is pinned: the ChaCha20 block function against the published RFC 7539 / all-zero-key vectors
// (tests/test_keygen.py).  What is restated from memory and cannot be checked offline: the order in which the generator
// is consumed (Fr::rand x3; then per key G1::rand, ...), ark-ff's rejection sampling of field elements (random bits are
// the MONTGOMERY representation), ark-ec's G::rand (x, then a bool for the larger root, cofactor cleared),
// from_random_bytes' masks and flag rules, and hash_to_g2's loop.  Keys made here are always VALID (every proof of
// knowledge verifies, under this library and under any correct verifier that is given g2_s); they equal the
// reference's byte for byte only if the recollection is exact.  The Rust shim therefore passes arkworks' own keys and
// g2_s points where it has them.
#include <cstdio>
#include <cstring>
#include <future>
#include <string>
#include <vector>

#include "../../include/ptau.h"
#include "kernels_impl.hpp"

namespace {

using namespace ptau;

// ---- ChaCha20 (RFC 7539 block function) as rand_chacha 0.3.1's ChaCha20Rng: 64-bit block counter from 0 in words 12-13,
// 64-bit stream id 0 in words 14-15, output = the keystream as little-endian 32-bit words in order ----
inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    memcpy(x, s, sizeof(x));
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int i = 0; i < 10; i++) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

struct ChaChaRng {
    uint32_t key[8];
    uint64_t block = 0;
    uint32_t buf[16];
    int idx = 16;
    explicit ChaChaRng(const uint8_t seed[32]) {
        for (int i = 0; i < 8; i++) key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    }
    uint32_t next_u32() {
        if (idx == 16) {
            chacha20_block(key, block++, buf);
            idx = 0;
        }
        return buf[idx++];
    }
    uint64_t next_u64() {   // rand_core BlockRng: two consecutive words, low first
        const uint64_t lo = next_u32();
        return lo | ((uint64_t)next_u32() << 32);
    }
    bool gen_bool() { return (int32_t)next_u32() < 0; }   // rand 0.8 Standard for bool: the sign bit of next_u32
    uint8_t gen_u8() { return (uint8_t)next_u32(); }     // rand 0.8 Standard for u8: next_u32 truncated
};

// The same modulus on packed 32-bit limbs with R = 2^(32 N32) = 2^(64 N64): ark-ff's Montgomery radix.
template <class Fq>
struct Mont32 { using type = typename Fq::Twin; };
template <class P>
struct Mont32<Fp<P>> { using type = Fp<P>; };

// ark-ff 0.4.2 `impl Distribution<Fp<P, N>> for Standard`: N random u64 taken as the MONTGOMERY representation, the bits
// above the modulus length cleared, rejected while >= modulus.  Returns the CANONICAL value as packed 32-bit words.
template <class M32>
void field_rand_canonical(ChaChaRng &rng, uint32_t *canon) {
    constexpr int N = M32::N;
    static_assert(N % 2 == 0, "ark-ff limbs are 64-bit");
    constexpr int shave = 32 * N - M32::Params::BITS;
    M32 t;
    for (;;) {
        for (int i = 0; i < N / 2; i++) {
            const uint64_t v = rng.next_u64();
            t.l[2 * i] = (uint32_t)v;
            t.l[2 * i + 1] = (uint32_t)(v >> 32);
        }
        const uint64_t mask = shave >= 64 ? 0ull : (~0ull >> shave);
        t.l[N - 2] &= (uint32_t)mask;
        t.l[N - 1] &= (uint32_t)(mask >> 32);
        if (M32::canonical_lt_mod(t.l)) break;
    }
    t.to_canonical(canon);
}

template <int NW>
void words_to_bytes(const uint32_t *w, uint8_t *out, int nbytes) {
    for (int i = 0; i < nbytes; i++) out[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
}

template <class G>
struct HostGroup {
    using F = typename G::F;
    using B = typename F::Base;
    using A = typename G::A;
    using J = typename G::J;
    using Cfg = typename J::CfgT;
    using M32 = typename Mont32<B>::type;
    static constexpr int DEG = F::DEG;
    static constexpr int BB = B::BYTES;

    // k * p, k = `bits`-bit integer in 32-bit words (MSB-first double-and-add with mixed additions)
    static J mul_words(const A &p, const uint32_t *k, int bits) {
        J acc = J::infinity();
        for (int i = bits - 1; i >= 0; i--) {
            acc = J::dbl(acc);
            if ((k[i >> 5] >> (i & 31)) & 1u) acc = J::add_mixed(acc, p);
        }
        return acc;
    }
    static A to_affine(const J &q) {
        A a;
        if (q.Z.is_zero()) {
            a.inf = true;
            a.x = F::zero();
            a.y = F::zero();
            return a;
        }
        const F zi = F::inverse(q.Z), zi2 = F::sqr(zi);
        a.x = F::mul(q.X, zi2);
        a.y = F::mul(q.Y, F::mul(zi2, zi));
        a.inf = false;
        return a;
    }
    static A clear_cofactor(const A &p) { return to_affine(mul_words(p, Cfg::COFACTOR, Cfg::COFACTOR_BITS)); }

    // ark-ec get_point_from_x_unchecked(x, greatest): the root with the larger canonical value when `greatest`
    static bool point_from_x(const F &x, bool greatest, A &out) {
        F y;
        if (!F::sqrt(curve_rhs<F, Cfg>(x), y)) return false;
        const bool y_is_larger = y.sign_class() > 0;
        out.x = x;
        out.y = (y_is_larger == greatest) ? y : F::neg(y);
        out.inf = false;
        return true;
    }
    // ark-ec `impl Distribution<Projective<P>> for Standard` (base-field groups only: G1): x <- Fq::rand, greatest <- bool,
    // retry until x is on the curve, clear the cofactor
    static A rand_point(ChaChaRng &rng) {
        static_assert(DEG == 1, "G::rand is used for G1 only");
        for (;;) {
            uint32_t canon[M32::N];
            field_rand_canonical<M32>(rng, canon);
            uint8_t bytes[BB + 4];
            words_to_bytes<M32::N>(canon, bytes, BB);
            uint8_t fl = 0;
            F x;
            (void)F::read_bytes(bytes, false, &fl, x);
            const bool greatest = rng.gen_bool();
            A p;
            if (point_from_x(x, greatest, p)) return clear_cofactor(p);
        }
    }
    // ark-ff Fp::from_random_bytes_with_flags on one base-field slice of BB (+ flag bits in the last byte when
    // with_flags): the bits above the modulus length are cleared, the value must be canonical.
    static bool base_from_random_bytes(const uint8_t *in, bool with_flags, uint8_t *flags, B &out) {
        uint8_t buf[BB];
        memcpy(buf, in, BB);
        constexpr int top_bits = B::Params::BITS - 8 * (BB - 1);   // modulus bits that live in the last byte
        if (with_flags) *flags = buf[BB - 1] & 0xC0u;
        buf[BB - 1] &= (uint8_t)((1u << top_bits) - 1u);
        uint8_t dummy = 0;
        return B::read_bytes(buf, false, &dummy, out);
    }
    // ark-ec Affine::from_random_bytes on `COMP` bytes: x from the slices (flags in the last one); infinity only for
    // x = 0 with the infinity flag alone; both flags set -> None; else get_point_from_x_unchecked(x, flags.is_positive()).
    static bool from_random_bytes(const uint8_t *bytes, A &out) {
        B c[3];
        uint8_t flags = 0;
        for (int i = 0; i < DEG; i++)
            if (!base_from_random_bytes(bytes + i * BB, i == DEG - 1, &flags, c[i])) return false;
        F x;
        if constexpr (DEG == 1) x = c[0];
        else if constexpr (DEG == 2) x = F::make(c[0], c[1]);
        else x = F::make(c[0], c[1], c[2]);
        const bool neg = (flags & 0x80u) != 0, inf = (flags & 0x40u) != 0;
        if (neg && inf) return false;
        if (inf) {
            if (!x.is_zero()) return false;
            out.inf = true;
            out.x = F::zero();
            out.y = F::zero();
            return true;
        }
        return point_from_x(x, /*greatest = y_is_positive*/ !neg, out);
    }
    // setup_utils::hash_to_g2: ChaChaRng from the first 32 bytes of the digest; COMP random bytes (one u8 per next_u32)
    // until from_random_bytes yields a point whose cofactor multiple is not the identity.
    static A hash_to_group(const uint8_t digest[64]) {
        ChaChaRng rng(digest);
        std::vector<uint8_t> bytes(G::COMP);
        for (;;) {
            for (auto &b : bytes) b = rng.gen_u8();
            A p;
            if (!from_random_bytes(bytes.data(), p)) continue;
            if (p.inf) continue;
            const A s = clear_cofactor(p);
            if (!s.inf) return s;
        }
    }
    static A mul_scalar(const A &p, const uint32_t *k, int bits) { return to_affine(mul_words(p, k, bits)); }
};

template <class G1, class G2>
struct KeyGen {
    using H1 = HostGroup<G1>;
    using H2 = HostGroup<G2>;
    using Fr = typename G1::Fr;
    static constexpr int U1 = G1::UNCOMP, U2 = G2::UNCOMP;

    // compute_g2_s: hash_to_g2(BLAKE2b(personalization | digest | g1_s | g1_s_x)), the two points uncompressed
    static typename G2::A g2_s(const uint8_t digest[64], const uint8_t *g1_s, const uint8_t *g1_s_x, uint8_t personalization) {
        std::vector<uint8_t> msg(1 + 64 + 2 * (size_t)U1);
        msg[0] = personalization;
        memcpy(&msg[1], digest, 64);
        memcpy(&msg[65], g1_s, U1);
        memcpy(&msg[65 + U1], g1_s_x, U1);
        uint8_t h[64];
        ptau_blake2b512(msg.data(), msg.size(), h);
        return H2::hash_to_group(h);
    }
    static int compute_g2_s_all(const uint8_t digest[64], const uint8_t *pk, uint8_t *out) {
        for (int x = 0; x < 3; x++) {
            const typename G2::A p = g2_s(digest, pk + (2 * x) * U1, pk + (2 * x + 1) * U1, (uint8_t)x);
            write_point<typename G2::F>(out + x * U2, p, false);
        }
        return PTAU_OK;
    }
    // tau, alpha, beta <- Fr::rand: the first draws of Phase1::key_generation (they do not depend on the digest)
    static void sample_private(ChaChaRng &rng, uint32_t (&k)[3][Fr::N], uint64_t *sk[3]) {
        for (int x = 0; x < 3; x++) {
            field_rand_canonical<Fr>(rng, k[x]);
            for (int i = 0; i < (Fr::N + 1) / 2; i++)
                sk[x][i] = (uint64_t)k[x][2 * i] | (2 * i + 1 < Fr::N ? (uint64_t)k[x][2 * i + 1] << 32 : 0ull);
        }
    }
    static int private_only(const uint8_t seed[32], uint64_t *sk[3]) {
        ChaChaRng rng(seed);
        uint32_t k[3][Fr::N];
        sample_private(rng, k, sk);
        return PTAU_OK;
    }
    // Phase1::key_generation
    static int generate(const uint8_t seed[32], const uint8_t digest[64], uint64_t *sk[3], uint8_t *pk) {
        ChaChaRng rng(seed);
        uint32_t k[3][Fr::N];
        sample_private(rng, k, sk);
        for (int x = 0; x < 3; x++) {
            const typename G1::A s = H1::rand_point(rng);
            const typename G1::A sx = H1::mul_scalar(s, k[x], Fr::Params::BITS);
            uint8_t *p_s = pk + (2 * x) * U1, *p_sx = pk + (2 * x + 1) * U1;
            write_point<typename G1::F>(p_s, s, false);
            write_point<typename G1::F>(p_sx, sx, false);
            const typename G2::A h = g2_s(digest, p_s, p_sx, (uint8_t)x);
            const typename G2::A hx = H2::mul_scalar(h, k[x], Fr::Params::BITS);
            write_point<typename G2::F>(pk + 6 * U1 + x * U2, hx, false);
        }
        return PTAU_OK;
    }
};

using BlsK = KeyGen<GroupTypes<BlsFqR29, BlsFr, BlsG1Cfg>, GroupTypes<BlsFqR29, BlsFr, BlsG2Cfg>>;
using Bw6K = KeyGen<GroupTypes<Bw6FqR29, Bw6Fr, Bw6G1Cfg>, GroupTypes<Bw6FqR29, Bw6Fr, Bw6G2Cfg>>;
using Mnt4K = KeyGen<GroupTypes<Mnt4FqR29, Mnt4Fr, Mnt4G1Cfg>, GroupTypes<Mnt4FqR29, Mnt4Fr, Mnt4G2Cfg>>;
using Mnt6K = KeyGen<GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg>, GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G2Cfg>>;

// Validity of serialized UNCOMPRESSED points as arkworks' checked deserialisation would see them: canonical coordinates,
// on the curve, in the prime-order subgroup ([r]P == O; cofactor-1 groups need no multiplication).  The point at infinity
// passes here: check_same_ratio rejects it.  One host thread per point (the few points of a PublicKey).
template <class G>
int validate_points(const uint8_t *pts, size_t n) {
    using H = HostGroup<G>;
    using F = typename G::F;
    using Cfg = typename H::Cfg;
    using Fr = typename G::Fr;
    std::vector<std::future<int>> futs;
    for (size_t i = 0; i < n; i++)
        futs.push_back(std::async(std::launch::async, [pts, i]() -> int {
            typename G::A p;
            if (!read_uncompressed<F>(pts + i * G::UNCOMP, p)) return PTAU_E_POINT;
            if (p.inf) return PTAU_OK;
            if (!(F::sqr(p.y) == curve_rhs<F, Cfg>(p.x))) return PTAU_E_POINT;
            if (Cfg::COFACTOR_BITS == 1 && Cfg::COFACTOR[0] == 1u) return PTAU_OK;
            return H::mul_words(p, Fr::Params::MOD, Fr::Params::BITS).is_inf() ? PTAU_OK : PTAU_E_NOT_IN_SUBGROUP;
        }));
    int rc = PTAU_OK;
    for (auto &f : futs) {
        const int r = f.get();
        if (r && !rc) rc = r;
    }
    return rc;
}

int fail(char *buf, size_t len, int code, const char *msg) {
    if (buf && len) snprintf(buf, len, "%s", msg);
    return code;
}

}  // namespace

namespace ptau {
// capi.hip: the PublicKey at the tail of a response and a caller-supplied g2_s are attacker-controlled bytes that go
// straight into pairings — validated first, as the reference's checked deserialisation of the PublicKey does.
int host_validate_points(int curve, int group, const uint8_t *pts, size_t n) {
    switch (curve * 2 + (group - 1)) {
    case 0: return validate_points<GroupTypes<BlsFqR29, BlsFr, BlsG1Cfg>>(pts, n);
    case 1: return validate_points<GroupTypes<BlsFqR29, BlsFr, BlsG2Cfg>>(pts, n);
    case 2: return validate_points<GroupTypes<Bw6FqR29, Bw6Fr, Bw6G1Cfg>>(pts, n);
    case 3: return validate_points<GroupTypes<Bw6FqR29, Bw6Fr, Bw6G2Cfg>>(pts, n);
    case 4: return validate_points<GroupTypes<Mnt4FqR29, Mnt4Fr, Mnt4G1Cfg>>(pts, n);
    case 5: return validate_points<GroupTypes<Mnt4FqR29, Mnt4Fr, Mnt4G2Cfg>>(pts, n);
    case 6: return validate_points<GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg>>(pts, n);
    case 7: return validate_points<GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G2Cfg>>(pts, n);
    }
    return PTAU_E_ARG;
}
}  // namespace ptau

extern "C" {

int ptau_chacha20_rng_words(const uint8_t seed[32], uint64_t word_offset, uint32_t *out, size_t n_words) {
    if (!seed || (!out && n_words)) return PTAU_E_ARG;
    ChaChaRng rng(seed);
    rng.block = word_offset / 16;
    for (uint64_t i = 0; i < word_offset % 16; i++) (void)rng.next_u32();
    for (size_t i = 0; i < n_words; i++) out[i] = rng.next_u32();
    return PTAU_OK;
}

int ptau_compute_g2_s(int curve, const uint8_t digest[64], const uint8_t *public_key, size_t public_key_len, uint8_t *g2_s_out,
                      char *errbuf, size_t err_len) {
    if (!digest || !public_key || !g2_s_out) return fail(errbuf, err_len, PTAU_E_ARG, "null pointer argument");
    const int u1 = ptau_point_size(curve, 1, 0), u2 = ptau_point_size(curve, 2, 0);
    if (u1 < 0) return fail(errbuf, err_len, PTAU_E_ARG, "unknown curve id");
    if (public_key_len != 6 * (size_t)u1 + 3 * (size_t)u2) return fail(errbuf, err_len, PTAU_E_LEN, "public key length does not match the curve");
    switch (curve) {
    case PTAU_CURVE_BLS12_377: return BlsK::compute_g2_s_all(digest, public_key, g2_s_out);
    case PTAU_CURVE_BW6_761: return Bw6K::compute_g2_s_all(digest, public_key, g2_s_out);
    case PTAU_CURVE_MNT4_753: return Mnt4K::compute_g2_s_all(digest, public_key, g2_s_out);
    case PTAU_CURVE_MNT6_753: return Mnt6K::compute_g2_s_all(digest, public_key, g2_s_out);
    }
    return fail(errbuf, err_len, PTAU_E_ARG, "unknown curve id");
}

int ptau_private_key_from_seed(int curve, const uint8_t *seed, size_t seed_len, uint64_t *tau, uint64_t *alpha, uint64_t *beta,
                               char *errbuf, size_t err_len) {
    if (!seed || !tau || !alpha || !beta) return fail(errbuf, err_len, PTAU_E_ARG, "null pointer argument");
    if (seed_len < 32) return fail(errbuf, err_len, PTAU_E_LEN, "seed shorter than 32 bytes (derive_rng_from_seed takes the first 32)");
    uint64_t *sk[3] = {tau, alpha, beta};
    switch (curve) {
    case PTAU_CURVE_BLS12_377: return BlsK::private_only(seed, sk);
    case PTAU_CURVE_BW6_761: return Bw6K::private_only(seed, sk);
    case PTAU_CURVE_MNT4_753: return Mnt4K::private_only(seed, sk);
    case PTAU_CURVE_MNT6_753: return Mnt6K::private_only(seed, sk);
    }
    return fail(errbuf, err_len, PTAU_E_ARG, "unknown curve id");
}

int ptau_key_generation(int curve, const uint8_t *seed, size_t seed_len, const uint8_t digest[64], uint64_t *tau, uint64_t *alpha,
                        uint64_t *beta, uint8_t *public_key, size_t public_key_len, char *errbuf, size_t err_len) {
    if (!seed || !digest || !tau || !alpha || !beta || !public_key) return fail(errbuf, err_len, PTAU_E_ARG, "null pointer argument");
    if (seed_len < 32) return fail(errbuf, err_len, PTAU_E_LEN, "seed shorter than 32 bytes (derive_rng_from_seed takes the first 32)");
    const int u1 = ptau_point_size(curve, 1, 0), u2 = ptau_point_size(curve, 2, 0);
    if (u1 < 0) return fail(errbuf, err_len, PTAU_E_ARG, "unknown curve id");
    if (public_key_len != 6 * (size_t)u1 + 3 * (size_t)u2) return fail(errbuf, err_len, PTAU_E_LEN, "public key buffer length does not match the curve");
    uint64_t *sk[3] = {tau, alpha, beta};
    switch (curve) {
    case PTAU_CURVE_BLS12_377: return BlsK::generate(seed, digest, sk, public_key);
    case PTAU_CURVE_BW6_761: return Bw6K::generate(seed, digest, sk, public_key);
    case PTAU_CURVE_MNT4_753: return Mnt4K::generate(seed, digest, sk, public_key);
    case PTAU_CURVE_MNT6_753: return Mnt6K::generate(seed, digest, sk, public_key);
    }
    return fail(errbuf, err_len, PTAU_E_ARG, "unknown curve id");
}

}  // extern "C"
