// BLAKE2b-512 — see blake2b_host.hpp.  Plain C++ (g++), no HIP.
#include "blake2b_host.hpp"

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstring>

#include "../../include/ptau.h"

namespace ptau {
namespace {

const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                        0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
constexpr uint8_t SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

#define PTAU_B2_SCALAR_BODY \
    uint64_t m[16], v[16]; \
    memcpy(m, blk, 128); \
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; } \
    v[12] ^= t0; \
    v[13] ^= t1; \
    if (last) v[14] = ~v[14]; \
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) { \
        v[a] += v[b] + x; v[d] = rotr(v[d] ^ v[a], 32); \
        v[c] += v[d];     v[b] = rotr(v[b] ^ v[c], 24); \
        v[a] += v[b] + y; v[d] = rotr(v[d] ^ v[a], 16); \
        v[c] += v[d];     v[b] = rotr(v[b] ^ v[c], 63); \
    }; \
    for (int r = 0; r < 12; r++) { \
        const uint8_t *s = SIGMA[r]; \
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]); \
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]); \
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]); \
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]); \
    } \
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8]; \

void compress_portable(uint64_t h[8], const uint8_t *blk, uint64_t t0, uint64_t t1, bool last) { PTAU_B2_SCALAR_BODY }
// the same source compiled for BMI2 (rorx: rotate without destroying the source register)
__attribute__((target("bmi2"))) void compress_bmi2(uint64_t h[8], const uint8_t *blk, uint64_t t0, uint64_t t1, bool last) { PTAU_B2_SCALAR_BODY }

// the four G functions of a column (then diagonal) step side by side: rows a, b, c, d of the state in one 256-bit register each
__attribute__((target("avx2"))) void compress_avx2(uint64_t h[8], const uint8_t *blk, uint64_t t0, uint64_t t1, bool last) {
    uint64_t m[16];
    memcpy(m, blk, 128);
    const __m256i r24 = _mm256_setr_epi8(3, 4, 5, 6, 7, 0, 1, 2, 11, 12, 13, 14, 15, 8, 9, 10, 3, 4, 5, 6, 7, 0, 1, 2, 11, 12, 13, 14, 15, 8, 9, 10);
    const __m256i r16 = _mm256_setr_epi8(2, 3, 4, 5, 6, 7, 0, 1, 10, 11, 12, 13, 14, 15, 8, 9, 2, 3, 4, 5, 6, 7, 0, 1, 10, 11, 12, 13, 14, 15, 8, 9);
    __m256i a = _mm256_loadu_si256((const __m256i *)&h[0]);
    __m256i b = _mm256_loadu_si256((const __m256i *)&h[4]);
    __m256i c = _mm256_loadu_si256((const __m256i *)&IV[0]);
    __m256i d = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)&IV[4]),
                                 _mm256_set_epi64x(0, last ? -1LL : 0, (long long)t1, (long long)t0));
    const __m256i a0 = a, b0 = b;
#define PTAU_B2_G(x, y)                                                                              \
    a = _mm256_add_epi64(_mm256_add_epi64(a, b), x);                                                 \
    d = _mm256_xor_si256(d, a); d = _mm256_shuffle_epi32(d, _MM_SHUFFLE(2, 3, 0, 1));                \
    c = _mm256_add_epi64(c, d);                                                                      \
    b = _mm256_xor_si256(b, c); b = _mm256_shuffle_epi8(b, r24);                                     \
    a = _mm256_add_epi64(_mm256_add_epi64(a, b), y);                                                 \
    d = _mm256_xor_si256(d, a); d = _mm256_shuffle_epi8(d, r16);                                     \
    c = _mm256_add_epi64(c, d);                                                                      \
    b = _mm256_xor_si256(b, c); b = _mm256_or_si256(_mm256_srli_epi64(b, 63), _mm256_add_epi64(b, b));
#pragma GCC unroll 12
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = SIGMA[r];
        __m256i x = _mm256_set_epi64x((long long)m[s[6]], (long long)m[s[4]], (long long)m[s[2]], (long long)m[s[0]]);
        __m256i y = _mm256_set_epi64x((long long)m[s[7]], (long long)m[s[5]], (long long)m[s[3]], (long long)m[s[1]]);
        PTAU_B2_G(x, y)
        b = _mm256_permute4x64_epi64(b, _MM_SHUFFLE(0, 3, 2, 1));
        c = _mm256_permute4x64_epi64(c, _MM_SHUFFLE(1, 0, 3, 2));
        d = _mm256_permute4x64_epi64(d, _MM_SHUFFLE(2, 1, 0, 3));
        x = _mm256_set_epi64x((long long)m[s[14]], (long long)m[s[12]], (long long)m[s[10]], (long long)m[s[8]]);
        y = _mm256_set_epi64x((long long)m[s[15]], (long long)m[s[13]], (long long)m[s[11]], (long long)m[s[9]]);
        PTAU_B2_G(x, y)
        b = _mm256_permute4x64_epi64(b, _MM_SHUFFLE(2, 1, 0, 3));
        c = _mm256_permute4x64_epi64(c, _MM_SHUFFLE(1, 0, 3, 2));
        d = _mm256_permute4x64_epi64(d, _MM_SHUFFLE(0, 3, 2, 1));
    }
#undef PTAU_B2_G
    a = _mm256_xor_si256(_mm256_xor_si256(a, c), a0);
    b = _mm256_xor_si256(_mm256_xor_si256(b, d), b0);
    _mm256_storeu_si256((__m256i *)&h[0], a);
    _mm256_storeu_si256((__m256i *)&h[4], b);
}

// AVX-512 (F + VL): the rotations are single instructions (vprorq) and the message schedule is two vpermi2q per round — each
// picks eight of the block's sixteen words out of two 512-bit registers (the AVX2 form assembles them with vpinsrq: 16
// loads + 12 inserts per round, its bottleneck on Zen).
struct alignas(64) B2Idx { uint64_t v[12][2][8]; };
constexpr B2Idx make_b2idx() {
    B2Idx t{};
    for (int r = 0; r < 12; r++)
        for (int h = 0; h < 2; h++)
            for (int k = 0; k < 4; k++) {
                t.v[r][h][k] = SIGMA[r][8 * h + 2 * k];           // x: the first word of each G
                t.v[r][h][4 + k] = SIGMA[r][8 * h + 2 * k + 1];   // y: the second
            }
    return t;
}
constexpr B2Idx B2IDX = make_b2idx();
__attribute__((target("avx512f,avx512vl,avx2"))) void compress_avx512(uint64_t h[8], const uint8_t *blk, uint64_t t0, uint64_t t1, bool last) {
    const __m512i m0 = _mm512_loadu_si512((const void *)blk), m1 = _mm512_loadu_si512((const void *)(blk + 64));
    __m256i a = _mm256_loadu_si256((const __m256i *)&h[0]);
    __m256i b = _mm256_loadu_si256((const __m256i *)&h[4]);
    __m256i c = _mm256_loadu_si256((const __m256i *)&IV[0]);
    __m256i d = _mm256_xor_si256(_mm256_loadu_si256((const __m256i *)&IV[4]),
                                 _mm256_set_epi64x(0, last ? -1LL : 0, (long long)t1, (long long)t0));
    const __m256i a0 = a, b0 = b;
#define PTAU_B2_G512(x, y)                                                                  \
    a = _mm256_add_epi64(_mm256_add_epi64(a, b), x);                                         \
    d = _mm256_ror_epi64(_mm256_xor_si256(d, a), 32);                                        \
    c = _mm256_add_epi64(c, d);                                                              \
    b = _mm256_ror_epi64(_mm256_xor_si256(b, c), 24);                                        \
    a = _mm256_add_epi64(_mm256_add_epi64(a, b), y);                                         \
    d = _mm256_ror_epi64(_mm256_xor_si256(d, a), 16);                                        \
    c = _mm256_add_epi64(c, d);                                                              \
    b = _mm256_ror_epi64(_mm256_xor_si256(b, c), 63);
#pragma GCC unroll 12
    for (int r = 0; r < 12; r++) {
        const __m512i col = _mm512_permutex2var_epi64(m0, _mm512_load_si512((const void *)B2IDX.v[r][0]), m1);
        const __m512i dia = _mm512_permutex2var_epi64(m0, _mm512_load_si512((const void *)B2IDX.v[r][1]), m1);
        PTAU_B2_G512(_mm512_castsi512_si256(col), _mm512_extracti64x4_epi64(col, 1))
        b = _mm256_permute4x64_epi64(b, _MM_SHUFFLE(0, 3, 2, 1));
        c = _mm256_permute4x64_epi64(c, _MM_SHUFFLE(1, 0, 3, 2));
        d = _mm256_permute4x64_epi64(d, _MM_SHUFFLE(2, 1, 0, 3));
        PTAU_B2_G512(_mm512_castsi512_si256(dia), _mm512_extracti64x4_epi64(dia, 1))
        b = _mm256_permute4x64_epi64(b, _MM_SHUFFLE(2, 1, 0, 3));
        c = _mm256_permute4x64_epi64(c, _MM_SHUFFLE(1, 0, 3, 2));
        d = _mm256_permute4x64_epi64(d, _MM_SHUFFLE(0, 3, 2, 1));
    }
#undef PTAU_B2_G512
    a = _mm256_xor_si256(_mm256_xor_si256(a, c), a0);
    b = _mm256_xor_si256(_mm256_xor_si256(b, d), b0);
    _mm256_storeu_si256((__m256i *)&h[0], a);
    _mm256_storeu_si256((__m256i *)&h[4], b);
}

// Which form is faster depends on the core (measured: AVX2 +14 % on a 2.1 GHz Xeon, -38 % on an EPYC 9575F, whose scalar
// pipes beat the vpinsrq-built message vectors), so the first use times every available form on 512 KB and keeps the winner.
std::atomic<int> g_impl{-1};
std::once_flag g_pick_once;
constexpr int N_IMPL = 4;
inline void compress_with(int impl, uint64_t h[8], const uint8_t *blk, uint64_t t0, uint64_t t1, bool last) {
    if (impl == 3) compress_avx512(h, blk, t0, t1, last);
    else if (impl == 1) compress_avx2(h, blk, t0, t1, last);
    else if (impl == 2) compress_bmi2(h, blk, t0, t1, last);
    else compress_portable(h, blk, t0, t1, last);
}
bool have_impl(int impl) {
    switch (impl) {
    case 0: return true;
    case 1: return __builtin_cpu_supports("avx2");
    case 2: return __builtin_cpu_supports("bmi2");
    case 3: return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx2");
    }
    return false;
}
// The timing race runs ONCE per process (std::call_once: the two hashing threads of a cold first verification start
// together), on a local probe block; a forced choice made earlier (blake2b_implementation) is kept.
int pick() {
    const int cur = g_impl.load(std::memory_order_acquire);
    if (cur >= 0) return cur;
    std::call_once(g_pick_once, [] {
        if (g_impl.load(std::memory_order_acquire) >= 0) return;
        uint8_t probe[128] = {0};
        double best[N_IMPL] = {1e30, 1e30, 1e30, 1e30};
        for (int rep = 0; rep < 3; rep++)
            for (int impl = 0; impl < N_IMPL; impl++) {
                if (!have_impl(impl)) continue;
                uint64_t hh[8];
                memcpy(hh, IV, sizeof hh);
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 4096; i++) compress_with(impl, hh, probe, 128ull * i, 0, false);
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                probe[rep] ^= (uint8_t)hh[0];   // keep the work observable
                if (dt < best[impl]) best[impl] = dt;
            }
        int w = 0;
        for (int impl = 1; impl < N_IMPL; impl++)
            if (best[impl] < best[w]) w = impl;
        g_impl.store(w, std::memory_order_release);
    });
    return g_impl.load(std::memory_order_acquire);
}
inline void compress(uint64_t h[8], const uint8_t *blk, uint64_t t0, uint64_t t1, bool last) { compress_with(pick(), h, blk, t0, t1, last); }

}  // namespace

int blake2b_implementation(int force) {
    if (force >= 0 && force < N_IMPL && have_impl(force)) g_impl.store(force, std::memory_order_release);
    return pick();
}

Blake2b::Blake2b() {
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010040ULL;   // digest 64 bytes, no key, fanout = depth = 1
}
void Blake2b::update(const uint8_t *p, size_t n) {
    while (n) {
        if (fill == 128) {   // buffer full and more data follows: not the last block
            t0 += 128;
            if (t0 < 128) t1++;
            compress(h, buf, t0, t1, false);
            fill = 0;
        }
        if (fill == 0) {   // whole blocks straight from the caller's buffer, the last one kept back (it may be THE last)
            while (n > 128) {
                t0 += 128;
                if (t0 < 128) t1++;
                compress(h, p, t0, t1, false);
                p += 128;
                n -= 128;
            }
        }
        const size_t k = std::min(n, (size_t)128 - fill);
        memcpy(buf + fill, p, k);
        fill += k;
        p += k;
        n -= k;
    }
}
void Blake2b::finish(uint8_t out[64]) {
    t0 += fill;
    if (t0 < fill) t1++;
    memset(buf + fill, 0, 128 - fill);
    compress(h, buf, t0, t1, true);
    memcpy(out, h, 64);
}

}  // namespace ptau

extern "C" void ptau_blake2b512(const uint8_t *data, size_t len, uint8_t out[64]) {
    ptau::Blake2b b;
    b.update(data, len);
    b.finish(out);
}
extern "C" int ptau_blake2b_implementation(int force) { return ptau::blake2b_implementation(force); }
