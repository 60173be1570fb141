/* CPU oracle for the chunked powers-of-tau hot path — TEST INFRASTRUCTURE ONLY (see ptau_oracle.h).
 *
 * Restates (upstream sources are un-vendored, parity UNPINNED):
 *   ark-ff 0.4.2  Fp / QuadExtField / CubicExtField (Montgomery form)         -> fp_*, fe_*
 *   ark-ec 0.4.2  short_weierstrass Jacobian double / mixed add,
 *                 sw_double_and_add_affine (MSB-first double-and-add)          -> jac_*
 *   setup-utils   generate_powers_of_tau, batch_exp (+ batch_normalization),
 *                 read_batch / write_batch                                      -> oracle_apply_powers
 *   phase1        Phase1Parameters, Phase1::computation, Phase1::decompress     -> oracle_params / _computation / _decompress
 *   blake2 0.10.6 BLAKE2b-512                                                   -> oracle_blake2b512
 * Reference call sites: /root/reference/src/bin/contribute.rs:811-823, 968-986;
 * /root/reference/src/utils.rs:326-352.
 *
 * 64-bit limbs, runtime limb count; every Montgomery constant is derived at start-up from the
 * modulus alone (independent of the constants the HIP library is generated with).
 */
#include "ptau_oracle.h"

#include <immintrin.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define MAXL 12 /* 768 bits */
#define MAXD 3

/* ------------------------------------------------------------------ prime field */
typedef struct {
    int n, bits;
    u64 p[MAXL], inv, r1[MAXL], r2[MAXL];
} fpctx;

static int bn_geq(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static u64 bn_add(u64 *z, const u64 *a, const u64 *b, int n) {
    u64 c = 0;
    for (int i = 0; i < n; i++) {
        u128 v = (u128)a[i] + b[i] + c;
        z[i] = (u64)v;
        c = (u64)(v >> 64);
    }
    return c;
}
static u64 bn_sub(u64 *z, const u64 *a, const u64 *b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; i++) {
        u128 v = (u128)a[i] - b[i] - br;
        z[i] = (u64)v;
        br = (u64)(v >> 64) & 1;
    }
    return br;
}
static int bn_is_zero(const u64 *a, int n) {
    u64 o = 0;
    for (int i = 0; i < n; i++) o |= a[i];
    return o == 0;
}
static int bn_bitlen(const u64 *a, int n) {
    for (int i = n - 1; i >= 0; i--)
        if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]);
    return 0;
}
static int bn_bit(const u64 *a, int i) { return (int)((a[i >> 6] >> (i & 63)) & 1); }

/* fixed-N bodies (carry chains the compiler keeps in registers); the generic loop serves other sizes */
#define DEF_FP_ADDSUB(N)                                                                      \
    static void fp_add_##N(const fpctx *c, u64 *z, const u64 *a, const u64 *b) {              \
        u64 t[N], s[N];                                                                       \
        unsigned char cy = 0, br = 0;                                                         \
        _Pragma("GCC unroll 12") for (int i = 0; i < N; i++) cy = _addcarry_u64(cy, a[i], b[i], (unsigned long long *)&t[i]); \
        _Pragma("GCC unroll 12") for (int i = 0; i < N; i++) br = _subborrow_u64(br, t[i], c->p[i], (unsigned long long *)&s[i]); \
        const int reduced = cy | !br; /* a + b >= p */                                        \
        _Pragma("GCC unroll 12") for (int i = 0; i < N; i++) z[i] = reduced ? s[i] : t[i];    \
    }                                                                                         \
    static void fp_sub_##N(const fpctx *c, u64 *z, const u64 *a, const u64 *b) {              \
        u64 t[N];                                                                             \
        unsigned char br = 0, cy = 0;                                                         \
        _Pragma("GCC unroll 12") for (int i = 0; i < N; i++) br = _subborrow_u64(br, a[i], b[i], (unsigned long long *)&t[i]); \
        const u64 mask = (u64)0 - br;                                                         \
        _Pragma("GCC unroll 12") for (int i = 0; i < N; i++) cy = _addcarry_u64(cy, t[i], c->p[i] & mask, (unsigned long long *)&z[i]); \
    }
DEF_FP_ADDSUB(4)
DEF_FP_ADDSUB(6)
DEF_FP_ADDSUB(12)
static void fp_add(const fpctx *c, u64 *z, const u64 *a, const u64 *b) {
    switch (c->n) {
    case 4: fp_add_4(c, z, a, b); return;
    case 6: fp_add_6(c, z, a, b); return;
    case 12: fp_add_12(c, z, a, b); return;
    }
    u64 t[MAXL];
    u64 cy = bn_add(t, a, b, c->n);
    if (cy || bn_geq(t, c->p, c->n)) bn_sub(t, t, c->p, c->n);
    memcpy(z, t, 8 * c->n);
}
static void fp_sub(const fpctx *c, u64 *z, const u64 *a, const u64 *b) {
    switch (c->n) {
    case 4: fp_sub_4(c, z, a, b); return;
    case 6: fp_sub_6(c, z, a, b); return;
    case 12: fp_sub_12(c, z, a, b); return;
    }
    u64 t[MAXL];
    if (bn_sub(t, a, b, c->n)) bn_add(t, t, c->p, c->n);
    memcpy(z, t, 8 * c->n);
}
static void fp_neg(const fpctx *c, u64 *z, const u64 *a) {
    if (bn_is_zero(a, c->n)) { memset(z, 0, 8 * c->n); return; }
    bn_sub(z, c->p, a, c->n);
}
/* CIOS Montgomery product z = x*y/R mod p; fixed-N bodies so the compiler unrolls them */
#define DEF_FP_MUL(N)                                                                         \
    static void fp_mul_##N(const fpctx *c, u64 *z, const u64 *x, const u64 *y) {              \
        u64 t[N + 2] = {0};                                                                   \
        _Pragma("GCC unroll 12") for (int i = 0; i < N; i++) {                                \
            u64 carry = 0;                                                                    \
            _Pragma("GCC unroll 12") for (int j = 0; j < N; j++) {                            \
                u128 v = (u128)x[j] * y[i] + t[j] + carry;                                    \
                t[j] = (u64)v;                                                                \
                carry = (u64)(v >> 64);                                                       \
            }                                                                                 \
            u128 v = (u128)t[N] + carry;                                                      \
            t[N] = (u64)v;                                                                    \
            t[N + 1] = (u64)(v >> 64);                                                        \
            u64 m = t[0] * c->inv;                                                            \
            v = (u128)m * c->p[0] + t[0];                                                     \
            carry = (u64)(v >> 64);                                                           \
            _Pragma("GCC unroll 12") for (int j = 1; j < N; j++) {                            \
                v = (u128)m * c->p[j] + t[j] + carry;                                         \
                t[j - 1] = (u64)v;                                                            \
                carry = (u64)(v >> 64);                                                       \
            }                                                                                 \
            v = (u128)t[N] + carry;                                                           \
            t[N - 1] = (u64)v;                                                                \
            t[N] = t[N + 1] + (u64)(v >> 64);                                                 \
        }                                                                                     \
        if (t[N] || bn_geq(t, c->p, N)) bn_sub(t, t, c->p, N);                                \
        for (int i = 0; i < N; i++) z[i] = t[i];                                              \
    }
DEF_FP_MUL(4)
DEF_FP_MUL(6)
DEF_FP_MUL(12)
static void fp_mul(const fpctx *c, u64 *z, const u64 *x, const u64 *y) {
    switch (c->n) {
    case 4: fp_mul_4(c, z, x, y); break;
    case 6: fp_mul_6(c, z, x, y); break;
    default: fp_mul_12(c, z, x, y); break;
    }
}
static void fp_to_mont(const fpctx *c, u64 *z, const u64 *a) { fp_mul(c, z, a, c->r2); }
static void fp_from_mont(const fpctx *c, u64 *z, const u64 *a) {
    u64 one[MAXL] = {1};
    fp_mul(c, z, a, one);
}
static void fp_pow(const fpctx *c, u64 *z, const u64 *a, const u64 *e, int elimbs) {
    u64 res[MAXL], base[MAXL];
    memcpy(res, c->r1, 8 * c->n);
    memcpy(base, a, 8 * c->n);
    int bl = bn_bitlen(e, elimbs);
    for (int i = bl - 1; i >= 0; i--) {
        fp_mul(c, res, res, res);
        if (bn_bit(e, i)) fp_mul(c, res, res, base);
    }
    memcpy(z, res, 8 * c->n);
}
static void fp_inv(const fpctx *c, u64 *z, const u64 *a) {
    u64 e[MAXL], two[MAXL] = {2};
    bn_sub(e, c->p, two, c->n);
    fp_pow(c, z, a, e, c->n);
}

static int hexval(char ch) {
    if (ch >= '0' && ch <= '9') return ch - '0';
    if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
    return -1;
}
static void fp_init(fpctx *c, const char *hex) {
    memset(c, 0, sizeof *c);
    size_t len = strlen(hex);
    for (size_t i = 0; i < len; i++) {
        int v = hexval(hex[len - 1 - i]);
        c->p[i / 16] |= (u64)v << (4 * (i % 16));
    }
    c->bits = bn_bitlen(c->p, MAXL);
    c->n = (c->bits + 63) / 64;
    /* inv = -p^{-1} mod 2^64 (Newton) */
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - c->p[0] * x;
    c->inv = (u64)0 - x;
    /* R mod p by 64n doublings of 1, R^2 by 64n more */
    u64 t[MAXL] = {1};
    for (int i = 0; i < 128 * c->n; i++) {
        fp_add(c, t, t, t);
        if (i == 64 * c->n - 1) memcpy(c->r1, t, 8 * c->n);
    }
    memcpy(c->r2, t, 8 * c->n);
}

/* ------------------------------------------------------------------ extension field Fq[u]/(u^d - nr) */
typedef struct { u64 c[MAXD][MAXL]; } fe;
typedef struct {
    const fpctx *fp;
    int deg;
    u64 nr[MAXL]; /* Montgomery */
} fectx;

static void fe_zero(fe *z) { memset(z, 0, sizeof *z); }
static void fe_one(const fectx *F, fe *z) { fe_zero(z); memcpy(z->c[0], F->fp->r1, 8 * F->fp->n); }
static int fe_is_zero(const fectx *F, const fe *a) {
    for (int i = 0; i < F->deg; i++)
        if (!bn_is_zero(a->c[i], F->fp->n)) return 0;
    return 1;
}
static int fe_eq(const fectx *F, const fe *a, const fe *b) {
    for (int i = 0; i < F->deg; i++)
        if (memcmp(a->c[i], b->c[i], 8 * F->fp->n)) return 0;
    return 1;
}
static void fe_add(const fectx *F, fe *z, const fe *a, const fe *b) {
    for (int i = 0; i < F->deg; i++) fp_add(F->fp, z->c[i], a->c[i], b->c[i]);
}
static void fe_sub(const fectx *F, fe *z, const fe *a, const fe *b) {
    for (int i = 0; i < F->deg; i++) fp_sub(F->fp, z->c[i], a->c[i], b->c[i]);
}
static void fe_neg(const fectx *F, fe *z, const fe *a) {
    for (int i = 0; i < F->deg; i++) fp_neg(F->fp, z->c[i], a->c[i]);
}
static void fe_dbl(const fectx *F, fe *z, const fe *a) { fe_add(F, z, a, a); }
static void fe_mul(const fectx *F, fe *z, const fe *a, const fe *b) {
    const fpctx *c = F->fp;
    if (F->deg == 1) { /* fp_mul is alias-safe */
        fp_mul(c, z->c[0], a->c[0], b->c[0]);
        return;
    }
    fe r;
    u64 t[MAXL], s[MAXL];
    if (F->deg == 2) {
        fp_mul(c, r.c[0], a->c[0], b->c[0]);
        fp_mul(c, t, a->c[1], b->c[1]);
        fp_mul(c, t, t, F->nr);
        fp_add(c, r.c[0], r.c[0], t);
        fp_mul(c, r.c[1], a->c[0], b->c[1]);
        fp_mul(c, t, a->c[1], b->c[0]);
        fp_add(c, r.c[1], r.c[1], t);
    } else {
        /* c0 = a0b0 + nr(a1b2 + a2b1) */
        fp_mul(c, t, a->c[1], b->c[2]);
        fp_mul(c, s, a->c[2], b->c[1]);
        fp_add(c, t, t, s);
        fp_mul(c, t, t, F->nr);
        fp_mul(c, s, a->c[0], b->c[0]);
        fp_add(c, r.c[0], t, s);
        /* c1 = a0b1 + a1b0 + nr a2b2 */
        fp_mul(c, t, a->c[2], b->c[2]);
        fp_mul(c, t, t, F->nr);
        fp_mul(c, s, a->c[0], b->c[1]);
        fp_add(c, t, t, s);
        fp_mul(c, s, a->c[1], b->c[0]);
        fp_add(c, r.c[1], t, s);
        /* c2 = a0b2 + a1b1 + a2b0 */
        fp_mul(c, t, a->c[0], b->c[2]);
        fp_mul(c, s, a->c[1], b->c[1]);
        fp_add(c, t, t, s);
        fp_mul(c, s, a->c[2], b->c[0]);
        fp_add(c, r.c[2], t, s);
    }
    *z = r;
}
static void fe_sqr(const fectx *F, fe *z, const fe *a) { fe_mul(F, z, a, a); }
static void fe_inv(const fectx *F, fe *z, const fe *a) {
    const fpctx *c = F->fp;
    fe r;
    fe_zero(&r);
    u64 t[MAXL], s[MAXL], n[MAXL];
    if (F->deg == 1) {
        fp_inv(c, r.c[0], a->c[0]);
    } else if (F->deg == 2) {
        /* 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 - nr a1^2) */
        fp_mul(c, t, a->c[0], a->c[0]);
        fp_mul(c, s, a->c[1], a->c[1]);
        fp_mul(c, s, s, F->nr);
        fp_sub(c, t, t, s);
        fp_inv(c, n, t);
        fp_mul(c, r.c[0], a->c[0], n);
        fp_mul(c, t, a->c[1], n);
        fp_neg(c, r.c[1], t);
    } else {
        u64 t0[MAXL], t1[MAXL], t2[MAXL];
        const u64 *a0 = a->c[0], *a1 = a->c[1], *a2 = a->c[2];
        fp_mul(c, t0, a0, a0); fp_mul(c, s, a1, a2); fp_mul(c, s, s, F->nr); fp_sub(c, t0, t0, s);
        fp_mul(c, t1, a2, a2); fp_mul(c, t1, t1, F->nr); fp_mul(c, s, a0, a1); fp_sub(c, t1, t1, s);
        fp_mul(c, t2, a1, a1); fp_mul(c, s, a0, a2); fp_sub(c, t2, t2, s);
        fp_mul(c, t, a2, t1); fp_mul(c, s, a1, t2); fp_add(c, t, t, s); fp_mul(c, t, t, F->nr);
        fp_mul(c, s, a0, t0); fp_add(c, t, t, s);
        fp_inv(c, n, t);
        fp_mul(c, r.c[0], t0, n); fp_mul(c, r.c[1], t1, n); fp_mul(c, r.c[2], t2, n);
    }
    *z = r;
}
static void fe_pow(const fectx *F, fe *z, const fe *a, const u64 *e, int elimbs) {
    fe res, base = *a;
    fe_one(F, &res);
    int bl = bn_bitlen(e, elimbs);
    for (int i = bl - 1; i >= 0; i--) {
        fe_sqr(F, &res, &res);
        if (bn_bit(e, i)) fe_mul(F, &res, &res, &base);
    }
    *z = res;
}

/* ------------------------------------------------------------------ curves */
#define MAXE (MAXL * MAXD) /* limbs of q^d */
typedef struct {
    fectx F;
    fe a, b;
    int a_is_zero, comp, uncomp;
    /* Tonelli-Shanks data over the full field order */
    int ts_s;
    u64 ts_t[MAXE], ts_t1h[MAXE], ts_half[MAXE]; /* t, (t+1)/2, (order-1)/2 */
    int elimbs;
    fe ts_c; /* nonsquare^t */
} group;
typedef struct {
    fpctx fq, fr;
    group g[2];
    int fq_bytes, fr_bytes;
} curve;

static const char *Q_HEX[4] = {
    "1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001",
    "122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b",
    "1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db117e776f218059db80f0da5cb537e38685acce9767254a4638810719ac425f0e39d54522cdd119f5e9063de245e8001",
    "1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db26c5c28c859a99b3eebca9429212636b9dff97634993aa4d6c381bc3f0057974ea099170fa13a4fd90776e240000001"};
static const char *R_HEX[4] = {
    "12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001",
    "1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001",
    "1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db26c5c28c859a99b3eebca9429212636b9dff97634993aa4d6c381bc3f0057974ea099170fa13a4fd90776e240000001",
    "1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db117e776f218059db80f0da5cb537e38685acce9767254a4638810719ac425f0e39d54522cdd119f5e9063de245e8001"};
/* G1 b coefficient (hex, canonical); a given as small int */
static const char *MNT4_B_HEX = "1373684a8c9dcae7a016ac5d7748d3313cd8e39051c596560835df0c9e50a5b59b882a92c78dc537e51a16703ec9855c77fc3d8bb21c8d68bb8cfb9db4b8c8fba773111c36c8b1b4e8f1ece940ef9eaad265458e06372009c9a0491678ef4";
static const char *MNT6_B_HEX = "7da285e70863c79d56446237ce2e1468d14ae9bb64b2bb01b10e60a5d5dfe0a25714b7985993f62f03b22a9a3c737a1a1e0fcf2c43d7bf847957c34cca1e3585f9a80a95f401867c4e80f4747fde5aba7505ba6fcf2485540b13dfc8468a";

static curve CV[4];
static pthread_once_t cv_once = PTHREAD_ONCE_INIT;

static void fp_from_hex_mont(const fpctx *c, u64 *z, const char *hex) {
    u64 t[MAXL] = {0};
    size_t len = strlen(hex);
    for (size_t i = 0; i < len; i++) t[i / 16] |= (u64)hexval(hex[len - 1 - i]) << (4 * (i % 16));
    fp_to_mont(c, z, t);
}
static void fp_from_small_mont(const fpctx *c, u64 *z, long v) {
    u64 t[MAXL] = {0};
    t[0] = (u64)(v < 0 ? -v : v);
    fp_to_mont(c, z, t);
    if (v < 0) fp_neg(c, z, z);
}

/* small bignum helpers for q^d */
static void bn_mul_full(u64 *z, const u64 *a, int na, const u64 *b, int nb) {
    memset(z, 0, 8 * (na + nb));
    for (int i = 0; i < na; i++) {
        u64 carry = 0;
        for (int j = 0; j < nb; j++) {
            u128 v = (u128)a[i] * b[j] + z[i + j] + carry;
            z[i + j] = (u64)v;
            carry = (u64)(v >> 64);
        }
        z[i + nb] = carry;
    }
}
static void bn_shr(u64 *a, int n, int s) {
    int w = s / 64, b = s % 64;
    for (int i = 0; i < n; i++) {
        u64 lo = i + w < n ? a[i + w] : 0, hi = i + w + 1 < n ? a[i + w + 1] : 0;
        a[i] = b ? (lo >> b) | (hi << (64 - b)) : lo;
    }
}

static int fe_is_square(const group *G, const fe *a) {
    if (fe_is_zero(&G->F, a)) return 1;
    fe t, one;
    fe_pow(&G->F, &t, a, G->ts_half, G->elimbs);
    fe_one(&G->F, &one);
    return fe_eq(&G->F, &t, &one);
}

static void group_init_ts(group *G) {
    const fpctx *c = G->F.fp;
    int d = G->F.deg, n = c->n;
    u64 ord[MAXE + MAXL] = {0}, tmp[MAXE + MAXL];
    memcpy(ord, c->p, 8 * n);
    int len = n;
    for (int i = 1; i < d; i++) {
        bn_mul_full(tmp, ord, len, c->p, n);
        len += n;
        memcpy(ord, tmp, 8 * len);
    }
    G->elimbs = len;
    ord[0] -= 1; /* q^d is odd, no borrow */
    memcpy(G->ts_half, ord, 8 * len);
    bn_shr(G->ts_half, len, 1);
    int s = 0;
    while (!bn_bit(ord, s)) s++;
    G->ts_s = s;
    memcpy(G->ts_t, ord, 8 * len);
    bn_shr(G->ts_t, len, s);
    memcpy(G->ts_t1h, G->ts_t, 8 * len);
    G->ts_t1h[0] += 1; /* t odd -> t+1 may carry */
    if (G->ts_t1h[0] == 0)
        for (int i = 1; i < len && ++G->ts_t1h[i] == 0; i++) {}
    bn_shr(G->ts_t1h, len, 1);
    /* find a non-square: (k, 1, 0..) pattern, k = 1,2,... */
    for (long k = 1;; k++) {
        fe z;
        fe_zero(&z);
        fp_from_small_mont(c, z.c[0], k);
        if (d > 1) fp_from_small_mont(c, z.c[1], 1);
        if (!fe_is_square(G, &z)) {
            fe_pow(&G->F, &G->ts_c, &z, G->ts_t, len);
            break;
        }
    }
}

static void curves_init(void) {
    for (int id = 0; id < 4; id++) {
        curve *C = &CV[id];
        fp_init(&C->fq, Q_HEX[id]);
        fp_init(&C->fr, R_HEX[id]);
        C->fq_bytes = (C->fq.bits + 7) / 8;
        C->fr_bytes = (C->fr.bits + 7) / 8;
        for (int w = 0; w < 2; w++) {
            group *G = &C->g[w];
            memset(G, 0, sizeof *G);
            G->F.fp = &C->fq;
            G->F.deg = 1;
        }
        group *G1 = &C->g[0], *G2 = &C->g[1];
        const fpctx *q = &C->fq;
        switch (id) {
        case ORACLE_BLS12_377: {
            fp_from_small_mont(q, G1->b.c[0], 1);
            G2->F.deg = 2;
            fp_from_small_mont(q, G2->F.nr, -5);
            /* b' = (0, 1/u) with u^2 = -5  =>  1/u = u / (-5) = (0, -1/5) */
            u64 five[MAXL], inv5[MAXL];
            fp_from_small_mont(q, five, 5);
            fp_inv(q, inv5, five);
            fp_neg(q, G2->b.c[1], inv5);
            break;
        }
        case ORACLE_BW6_761:
            fp_from_small_mont(q, G1->b.c[0], -1);
            fp_from_small_mont(q, G2->b.c[0], 4);
            break;
        case ORACLE_MNT4_753: {
            fp_from_small_mont(q, G1->a.c[0], 2);
            fp_from_hex_mont(q, G1->b.c[0], MNT4_B_HEX);
            G2->F.deg = 2;
            fp_from_small_mont(q, G2->F.nr, 13);
            fp_from_small_mont(q, G2->a.c[0], 26);             /* a * 13 */
            fp_mul(q, G2->b.c[1], G1->b.c[0], G2->F.nr);       /* (0, b * 13) */
            break;
        }
        case ORACLE_MNT6_753: {
            fp_from_small_mont(q, G1->a.c[0], 11);
            fp_from_hex_mont(q, G1->b.c[0], MNT6_B_HEX);
            G2->F.deg = 3;
            fp_from_small_mont(q, G2->F.nr, 11);
            fp_from_small_mont(q, G2->a.c[2], 11);             /* (0, 0, a) */
            fp_mul(q, G2->b.c[0], G1->b.c[0], G2->F.nr);       /* (b * 11, 0, 0) */
            break;
        }
        }
        for (int w = 0; w < 2; w++) {
            group *G = &C->g[w];
            G->a_is_zero = fe_is_zero(&G->F, &G->a);
            G->comp = G->F.deg * C->fq_bytes;
            G->uncomp = 2 * G->comp;
            group_init_ts(G);
        }
    }
}
static const curve *get_curve(int id) {
    if (id < 0 || id > 3) return NULL;
    pthread_once(&cv_once, curves_init);
    return &CV[id];
}

/* ------------------------------------------------------------------ Jacobian arithmetic */
typedef struct { fe X, Y, Z; } jac;
typedef struct { fe x, y; int inf; } aff;

static void jac_set_inf(const group *G, jac *P) {
    fe_one(&G->F, &P->X);
    fe_one(&G->F, &P->Y);
    fe_zero(&P->Z);
}
static int jac_is_inf(const group *G, const jac *P) { return fe_is_zero(&G->F, &P->Z); }

static void jac_double(const group *G, jac *R, const jac *P) {
    const fectx *F = &G->F;
    if (jac_is_inf(G, P)) { *R = *P; return; }
    fe XX, YY, YYYY, S, M, T, t, Z3;
    fe_sqr(F, &XX, &P->X);
    fe_sqr(F, &YY, &P->Y);
    fe_sqr(F, &YYYY, &YY);
    fe_add(F, &t, &P->X, &YY);
    fe_sqr(F, &t, &t);
    fe_sub(F, &t, &t, &XX);
    fe_sub(F, &t, &t, &YYYY);
    fe_dbl(F, &S, &t);
    fe_dbl(F, &M, &XX);
    fe_add(F, &M, &M, &XX);
    if (!G->a_is_zero) {
        fe ZZ;
        fe_sqr(F, &ZZ, &P->Z);
        fe_sqr(F, &ZZ, &ZZ);
        fe_mul(F, &ZZ, &ZZ, &G->a);
        fe_add(F, &M, &M, &ZZ);
    }
    fe_mul(F, &Z3, &P->Y, &P->Z);
    fe_dbl(F, &Z3, &Z3);
    fe_sqr(F, &T, &M);
    fe_sub(F, &T, &T, &S);
    fe_sub(F, &T, &T, &S);
    fe_sub(F, &t, &S, &T);
    fe_mul(F, &t, &t, &M);
    fe_dbl(F, &YYYY, &YYYY);
    fe_dbl(F, &YYYY, &YYYY);
    fe_dbl(F, &YYYY, &YYYY);
    R->X = T;
    fe_sub(F, &R->Y, &t, &YYYY);
    R->Z = Z3;
}
/* R = P + Q, Q affine (madd-2007-bl) */
static void jac_add_mixed(const group *G, jac *R, const jac *P, const aff *Q) {
    const fectx *F = &G->F;
    if (Q->inf) { *R = *P; return; }
    if (jac_is_inf(G, P)) {
        R->X = Q->x;
        R->Y = Q->y;
        fe_one(F, &R->Z);
        return;
    }
    fe Z1Z1, U2, S2, H, HH, I, J, rr, V, t;
    fe_sqr(F, &Z1Z1, &P->Z);
    fe_mul(F, &U2, &Q->x, &Z1Z1);
    fe_mul(F, &S2, &Q->y, &P->Z);
    fe_mul(F, &S2, &S2, &Z1Z1);
    fe_sub(F, &H, &U2, &P->X);
    fe_sub(F, &rr, &S2, &P->Y);
    if (fe_is_zero(F, &H)) {
        if (fe_is_zero(F, &rr)) { jac_double(G, R, P); return; }
        jac_set_inf(G, R);
        return;
    }
    fe_dbl(F, &rr, &rr);
    fe_sqr(F, &HH, &H);
    fe_dbl(F, &I, &HH);
    fe_dbl(F, &I, &I);
    fe_mul(F, &J, &H, &I);
    fe_mul(F, &V, &P->X, &I);
    fe X3, Y3, Z3;
    fe_sqr(F, &X3, &rr);
    fe_sub(F, &X3, &X3, &J);
    fe_sub(F, &X3, &X3, &V);
    fe_sub(F, &X3, &X3, &V);
    fe_sub(F, &t, &V, &X3);
    fe_mul(F, &Y3, &rr, &t);
    fe_mul(F, &t, &P->Y, &J);
    fe_dbl(F, &t, &t);
    fe_sub(F, &Y3, &Y3, &t);
    fe_add(F, &Z3, &P->Z, &H);
    fe_sqr(F, &Z3, &Z3);
    fe_sub(F, &Z3, &Z3, &Z1Z1);
    fe_sub(F, &Z3, &Z3, &HH);
    R->X = X3; R->Y = Y3; R->Z = Z3;
}
/* general Jacobian add via normalising Q is avoided: MSM uses this full add (add-2007-bl) */
static void jac_add(const group *G, jac *R, const jac *P, const jac *Q) {
    const fectx *F = &G->F;
    if (jac_is_inf(G, P)) { *R = *Q; return; }
    if (jac_is_inf(G, Q)) { *R = *P; return; }
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t;
    fe_sqr(F, &Z1Z1, &P->Z);
    fe_sqr(F, &Z2Z2, &Q->Z);
    fe_mul(F, &U1, &P->X, &Z2Z2);
    fe_mul(F, &U2, &Q->X, &Z1Z1);
    fe_mul(F, &S1, &P->Y, &Q->Z); fe_mul(F, &S1, &S1, &Z2Z2);
    fe_mul(F, &S2, &Q->Y, &P->Z); fe_mul(F, &S2, &S2, &Z1Z1);
    fe_sub(F, &H, &U2, &U1);
    fe_sub(F, &rr, &S2, &S1);
    if (fe_is_zero(F, &H)) {
        if (fe_is_zero(F, &rr)) { jac_double(G, R, P); return; }
        jac_set_inf(G, R);
        return;
    }
    fe_dbl(F, &rr, &rr);
    fe_dbl(F, &I, &H); fe_sqr(F, &I, &I);
    fe_mul(F, &J, &H, &I);
    fe_mul(F, &V, &U1, &I);
    fe X3, Y3, Z3;
    fe_sqr(F, &X3, &rr);
    fe_sub(F, &X3, &X3, &J); fe_sub(F, &X3, &X3, &V); fe_sub(F, &X3, &X3, &V);
    fe_sub(F, &t, &V, &X3);
    fe_mul(F, &Y3, &rr, &t);
    fe_mul(F, &t, &S1, &J); fe_dbl(F, &t, &t);
    fe_sub(F, &Y3, &Y3, &t);
    fe_add(F, &Z3, &P->Z, &Q->Z); fe_sqr(F, &Z3, &Z3);
    fe_sub(F, &Z3, &Z3, &Z1Z1); fe_sub(F, &Z3, &Z3, &Z2Z2);
    fe_mul(F, &Z3, &Z3, &H);
    R->X = X3; R->Y = Y3; R->Z = Z3;
}
/* ark-ec sw_double_and_add_affine: MSB-first, skip leading zeros */
static void jac_mul_affine(const group *G, jac *R, const aff *P, const u64 *k, int klimbs) {
    jac acc;
    jac_set_inf(G, &acc);
    int bl = bn_bitlen(k, klimbs);
    for (int i = bl - 1; i >= 0; i--) {
        jac_double(G, &acc, &acc);
        if (bn_bit(k, i)) jac_add_mixed(G, &acc, &acc, P);
    }
    *R = acc;
}
/* setup-utils batch_exp's batch_normalization: Montgomery trick, infinities skipped */
static void batch_to_affine(const group *G, aff *out, const jac *in, u64 n) {
    const fectx *F = &G->F;
    fe *pre = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    fe acc;
    fe_one(F, &acc);
    for (u64 i = 0; i < n; i++) {
        pre[i] = acc;
        if (!jac_is_inf(G, &in[i])) fe_mul(F, &acc, &acc, &in[i].Z);
    }
    fe inv;
    fe_inv(F, &inv, &acc);
    for (u64 i = n; i-- > 0;) {
        if (jac_is_inf(G, &in[i])) {
            out[i].inf = 1;
            fe_zero(&out[i].x);
            fe_zero(&out[i].y);
            continue;
        }
        fe zi, zi2;
        fe_mul(F, &zi, &inv, &pre[i]);
        fe_mul(F, &inv, &inv, &in[i].Z);
        fe_sqr(F, &zi2, &zi);
        fe_mul(F, &out[i].x, &in[i].X, &zi2);
        fe_mul(F, &zi2, &zi2, &zi);
        fe_mul(F, &out[i].y, &in[i].Y, &zi2);
        out[i].inf = 0;
    }
    free(pre);
}

/* ------------------------------------------------------------------ serde */
static void fe_to_bytes(const curve *C, const fectx *F, const fe *a, uint8_t *out, uint8_t flags) {
    int nb = C->fq_bytes;
    for (int i = 0; i < F->deg; i++) {
        u64 t[MAXL + 1] = {0};
        fp_from_mont(F->fp, t, a->c[i]);
        memcpy(out + i * nb, t, nb); /* little-endian host */
    }
    out[F->deg * nb - 1] |= flags;
}
static int fe_from_bytes(const curve *C, const fectx *F, fe *a, const uint8_t *in, int with_flags, uint8_t *flags) {
    int nb = C->fq_bytes;
    fe_zero(a);
    for (int i = 0; i < F->deg; i++) {
        u64 t[MAXL + 1] = {0};
        memcpy(t, in + i * nb, nb);
        if (i == F->deg - 1 && with_flags) {
            uint8_t *tb = (uint8_t *)t;
            *flags = tb[nb - 1] & 0xC0;
            tb[nb - 1] &= 0x3F;
        }
        if (bn_geq(t, F->fp->p, F->fp->n)) return ORACLE_E_POINT;
        fp_to_mont(F->fp, a->c[i], t);
    }
    return 0;
}
/* y > -y in arkworks' Ord (highest coefficient first, canonical integers) */
static int y_is_negative(const fectx *F, const fe *y) {
    fe ny;
    fe_neg(F, &ny, y);
    for (int i = F->deg - 1; i >= 0; i--) {
        u64 a[MAXL], b[MAXL];
        fp_from_mont(F->fp, a, y->c[i]);
        fp_from_mont(F->fp, b, ny.c[i]);
        if (memcmp(a, b, 8 * F->fp->n) == 0) continue;
        return bn_geq(a, b, F->fp->n);
    }
    return 0;
}
static void write_point(const curve *C, const group *G, const aff *P, uint8_t *out, int compressed) {
    const fectx *F = &G->F;
    memset(out, 0, compressed ? G->comp : G->uncomp);
    if (P->inf) {
        out[(compressed ? G->comp : G->uncomp) - 1] |= 0x40;
        return;
    }
    uint8_t flags = y_is_negative(F, &P->y) ? 0x80 : 0;
    if (compressed) {
        fe_to_bytes(C, F, &P->x, out, flags);
    } else {
        fe_to_bytes(C, F, &P->x, out, 0);
        fe_to_bytes(C, F, &P->y, out + G->comp, flags);
    }
}
static int fe_sqrt(const group *G, fe *z, const fe *a) {
    const fectx *F = &G->F;
    if (fe_is_zero(F, a)) { fe_zero(z); return 0; }
    if (!fe_is_square(G, a)) return ORACLE_E_NOT_ON_CURVE;
    fe x, b, c = G->ts_c, one;
    fe_one(F, &one);
    fe_pow(F, &x, a, G->ts_t1h, G->elimbs);
    fe_pow(F, &b, a, G->ts_t, G->elimbs);
    int m = G->ts_s;
    while (!fe_eq(F, &b, &one)) {
        int i = 0;
        fe b2 = b;
        while (!fe_eq(F, &b2, &one)) { fe_sqr(F, &b2, &b2); i++; }
        fe e = c;
        for (int k = 0; k < m - i - 1; k++) fe_sqr(F, &e, &e);
        fe_mul(F, &x, &x, &e);
        fe_sqr(F, &c, &e);
        fe_mul(F, &b, &b, &c);
        m = i;
    }
    *z = x;
    return 0;
}
static int read_point(const curve *C, const group *G, aff *P, const uint8_t *in, int compressed) {
    const fectx *F = &G->F;
    uint8_t flags = 0;
    int rc;
    if (compressed) {
        if ((rc = fe_from_bytes(C, F, &P->x, in, 1, &flags))) return rc;
        if (flags & 0x40) { P->inf = 1; fe_zero(&P->x); fe_zero(&P->y); return 0; }
        fe rhs, t;
        fe_sqr(F, &rhs, &P->x);
        fe_mul(F, &rhs, &rhs, &P->x);
        fe_mul(F, &t, &G->a, &P->x);
        fe_add(F, &rhs, &rhs, &t);
        fe_add(F, &rhs, &rhs, &G->b);
        if ((rc = fe_sqrt(G, &P->y, &rhs))) return rc;
        int neg = y_is_negative(F, &P->y);
        if (neg != !!(flags & 0x80)) fe_neg(F, &P->y, &P->y);
        P->inf = 0;
        return 0;
    }
    if ((rc = fe_from_bytes(C, F, &P->x, in, 0, &flags))) return rc;
    if ((rc = fe_from_bytes(C, F, &P->y, in + G->comp, 1, &flags))) return rc;
    P->inf = !!(flags & 0x40);
    if (P->inf) { fe_zero(&P->x); fe_zero(&P->y); }
    return 0;
}

/* ------------------------------------------------------------------ public API */
int oracle_point_size(int curve_id, int grp, int compressed) {
    const curve *C = get_curve(curve_id);
    if (!C || grp < 1 || grp > 2) return ORACLE_E_ARG;
    return compressed ? C->g[grp - 1].comp : C->g[grp - 1].uncomp;
}
int oracle_fr_limbs(int curve_id) {
    const curve *C = get_curve(curve_id);
    return C ? C->fr.n : ORACLE_E_ARG;
}

int oracle_params(int curve_id, int mode, uint64_t chunk_index, uint64_t chunk_size, uint32_t power,
                  uint64_t out[6]) {
    const curve *C = get_curve(curve_id);
    if (!C || power > 40 || (mode == 0 && chunk_size == 0)) return ORACLE_E_ARG;
    u64 L = 1ull << power, G1L = 2 * L - 1, start = mode == 0 ? chunk_index * chunk_size : 0;
    u64 g1 = mode ? G1L : (start >= G1L ? 0 : (G1L - start < chunk_size ? G1L - start : chunk_size));
    u64 ot = mode ? L : (start >= L ? 0 : (L - start < chunk_size ? L - start : chunk_size));
    u64 g1u = C->g[0].uncomp, g2u = C->g[1].uncomp, g1c = C->g[0].comp, g2c = C->g[1].comp;
    out[0] = g1;
    out[1] = ot;
    out[4] = 6 * g1u + 3 * g2u;
    out[2] = 64 + g1 * g1u + ot * (g2u + 2 * g1u) + g2u;
    out[3] = 64 + g1 * g1c + ot * (g2c + 2 * g1c) + g2c + out[4];
    out[5] = start;
    return 0;
}

typedef struct {
    const curve *C;
    const group *G;
    const uint8_t *in;
    uint8_t *out;
    int cin, cout;
    u64 lo, hi, start;
    const u64 *tau, *coeff;
    int rc, joinable;
} ap_job;

static void *apply_powers_worker(void *arg) {
    ap_job *J = (ap_job *)arg;
    const curve *C = J->C;
    const group *G = J->G;
    const fpctx *fr = &C->fr;
    u64 n = J->hi - J->lo;
    if (!n) return NULL;
    int isz = J->cin ? G->comp : G->uncomp, osz = J->cout ? G->comp : G->uncomp;
    jac *res = (jac *)malloc(sizeof(jac) * n);
    aff *affs = (aff *)malloc(sizeof(aff) * n);
    /* generate_powers_of_tau: first element by pow, then running product */
    u64 taum[MAXL], e[MAXL], cm[MAXL], idx[MAXL] = {0};
    fp_to_mont(fr, taum, J->tau);
    idx[0] = J->start + J->lo;
    fp_pow(fr, e, taum, idx, 1);
    if (J->coeff) fp_to_mont(fr, cm, J->coeff);
    for (u64 i = 0; i < n; i++) {
        aff P;
        int rc = read_point(C, G, &P, J->in + (J->lo + i) * isz, J->cin);
        if (rc) { J->rc = rc; free(res); free(affs); return NULL; }
        u64 k[MAXL], kc[MAXL];
        memcpy(k, e, 8 * fr->n);
        if (J->coeff) fp_mul(fr, k, k, cm);
        fp_from_mont(fr, kc, k);
        jac_mul_affine(G, &res[i], &P, kc, fr->n);
        fp_mul(fr, e, e, taum);
    }
    batch_to_affine(G, affs, res, n);
    for (u64 i = 0; i < n; i++) write_point(C, G, &affs[i], J->out + (J->lo + i) * osz, J->cout);
    free(res);
    free(affs);
    return NULL;
}

int oracle_apply_powers(int curve_id, int grp, const uint8_t *in, int cin, uint8_t *out, int cout,
                        uint64_t n, uint64_t start, const uint64_t *tau, const uint64_t *coeff, int nthreads) {
    const curve *C = get_curve(curve_id);
    if (!C || grp < 1 || grp > 2 || !tau) return ORACLE_E_ARG;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if ((u64)nthreads > n) nthreads = n ? (int)n : 1;
    ap_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; t++) {
        ap_job *J = &jobs[t];
        J->C = C; J->G = &C->g[grp - 1]; J->in = in; J->out = out; J->cin = cin; J->cout = cout;
        J->lo = n * t / nthreads; J->hi = n * (t + 1) / nthreads; J->start = start;
        J->tau = tau; J->coeff = coeff; J->rc = 0;
        /* a thread that cannot be created (a box's thread cap) must not leave its slice unwritten: run it here */
        J->joinable = 0;
        if (nthreads == 1 || pthread_create(&th[t], NULL, apply_powers_worker, J) != 0) apply_powers_worker(J);
        else J->joinable = 1;
    }
    int rc = 0;
    for (int t = 0; t < nthreads; t++) {
        if (jobs[t].joinable) pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}

int oracle_computation(int curve_id, int mode, uint64_t chunk_index, uint64_t chunk_size, uint32_t power,
                       const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                       const uint64_t *tau, const uint64_t *alpha, const uint64_t *beta, int nthreads) {
    const curve *C = get_curve(curve_id);
    u64 pr[6];
    if (!C || oracle_params(curve_id, mode, chunk_index, chunk_size, power, pr)) return ORACLE_E_ARG;
    if (in_len != pr[2] || out_len != pr[3]) return ORACLE_E_LEN;
    u64 g1 = pr[0], ot = pr[1], start = pr[5];
    u64 g1u = C->g[0].uncomp, g2u = C->g[1].uncomp, g1c = C->g[0].comp, g2c = C->g[1].comp;
    const uint8_t *pi = in + 64;
    uint8_t *po = out + 64;
    int rc;
    if ((rc = oracle_apply_powers(curve_id, 1, pi, 0, po, 1, g1, start, tau, NULL, nthreads))) return rc;
    pi += g1 * g1u; po += g1 * g1c;
    if ((rc = oracle_apply_powers(curve_id, 2, pi, 0, po, 1, ot, start, tau, NULL, nthreads))) return rc;
    pi += ot * g2u; po += ot * g2c;
    if ((rc = oracle_apply_powers(curve_id, 1, pi, 0, po, 1, ot, start, tau, alpha, nthreads))) return rc;
    pi += ot * g1u; po += ot * g1c;
    if ((rc = oracle_apply_powers(curve_id, 1, pi, 0, po, 1, ot, start, tau, beta, nthreads))) return rc;
    pi += ot * g1u; po += ot * g1c;
    /* beta_g2 <- beta * beta_g2 : tau^0 * beta */
    u64 one[MAXL] = {1};
    return oracle_apply_powers(curve_id, 2, pi, 0, po, 1, 1, 0, one, beta, 1);
}

typedef struct {
    const curve *C;
    const group *G;
    const uint8_t *in;
    uint8_t *out;
    u64 lo, hi;
    int rc, joinable;
} dc_job;
static void *decompress_worker(void *arg) {
    dc_job *J = (dc_job *)arg;
    for (u64 i = J->lo; i < J->hi; i++) {
        aff P;
        int rc = read_point(J->C, J->G, &P, J->in + i * J->G->comp, 1);
        if (rc) { J->rc = rc; return NULL; }
        write_point(J->C, J->G, &P, J->out + i * J->G->uncomp, 0);
    }
    return NULL;
}
int oracle_decompress(int curve_id, int grp, const uint8_t *in, uint8_t *out, uint64_t n, int nthreads) {
    const curve *C = get_curve(curve_id);
    if (!C || grp < 1 || grp > 2) return ORACLE_E_ARG;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if ((u64)nthreads > n) nthreads = n ? (int)n : 1;
    dc_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; t++) {
        dc_job *J = &jobs[t];
        J->C = C; J->G = &C->g[grp - 1]; J->in = in; J->out = out;
        J->lo = n * t / nthreads; J->hi = n * (t + 1) / nthreads; J->rc = 0;
        J->joinable = 0;
        if (nthreads == 1 || pthread_create(&th[t], NULL, decompress_worker, J) != 0) decompress_worker(J);
        else J->joinable = 1;
    }
    int rc = 0;
    for (int t = 0; t < nthreads; t++) {
        if (jobs[t].joinable) pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}

int oracle_scalar_mul(int curve_id, int grp, const uint8_t *in, int cin, uint8_t *out, int cout, uint64_t n,
                      const uint64_t *scalars, int lps, int shared) {
    const curve *C = get_curve(curve_id);
    if (!C || grp < 1 || grp > 2 || lps < 1 || lps > MAXE) return ORACLE_E_ARG;
    const group *G = &C->g[grp - 1];
    int isz = cin ? G->comp : G->uncomp, osz = cout ? G->comp : G->uncomp;
    jac *res = (jac *)malloc(sizeof(jac) * (n ? n : 1));
    aff *affs = (aff *)malloc(sizeof(aff) * (n ? n : 1));
    int rc = 0;
    for (u64 i = 0; i < n && !rc; i++) {
        aff P;
        rc = read_point(C, G, &P, in + i * isz, cin);
        if (!rc) jac_mul_affine(G, &res[i], &P, scalars + (shared ? 0 : i * lps), lps);
    }
    if (!rc) {
        batch_to_affine(G, affs, res, n);
        for (u64 i = 0; i < n; i++) write_point(C, G, &affs[i], out + i * osz, cout);
    }
    free(res);
    free(affs);
    return rc;
}

int oracle_msm(int curve_id, int grp, const uint8_t *in, int cin, uint64_t n, const uint64_t *scalars, int lps,
               uint8_t *out, int cout) {
    const curve *C = get_curve(curve_id);
    if (!C || grp < 1 || grp > 2 || lps < 1 || lps > MAXE) return ORACLE_E_ARG;
    const group *G = &C->g[grp - 1];
    int isz = cin ? G->comp : G->uncomp;
    jac acc, t;
    jac_set_inf(G, &acc);
    for (u64 i = 0; i < n; i++) {
        aff P;
        int rc = read_point(C, G, &P, in + i * isz, cin);
        if (rc) return rc;
        jac_mul_affine(G, &t, &P, scalars + i * lps, lps);
        jac_add(G, &acc, &acc, &t);
    }
    aff A;
    batch_to_affine(G, &A, &acc, 1);
    write_point(C, G, &A, out, cout);
    return 0;
}

/* ------------------------------------------------------------------ BLAKE2b-512 (RFC 7693) */
static const u64 B2_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                             0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                             0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
#define ROTR64(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
#define B2G(a, b, c, d, x, y) \
    do { a = a + b + x; d = ROTR64(d ^ a, 32); c = c + d; b = ROTR64(b ^ c, 24); \
         a = a + b + y; d = ROTR64(d ^ a, 16); c = c + d; b = ROTR64(b ^ c, 63); } while (0)
static void b2_compress(u64 h[8], const uint8_t block[128], u128 t, int last) {
    u64 m[16], v[16];
    memcpy(m, block, 128);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= (u64)t;
    v[13] ^= (u64)(t >> 64);
    if (last) v[14] = ~v[14];
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = B2_SIGMA[r];
        B2G(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
        B2G(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
        B2G(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
        B2G(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
        B2G(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
        B2G(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
        B2G(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
        B2G(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
void oracle_blake2b512(const uint8_t *data, size_t len, uint8_t out[64]) {
    u64 h[8];
    memcpy(h, B2_IV, sizeof h);
    h[0] ^= 0x01010000ULL ^ 64;
    u128 t = 0;
    while (len > 128) {
        t += 128;
        b2_compress(h, data, t, 0);
        data += 128;
        len -= 128;
    }
    uint8_t last[128] = {0};
    memcpy(last, data, len);
    t += len;
    b2_compress(h, last, t, 1);
    memcpy(out, h, 64);
}
