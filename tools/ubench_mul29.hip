// Prototype of a reduced-radix (2^29) Montgomery multiplier for the 377-bit BLS12-377 base field on gfx950:
// 13 limbs of 29 bits, 64-bit column accumulators with NO carry instruction (26 products of < 2^58 fit 64 bits),
// against the 12 x 32-bit product-scanning form (v_mad_u64_u32 + v_addc_co_u32 per MAC) of fp.hpp.
// Checks itself against a host big-integer reference, prints G products/s.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int N = 13, LB = 29;
constexpr uint32_t MASK = (1u << LB) - 1;
struct P29 { uint32_t mod[N]; uint32_t inv; };

// plain C++: the compiler selects v_mad_u64_u32 and schedules the chain (one-instruction asm statements each drew a hazard s_nop)
__device__ __forceinline__ void mac(uint64_t &acc, uint32_t x, uint32_t y) { acc += (uint64_t)x * y; }
template <bool SQR>
__device__ __forceinline__ void mul29(const uint32_t (&a)[N], const uint32_t (&b)[N], const P29 &p, uint32_t (&out)[N]) {
    uint32_t m[N], r[N], d[N];
    if (SQR) {
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = a[i] << 1;
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (SQR) {
#pragma unroll
            for (int j = 0; 2 * j < i; j++) mac(acc, d[j], a[i - j]);
            if ((i & 1) == 0) mac(acc, a[i / 2], a[i / 2]);
        } else {
#pragma unroll
            for (int j = 0; j <= i; j++) mac(acc, a[j], b[i - j]);
        }
#pragma unroll
        for (int j = 0; j < i; j++) mac(acc, m[j], p.mod[i - j]);
        m[i] = ((uint32_t)acc * p.inv) & MASK;
        mac(acc, m[i], p.mod[0]);
        acc >>= LB;
    }
#pragma unroll
    for (int i = N; i < 2 * N; i++) {
        if (SQR) {
#pragma unroll
            for (int j = i - N + 1; 2 * j < i; j++) mac(acc, d[j], a[i - j]);
            if ((i & 1) == 0 && i / 2 < N) mac(acc, a[i / 2], a[i / 2]);
        } else {
#pragma unroll
            for (int j = i - N + 1; j < N; j++) mac(acc, a[j], b[i - j]);
        }
#pragma unroll
        for (int j = i - N + 1; j < N; j++) mac(acc, m[j], p.mod[i - j]);
        r[i - N] = i == 2 * N - 1 ? (uint32_t)acc : ((uint32_t)acc & MASK);
        acc >>= LB;
    }
    // result < 2p: one conditional subtraction (signed borrow chain over 29-bit limbs)
    uint32_t u[N];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        int32_t t = (int32_t)r[i] - (int32_t)p.mod[i] + c;
        c = t >> LB;
        u[i] = (uint32_t)t & MASK;
    }
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = c < 0 ? r[i] : u[i];
}

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *io, P29 p, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[N], b[N];
    for (int i = 0; i < N; i++) { a[i] = io[(t & 255) * N + i]; b[i] = io[((t + 1) & 255) * N + i]; }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { mul29<false>(a, b, p, a); mul29<false>(b, a, p, b); }
        if (MODE == 1) { mul29<true>(a, a, p, a); mul29<true>(b, b, p, b); }
    }
    for (int i = 0; i < N; i++) io[(256 + t) * N + i] = a[i] ^ b[i];
}

typedef unsigned __int128 u128;
static void host_mul(const uint32_t *a, const uint32_t *b, const P29 &p, uint32_t *out) {   // same algorithm, scalar
    uint32_t m[N], r[N];
    u128 acc = 0;
    for (int i = 0; i < 2 * N; i++) {
        for (int j = 0; j < N; j++) { int k2 = i - j; if (k2 < 0 || k2 >= N) continue; acc += (u128)a[j] * b[k2]; if (j < i && j < N && i - j < N && j < (i < N ? i : N)) {} }
        for (int j = 0; j < N && j < i; j++) { int k2 = i - j; if (k2 >= N) continue; acc += (u128)m[j] * p.mod[k2]; }
        if (i < N) { m[i] = ((uint32_t)acc * p.inv) & MASK; acc += (u128)m[i] * p.mod[0]; }
        else r[i - N] = i == 2 * N - 1 ? (uint32_t)acc : ((uint32_t)acc & MASK);
        acc >>= LB;
    }
    int32_t c = 0; uint32_t u[N];
    for (int i = 0; i < N; i++) { int32_t t = (int32_t)r[i] - (int32_t)p.mod[i] + c; c = t >> LB; u[i] = (uint32_t)t & MASK; }
    for (int i = 0; i < N; i++) out[i] = c < 0 ? r[i] : u[i];
}

int main() {
    // BLS12-377 q in 29-bit limbs
    const uint32_t q32[12] = {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu, 0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
    P29 p;
    for (int i = 0; i < N; i++) {
        int bit = i * LB; uint64_t v = 0;
        int li = bit >> 5, sh = bit & 31;
        v = q32[li] >> sh;
        if (li + 1 < 12) v |= (uint64_t)q32[li + 1] << (32 - sh);
        p.mod[i] = (uint32_t)v & MASK;
    }
    uint32_t inv = 1;   // -p^-1 mod 2^29 by Newton
    for (int i = 0; i < 6; i++) inv *= 2 - p.mod[0] * inv;
    p.inv = (0u - inv) & MASK;
    const int ITER = 2048;
    std::vector<uint32_t> h(256 * N);
    for (int i = 0; i < 256; i++) for (int j = 0; j < N; j++) h[i * N + j] = ((i * 2654435761u + j * 40503u) ^ (j << 20)) & (j == N - 1 ? 0x0fffffu : MASK);
    uint32_t *d;
    hipMalloc(&d, sizeof(uint32_t) * N * (256 + 4096 * 256));
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(256 * N);
    for (int blocks : {256, 512, 768, 1024, 2048}) {
        for (int mode = 0; mode < 2; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, p, ITER); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, p, ITER);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, p, ITER); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, p, ITER);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-10s blocks=%5d %8.3f ms  %7.2f Gmul/s\n", mode ? "SQR29" : "MUL29", blocks, ms, (double)blocks * 256 * ITER * 2 / ms / 1e6);
            if (blocks == 256) {
                hipMemcpy(o.data(), d + 256 * N, o.size() * 4, hipMemcpyDeviceToHost);
                int bad = 0;
                for (int i = 0; i < 256; i++) {
                    uint32_t a[N], b[N];
                    for (int j = 0; j < N; j++) { a[j] = h[(i & 255) * N + j]; b[j] = h[((i + 1) & 255) * N + j]; }
                    for (int it = 0; it < ITER; it++) {
                        if (mode == 0) { host_mul(a, b, p, a); host_mul(b, a, p, b); } else { host_mul(a, a, p, a); host_mul(b, b, p, b); }
                    }
                    for (int j = 0; j < N; j++) if ((a[j] ^ b[j]) != o[i * N + j]) { bad++; break; }
                }
                printf("   mismatches vs host: %d\n", bad);
            }
        }
    }
    return 0;
}
