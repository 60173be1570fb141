// Microbenchmark of the integer VALU instructions the Montgomery multiplier is made of (gfx950).
// Prints wave-instruction issue cost (cycles per wave64 instruction per SIMD) and chip-wide rates.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITER 4096
#define UNROLL 16

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[UNROLL];
    uint32_t c32[UNROLL];
    for (int i = 0; i < UNROLL; i++) { acc[i] = i + a; c32[i] = i ^ b; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (MODE == 0) acc[i] = (uint64_t)a * (uint32_t)(acc[i] >> 7) + acc[i];     // v_mad_u64_u32 dependent per chain
            if (MODE == 1) c32[i] = c32[i] * a + b;                                      // v_mul_lo_u32 + add (or v_mad_u32..)
            if (MODE == 2) c32[i] = __umulhi(c32[i], a) + b;                             // v_mul_hi_u32
            if (MODE == 3) c32[i] = c32[i] + a;                                          // v_add_u32
            if (MODE == 4) { uint64_t t = acc[i] + (((uint64_t)b << 32) | a); acc[i] = t; }   // 64-bit add (add_co + addc)
            if (MODE == 5) c32[i] = __umul24(c32[i], a) + b;             // v_mad_u32_u24
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i] + c32[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <int MODE>
void run(const char *name, int blocks, uint32_t *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITER * UNROLL;
    printf("%-28s %8.3f ms  %8.2f Tops/s (lane-ops)  blocks=%d\n", name, ms, ops / ms / 1e9, blocks);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 4096 * 256 * 4 * 4);
    for (int blocks : {256 * 4, 256 * 8}) {
        run<0>("v_mad_u64_u32", blocks, d);
        run<1>("v_mul_lo_u32(+add)", blocks, d);
        run<2>("v_mul_hi_u32(+add)", blocks, d);
        run<3>("v_add_u32", blocks, d);
        run<4>("64-bit add", blocks, d);
        run<5>("v_mad_u32_u24", blocks, d);
    }
    return 0;
}
