// Issue-rate microbenchmark (gfx950) of the instructions a big-integer multiplier can be built from,
// written with inline asm so the compiler cannot fold the loops.  Reports cycles per wave64
// instruction per SIMD (at the measured kernel time and 2.4 GHz) and chip-wide lane-ops/s.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITER 2048
#define CH 8   // independent dependency chains per thread

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CH];
    uint32_t hi[CH];
    double f[CH];
    const double fa = 1.0 + (double)a * 1e-9, fb = 1e-30 * b;
    for (int i = 0; i < CH; i++) { acc[i] = i + a; hi[i] = i ^ b; f[i] = (double)i; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(hi[i]) : "v"(a), "v"(b) : "vcc");
            if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fa), "v"(fb));
            if (MODE == 3) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(hi[i]) : "v"(a) : "vcc");
            if (MODE == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"((uint64_t)a));
            if (MODE == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(hi[i]) : "v"(a));
            if (MODE == 6) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(hi[i]) : "v"(a));
            if (MODE == 7) asm volatile("v_add_u32 %0, %1, %0" : "+v"(hi[i]) : "v"(a));
            if (MODE == 8) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi[i]) : "v"(a), "v"(b));
            if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(hi[i]) : "v"(a), "v"(b));
            if (MODE == 10) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(hi[i]) : "v"(a));
            if (MODE == 11) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(acc[i]), "+v"(hi[i]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; i++) s += acc[i] + hi[i] + (uint64_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <int MODE>
void run(const char *name, int blocks, uint32_t *d, int per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * ITER * CH * per_iter;       // wave-instructions
    double cyc = ms * 1e-3 * 2.4e9 * 1024;                            // SIMD-cycles available (256 CU x 4 SIMD)
    printf("%-34s %8.3f ms  %6.2f cyc/wave-inst/SIMD  %8.2f T lane-ops/s  blocks=%d\n", name, ms, cyc / winst,
           winst * 64 / ms / 1e9, blocks);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 8192 * 256 * 4);
    for (int blocks : {2048, 4096}) {
        run<0>("v_mad_u64_u32", blocks, d, 1);
        run<1>("v_mad_u64_u32 + v_addc_co", blocks, d, 2);
        run<11>("v_mad_u64_u32 + v_add_u32", blocks, d, 2);
        run<2>("v_fma_f64", blocks, d, 1);
        run<3>("v_add_co_u32", blocks, d, 1);
        run<4>("v_lshl_add_u64", blocks, d, 1);
        run<5>("v_mul_lo_u32", blocks, d, 1);
        run<6>("v_mul_hi_u32", blocks, d, 1);
        run<7>("v_add_u32", blocks, d, 1);
        run<8>("v_fma_f32", blocks, d, 1);
        run<9>("v_mad_u32_u24", blocks, d, 1);
        run<10>("v_alignbit_b32", blocks, d, 1);
    }
    return 0;
}
