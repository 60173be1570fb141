// Microbenchmark of the 12-limb Montgomery multiplier (BLS12-377 Fq) on gfx950:
//   call = Fp::mul (out-of-line FIPS, VGPR vector ABI)    inline = Fp::mul_inl (FIPS inlined)
// Prints Gmul/s and the equivalent v_mad_u64_u32 rate (288 MACs per product).
// Round-1 A/B (profiles/r01_ubench_mul.txt): the operand-scanning CIOS form this replaced ran at
// 38 Gmul/s, FIPS at 59-60 Gmul/s (= the v_mad_u64_u32 pipe peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp.hpp"
using namespace ptau;
using F = Fp<BlsFq>;
#define ITER 2048

template <int MODE>
__global__ void __launch_bounds__(256) k(F *io, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    F a = io[t & 255], b = io[(t + 1) & 255];
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { a = F::mul(a, b); b = F::mul(b, a); }
        if (MODE == 1) { a = F::mul_inl(a, b); b = F::mul_inl(b, a); }
        if (MODE == 2) { a = F::sqr(a); b = F::sqr(b); }
        if (MODE == 3) { a = F::sqr_inl(a); b = F::sqr_inl(b); }
        if (MODE == 4) { F t = F::mul_inl(a, b); F u = F::mul_inl(b, b); a = t; b = u; }   // two INDEPENDENT products per iteration (ILP 2)
    }
    io[256 + t] = F::add(a, b);
}

template <int MODE>
void run(const char *name, F *d, int blocks, F *host_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, ITER);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double muls = (double)blocks * 256 * ITER * 2;
    printf("%-14s blocks=%5d %8.3f ms  %7.2f Gmul/s  (%.2f T MAC/s at 288/mul)\n", name, blocks, ms, muls / ms / 1e6, muls * 288 / ms / 1e9);
    hipMemcpy(host_out, d + 256, sizeof(F) * 256, hipMemcpyDeviceToHost);
}

int main() {
    F h[256];
    for (int i = 0; i < 256; i++) { uint32_t c[12]; for (int j = 0; j < 12; j++) c[j] = (i * 2654435761u + j * 40503u) ^ (j << 20); c[11] &= 0x00ffffff; h[i] = F::from_canonical(c); }
    F *d; hipMalloc(&d, sizeof(F) * (256 + 4096 * 256));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    F o0[256], o1[256], o2[256], o3[256], o4[256];
    // 256 blocks = one wave per SIMD (no other wave hides the MAC dependency chain); 512 = two; >= 1024 = four and more
    for (int blocks : {256, 512, 1024, 2048, 3072}) {
        run<0>("FIPS call", d, blocks, o0);
        run<1>("FIPS inline", d, blocks, o1);
        run<2>("SQR call", d, blocks, o2);
        run<3>("SQR inline", d, blocks, o3);
        run<4>("FIPS inline x2 ILP", d, blocks, o4);
    }
    // host reference: the same templates compiled for the CPU
    int bad = 0;
    for (int i = 0; i < 256; i++) {
        F a = h[i & 255], b = h[(i + 1) & 255];
        for (int it = 0; it < ITER; it++) { a = F::mul_inl(a, b); b = F::mul_inl(b, a); }
        F e = F::add(a, b);
        if (!(e == o0[i]) || !(e == o1[i])) bad++;
        F a2 = h[i & 255], b2 = h[(i + 1) & 255];
        for (int it = 0; it < ITER; it++) { a2 = F::mul_inl(a2, a2); b2 = F::mul_inl(b2, b2); }
        F e2 = F::add(a2, b2);
        if (!(e2 == o2[i]) || !(e2 == o3[i])) bad++;
    }
    printf("mismatches vs host: %d\n", bad);
    return bad != 0;
}
