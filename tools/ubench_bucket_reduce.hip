// A/B of the two ways to sum the points of a bucket (north_star: "wavefront ballot / shfl reductions for the buckets";
// DESIGN §4.2b argues the per-thread form, this measures both on the same data):
//   run_sums  : one THREAD adds a run of 64 consecutive points by mixed additions (what msm_accumulate does);
//   wave_sums : one WAVE takes 64 consecutive points, one per lane, and reduces them over the lanes — log2 64 = 6 steps, in each
//               of which every lane fetches a partner's Jacobian point (39 dwords by __shfl_down) and executes a FULL addition.
//               (A segmented scan keyed by bucket — __ballot of the head flags — does the same work plus predicates; the plain
//               reduction is its best case.)
// Both sum the same 2^23 BLS12-377 G1 points ((i + 1) G, affine, Montgomery form) into 131072 sums of 64, which are compared.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I snark-setup-operator_amd/csrc tools/ubench_bucket_reduce.hip -o tools/ubench_bucket_reduce
//   rocprofv3 --kernel-trace --stats -- tools/ubench_bucket_reduce      (or plain: it prints HIP-event times)
#include <cstdio>
#include <vector>

#include "kernels_impl.hpp"

using namespace ptau;
using G = GroupTypes<BlsFqR29, BlsFr, BlsG1Cfg>;
using J = G::J;
using A = G::A;
using F = G::F;

__global__ void __launch_bounds__(256) make_points(const uint8_t *gen_bytes, A *pts, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    A g;
    read_uncompressed<F>(gen_bytes, g);
    g.inf = false;
    const uint32_t k = (uint32_t)(i + 1);
    J acc = J::infinity();
    for (int b = 31 - __builtin_clz(k); b >= 0; b--) {
        acc = J::dbl(acc);
        if ((k >> b) & 1u) acc = J::add_mixed(acc, g);
    }
    const F zi = F::inverse(acc.Z), zi2 = F::sqr(zi);
    A p;
    p.x = F::mul(acc.X, zi2);
    p.y = F::mul(acc.Y, F::mul(zi2, zi));
    p.inf = false;
    pts[i] = p;
}

__global__ void __launch_bounds__(256, 2) run_sums(const A *__restrict__ pts, J *__restrict__ out, uint64_t runs) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= runs) return;
    J acc = J::infinity();
    for (int e = 0; e < 64; e++) acc = J::add_mixed(acc, pts[t * 64 + e]);
    out[t] = acc;
}

__global__ void __launch_bounds__(256, 2) wave_sums(const A *__restrict__ pts, J *__restrict__ out, uint64_t waves) {
    const uint64_t gt = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t w = gt >> 6;
    const int lane = (int)(gt & 63);
    if (w >= waves) return;
    J acc = J::from_affine(pts[w * 64 + lane]);
    constexpr int WORDS = sizeof(J) / 4;
    for (int d = 32; d >= 1; d >>= 1) {
        J o;
        uint32_t *ow = reinterpret_cast<uint32_t *>(&o);
        const uint32_t *aw = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
        for (int j = 0; j < WORDS; j++) ow[j] = __shfl_down(aw[j], d, 64);
        acc = J::add(acc, o);   // every lane: the upper ones compute sums nobody reads
    }
    if (lane == 0) out[w] = acc;
}

__global__ void compare(const J *a, const J *b, uint64_t n, int *bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // same projective point: X1 Z2^2 == X2 Z1^2 and Y1 Z2^3 == Y2 Z1^3
    const F z1 = a[i].Z, z2 = b[i].Z, z1s = F::sqr(z1), z2s = F::sqr(z2);
    const bool ok = F::mul(a[i].X, z2s) == F::mul(b[i].X, z1s) && F::mul(a[i].Y, F::mul(z2s, z2)) == F::mul(b[i].Y, F::mul(z1s, z1));
    if (!ok) atomicAdd(bad, 1);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const uint64_t N = 1ull << 23, RUNS = N / 64;
    A *pts;
    J *s_run, *s_wave;
    uint8_t *d_gen;
    int *d_bad;
    CK(hipMalloc(&pts, N * sizeof(A)));
    CK(hipMalloc(&s_run, RUNS * sizeof(J)));
    CK(hipMalloc(&s_wave, RUNS * sizeof(J)));
    CK(hipMalloc(&d_gen, 96));
    CK(hipMalloc(&d_bad, 4));
    CK(hipMemcpy(d_gen, Generators::BLS_G1, 96, hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(make_points, dim3((unsigned)(N / 256)), dim3(256), 0, 0, d_gen, pts, N);
    CK(hipDeviceSynchronize());
    hipEvent_t e[2];
    CK(hipEventCreate(&e[0]));
    CK(hipEventCreate(&e[1]));
    float t_run = 0, t_wave = 0;
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(run_sums, dim3((unsigned)((RUNS + 255) / 256)), dim3(256), 0, 0, pts, s_run, RUNS);
        CK(hipEventRecord(e[1]));
        CK(hipEventSynchronize(e[1]));
        CK(hipEventElapsedTime(&ms, e[0], e[1]));
        if (rep) t_run += ms / 2;
        CK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(wave_sums, dim3((unsigned)(N / 256)), dim3(256), 0, 0, pts, s_wave, RUNS);
        CK(hipEventRecord(e[1]));
        CK(hipEventSynchronize(e[1]));
        CK(hipEventElapsedTime(&ms, e[0], e[1]));
        if (rep) t_wave += ms / 2;
    }
    hipLaunchKernelGGL(compare, dim3((unsigned)((RUNS + 255) / 256)), dim3(256), 0, 0, s_run, s_wave, RUNS, d_bad);
    int bad = -1;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("2^23 BLS12-377 G1 points summed in runs of 64 (131072 sums), gfx950, HIP events, mean of 2 after a warm-up:\n");
    printf("  run_sums  (one thread per run, 64 mixed additions)                     %8.3f ms\n", t_run);
    printf("  wave_sums (one wave per run, 6 shuffle steps of full additions / lane) %8.3f ms   = %.2fx\n", t_wave, t_wave / t_run);
    printf("  sums that differ between the two: %d of %llu\n", bad, (unsigned long long)RUNS);
    return bad ? 2 : 0;
}
