// Issue cost (gfx950) of the plain VALU instructions that surround the multiply-accumulates of a big-integer product:
// masks, shifts, three-operand adds, carries through VCC, selects, DPP moves.  Same harness as ubench_issue.hip: inline asm,
// 8 independent chains per thread, cycles per wave64 instruction per SIMD at 2.4 GHz.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITER 2048
#define CH 8

#define OPS(X)                                                                                                   \
    X(0, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %2, %3, %0", 1)                                                 \
    X(1, "v_add_u32", "v_add_u32 %1, %2, %1", 1)                                                                 \
    X(2, "v_sub_u32", "v_sub_u32 %1, %1, %2", 1)                                                                 \
    X(3, "v_and_b32", "v_and_b32 %1, %2, %1", 1)                                                                 \
    X(4, "v_and_b32 const", "v_and_b32 %1, 0x1fffffff, %1", 1)                                                   \
    X(5, "v_or_b32", "v_or_b32 %1, %2, %1", 1)                                                                   \
    X(6, "v_xor_b32", "v_xor_b32 %1, %2, %1", 1)                                                                 \
    X(7, "v_lshrrev_b32", "v_lshrrev_b32 %1, 3, %1", 1)                                                          \
    X(8, "v_lshlrev_b32", "v_lshlrev_b32 %1, 3, %1", 1)                                                          \
    X(9, "v_lshrrev_b64", "v_lshrrev_b64 %0, 29, %0", 1)                                                         \
    X(10, "v_lshlrev_b64", "v_lshlrev_b64 %0, 3, %0", 1)                                                         \
    X(11, "v_bfe_u32", "v_bfe_u32 %1, %1, 3, 29", 1)                                                             \
    X(12, "v_and_or_b32", "v_and_or_b32 %1, %1, %2, %3", 1)                                                      \
    X(13, "v_lshl_add_u32", "v_lshl_add_u32 %1, %1, 3, %2", 1)                                                   \
    X(14, "v_lshl_or_b32", "v_lshl_or_b32 %1, %1, 3, %2", 1)                                                     \
    X(15, "v_add3_u32", "v_add3_u32 %1, %1, %2, %3", 1)                                                          \
    X(16, "v_add_lshl_u32", "v_add_lshl_u32 %1, %1, %2, 3", 1)                                                   \
    X(17, "v_alignbit_b32", "v_alignbit_b32 %1, %1, %2, 29", 1)                                                  \
    X(18, "v_cndmask_b32 (sgpr mask)", "v_cndmask_b32_e64 %1, %1, %2, %5", 1)                                    \
    X(19, "v_mov_b32", "v_mov_b32 %1, %2", 1)                                                                    \
    X(20, "v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1)   \
    X(21, "v_add_co_u32", "v_add_co_u32 %1, vcc, %2, %1", 1)                                                     \
    X(22, "v_addc_co_u32", "v_addc_co_u32 %1, vcc, %2, %1, vcc", 1)                                              \
    X(23, "v_subb_co_u32", "v_subb_co_u32 %1, vcc, %1, %2, vcc", 1)                                              \
    X(24, "v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 0, %4", 1)                                                   \
    X(25, "v_cmp_lt_u32 (vcc)", "v_cmp_lt_u32 vcc, %1, %2", 1)                                                   \
    X(26, "v_min_u32", "v_min_u32 %1, %1, %2", 1)                                                                \
    X(27, "v_mul_u32_u24", "v_mul_u32_u24 %1, %1, %2", 1)                                                        \
    X(28, "v_mad_u32_u24", "v_mad_u32_u24 %1, %1, %2, %3", 1)                                                    \
    X(29, "v_mul_lo_u32", "v_mul_lo_u32 %1, %1, %2", 1)                                                          \
    X(30, "v_sub_u32 sdwa-free neg", "v_sub_u32 %1, 0, %1", 1)                                                   \
    X(31, "v_xad_u32", "v_xad_u32 %1, %1, %2, %3", 1)                                                            \
    X(32, "v_perm_b32", "v_perm_b32 %1, %1, %2, %3", 1)                                                          \
    X(33, "v_pk_add_u16", "v_pk_add_u16 %1, %1, %2", 1)                                                          \
    X(34, "v_add_u32 dpp", "v_add_u32_dpp %1, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1)         \
    X(35, "v_accvgpr_write+read", "v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %1, a0", 2)                  \
    X(36, "mad_u64 + and + lshrrev_b64", "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x1fffffff, %1\n\tv_lshrrev_b64 %0, 29, %0", 3) \
    X(37, "mad_u64 + 2x add_u32", "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1", 3)

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CH];
    uint32_t hi[CH];
    const uint64_t a64 = ((uint64_t)b << 32) | a;
    const uint64_t smask = __builtin_amdgcn_readfirstlane(seed) * 0x9e3779b97f4a7c15ull;   // a wave-uniform lane mask
    for (int i = 0; i < CH; i++) { acc[i] = i + a; hi[i] = i ^ b; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
#define X(ID, NAME, ASM, N) \
            if (MODE == ID) asm volatile(ASM : "+v"(acc[i]), "+v"(hi[i]) : "v"(a), "v"(b), "v"(a64), "s"(smask) : "vcc", "a0");
            OPS(X)
#undef X
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; i++) s += acc[i] + hi[i];
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
    double winst = (double)blocks * 4 * ITER * CH * per_iter;
    double cyc = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-34s %8.3f ms  %6.2f cyc/wave-inst/SIMD  blocks=%d\n", name, ms, cyc / winst, blocks);
    fflush(stdout);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 8192 * 256 * 4);
    const int blocks = 4096;
#define X(ID, NAME, ASM, N) run<ID>(NAME, blocks, d, N);
    OPS(X)
#undef X
    return 0;
}
