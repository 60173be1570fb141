// How many VGPRs of arguments does the AMDGPU function ABI pass in registers?  A noinline device function with four 14-word vector
// arguments (56 words): hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S tools/abi_args_probe.hip -o - | grep -c scratch_load
// -> the callee starts by loading 24 of them from scratch (s32-relative): 32 VGPRs travel in registers, the rest through memory.
// This is why the fused tower products take TWO 14-limb operands and fetch partner coefficients inside.
#include <hip/hip_runtime.h>
typedef uint32_t V14 __attribute__((ext_vector_type(14)));
__device__ __noinline__ V14 f4(V14 a, V14 b, V14 c, V14 d) {
    V14 r;
    for (int i = 0; i < 14; i++) r[i] = a[i] * b[13 - i] + c[i] * d[13 - i];
    return r;
}
__global__ void k(uint32_t *p) {
    V14 a, b, c, d;
    for (int i = 0; i < 14; i++) { a[i] = p[i]; b[i] = p[14 + i]; c[i] = p[28 + i]; d[i] = p[42 + i]; }
    V14 r = f4(a, b, c, d);
    for (int j = 0; j < 3; j++) r = f4(r, b, c, r);
    for (int i = 0; i < 14; i++) p[i] = r[i];
}
