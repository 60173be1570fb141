// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two read patterns of this repo's kernels
// (MI355X_MICROARCH.md: FETCH_SIZE reports half of a 16-B-per-lane streaming read; "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern"):
//   stream16   : lane i reads the uint4 at index i, i + T, ...                 (coalesced 16 B per lane)
//   lane_chunks: lane i reads 42 consecutive dwords of ITS OWN 168-byte chunk  (K1's window-table entry of one lane)
//   lane_bytes : lane i reads 96 consecutive BYTES of its own chunk              (K1's read of an uncompressed point)
// Each kernel reads BYTES bytes exactly once (buffers far larger than the 256 MiB Infinity Cache are used so that nothing
// is served on-die).  Usage:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./ubench_fetch
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fetch.hip -o tools/ubench_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void stream16(const uint4 *__restrict__ in, uint32_t *__restrict__ out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, T = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += T) {
        const uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void lane_chunks(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t lanes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    const uint32_t *p = in + i * 42;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 42; j++) acc += p[j];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void lane_bytes(const uint8_t *__restrict__ in, uint32_t *__restrict__ out, size_t lanes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    const uint8_t *p = in + i * 96;
    uint32_t acc = 0;
    for (int j = 0; j < 96; j++) acc += p[j];
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t BYTES = (size_t)1 << 30;   // 1 GiB per kernel, 4x the Infinity Cache
    void *buf;
    uint32_t *out;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    const size_t n16 = BYTES / 16, lanes42 = BYTES / 168, lanes96 = BYTES / 96;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, out, n16);
        hipLaunchKernelGGL(lane_chunks, dim3((unsigned)((lanes42 + 255) / 256)), dim3(256), 0, 0, (const uint32_t *)buf, out, lanes42);
        hipLaunchKernelGGL(lane_bytes, dim3((unsigned)((lanes96 + 255) / 256)), dim3(256), 0, 0, (const uint8_t *)buf, out, lanes96);
    }
    hipDeviceSynchronize();
    printf("bytes read per launch: stream16 %zu  lane_chunks %zu  lane_bytes %zu\n", n16 * 16, lanes42 * 168, lanes96 * 96);
    return 0;
}
