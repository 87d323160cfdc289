// fetch_calib.hip -- known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per access shape
// (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of a 16 B/lane coalesced stream; "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Shapes = the ones the library's kernels use:
//   read4 / read8 / read16   coalesced streams, 4 / 8 / 16 bytes per lane          (k_welch_carry loads 8 B per lane)
//   rows128                  8 B per lane, 16 lanes = one 128-byte row, every other 128-byte line (k_csdm_fused's tile rows)
//   write4 / write8 / write16
// Every kernel touches BYTES = 1 GiB of a 2 GiB buffer exactly once (far beyond the 256 MiB Infinity Cache).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/ubench/fetch_calib     (then WRITE_SIZE in its own pass)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define BYTES ((size_t)1 << 30)

template <typename T> __global__ void k_read(const T *__restrict__ p, size_t n, float *__restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        acc += reinterpret_cast<const float *>(&v)[0];
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <typename T> __global__ void k_write(T *__restrict__ p, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// rows of 128 bytes, `pitch` bytes apart: lane l of a wave reads 8 bytes at row (4 w + l / 16), byte 8 (l % 16)
__global__ void k_rows128(const char *__restrict__ p, size_t nrows, size_t pitch, float *__restrict__ sink) {
    float acc = 0.f;
    const size_t lane = threadIdx.x & 15, rsub = threadIdx.x >> 4;                 // 16 rows per 256-thread block step
    for (size_t r = (size_t)blockIdx.x * 16 + rsub; r < nrows; r += (size_t)gridDim.x * 16) {
        const float2 v = *reinterpret_cast<const float2 *>(p + r * pitch + lane * 8);
        acc += v.x;
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    char *buf;
    float *sink;
    if (hipMalloc(&buf, 2 * BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, 2 * BYTES);
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_read<float>), dim3(grid), dim3(256), 0, 0, (const float *)buf, BYTES / 4, sink);
        hipLaunchKernelGGL((k_read<float2>), dim3(grid), dim3(256), 0, 0, (const float2 *)buf, BYTES / 8, sink);
        hipLaunchKernelGGL((k_read<float4>), dim3(grid), dim3(256), 0, 0, (const float4 *)buf, BYTES / 16, sink);
        // 128-byte rows 256 bytes apart over the 2 GiB buffer: 8 Mi rows x 128 B = 1 GiB, every line touched once
        hipLaunchKernelGGL(k_rows128, dim3(grid), dim3(256), 0, 0, (const char *)buf, (size_t)(2 * BYTES / 256), (size_t)256, sink);
        hipLaunchKernelGGL((k_write<float>), dim3(grid), dim3(256), 0, 0, (float *)buf, BYTES / 4, 1.f);
        hipLaunchKernelGGL((k_write<float2>), dim3(grid), dim3(256), 0, 0, (float2 *)buf, BYTES / 8, make_float2(1.f, 2.f));
        hipLaunchKernelGGL((k_write<float4>), dim3(grid), dim3(256), 0, 0, (float4 *)buf, BYTES / 16, make_float4(1.f, 2.f, 3.f, 4.f));
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("fetch_calib: every k_read / k_write moves %zu bytes; k_rows128 reads %zu rows of 128 B = %zu bytes (rows 256 B apart: every other 128-byte line)\n",
           BYTES, (size_t)(2 * BYTES / 256), (size_t)(2 * BYTES / 256) * 128);
    return 0;
}
