// Micro-benchmark: LDS-DMA ingest rate per CU vs contiguous segment size.  A [P pixels][384 B] buffer (the dense
// block layout); each wave-instruction fetches 1 KiB made of SEG-byte contiguous pieces of consecutive pixels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int SEG, int DEPTH>
__global__ __launch_bounds__(512) void stream_k(const char* src, long npix, int stride, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPS = SEG / 16;            // lanes per segment
    constexpr int PPI = 64 / LPS;            // pixels per instruction
    const long per_wg = npix / gridDim.x;
    const long p0 = (long)blockIdx.x * per_wg;
    const int niter = (int)(per_wg / (PPI * 8 * DEPTH));
    float acc = 0.f;
    for (int it = 0; it < niter; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long px = p0 + ((long)(it * DEPTH + d) * 8 + wave) * PPI + lane / LPS;
            const char* g = src + px * stride + (lane % LPS) * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(smem + (d * 8 + wave) * 1024), 16, 0, 0);
        }
        __syncthreads();
        acc += *(float*)(smem + threadIdx.x * 4);
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <int SEG, int DEPTH>
void run(const char* buf, long npix, int stride, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    size_t smem = DEPTH * 8 * 1024;
    hipFuncSetAttribute((const void*)stream_k<SEG, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_k<SEG, DEPTH>), dim3(grid), dim3(512), smem, 0, buf, npix, stride, sink);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((stream_k<SEG, DEPTH>), dim3(grid), dim3(512), smem, 0, buf, npix, stride, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    constexpr int PPI = 64 / (SEG / 16);
    const long per_wg = npix / grid; const int niter = (int)(per_wg / (PPI * 8 * DEPTH));
    const double bytes = (double)grid * niter * DEPTH * 8 * 1024;
    printf("seg %3d B  depth %2d KiB-per-wave: %7.1f us  %6.2f TB/s  (%5.1f GB/s/CU)\n", SEG, DEPTH, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}
int main() {
    const long npix = 16L * 256 * 256; const int stride = 384;
    char* buf; float* sink;
    hipMalloc(&buf, npix * stride + 4096); hipMemset(buf, 1, npix * stride); hipMalloc(&sink, 64);
    run<64, 4>(buf, npix, stride, sink);  run<64, 8>(buf, npix, stride, sink);  run<64, 16>(buf, npix, stride, sink);
    run<128, 4>(buf, npix, stride, sink); run<128, 8>(buf, npix, stride, sink); run<128, 16>(buf, npix, stride, sink);
    run<256, 8>(buf, npix, stride, sink); run<256, 16>(buf, npix, stride, sink);
    return 0;
}
