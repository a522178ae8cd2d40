// Micro-benchmark: do an MFMA stream and an LDS-DMA stream overlap on one CU, and what does the shader clock do?
// 8 MFMA waves (dependent-free 32x32x16 bf16 chains, no memory) + 4 loader waves (LDS-DMA streaming 1 KiB pieces
// from a large buffer, 16 pieces in flight per wave).  Modes: 1 = MFMA only, 2 = DMA only, 3 = both.
// Clock = delta(s_memtime) / delta(s_memrealtime) * 100 MHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void dma_buf16(const void* base, int nrec, int voff, int soff, lptr_t lds) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, voff, soff, 0, 0);
}
// LV: 0 = per-lane 64-bit addresses (VALU per DMA), 1 = buffer descriptor + scalar offsets (no VALU per DMA),
//     2 = as 1 with s_setprio 3 in the loader waves, 3 = as 0 with s_setprio 3
template <int LV>
__global__ __launch_bounds__(768) void k(const char* src, long bytes_per_wg, int mfma_iters, int mode, float* sink, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (wave >= 8) {
        if (mode & 2) {
            const int iw = wave - 8;
            const char* base = src + (long)blockIdx.x * bytes_per_wg;
            const int niter = (int)(bytes_per_wg / (4 * 16 * 1024));
            if (LV >= 2) __builtin_amdgcn_s_setprio(3);
            for (int it = 0; it < niter; ++it) {
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    if (LV == 0 || LV == 3) {
                        const char* g = base + ((long)(it * 16 + d) * 4 + iw) * 1024 + lane * 16;
                        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(smem + (d * 4 + iw) * 1024), 16, 0, 0);
                    } else {
                        dma_buf16(base, (int)bytes_per_wg, lane * 16, ((it * 16 + d) * 4 + iw) * 1024, (lptr_t)(smem + (d * 4 + iw) * 1024));
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
            }
        }
    } else if (mode & 1) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        bf16x8 x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(lane + i); y[i] = (__bf16)(float)(lane - i); }
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
        if (s == 123.456f) sink[0] = s;
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { clk[(blockIdx.x * 12 + wave) * 2] = t1 - t0; clk[(blockIdx.x * 12 + wave) * 2 + 1] = r1 - r0; }
}
int main() {
    const long total = 1L << 30;                 // 1 GiB stream
    char* buf; float* sink; unsigned long long* clk;
    hipMalloc(&buf, total + 4096); hipMemset(buf, 1, total); hipMalloc(&sink, 64);
    hipMalloc(&clk, 256 * 12 * 2 * 8);
    const size_t smem = 64 * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int mfma_iters = 4000;                 // 64k MFMAs per wave
    unsigned long long* h = (unsigned long long*)malloc(256 * 12 * 2 * 8);
    for (int cfg = 0; cfg < 10; ++cfg) {
        const int mode = cfg < 3 ? cfg + 1 : 3, lv = cfg == 9 ? 3 : cfg < 3 ? 0 : (cfg - 3) % 3;
        const int iters = cfg >= 6 ? mfma_iters / 8 : mfma_iters;
        if (cfg >= 3 && cfg < 6 && lv == 0) continue;
        auto kern = lv == 0 ? k<0> : lv == 1 ? k<1> : lv == 2 ? k<2> : k<3>;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(768), smem, 0, buf, total / 256, iters, mode, sink, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, 256 * 12 * 2 * 8, hipMemcpyDeviceToHost);
        double cm = 0, cl = 0; int nm = 0, nl = 0;
        double tm = 0, tl = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 12; ++w) {
            const double cyc = (double)h[(b * 12 + w) * 2], rt = (double)h[(b * 12 + w) * 2 + 1];
            if (rt < 10) continue;
            if (w < 8) { cm += cyc / rt * 100.0; tm += rt / 100.0; ++nm; } else { cl += cyc / rt * 100.0; tl += rt / 100.0; ++nl; }
        }
        const double flops = (mode & 1) ? 256.0 * 8 * iters * 16 * 32768.0 : 0, bytes = (mode & 2) ? (double)total : 0;
        printf("mode %d lv %d iters %d: %8.1f us | MFMA waves: avg %7.1f us, clock %6.0f MHz, %7.1f TFLOP/s | loader waves: avg %7.1f us, clock %6.0f MHz, %5.2f TB/s\n",
               mode, lv, iters, ms * 1e3, nm ? tm / nm : 0, nm ? cm / nm : 0, nm ? flops / (tm / nm) / 1e6 : 0, nl ? tl / nl : 0, nl ? cl / nl : 0, nl ? bytes / (tl / nl) / 1e6 : 0);
    }
    return 0;
}
