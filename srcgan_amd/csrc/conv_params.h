// Shared by the convolution kernels: launch parameters and the fused epilogue.
#pragma once
#include "common.h"

// Tensor addressing: byte offset of channel c of pixel index q = q*pix + (c / KCE)*plane + (c % KCE)*sizeof(T), KCE = the
// 64-byte channel chunk.  Interleaved NHWC is pix = Cs*sizeof(T), plane = 64; the "blocked" layout used for the
// dense-block buffers is pix = 64, plane = npixels*64 (each 64-byte chunk of every pixel contiguous across pixels:
// operand fetches become >=128-byte contiguous; measured 3.4 TB/s -> 6.7 TB/s LDS-DMA ingest, scripts/hip/dma_stream_test.hip).
struct ConvP {
    const void* x; const void* wp; const float* bias; void* y;
    const void* r1; const void* r2; const void* mz;
    int B, H, W, Cin, xcoff;
    int OH, OW, Cout, YH, YW, ycoff;
    int pad_y, pad_x, os, oa, ob;
    int r1coff, r1cend, r2coff, r2cend, mzcoff, mzc0;
    long xpix, xplane, ypix, yplane, r1pix, r1plane, r2pix, r2plane, mzpix, mzplane;     // bytes
    float alpha, beta1, beta2, slope, mslope;
    int act, vec, nchunk, tiles_x, tiles_y, ctiles;
    int vec16;    // every epilogue tensor allows 16-byte accesses per lane (LDS-transposed epilogue)
    int dbg;      // diagnostic builds only: 1 = skip MFMAs, 2 = skip operand DMA after the first chunk, 4 = skip epilogue
};

template <typename T>
__device__ __forceinline__ size_t chan_off(int c, long plane) {        // byte offset of channel c inside a pixel record
    constexpr int KCE = DT<T>::KCE;
    return (size_t)(c / KCE) * plane + (size_t)(c % KCE) * sizeof(T);
}

// Fused epilogue for a 32x32 MFMA result tile set.  acc[m][q][4g+i] = D[cout = 32m + 8g + 4h + i][pixel = r]
// (M = Cout, N = 32 consecutive output pixels of row oyb+q): each lane owns 4 consecutive channels of one pixel.
//   v = alpha*(acc+bias) + beta1*r1 + beta2*r2 ; LeakyReLU ; * LeakyReLU'(mz) ; strided (pixel-shuffle) store
template <typename T, int MT, int PT>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, const f32x16 (&acc)[MT][PT], int b, int ct, int oyb, int ox0, int r, int h) {
    constexpr int COT = 32 * MT;
    const int wave = 0; (void)wave;
    // ---- epilogue.  acc[m][q][4g+i] = D[cout = 32m + 8g + 4h + i][pixel = r]
    const int ox = ox0 + r;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int oy = oyb + q;
        if (oy >= p.OH || ox >= p.OW) continue;
        const size_t opix = ((size_t)b * p.YH + (size_t)oy * p.os + p.oa) * p.YW + (size_t)ox * p.os + p.ob;
        char* yp = (char*)p.y + opix * p.ypix;
        const char* r1p = p.r1 ? (const char*)p.r1 + opix * p.r1pix : nullptr;
        const char* r2p = p.r2 ? (const char*)p.r2 + opix * p.r2pix : nullptr;
        const char* mzp = p.mz ? (const char*)p.mz + opix * p.mzpix : nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co0 = ct * COT + m * 32 + 8 * g + 4 * h;
                if (co0 >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[m][q][4 * g + i];
                if (p.vec) {
                    if (p.bias) { f32x4 bv = *(const f32x4*)(p.bias + co0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += bv[i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= p.alpha;
                    if (r1p && co0 < p.r1cend) { float rv[4]; load4<T>((const T*)(r1p + chan_off<T>(p.r1coff + co0, p.r1plane)), rv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += p.beta1 * rv[i]; }
                    if (r2p && co0 < p.r2cend) { float rv[4]; load4<T>((const T*)(r2p + chan_off<T>(p.r2coff + co0, p.r2plane)), rv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += p.beta2 * rv[i]; }
                    if (p.act) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope; }
                    if (mzp && co0 >= p.mzc0) { float zv[4]; load4<T>((const T*)(mzp + chan_off<T>(p.mzcoff + co0, p.mzplane)), zv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] *= (zv[i] > 0.f ? 1.f : p.mslope); }
                    store4<T>((T*)(yp + chan_off<T>(p.ycoff + co0, p.yplane)), v);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co0 + i;
                        if (co >= p.Cout) continue;
                        float u = v[i];
                        if (p.bias) u += p.bias[co];
                        u *= p.alpha;
                        if (r1p && co < p.r1cend) u += p.beta1 * to_f(*(const T*)(r1p + chan_off<T>(p.r1coff + co, p.r1plane)));
                        if (r2p && co < p.r2cend) u += p.beta2 * to_f(*(const T*)(r2p + chan_off<T>(p.r2coff + co, p.r2plane)));
                        if (p.act) u = u > 0.f ? u : u * p.slope;
                        if (mzp && co >= p.mzc0) u *= (to_f(*(const T*)(mzp + chan_off<T>(p.mzcoff + co, p.mzplane))) > 0.f ? 1.f : p.mslope);
                        *(T*)(yp + chan_off<T>(p.ycoff + co, p.yplane)) = from_f<T>(u);
                    }
                }
            }
        }
    }
}

// LDS-transposed epilogue.  The MFMA result has one pixel per lane and 4 channels per register group, so a direct
// store scatters 8-byte pieces over 32 cache lines per instruction (measured: ~1.5 TB/s, half of a conv1 launch).
// Here each wave parks its f32 tile in a private LDS region [pixel][COT] and reads it back with LPP = COT/EPP lanes
// per pixel, so every global access (store, residual loads, mask load) is 16 bytes per lane and the lanes of a pixel
// are contiguous: 64-128-byte segments like the operand loads.  Caller must have passed a workgroup barrier after
// the last read of the region being reused.  Requires p.vec16.
template <typename T, int MT, int PT>
__device__ __forceinline__ void conv_epilogue_lds(const ConvP& p, const f32x16 (&acc)[MT][PT], char* lds_wave, int b, int ct,
                                                  int oyb, int ox0, int lane) {
    constexpr int COT = 32 * MT, EPP = DT<T>::EPP, LPP = COT / EPP, PPP = 64 / LPP;   // pixels per pass
    constexpr int RS = COT * 4 + 16;                                                  // padded row stride (bytes)
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
                *(f32x4*)(lds_wave + (q * 32 + r) * RS + (m * 32 + 8 * g + 4 * h) * 4) = v;
            }
    const int cpart = lane % LPP, c0 = cpart * EPP, co0 = ct * COT + c0;
    if (co0 >= p.Cout) return;
    float bias[EPP];
#pragma unroll
    for (int i = 0; i < EPP; ++i) bias[i] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int i = 0; i < EPP; i += 4) { const f32x4 bv = *(const f32x4*)(p.bias + co0 + i); bias[i] = bv[0]; bias[i + 1] = bv[1]; bias[i + 2] = bv[2]; bias[i + 3] = bv[3]; }
    }
    const bool use_r1 = p.r1 && co0 < p.r1cend, use_r2 = p.r2 && co0 < p.r2cend, use_mz = p.mz && co0 >= p.mzc0;
#pragma unroll
    for (int pass = 0; pass < PT * 32 / PPP; ++pass) {
        const int pix = pass * PPP + lane / LPP;
        const int oy = oyb + pix / 32, ox = ox0 + (pix & 31);
        float v[EPP];
#pragma unroll
        for (int i = 0; i < EPP; i += 4) {
            const f32x4 t = *(const f32x4*)(lds_wave + pix * RS + (c0 + i) * 4);
            v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
        if (oy >= p.OH || ox >= p.OW) continue;
        const size_t opix = ((size_t)b * p.YH + (size_t)oy * p.os + p.oa) * p.YW + (size_t)ox * p.os + p.ob;
        typedef __attribute__((ext_vector_type(EPP))) T vecT;
#pragma unroll
        for (int i = 0; i < EPP; ++i) v[i] = (v[i] + bias[i]) * p.alpha;
        if (use_r1) { const vecT t = *(const vecT*)((const char*)p.r1 + opix * p.r1pix + chan_off<T>(p.r1coff + co0, p.r1plane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta1 * to_f(t[i]); }
        if (use_r2) { const vecT t = *(const vecT*)((const char*)p.r2 + opix * p.r2pix + chan_off<T>(p.r2coff + co0, p.r2plane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta2 * to_f(t[i]); }
        if (p.act) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope; }
        if (use_mz) { const vecT t = *(const vecT*)((const char*)p.mz + opix * p.mzpix + chan_off<T>(p.mzcoff + co0, p.mzplane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] *= (to_f(t[i]) > 0.f ? 1.f : p.mslope); }
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(v[i]);
        *(vecT*)((char*)p.y + opix * p.ypix + chan_off<T>(p.ycoff + co0, p.yplane)) = o;
    }
    // consume the bias registers on every path: a load left "pending" at the end of the epilogue makes hipcc drain
    // vmcnt(0) at the next write of those registers -- inside the main loop, once per stage (measured: it
    // serialised the LDS-DMA ring).
#pragma unroll
    for (int i = 0; i < EPP; ++i) asm volatile("" :: "v"(bias[i]));
}

// One output row (32 pixels) of the LDS-transposed epilogue: transpose space = 32 * (COT*4+16) bytes per wave.
template <typename T, int MT, int PT>
__device__ __forceinline__ void conv_epilogue_lds_row_impl(const ConvP& p, const f32x16 (&acc)[MT][PT], int q, char* lds_wave, int b, int ct,
                                                           int oy, int ox0, int lane) {
    constexpr int COT = 32 * MT, EPP = DT<T>::EPP, LPP = COT / EPP, PPP = 64 / LPP;
    constexpr int RS = COT * 4 + 16;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
            *(f32x4*)(lds_wave + r * RS + (m * 32 + 8 * g + 4 * h) * 4) = v;
        }
    const int cpart = lane % LPP, c0 = cpart * EPP, co0 = ct * COT + c0;
    const bool cok = co0 < p.Cout;
    float bias[EPP];
#pragma unroll
    for (int i = 0; i < EPP; ++i) bias[i] = 0.f;
    if (p.bias && cok) {
#pragma unroll
        for (int i = 0; i < EPP; i += 4) { const f32x4 bv = *(const f32x4*)(p.bias + co0 + i); bias[i] = bv[0]; bias[i + 1] = bv[1]; bias[i + 2] = bv[2]; bias[i + 3] = bv[3]; }
    }
    const bool use_r1 = p.r1 && co0 < p.r1cend, use_r2 = p.r2 && co0 < p.r2cend, use_mz = p.mz && co0 >= p.mzc0;
#pragma unroll
    for (int pass = 0; pass < 32 / PPP; ++pass) {
        const int pix = pass * PPP + lane / LPP;
        const int ox = ox0 + pix;
        float v[EPP];
#pragma unroll
        for (int i = 0; i < EPP; i += 4) {
            const f32x4 t = *(const f32x4*)(lds_wave + pix * RS + (c0 + i) * 4);
            v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
        if (!cok || oy >= p.OH || ox >= p.OW) continue;
        const size_t opix = ((size_t)b * p.YH + (size_t)oy * p.os + p.oa) * p.YW + (size_t)ox * p.os + p.ob;
        typedef __attribute__((ext_vector_type(EPP))) T vecT;
#pragma unroll
        for (int i = 0; i < EPP; ++i) v[i] = (v[i] + bias[i]) * p.alpha;
        if (use_r1) { const vecT t = *(const vecT*)((const char*)p.r1 + opix * p.r1pix + chan_off<T>(p.r1coff + co0, p.r1plane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta1 * to_f(t[i]); }
        if (use_r2) { const vecT t = *(const vecT*)((const char*)p.r2 + opix * p.r2pix + chan_off<T>(p.r2coff + co0, p.r2plane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta2 * to_f(t[i]); }
        if (p.act) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope; }
        if (use_mz) { const vecT t = *(const vecT*)((const char*)p.mz + opix * p.mzpix + chan_off<T>(p.mzcoff + co0, p.mzplane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] *= (to_f(t[i]) > 0.f ? 1.f : p.mslope); }
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(v[i]);
        *(vecT*)((char*)p.y + opix * p.ypix + chan_off<T>(p.ycoff + co0, p.yplane)) = o;
    }
    // consume the bias registers on every path: a load left "pending" at the end of the epilogue makes hipcc drain
    // vmcnt(0) at the next write of those registers -- inside the main loop, once per stage (measured: it
    // serialised the LDS-DMA ring).
#pragma unroll
    for (int i = 0; i < EPP; ++i) asm volatile("" :: "v"(bias[i]));
}
template <typename T, int MT, int PT>
__device__ __forceinline__ void conv_epilogue_lds_row(const ConvP& p, const f32x16 (&acc)[MT][PT], int q, char* lds_wave, int b, int ct,
                                                      int oy, int ox0, int lane) {
    conv_epilogue_lds_row_impl<T, MT, PT>(p, acc, q, lds_wave, b, ct, oy, ox0, lane);
}

// Half-row (16 pixels) variant for kernels whose free LDS slot is small: transpose space = 16 * (COT*4+16) bytes per wave.
template <typename T, int MT, int PT>
__device__ __forceinline__ void conv_epilogue_lds_half(const ConvP& p, const f32x16 (&acc)[MT][PT], int q, int half, char* lds_wave,
                                                       int b, int ct, int oy, int ox0, int lane) {
    constexpr int COT = 32 * MT, EPP = DT<T>::EPP, LPP = COT / EPP, PPP = 64 / LPP;
    constexpr int RS = COT * 4 + 16;
    const int r = lane & 31, h = lane >> 5;
    if ((r >> 4) == half) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
                *(f32x4*)(lds_wave + (r & 15) * RS + (m * 32 + 8 * g + 4 * h) * 4) = v;
            }
    }
    const int cpart = lane % LPP, c0 = cpart * EPP, co0 = ct * COT + c0;
    const bool cok = co0 < p.Cout;
    float bias[EPP];
#pragma unroll
    for (int i = 0; i < EPP; ++i) bias[i] = 0.f;
    if (p.bias && cok) {
#pragma unroll
        for (int i = 0; i < EPP; i += 4) { const f32x4 bv = *(const f32x4*)(p.bias + co0 + i); bias[i] = bv[0]; bias[i + 1] = bv[1]; bias[i + 2] = bv[2]; bias[i + 3] = bv[3]; }
    }
    const bool use_r1 = p.r1 && co0 < p.r1cend, use_r2 = p.r2 && co0 < p.r2cend, use_mz = p.mz && co0 >= p.mzc0;
#pragma unroll
    for (int pass = 0; pass < (16 + PPP - 1) / PPP; ++pass) {
        const int pix = pass * PPP + lane / LPP;
        if (PPP > 16 && pix >= 16) continue;
        const int ox = ox0 + half * 16 + pix;
        float v[EPP];
#pragma unroll
        for (int i = 0; i < EPP; i += 4) {
            const f32x4 t = *(const f32x4*)(lds_wave + pix * RS + (c0 + i) * 4);
            v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
        if (!cok || oy >= p.OH || ox >= p.OW) continue;
        const size_t opix = ((size_t)b * p.YH + (size_t)oy * p.os + p.oa) * p.YW + (size_t)ox * p.os + p.ob;
        typedef __attribute__((ext_vector_type(EPP))) T vecT;
#pragma unroll
        for (int i = 0; i < EPP; ++i) v[i] = (v[i] + bias[i]) * p.alpha;
        if (use_r1) { const vecT t = *(const vecT*)((const char*)p.r1 + opix * p.r1pix + chan_off<T>(p.r1coff + co0, p.r1plane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta1 * to_f(t[i]); }
        if (use_r2) { const vecT t = *(const vecT*)((const char*)p.r2 + opix * p.r2pix + chan_off<T>(p.r2coff + co0, p.r2plane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta2 * to_f(t[i]); }
        if (p.act) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope; }
        if (use_mz) { const vecT t = *(const vecT*)((const char*)p.mz + opix * p.mzpix + chan_off<T>(p.mzcoff + co0, p.mzplane));
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] *= (to_f(t[i]) > 0.f ? 1.f : p.mslope); }
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(v[i]);
        *(vecT*)((char*)p.y + opix * p.ypix + chan_off<T>(p.ycoff + co0, p.yplane)) = o;
    }
    // consume the bias registers on every path: a load left "pending" at the end of the epilogue makes hipcc drain
    // vmcnt(0) at the next write of those registers -- inside the main loop, once per stage (measured: it
    // serialised the LDS-DMA ring).
#pragma unroll
    for (int i = 0; i < EPP; ++i) asm volatile("" :: "v"(bias[i]));
}
