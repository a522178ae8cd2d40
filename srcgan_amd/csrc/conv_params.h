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
    int buf16;    // vec16, and every epilogue tensor's channel plane is below 2 GiB: the row epilogue's 32-bit buffer offsets (conv_epilogue_lds_row)
    int rev;      // images are walked last to first
    long wpar;    // dgrad_s2k4 (conv_par4.hip) and the four-parity 1x1 form: bytes between the packs of consecutive output parities
    int npar;     // conv_igemm_k: 4 = the channel-tile index also selects an output parity (oa, ob) = (par >> 1, par & 1) and its weight pack;
                  // 2 = it selects the ROW parity oa only: a 128-row tile holds both column parities of a pixel pair (osx, wsplit)
    int osx;      // row epilogue: output pixel step along x (0 = os).  The pair form writes [.., 2H, W, 2 x C] with osx = 1, os = 2
    long wsplit;  // != 0: rows 64..127 of a 128-row weight tile come from a second 64-row pack wsplit bytes behind the first (1x1 kernels)
    unsigned char* sgn_out; const unsigned char* sgn_in;   // LeakyReLU sign masks, Cout / 8 bytes per output pixel: byte c / 8, bit c % 8 (loader-specialised
                                                           // 3x3 kernel: Cout == 32 either way, Cout == 64 read only; row epilogue of the generic kernel: written)
    int dbg;      // diagnostic builds only: 1 = skip MFMAs, 2 = skip operand DMA after the first chunk, 4 = skip epilogue
    unsigned long long* trace;   // diagnostic: per-barrier timestamps of workgroup 0 (SRCGAN_TRACE=1), else null
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
// Every operand access is a BUFFER access: one descriptor per operand and row whose base (image, row, first pixel of the tile, plane
// of the tile's first channel) is wave-uniform, the pass as scalar offset, and one 32-bit per-lane offset (pixel within the pass,
// channel) that is swapped for an out-of-range one where the lane has no work -- the range check returns zeros / drops the store.
// No exec-masked blocks: the older form (global loads and stores behind `if (in range)`, a 64-bit multiply per access) made hipcc
// wait for loads one by one and put `s_waitcnt vmcnt(0)` -- which on gfx9 counts stores -- at the head of every pass (ISA of the
// 4x4 stride-2 layers), in kernels whose workgroups run 2-8 K chunks between prologue and epilogue.  All operands of the row are
// requested before the transposition.  Needs per-lane offsets below 2^32: channel planes below 2 GiB (checked at launch).
template <typename T, int MT, int PT, bool OPS = true>
__device__ __forceinline__ void conv_epilogue_lds_row_impl(const ConvP& p, const f32x16 (&acc)[MT][PT], int q, char* lds_wave, int b, int ct,
                                                           int oy, int ox0, int lane) {
    constexpr int COT = 32 * MT, EPP = DT<T>::EPP, LPP = COT / EPP, PPP = 64 / LPP, NP = 32 / PPP;
    constexpr int RS = COT * 4 + 16;
    constexpr unsigned OOB = 0xffffffffu, FLAGS = 0x00020000u;
    const int osx = p.osx ? p.osx : p.os;          // (OPS == false: an instance without residual / mask operands -- no loads, no registers for them)
    static_assert(EPP * sizeof(T) == 16, "16-byte operand accesses");
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int r = lane & 31, h = lane >> 5;
    const int lx = lane / LPP, c0 = (lane % LPP) * EPP, co0 = ct * COT + c0;
    const bool cok = co0 < p.Cout;
    float bias[EPP];
    {
        const __amdgpu_buffer_rsrc_t db = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.Cout * 4 : 0, FLAGS);
#pragma unroll
        for (int i = 0; i < EPP; i += 4) {
            const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(db, (co0 + i) * 4, 0, 0));
            bias[i] = bv[0]; bias[i + 1] = bv[1]; bias[i + 2] = bv[2]; bias[i + 3] = bv[3];
        }
    }
    const bool use_r1 = p.r1 && cok && co0 < p.r1cend, use_r2 = p.r2 && cok && co0 < p.r2cend, use_mz = p.mz && cok && co0 >= p.mzc0;
    const long rowpix = ((long)b * p.YH + (long)oy * p.os + p.oa) * p.YW + p.ob + (long)ox0 * osx;        // wave-uniform: first pixel of pass 0
    const unsigned nrec = oy < p.OH ? 0xfffffff0u : 0u;
    const int xrem = p.OW - ox0 - lx;                // pass k is in range for this lane iff k * PPP < xrem
    const long lstep = (long)lx * osx;
    const int ctc = ct * COT;
    const long uy = (long)chan_off<T>(p.ycoff + ctc, p.yplane), u1 = (long)chan_off<T>(p.r1coff + ctc, p.r1plane);
    const long u2 = (long)chan_off<T>(p.r2coff + ctc, p.r2plane), um = (long)chan_off<T>(p.mzcoff + ctc, p.mzplane);
    const unsigned ly = cok ? (unsigned)(lstep * p.ypix + (long)chan_off<T>(p.ycoff + co0, p.yplane) - uy) : OOB;
    const unsigned l1 = use_r1 ? (unsigned)(lstep * p.r1pix + (long)chan_off<T>(p.r1coff + co0, p.r1plane) - u1) : OOB;
    const unsigned l2 = use_r2 ? (unsigned)(lstep * p.r2pix + (long)chan_off<T>(p.r2coff + co0, p.r2plane) - u2) : OOB;
    const unsigned lm = use_mz ? (unsigned)(lstep * p.mzpix + (long)chan_off<T>(p.mzcoff + co0, p.mzplane) - um) : OOB;
    const __amdgpu_buffer_rsrc_t dy = __builtin_amdgcn_make_buffer_rsrc((char*)p.y + uy + rowpix * p.ypix, 0, nrec, FLAGS);
    const __amdgpu_buffer_rsrc_t d1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.r1 + u1 + rowpix * p.r1pix), 0, p.r1 ? nrec : 0u, FLAGS);
    const __amdgpu_buffer_rsrc_t d2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.r2 + u2 + rowpix * p.r2pix), 0, p.r2 ? nrec : 0u, FLAGS);
    const __amdgpu_buffer_rsrc_t dm = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.mz + um + rowpix * p.mzpix), 0, p.mz ? nrec : 0u, FLAGS);
    const int sy = PPP * osx * (int)p.ypix, s1 = PPP * osx * (int)p.r1pix, s2 = PPP * osx * (int)p.r2pix, sm = PPP * osx * (int)p.mzpix;
    u32x4 r1v[OPS ? NP : 1], r2v[OPS ? NP : 1], mzv[OPS ? NP : 1];
    if constexpr (OPS) {
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const bool in = pass * PPP < xrem;
            r1v[pass] = __builtin_amdgcn_raw_buffer_load_b128(d1, in ? l1 : OOB, pass * s1, 0);
            r2v[pass] = __builtin_amdgcn_raw_buffer_load_b128(d2, in ? l2 : OOB, pass * s2, 0);
            mzv[pass] = __builtin_amdgcn_raw_buffer_load_b128(dm, in ? lm : OOB, pass * sm, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
            *(f32x4*)(lds_wave + r * RS + (m * 32 + 8 * g + 4 * h) * 4) = v;
        }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        float v[EPP];
#pragma unroll
        for (int i = 0; i < EPP; i += 4) {
            const f32x4 t = *(const f32x4*)(lds_wave + (pass * PPP + lx) * RS + (c0 + i) * 4);
            v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
        const vecT a1 = __builtin_bit_cast(vecT, r1v[OPS ? pass : 0]), a2 = __builtin_bit_cast(vecT, r2v[OPS ? pass : 0]), am = __builtin_bit_cast(vecT, mzv[OPS ? pass : 0]);
#pragma unroll
        for (int i = 0; i < EPP; ++i) v[i] = (v[i] + bias[i]) * p.alpha;
        if (OPS && p.r1) {                     // wave-uniform conditions: scalar branches, not exec masks (lanes outside their operand hold zeros)
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta1 * to_f(a1[i]); }
        if (OPS && p.r2) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] += p.beta2 * to_f(a2[i]); }
        if (p.act) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope; }
        if (OPS && p.mz) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] *= (!use_mz || to_f(am[i]) > 0.f) ? 1.f : p.mslope; }
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(v[i]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dy, pass * PPP < xrem ? ly : OOB, pass * sy, 0);
        if constexpr (EPP == 8) {
            if (p.sgn_out) {        // LeakyReLU sign mask of the stored activation: Cout / 8 bytes per (epilogue) pixel, this lane's 8 channels = one byte
                unsigned m = 0;
#pragma unroll
                for (int i = 0; i < EPP; ++i) m |= (v[i] > 0.f ? 1u : 0u) << i;
                const int mb = p.Cout >> 3;
                const __amdgpu_buffer_rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(p.sgn_out + rowpix * mb, 0, nrec, FLAGS);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)m, ds, (cok && pass * PPP < xrem) ? (unsigned)(lstep * mb + (co0 >> 3)) : OOB, pass * PPP * osx * mb, 0);
            }
        }
    }
}
template <typename T, int MT, int PT, bool OPS = true>
__device__ __forceinline__ void conv_epilogue_lds_row(const ConvP& p, const f32x16 (&acc)[MT][PT], int q, char* lds_wave, int b, int ct,
                                                      int oy, int ox0, int lane) {
    conv_epilogue_lds_row_impl<T, MT, PT, OPS>(p, acc, q, lds_wave, b, ct, oy, ox0, lane);
}

// Epilogue of the loader-specialised 3x3 kernel: all PT rows of a wave.  Differences from the per-row form above, each
// from a timestamp trace of the kernel (the epilogue was 25-30 % of a unit's time, ~2000 instructions per wave):
//  * every address is split into a wave-uniform 64-bit part (image, row, tile column, pass: scalar ALU) and a per-lane
//    part (pixel within the pass, channel) that does not depend on the unit -- no per-pass 64-bit vector multiplies;
//  * the bias comes from a copy staged in LDS at kernel start (a global load per row exposed ~1 us of latency);
//  * the residual (r1) and activation-mask (mz) operands of ALL passes of a row are requested before the row's
//    accumulators go through the LDS transposition, so one memory latency is exposed per row, not one per pass.
// b, ct, oy0, ox0 must be wave-uniform.
// EM: which optional operands this instantiation can take (1 = r1, 2 = r2, 4 = mz, 8 = sign mask read, 16 = sign mask write;
// the masks need MT == 1, 8 channels per lane and os == 1: byte (c0 / 8) of the pixel's u32); the others compile away, with
// their addressing and the scalar registers it pins (the all-operand form spills ~100 SGPRs and runs ~1000 instructions).
// The residual operands of a 64-row tile (conv5 / block-input gradient), ALL rows and passes of the wave, as requested by
// conv_lds_rows_request: the 3x3 kernel calls it BEFORE the barrier in front of its epilogue, so the operands' memory latency runs
// under the barrier wait (traced ~1.4 us per unit) instead of after it.  Requested inside the epilogue, the first operand cost
// 10 us per 192 -> 64 launch (194.6 -> 204.3 us) and the second one, requested row by row for want of registers beside the loop's
// fragment rings -- which are dead at the barrier --, another 27 us (206.7 -> 233.5 us; scripts/ab_conv.py f192n / f192 / b192 / b192r).
template <typename T, int MT, int PT, int EM>
struct ConvRowsPre {
    static constexpr int EPP = DT<T>::EPP, NP = 32 / (64 / (32 * MT / EPP));
    static constexpr bool ON = MT == 2 && sizeof(T) == 2 && NP * 4 * 2 <= 32 && (EM & 3) != 0 && (EM & ~3) == 0;
    u32x4 r1[ON && (EM & 1) ? PT : 1][ON && (EM & 1) ? NP : 1];
    u32x4 r2[ON && (EM & 2) ? PT : 1][ON && (EM & 2) ? NP : 1];
};
// Buffer loads, not global loads behind `if (in range)`: exec-masked blocks made hipcc wait for every load on its own
// (s_waitcnt vmcnt(0) + a scratch spill after each of the 16).  One descriptor per row and operand -- base = the row segment's
// first pixel in the plane of the tile's first channel, everything wave-uniform --, the pass as scalar offset, one per-lane
// 32-bit offset (pixel within the pass, channel, plane) that is swapped for an out-of-range one where the lane has no work: the
// hardware range check returns zeros for it.  Needs lane offsets < 2^32: planes below 2 GiB (checked at launch).
template <typename T, int MT, int PT, int EM>
__device__ __forceinline__ void conv_lds_rows_request(const ConvP& p, ConvRowsPre<T, MT, PT, EM>& pre, int b, int ct, int oy0, int ox0, int lane) {
    if constexpr (ConvRowsPre<T, MT, PT, EM>::ON) {
        constexpr int COT = 32 * MT, EPP = DT<T>::EPP, LPP = COT / EPP, PPP = 64 / LPP, NP = 32 / PPP;
        constexpr unsigned OOB = 0xffffffffu;
        const int lx = lane / LPP, co0 = ct * COT + (lane % LPP) * EPP;
        const bool cok = co0 < p.Cout;
        const bool use_r1 = (EM & 1) && p.r1 && cok && co0 < p.r1cend, use_r2 = (EM & 2) && p.r2 && cok && co0 < p.r2cend;
        const long lstep = (long)lx * p.os;
        const long u1 = (EM & 1) ? (long)chan_off<T>(p.r1coff + ct * COT, p.r1plane) : 0, u2 = (EM & 2) ? (long)chan_off<T>(p.r2coff + ct * COT, p.r2plane) : 0;
        const unsigned l1 = use_r1 ? (unsigned)(lstep * p.r1pix + (long)chan_off<T>(p.r1coff + co0, p.r1plane) - u1) : OOB;
        const unsigned l2 = use_r2 ? (unsigned)(lstep * p.r2pix + (long)chan_off<T>(p.r2coff + co0, p.r2plane) - u2) : OOB;
        const int xrem = p.OW - ox0 - lx;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const int oy = oy0 + q;
            const long rowpix = ((long)b * p.YH + (long)oy * p.os + p.oa) * p.YW + (long)ox0 * p.os + p.ob;
            const unsigned nrec = oy < p.OH ? 0xfffffff0u : 0u;
            [[maybe_unused]] __amdgpu_buffer_rsrc_t d1, d2;
            if constexpr ((EM & 1) != 0) d1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.r1 + u1 + rowpix * p.r1pix), 0, (EM & 1) && p.r1 ? nrec : 0u, 0x00020000);
            if constexpr ((EM & 2) != 0) d2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.r2 + u2 + rowpix * p.r2pix), 0, (EM & 2) && p.r2 ? nrec : 0u, 0x00020000);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const bool ok = pass * PPP < xrem;
                if constexpr ((EM & 1) != 0) pre.r1[q][pass] = __builtin_amdgcn_raw_buffer_load_b128(d1, ok ? l1 : OOB, pass * PPP * p.os * (int)p.r1pix, 0);
                if constexpr ((EM & 2) != 0) pre.r2[q][pass] = __builtin_amdgcn_raw_buffer_load_b128(d2, ok ? l2 : OOB, pass * PPP * p.os * (int)p.r2pix, 0);
            }
        }
    }
}

template <typename T, int MT, int PT, int EM = 7>
__device__ __forceinline__ void conv_epilogue_lds_rows(const ConvP& p, const f32x16 (&acc)[MT][PT], char* lds_wave, const char* lds_bias,
                                                       int b, int ct, int oy0, int ox0, int lane, const ConvRowsPre<T, MT, PT, EM>& pre) {
    constexpr int COT = 32 * MT, EPP = DT<T>::EPP, LPP = COT / EPP, PPP = 64 / LPP, NP = 32 / PPP;
    constexpr int RS = COT * 4 + 16;
    constexpr bool PF = NP * 4 * 2 <= 32;            // prefetch the operands of a whole row when that costs <= 32 VGPRs
    constexpr int NPF = PF ? NP : 1;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int r = lane & 31, h = lane >> 5;
    const int lx = lane / LPP, c0 = (lane % LPP) * EPP, co0 = ct * COT + c0;
    const bool cok = co0 < p.Cout;
    float bias[EPP];
#pragma unroll
    for (int i = 0; i < EPP; i += 4) {
        const f32x4 bv = *(const f32x4*)(lds_bias + (co0 + i) * 4);
        bias[i] = bv[0]; bias[i + 1] = bv[1]; bias[i + 2] = bv[2]; bias[i + 3] = bv[3];
    }
    const bool use_r1 = (EM & 1) && p.r1 && co0 < p.r1cend, use_r2 = (EM & 2) && p.r2 && co0 < p.r2cend, use_mz = (EM & 4) && p.mz && co0 >= p.mzc0;
    // per-lane byte offsets (pixel lx of a pass, channel co0): the same for every unit, row and pass
    const long lstep = (long)lx * p.os;
    const long ly = lstep * p.ypix + chan_off<T>(p.ycoff + co0, p.yplane);
    const long l1 = use_r1 ? lstep * p.r1pix + chan_off<T>(p.r1coff + co0, p.r1plane) : 0;
    const long l2 = use_r2 ? lstep * p.r2pix + chan_off<T>(p.r2coff + co0, p.r2plane) : 0;
    const long lm = use_mz ? lstep * p.mzpix + chan_off<T>(p.mzcoff + co0, p.mzplane) : 0;
    constexpr bool SGI = (EM & 8) != 0, SGO = (EM & 16) != 0;
    static_assert(!(SGI || SGO) || EPP == 8, "sign masks: 8 channels per lane");
    constexpr int MB = COT / 8;                      // mask bytes per pixel
    const long ls = (long)lx * MB + (c0 >> 3);         // mask byte of this lane's 8 channels
    const int xrem = p.OW - ox0 - lx;                // pass k is in range iff k * PPP < xrem
    // 64-row tiles (conv5 / block-input gradient: residual operands, 168-VGPR budget): the residual operands of ALL rows
    // and passes arrive in `pre` (requested in front of the pre-epilogue barrier, conv_lds_rows_request).
    constexpr bool PFALL = ConvRowsPre<T, MT, PT, EM>::ON;
    static_assert(!PFALL || PF, "prefetch sizing");
    // One explicit wait for these loads.  Without it the compiler puts `s_waitcnt vmcnt(0)` at the head of EVERY pass (the
    // passes are exec-masked blocks and its pending-load state is merged conservatively at their joins), and on gfx9
    // vmcnt counts stores too: each pass then waited for the previous pass's store to be acknowledged by memory
    // (~0.45 us each, 3.6 us per unit: the ISA showed the waits, the trace the time).
    if constexpr (PFALL) __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int oy = oy0 + q;
        const bool rok = cok && oy < p.OH;
        // wave-uniform: first output pixel of the row segment, and the pixel step between passes
        const long rowpix = ((long)b * p.YH + (long)oy * p.os + p.oa) * p.YW + (long)ox0 * p.os + p.ob;
        const long pstep = (long)PPP * p.os;
        vecT r1v[NPF], mzv[NPF];
        unsigned sgv[NPF];
        if (PF && !PFALL) {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const bool ok = rok && pass * PPP < xrem;
                const long px = rowpix + pass * pstep;
                if (use_r1 && ok) r1v[pass % NPF] = *(const vecT*)((const char*)p.r1 + px * p.r1pix + l1);
                if (use_mz && ok) mzv[pass % NPF] = *(const vecT*)((const char*)p.mz + px * p.mzpix + lm);
                if (SGI && ok) sgv[pass % NPF] = p.sgn_in[px * MB + ls];
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[m][q][4 * g], acc[m][q][4 * g + 1], acc[m][q][4 * g + 2], acc[m][q][4 * g + 3]};
                *(f32x4*)(lds_wave + r * RS + (m * 32 + 8 * g + 4 * h) * 4) = v;
            }
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            float v[EPP];
#pragma unroll
            for (int i = 0; i < EPP; i += 4) {
                const f32x4 t = *(const f32x4*)(lds_wave + (pass * PPP + lx) * RS + (c0 + i) * 4);
                v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
            }
            if (!rok || pass * PPP >= xrem) continue;
            const long px = rowpix + pass * pstep;
            if (!PF) {
                if (use_r1) r1v[0] = *(const vecT*)((const char*)p.r1 + px * p.r1pix + l1);
                if (use_mz) mzv[0] = *(const vecT*)((const char*)p.mz + px * p.mzpix + lm);
                if (SGI) sgv[0] = p.sgn_in[px * MB + ls];
            }
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = (v[i] + bias[i]) * p.alpha;
            if (use_r1) {
                vecT t;
                if constexpr (PFALL && (EM & 1) != 0) t = __builtin_bit_cast(vecT, pre.r1[q][pass]);
                else t = r1v[pass % NPF];
#pragma unroll
                for (int i = 0; i < EPP; ++i) v[i] += p.beta1 * to_f(t[i]); }
            if (use_r2) {
                vecT t;
                if constexpr (PFALL && (EM & 2) != 0) t = __builtin_bit_cast(vecT, pre.r2[q][pass]);
                else t = *(const vecT*)((const char*)p.r2 + px * p.r2pix + l2);
#pragma unroll
                for (int i = 0; i < EPP; ++i) v[i] += p.beta2 * to_f(t[i]); }
            if (p.act) {
#pragma unroll
                for (int i = 0; i < EPP; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope; }
            if (use_mz) {
#pragma unroll
                for (int i = 0; i < EPP; ++i) v[i] *= (to_f(mzv[pass % NPF][i]) > 0.f ? 1.f : p.mslope); }
            if (SGI) {
#pragma unroll
                for (int i = 0; i < EPP; ++i) v[i] *= ((sgv[pass % NPF] >> i) & 1u) ? 1.f : p.mslope; }
            if (SGO) {
                unsigned m = 0;
#pragma unroll
                for (int i = 0; i < EPP; ++i) m |= (v[i] > 0.f ? 1u : 0u) << i;
                p.sgn_out[px * MB + ls] = (unsigned char)m;
            }
            vecT o;
#pragma unroll
            for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(v[i]);
            *(vecT*)((char*)p.y + px * p.ypix + ly) = o;
        }
    }
}

// Direct epilogue of the loader-specialised 3x3 kernel for the dense-block convolutions (bf16, 32 output channels, unscaled
// output, no residual operands: EM = 0, 8 (sign mask read) or 16 (sign mask write)).  No LDS transposition: a lane keeps the
// accumulator layout -- 4 consecutive channels of pixel r for 4 channel groups -- and stores four 8-byte pieces; the lanes
// r and r + 32 fill the two halves of each 16 bytes, the four pieces of a 64-byte pixel record leave back to back and merge
// in L2.  The transposed form cost 1.4 us of a 5.6 us unit (Cin = 64; trace) plus a workgroup barrier in front of it (its
// scratch is a stage buffer); this one is ~150 VALU instructions per wave and needs no barrier, so a wave goes from its last
// MFMA straight to its stores.  Bias comes from the LDS copy; the sign mask word of a pixel is read by both of its lanes, and
// written by the h == 0 lane after a cross-half exchange.
// X16 (ycoff % 8 == 0): the two lanes of a pixel (r, r + 32) first exchange halves with v_permlane32_swap so that each holds 8
// CONSECUTIVE channels, and store 16 bytes: two instructions per row, each filling 32 contiguous bytes of every pixel record, instead
// of four 8-byte ones.  The texture addresser's time goes with the store instructions (scripts/hip/ingest_test.hip, the kernel's
// skeleton with all three streams running: 66.6 us with the 8-byte pattern, 62.8 with this one, 62.1 with fully transposed 1-KiB
// stores, 48.8 without stores), and it is shared with the loaders' DMA.
// sign words of the PT output rows of lane's pixel (EM & 8), all ones where the pixel is outside the image.  The 3x3 kernel requests
// them at the START of a unit (SG_SIGN_EARLY): requested in the epilogue, their memory latency (and the vmcnt(0) the compiler then
// placed at the head of the next chunk loop: on gfx9 vmcnt counts the epilogue's stores too) was exposed once per unit.
template <int PT>
__device__ __forceinline__ void conv_direct32_sign_request(const ConvP& p, unsigned (&sgr)[PT], int b, int oy0, int ox0, int lane) {
    const int ox = ox0 + (lane & 31);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        sgr[q] = 0xffffffffu;
        if (ox < p.OW && oy0 + q < p.OH) sgr[q] = ((const unsigned*)p.sgn_in)[((long)b * p.OH + oy0 + q) * p.OW + ox];
    }
}
// One output row of the direct epilogue.  BIAS / ACT / SCALE: which of  v = alpha * (acc + bias) ; LeakyReLU  this launch needs (wave-uniform,
// decided once per epilogue): the dense-block forward convolutions take (bias, act, alpha = 1), the gradient slices nothing but the
// sign mask.  The epilogue is VALU-bound -- round 3 measured it: with the epilogue compiled out a 128 -> 32 launch takes 66 us,
// with its arithmetic but no stores 82, complete 86 (scripts/ab_conv.py, SRCGAN_DBG 4 / 32), ~400 VALU instructions per row pair --
// so every form does only its own arithmetic:
//  * LeakyReLU as max(v, slope * v) (0 <= slope <= 1: checked by the launcher) -- 2 instructions instead of compare + multiply + select;
//  * sign bits by  v_cmp + v_addc (m = 2 m + carry)  in element order, nibble g shifted to bit 8 g on the way: 2 instructions per
//    element instead of compare + select + variable shift + or; the lane's half (4 h) is applied once per row;
//  * sign words read pre-shifted by 4 h, so an element's bit is an AND with a constant.
template <typename T, int EM, bool X16, bool BIAS, bool ACT, bool SCALE>
__device__ __forceinline__ void conv_direct32_row(const ConvP& p, const f32x16& a, const f32x4 (&bias)[4], unsigned sgs, int h4, bool xok, bool first_half,
                                                  char* yp, long lch0, unsigned* sgn_dst) {
    typedef __attribute__((ext_vector_type(4))) T vec4T;
    u32x2 pk[4];
    unsigned m = 0u;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        const int g = (EM & 16) ? 3 - gg : gg;              // sign accumulation runs from the highest element down: bit e = element e
        vec4T o;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = a[4 * g + i];
            if (BIAS) v[i] += bias[g][i];
            if (SCALE) v[i] *= p.alpha;
            if (ACT) v[i] = fmaxf(v[i], v[i] * p.slope);
            if (EM & 8) v[i] *= (sgs & (1u << (8 * g + i))) ? 1.f : p.mslope;
        }
        if (EM & 16) {
            if (gg > 0) m <<= 4;                             // leave the 4-bit gap of the other lane half between two groups
#pragma unroll
            for (int i = 3; i >= 0; --i)
                asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v[i]) : "vcc");
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = from_f<T>(v[i]);
        if constexpr (X16) pk[g] = __builtin_bit_cast(u32x2, o);
        else {
            if (xok && !SG_DBG(p, 32)) *(vec4T*)(yp + lch0 + 16 * g) = o;      // dbg 32: everything but the stores (timing experiment)
            if (SG_DBG(p, 32)) asm volatile("" :: "v"(o));
        }
    }
    if constexpr (X16) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // lanes h = 0 hold channels 16j + {0..3} (pk[2j]) and 16j + 8 + {0..3} (pk[2j+1]); lanes h = 1 the +4 ones.  After the
            // swap (upper half of x <-> lower half of y) a lane holds x', y' = channels 16j + 8h + {0..3}, + {4..7}.
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
            const u32x4 o16 = {s0[0], s1[0], s0[1], s1[1]};
// Nontemporal stores: in the skeleton (scripts/hip/ingest_test.hip) they take the stores' cost away (69 -> 62.5 us); in this kernel
// they are SLOWER (back to back 64->32 49 -> 52 us, 96->32 61 -> 66; training step 137.3 -> 140.1 ms, same-box A/B): the block's next
// convolution reads these 32 channels straight back, and the default policy keeps them in the Infinity Cache.  Off.
#ifndef SG_DIRECT_NT
#define SG_DIRECT_NT 0
#endif
            if (xok && !SG_DBG(p, 32)) {
                if (SG_DIRECT_NT) __builtin_nontemporal_store(o16, (u32x4*)(yp + lch0 + 32 * j));
                else *(u32x4*)(yp + lch0 + 32 * j) = o16;
            }
            if (SG_DBG(p, 32)) asm volatile("" :: "v"(o16));
        }
    }
    if (EM & 16) {
        // m: nibble g at bits 8 g .. 8 g + 3 = this lane's channels 8 g + 4 h + {0..3} -> shift by 4 h and merge the two halves of the
        // pixel (permlane32_swap of the word with itself: every lane ends up with lower | upper; no LDS round trip)
        const unsigned mine = m << h4;
        const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
        if (xok && first_half) *sgn_dst = sw[0] | sw[1];
    }
}

template <typename T, int PT, int EM, bool X16 = false, bool PRE = false>
__device__ __forceinline__ void conv_epilogue_direct32(const ConvP& p, const f32x16 (&acc)[1][PT], const char* lds_bias,
                                                       int b, int oy0, int ox0, int lane, const unsigned* sgpre = nullptr) {
    const int r = lane & 31, h = lane >> 5;
    const int ox = ox0 + r;
    const bool xok = ox < p.OW;
    // the 32 channels of the slice lie inside one 64-byte chunk (ycoff % 32 == 0 for blocked tensors; interleaved: linear)
    const long lch0 = (long)chan_off<T>(p.ycoff + (X16 ? 8 : 4) * h, p.yplane);
    // every row's sign word is requested before the first store: vmcnt counts stores too on gfx9, a load issued after a
    // row's stores could only be waited for together with them
    unsigned sgr[PT];
    if constexpr (PRE) {
        // ONE unconditional wait for the words requested at the unit's start (landed long ago), in front of every store of this unit:
        // left to the compiler, the wait for row 1's word came after row 0's stores (vmcnt retires in order -> it waited for their
        // acknowledgement), and another one guarded the registers at the next unit's request.
        __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
        for (int q = 0; q < PT; ++q) sgr[q] = sgpre[q];
    } else {
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            sgr[q] = 0xffffffffu;
            if ((EM & 8) && xok && oy0 + q < p.OH) sgr[q] = ((const unsigned*)p.sgn_in)[((long)b * p.OH + oy0 + q) * p.OW + ox];
        }
        if (EM & 8) __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    // bias of this lane's 16 channels (8 g + 4 h + i): ONE set of LDS reads per unit (it was read again for every row)
    const bool has_bias = p.bias != nullptr;
    f32x4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = has_bias ? *(const f32x4*)(lds_bias + (8 * g + 4 * h) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int h4 = 4 * h;
    const int form = (p.alpha != 1.f) ? 2 : (has_bias || p.act) ? 1 : 0;        // wave-uniform
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int oy = oy0 + q;
        if (oy >= p.OH) continue;                         // wave-uniform
        const long pix = ((long)b * p.OH + oy) * p.OW + ox;
        char* yp = (char*)p.y + pix * p.ypix;
        unsigned* sd = (EM & 16) ? (unsigned*)p.sgn_out + pix : nullptr;
        const unsigned sgs = sgr[q] >> h4;
        if (form == 0) conv_direct32_row<T, EM, X16, false, false, false>(p, acc[0][q], bias, sgs, h4, xok, h == 0, yp, lch0, sd);
        else if (form == 1) {
            if (p.act) conv_direct32_row<T, EM, X16, true, true, false>(p, acc[0][q], bias, sgs, h4, xok, h == 0, yp, lch0, sd);
            else conv_direct32_row<T, EM, X16, true, false, false>(p, acc[0][q], bias, sgs, h4, xok, h == 0, yp, lch0, sd);
        } else {
            if (p.act) conv_direct32_row<T, EM, X16, true, true, true>(p, acc[0][q], bias, sgs, h4, xok, h == 0, yp, lch0, sd);
            else conv_direct32_row<T, EM, X16, true, false, true>(p, acc[0][q], bias, sgs, h4, xok, h == 0, yp, lch0, sd);
        }
    }
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
