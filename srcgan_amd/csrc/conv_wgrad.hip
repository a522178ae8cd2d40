// Weight-gradient kernel for gfx950: dW[co][tap][ci] = sum_pixels dy[p][co] * x[p*s + tap - pad][ci].
// GEMM with K = pixels (millions), M = Cout tile, N = 32 input channels, per filter tap.
// Replaces the wgrad third of aten::convolution_backward behind reference
// src/model/rddb.py:52-58 (RDB convs), rddb.py:28-38 (deconv), model/model.py:612-634 (PatchGAN).
//
// Decomposition: grid = (Cin/32 tiles, Cout tiles, nsplit pixel ranges).  A workgroup walks its
// range of 8x32 (or 4x32) output-pixel tiles; per tile it stages the dy tile and the x halo tile
// (NHWC, 32-channel planes, *unpadded* 64 B / 128 B pixels) in LDS; its waves split the filter taps and
// keep every tap's 32x32 f32 accumulator in registers across the whole range (split-K in registers,
// no atomics).  Partial results go to an f32 slab; a second kernel reduces the nsplit slabs in a
// fixed order (deterministic) and scatters into the canonical torch gradient layout.
//
// Operand fetch: K (pixels) must run along the lane's fragment, but NHWC keeps channels contiguous,
// so bf16 fragments are read with ds_read_b64_tr_b16 (4 pixels x 16 channels transposed per 16 lanes;
// 4 consecutive 64-B pixels = one 256-B bank row -> conflict free).  f32 uses plain ds_read_b32
// (mfma_32x32x2_f32 takes one element per lane).
#include "common.h"
#include <type_traits>

struct WgradP {
    const void* dy; const void* x; float* slab;
    int B, H, W, Cin, xCs, xcoff;
    int OH, OW, Cout, dyCs, dycoff;
    int pad_y, pad_x;
    int nsplit, tiles_x, tiles_y, ntiles, citiles, ctiles, want_bias;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    // two transposed 4x16 reads -> 8 consecutive-k bf16 for this lane's row/column
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NCW compute waves split the taps; when BIASW an extra wave helps staging and (for the first Cin tile)
// accumulates the bias gradient as a pseudo-tap whose B operand is all ones: D[co][*] = sum_p dy[p][co].
template <typename T, int KH, int KW, int S, int MT, int TH, int NCW, bool BIASW>
__global__ __launch_bounds__((NCW + (BIASW ? 1 : 0)) * 64) void conv_wgrad_k(const WgradP p) {
    using D = DT<T>;
    constexpr int NW = NCW + (BIASW ? 1 : 0);
    constexpr int TW = 32, NTAP = KH * KW, COT = 32 * MT;
    constexpr int IHT = (TH - 1) * S + KH, IWT = (TW - 1) * S + KW;
    constexpr int PB = 32 * (int)sizeof(T);          // bytes per pixel per 32-channel plane
    constexpr int PPP = PB / 16;                     // 16-byte pieces per pixel
    constexpr int TPW = (NTAP + NCW - 1) / NCW;      // taps per compute wave
    constexpr int NT = NW * 64;
    constexpr int NPD = MT * TH * TW * PPP;          // pieces of the dy tile
    constexpr int NPX = IHT * IWT * PPP;             // pieces of the x halo tile
    constexpr int DIT = (NPD + NT - 1) / NT, XIT = (NPX + NT - 1) / NT;
    static_assert(NT % PPP == 0, "piece part must be thread-invariant");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_d = smem;                              // [MT][TH*TW][PB]
    char* lds_x = smem + MT * TH * TW * PB;          // [IHT*IWT][PB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    // Workgroup -> (input-channel tile, output-channel tile, pixel range).  The citiles x ctiles workgroups of one pixel range read
    // the same dy / x tiles (each its own 64-byte channel slice of every pixel record): they get consecutive slots of ONE XCD, so
    // the lines one of them pulls in serve the others from that L2.  (With grid = (citiles, ctiles, nsplit) neighbours in the
    // channel dimensions landed on different XCDs -- round-robin by linear id -- and every L2 fetched the lines again.)
#ifndef SG_WG_XCD
#define SG_WG_XCD 1
#endif
    int cit, ct, split;
    if (SG_WG_XCD) {
        const int L = blockIdx.x, pairs = p.citiles * p.ctiles;
        const int j = L >> 3, pr = j % pairs;
        split = (j / pairs) * 8 + (L & 7);
        if (split >= p.nsplit) return;
        cit = pr % p.citiles; ct = pr / p.citiles;
    } else { cit = blockIdx.x; ct = blockIdx.y; split = blockIdx.z; }
    const int part = tid % PPP;
    const bool bias_wave = BIASW && wave == NCW;
    const bool do_bias = bias_wave && cit == 0 && p.want_bias;

    f32x16 acc[TPW][MT];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][m][i] = 0.f;

    const int t_begin = (int)((long)p.ntiles * split / p.nsplit);
    const int t_end = (int)((long)p.ntiles * (split + 1) / p.nsplit);

    // branch-free staging through registers, one tile ahead of the MFMAs
    u32x4 dreg[DIT], xreg[XIT];
    unsigned dmask = 0, xmask = 0;          // validity bits of the pieces held in dreg / xreg
    auto issue = [&](int t) {
        int q = t;
        const int tx = q % p.tiles_x; q /= p.tiles_x;
        const int ty = q % p.tiles_y;
        const int b = q / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int gy0 = oy0 * S - p.pad_y, gx0 = ox0 * S - p.pad_x;
        const char* dyb = (const char*)p.dy + ((size_t)b * p.OH * p.OW * p.dyCs + p.dycoff) * sizeof(T);
        const char* xb = (const char*)p.x + ((size_t)b * p.H * p.W * p.xCs + p.xcoff) * sizeof(T);
        dmask = 0; xmask = 0;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * NT + tid;
            const int pix = (pc / PPP) % (TH * TW);
            const int m = pc / (PPP * TH * TW);
            const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
            const int ch = ct * COT + m * 32 + part * D::EPP;
            const bool ok = pc < NPD && oy < p.OH && ox < p.OW && ch < p.Cout;
            dmask |= (ok ? 1u : 0u) << it;
            dreg[it] = *(const u32x4*)(dyb + (ok ? ((size_t)(oy * p.OW + ox) * p.dyCs + ch) * sizeof(T) : 0));
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pc = it * NT + tid;
            const int pix = pc / PPP;
            const int iy = pix / IWT, ix = pix - iy * IWT;
            const int gy = gy0 + iy, gx = gx0 + ix;
            const int ch = cit * 32 + part * D::EPP;
            const bool ok = pc < NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && ch < p.Cin;
            xmask |= (ok ? 1u : 0u) << it;
            xreg[it] = *(const u32x4*)(xb + (ok ? ((size_t)(gy * p.W + gx) * p.xCs + ch) * sizeof(T) : 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * NT + tid;
            u32x4 v = dreg[it];
            if (!((dmask >> it) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            if (DIT * NT == NPD || pc < NPD) *(u32x4*)(lds_d + (pc / PPP) * PB + part * 16) = v;   // (m*TH*TW + pix) == pc / PPP
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pc = it * NT + tid;
            u32x4 v = xreg[it];
            if (!((xmask >> it) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            if (XIT * NT == NPX || pc < NPX) *(u32x4*)(lds_x + (pc / PPP) * PB + part * 16) = v;
        }
    };

    if (t_begin < t_end) issue(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        commit();
        __syncthreads();
        if (t + 1 < t_end) issue(t + 1);

        if constexpr (std::is_same<T, float>::value) {
            // mfma_32x32x2_f32: lane (r,h) supplies A[row r][k = h], B[k = h][col r]; one MFMA eats 2 pixels.
            if (!bias_wave) {
#pragma unroll 4
                for (int kk = 0; kk < TH * TW / 2; ++kk) {
                    const int pix = 2 * kk + h;
                    const int py = pix / TW, px = pix % TW;
                    float a[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) a[m] = *(const float*)(lds_d + (m * TH * TW + pix) * PB + r * 4);
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) {
                        const int tap = wave + tl * NCW;
                        if (tap < NTAP) {
                            const int ky = tap / KW, kx = tap % KW;
                            const float bv = *(const float*)(lds_x + ((py * S + ky) * IWT + px * S + kx) * PB + r * 4);
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                acc[tl][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv, acc[tl][m], 0, 0, 0);
                        }
                    }
                }
            } else if (do_bias) {
#pragma unroll 4
                for (int kk = 0; kk < TH * TW / 2; ++kk) {
                    const int pix = 2 * kk + h;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float a = *(const float*)(lds_d + (m * TH * TW + pix) * PB + r * 4);
                        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, acc[0][m], 0, 0, 0);
                    }
                }
            }
        } else {
            // mfma_32x32x16_bf16: lane (r,h) needs k = 8h..8h+7 (pixels) for its channel r.
            // transposed read geometry: 16-lane group gq: channels 16*(gq&1)+i, lane 4q+pp supplies
            // the address of pixel row q, channels 4pp..4pp+3 of the group's 16.
            const int gq = lane >> 4, idx = lane & 15, qq = idx >> 2, pp = idx & 3;
            const int choff = ((gq & 1) * 16 + 4 * pp) * 2;
            if (!bias_wave) {
#pragma unroll 2
                for (int kk = 0; kk < TH * 2; ++kk) {
                    const int py = kk >> 1, xh = (kk & 1) * 16;
                    const int px0 = xh + 8 * h + qq;           // first of this lane's two address pixels
                    bf16x8 a[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const char* base = lds_d + (m * TH * TW + py * TW + px0) * PB + choff;
                        a[m] = tr_frag(base, base + 4 * PB);
                    }
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) {
                        const int tap = wave + tl * NCW;
                        if (tap < NTAP) {
                            const int ky = tap / KW, kx = tap % KW;
                            const char* base = lds_x + ((py * S + ky) * IWT + px0 * S + kx) * PB + choff;
                            const bf16x8 bv = tr_frag(base, base + 4 * S * PB);
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                acc[tl][m] = sg_mfma16<T>(a[m], bv, acc[tl][m]);
                        }
                    }
                }
            } else if (do_bias) {
                const bf16x8 ones = sg_ones16<T>();
#pragma unroll 2
                for (int kk = 0; kk < TH * 2; ++kk) {
                    const int py = kk >> 1, xh = (kk & 1) * 16;
                    const int px0 = xh + 8 * h + qq;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const char* base = lds_d + (m * TH * TW + py * TW + px0) * PB + choff;
                        acc[0][m] = sg_mfma16<T>(tr_frag(base, base + 4 * PB), ones, acc[0][m]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- partial slab: [split][ct][cit][tap (NTAP+1)][co (COT)][ci (32)];  acc[.][m][4g+i] = D[co = 32m+8g+4h+i][ci = r]
    // (tap index NTAP = bias pseudo-tap, only written for cit == 0)
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        int tap = wave + tl * NCW;
        if (bias_wave) { if (tl > 0 || !do_bias) continue; tap = NTAP; }
        else if (tap >= NTAP) continue;
        float* sp = p.slab + ((((size_t)split * p.ctiles + ct) * p.citiles + cit) * (NTAP + 1) + tap) * (COT * 32);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = m * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                sp[co * 32 + r] = acc[tl][m][i];
            }
    }
}

// Weight gradient of a 3x3 s1 p1 convolution with at most THREE output channels (conv_last, rddb.py:98,113) in 16-bit modes.
// conv_wgrad_k pads the 3 output channels to a 32-row MFMA tile and runs one MFMA chain per tap: 144 MFMAs and 128 fragment reads
// per 8 x 32-pixel tile and 32 input channels, 10 x the useful work, 1.22 ms per step at the bench size for 2.4 GB of operands.
// Here the 27 (tap, output channel) pairs are the N dimension of ONE product per 16 pixels:
//   D[ci][n = 3 tap + co] = sum over x pixels (iy, ix) of  x[iy][ix][ci] * dy[iy - ky + 1][ix - kx + 1][co]
// A = the x tile (no halo; transposed fragment reads as in conv_wgrad_k), B = 8 consecutive pixels of a dy channel, read with one
// aligned ds_read_b128 from nine LDS planes (kx, co) that hold the dy halo tile pre-shifted by kx.  16 MFMAs per tile.
// The partial sums go to the same slab layout as conv_wgrad_k's ([split][cit][tap][co (32)][ci]), so wgrad_reduce_k serves both.
// KS x KS taps (3 or 4; stride 1, pad 1), NCO output channels: KS * KS * NCO <= 32 columns.  The 4 x 4 form with one channel is the
// PatchGAN's prediction layer (model/model.py:634).
template <typename T, int KS, int NCO>
__global__ __launch_bounds__(256) void wgrad_c3_k(const WgradP p) {
    using D = DT<T>;
    static_assert(sizeof(T) == 2, "16-bit modes");
    static_assert(KS * KS * NCO <= 32 && NCO <= 3, "(tap, channel) pairs are the 32 columns of one MFMA");
    constexpr int TH = 8, TW = 32, PB = 64, DH = TH + KS - 1, DW = TW + KS - 1, NN = KS * KS * NCO;
    constexpr int NPX = TH * TW * 4, NDP = DH * DW;
    constexpr int XIT = NPX / 256, DIT = (NDP + 255) / 256;
    __shared__ __attribute__((aligned(16))) char lds_x[TH * TW * PB];          // 16 KiB; reused for the cross-wave sum at the end
    __shared__ __attribute__((aligned(16))) T dyp[KS][NCO][DH][TW];            // [kx][co][halo row][column]: dy[.., column - kx + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int cit, split;
    {
        const int L = blockIdx.x, pairs = p.citiles;
        const int j = L >> 3, pr = j % pairs;
        split = (j / pairs) * 8 + (L & 7);
        if (split >= p.nsplit) return;
        cit = pr;
    }
    const int part = tid & 3;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int t_begin = (int)((long)p.ntiles * split / p.nsplit);
    const int t_end = (int)((long)p.ntiles * (split + 1) / p.nsplit);
    u32x4 xreg[XIT], dreg[DIT];
    unsigned xmask = 0, dmask = 0;
    auto issue = [&](int t) {
        int q = t;
        const int tx = q % p.tiles_x; q /= p.tiles_x;
        const int ty = q % p.tiles_y;
        const int b = q / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        const char* dyb = (const char*)p.dy + ((size_t)b * p.OH * p.OW * p.dyCs + p.dycoff) * sizeof(T);
        const char* xb = (const char*)p.x + ((size_t)b * p.H * p.W * p.xCs + p.xcoff) * sizeof(T);
        xmask = 0; dmask = 0;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pix = (it * 256 + tid) >> 2;
            const int gy = y0 + pix / TW, gx = x0 + pix % TW;
            const bool ok = gy < p.H && gx < p.W;
            xmask |= (ok ? 1u : 0u) << it;
            xreg[it] = *(const u32x4*)(xb + (ok ? ((size_t)(gy * p.W + gx) * p.xCs + cit * 32 + part * D::EPP) * sizeof(T) : 0));
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * 256 + tid;
            const int lr = pc / DW, lc = pc - lr * DW;
            const int gy = y0 - (KS - 2) + lr, gx = x0 - (KS - 2) + lc;
            const bool ok = pc < NDP && (unsigned)gy < (unsigned)p.OH && (unsigned)gx < (unsigned)p.OW;
            dmask |= (ok ? 1u : 0u) << it;
            dreg[it] = *(const u32x4*)(dyb + (ok ? (size_t)(gy * p.OW + gx) * p.dyCs * sizeof(T) : 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            u32x4 v = xreg[it];
            if (!((xmask >> it) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            *(u32x4*)(lds_x + ((it * 256 + tid) >> 2) * PB + part * 16) = v;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * 256 + tid;
            if (pc >= NDP) continue;
            u32x4 v = dreg[it];
            if (!((dmask >> it) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            const int lr = pc / DW, lc = pc - lr * DW;
            const unsigned short c0 = (unsigned short)(v[0] & 0xffffu), c1 = (unsigned short)(v[0] >> 16), c2 = (unsigned short)(v[1] & 0xffffu);
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int c = lc + kx - (KS - 1);
                if (c < 0 || c >= TW) continue;
                *(unsigned short*)&dyp[kx][0][lr][c] = c0;
                if (NCO > 1) *(unsigned short*)&dyp[kx][NCO > 1 ? 1 : 0][lr][c] = c1;
                if (NCO > 2) *(unsigned short*)&dyp[kx][NCO > 2 ? 2 : 0][lr][c] = c2;
            }
        }
    };
    const int gq = lane >> 4, idx = lane & 15, qq = idx >> 2, pp = idx & 3;
    const int choff = ((gq & 1) * 16 + 4 * pp) * 2;
    const int n = r < NN ? r : NN - 1, tap = n / NCO, co = n - NCO * tap, ky = tap / KS, kx = tap - KS * ky;
    if (t_begin < t_end) issue(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        commit();
        __syncthreads();
        if (t + 1 < t_end) issue(t + 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                       // this wave's rows 2 wave, 2 wave + 1, two 16-pixel blocks each
            const int py = 2 * wave + (k >> 1), xh = (k & 1) * 16;
            const char* base = lds_x + (py * TW + xh + 8 * h + qq) * PB + choff;
            const bf16x8 a = tr_frag(base, base + 4 * PB);
            const bf16x8 bv = *(const bf16x8*)&dyp[kx][co][py - ky + KS - 1][xh + 8 * h];
            acc = sg_mfma16<T>(a, bv, acc);
        }
        __syncthreads();
    }
    // ---- sum of the four waves' 32 x 32 results, then the 27 useful columns to the slab
    float* red = (float*)lds_x;                             // [wave][ci][n]
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 32 + 8 * (i >> 2) + 4 * h + (i & 3)) * 32 + r] = acc[i];
    __syncthreads();
    float* sp = p.slab + ((size_t)split * p.citiles + cit) * (KS * KS + 1) * 1024;
    for (int e = tid; e < NN * 32; e += 256) {
        const int nn = e >> 5, ci = e & 31;
        const float v = (red[(0 * 32 + ci) * 32 + nn] + red[(1 * 32 + ci) * 32 + nn]) + (red[(2 * 32 + ci) * 32 + nn] + red[(3 * 32 + ci) * 32 + nn]);
        sp[(nn / NCO) * 1024 + (nn % NCO) * 32 + ci] = v;
    }
}

struct WredP {
    const float* slab; float* grad; float* bias_grad;
    int nsplit, ctiles, citiles, ntap, kw, COT, Cout, Cin;
    long sr, sk, sty, stx, off;
    float alpha; int accumulate;
};

// 64 slab elements x 4 split lanes per block; fixed summation order -> deterministic
__global__ __launch_bounds__(256) void wgrad_reduce_k(const WredP p) {
    __shared__ float red[4][64];
    const long per_split = (long)p.ctiles * p.citiles * (p.ntap + 1) * p.COT * 32;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < per_split) {                      // four loads in flight per lane; the order stays fixed
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int sp = sl;
        for (; sp + 12 < p.nsplit; sp += 16) {
            a0 += p.slab[(size_t)sp * per_split + e];
            a1 += p.slab[(size_t)(sp + 4) * per_split + e];
            a2 += p.slab[(size_t)(sp + 8) * per_split + e];
            a3 += p.slab[(size_t)(sp + 12) * per_split + e];
        }
        for (; sp < p.nsplit; sp += 4) a0 += p.slab[(size_t)sp * per_split + e];
        s = (a0 + a1) + (a2 + a3);
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl != 0 || e >= per_split) return;
    s = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    long q = e;
    const int ci_l = (int)(q % 32); q /= 32;
    const int co_l = (int)(q % p.COT); q /= p.COT;
    const int tap = (int)(q % (p.ntap + 1)); q /= (p.ntap + 1);
    const int cit = (int)(q % p.citiles);
    const int ct = (int)(q / p.citiles);
    const int co = ct * p.COT + co_l, ci = cit * 32 + ci_l;
    if (co >= p.Cout) return;
    const float v = s * p.alpha;
    if (tap == p.ntap) {                      // bias pseudo-tap: every column holds the same sum
        if (p.bias_grad && cit == 0 && ci_l == 0) p.bias_grad[co] = p.accumulate ? p.bias_grad[co] + v : v;
        return;
    }
    if (ci >= p.Cin) return;
    const int ky = tap / p.kw, kx = tap % p.kw;
    float* g = p.grad + p.off + co * p.sr + ci * p.sk + ky * p.sty + kx * p.stx;
    *g = p.accumulate ? (*g + v) : v;
}

// ------------------------------------------------------------------ host side
static inline int wg_cot(int Cout) { return Cout <= 32 ? 32 : 64; }

extern "C" size_t srcgan_conv_wgrad_slab_bytes(int Cout, int Cin, int kh, int kw, int nsplit) {
    const int cot = wg_cot(Cout);
    return (size_t)nsplit * cdiv(Cout, cot) * cdiv(Cin, 32) * (kh * kw + 1) * cot * 32 * sizeof(float);
}

extern "C" int srcgan_conv_wgrad_nsplit(int B, int OH, int OW, int Cout, int Cin, int stride) {
    const int th = stride == 2 ? 4 : 8;
    const long ntiles = (long)B * cdiv(OH, th) * cdiv(OW, 32);
    const long pairs = (long)cdiv(Cout, wg_cot(Cout)) * cdiv(Cin, 32);
    long want = cdivl(1024, pairs);          // ~4 workgroups per CU over the chip (1280-4096 measured no better)
    if (want > ntiles) want = ntiles;
    if (want > 256) want = 256;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename T, int KH, int KW, int S, int MT, int TH, int NCW, bool BIASW>
static int launch_wgrad(WgradP p, hipStream_t st) {
    constexpr int NW = NCW + (BIASW ? 1 : 0);
    constexpr int TW = 32, PB = 32 * (int)sizeof(T);
    constexpr int IHT = (TH - 1) * S + KH, IWT = (TW - 1) * S + KW;
    constexpr size_t SMEM = (size_t)MT * TH * TW * PB + (size_t)IHT * IWT * PB;
    static_assert(SMEM <= 160 * 1024, "wgrad tile exceeds LDS");
    auto kern = conv_wgrad_k<T, KH, KW, S, MT, TH, NCW, BIASW>;
    static bool attr_set = false;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set = true;
    }
    p.tiles_x = cdiv(p.OW, TW);
    p.tiles_y = cdiv(p.OH, TH);
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.nsplit > p.ntiles) p.nsplit = p.ntiles;
    dim3 grid((unsigned)p.citiles, (unsigned)p.ctiles, (unsigned)p.nsplit);
    if (SG_WG_XCD) grid = dim3((unsigned)(p.citiles * p.ctiles * ((p.nsplit + 7) / 8) * 8), 1, 1);
    char cls[96];
    snprintf(cls, sizeof(cls), "conv_wgrad<%s,%dx%d,s%d,MT%d>", sizeof(T) == 4 ? "f32" : (__is_same(T, __bf16) ? "bf16" : "f16"), KH, KW, S, MT);
    const double px = (double)p.B * p.OH * p.OW;
    const int tok = sg_prof_start(cls, 2.0 * px * KH * KW * p.Cin * p.Cout,
                                  ((double)p.B * p.H * p.W * p.Cin + px * p.Cout) * sizeof(T), st);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), SMEM, st, p);
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return 0;
}

template <typename T, int MT>
static int dispatch_wgrad(const WgradP& p, int kh, int kw, int s, hipStream_t st) {
#define SG_CASE(KH_, KW_, S_, TH_, NCW_, BIASW_) \
    if (kh == KH_ && kw == KW_ && s == S_) return launch_wgrad<T, KH_, KW_, S_, MT, TH_, NCW_, BIASW_>(p, st);
    SG_CASE(3, 3, 1, 8, 3, true)
    SG_CASE(2, 2, 2, 4, 4, false)
    SG_CASE(2, 2, 1, 8, 4, false)       // first discriminator layer in its space-to-depth form
    SG_CASE(4, 4, 2, 2, 4, false)       // 2-row tiles: 33 KB of LDS, 4 workgroups per CU (4 rows: 58 KB, 2 per CU; 128->256 @256: 553 -> 407 us)
    // 4-row tiles: 32 KB of LDS per workgroup -> 4-5 workgroups (16-20 waves) per CU hide the register-staged global loads;
    // 8-row tiles (57 KB, 2 workgroups per CU) ran the 256->512 layer at 638 TFLOP/s against 840 (scripts/microbench_dwgrad.py)
    SG_CASE(4, 4, 1, 4, 4, false)
    SG_CASE(3, 3, 2, 4, 3, true)
    SG_CASE(1, 1, 2, 4, 4, false)
    SG_CASE(1, 1, 1, 8, 4, false)
    SG_CASE(5, 5, 1, 8, 7, false)
    if constexpr (MT == 1) { SG_CASE(9, 9, 1, 8, 9, false) }
    if constexpr (MT == 1) { SG_CASE(7, 7, 2, 4, 8, false) }     // 49 taps: 7 per wave x 16 accumulator registers, 32-row tiles only
#undef SG_CASE
    SG_FAIL("srcgan_conv_wgrad: unsupported kernel %dx%d stride %d", kh, kw, s);
}

extern "C" int srcgan_conv_wgrad(const srcgan_wgrad_desc* d, void* stream) {
    SG_REQUIRE(d && d->dy && d->x && d->slab && d->grad, "srcgan_conv_wgrad: null pointer");
    SG_REQUIRE(!d->bias_grad || (d->kh == 3 && d->kw == 3), "srcgan_conv_wgrad: fused bias gradient is implemented for 3x3 kernels only");
    SG_REQUIRE(sg_dtype_ok(d->dtype), "srcgan_conv_wgrad: bad dtype %d", d->dtype);
    const int esz = d->dtype == SRCGAN_F32 ? 4 : 2, epp = 16 / esz;
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0 && d->Cin > 0 && d->Cout > 0 && d->nsplit > 0,
               "srcgan_conv_wgrad: non-positive dimension");
    SG_REQUIRE(d->x_cs % epp == 0 && d->x_coff % epp == 0 && d->dy_cs % epp == 0 && d->dy_coff % epp == 0,
               "srcgan_conv_wgrad: channel strides/offsets must be multiples of %d", epp);
    SG_REQUIRE(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->dy % 16) == 0, "srcgan_conv_wgrad: tensors must be 16-byte aligned");
    // channels are fetched in 16-byte pieces: round the read extents up, the tensors must own (zero) padding
    const int cin_r = cdiv(d->Cin, epp) * epp, cout_r = cdiv(d->Cout, epp) * epp;
    SG_REQUIRE(d->x_coff + cin_r <= d->x_cs && d->dy_coff + cout_r <= d->dy_cs,
               "srcgan_conv_wgrad: channel slice (rounded to %d) exceeds stride", epp);
    WgradP p;
    memset(&p, 0, sizeof(p));
    p.dy = d->dy; p.x = d->x; p.slab = d->slab;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = cin_r; p.xCs = d->x_cs; p.xcoff = d->x_coff;
    p.OH = d->OH; p.OW = d->OW; p.Cout = cout_r; p.dyCs = d->dy_cs; p.dycoff = d->dy_coff;
    p.pad_y = d->pad_y; p.pad_x = d->pad_x; p.want_bias = d->bias_grad != nullptr;
    const int cot = d->kh >= 7 ? 32 : wg_cot(d->Cout);     // 49 / 81 taps: the per-wave accumulators only fit for 32-row tiles
    p.ctiles = cdiv(d->Cout, cot); p.citiles = cdiv(d->Cin, 32);
    const int th = d->stride == 2 ? 4 : 8;
    const long ntiles = (long)d->B * cdiv(d->OH, th) * cdiv(d->OW, 32);
    p.nsplit = d->nsplit > ntiles ? (int)ntiles : d->nsplit;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool c3 = d->kh == 3 && d->kw == 3 && d->Cout <= 3 && d->OH == d->H && d->OW == d->W;
    const bool c1 = d->kh == 4 && d->kw == 4 && d->Cout == 1 && d->OH == d->H - 1 && d->OW == d->W - 1;
    if (sg_is16(d->dtype) && (c3 || c1) && d->stride == 1 && d->pad_y == 1 && d->pad_x == 1 && !d->bias_grad && d->Cin % 32 == 0) {
        // at most three output channels (conv_last; the PatchGAN's 4x4 prediction layer): the (tap, channel) pairs as the N dimension of
        // one product (wgrad_c3_k).  Tiles are taken over the INPUT pixels
        p.tiles_x = cdiv(p.W, 32); p.tiles_y = cdiv(p.H, 8); p.ntiles = p.B * p.tiles_x * p.tiles_y;
        p.nsplit = d->nsplit;
        if (p.nsplit > p.ntiles) p.nsplit = p.ntiles;
        const dim3 grid((unsigned)(p.citiles * ((p.nsplit + 7) / 8) * 8), 1, 1);
        const double px = (double)p.B * p.OH * p.OW;
        const int tok = sg_prof_start(d->dtype == SRCGAN_F16 ? (c3 ? "conv_wgrad<f16,3x3,s1,c3>" : "conv_wgrad<f16,4x4,s1,c1>") : (c3 ? "conv_wgrad<bf16,3x3,s1,c3>" : "conv_wgrad<bf16,4x4,s1,c1>"),
                                      2.0 * px * d->kh * d->kw * d->Cin * d->Cout,
                                      ((double)p.B * p.H * p.W * d->Cin + px * d->Cout) * 2, st);
        if (c3) {
            if (d->dtype == SRCGAN_F16) hipLaunchKernelGGL((wgrad_c3_k<_Float16, 3, 3>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((wgrad_c3_k<__bf16, 3, 3>), grid, dim3(256), 0, st, p);
        } else {
            if (d->dtype == SRCGAN_F16) hipLaunchKernelGGL((wgrad_c3_k<_Float16, 4, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((wgrad_c3_k<__bf16, 4, 1>), grid, dim3(256), 0, st, p);
        }
        sg_prof_stop(tok, st);
        SG_LAUNCH_CHECK();
        rc = 0;
    } else
    if (cot == 32) rc = d->dtype == SRCGAN_F32 ? dispatch_wgrad<float, 1>(p, d->kh, d->kw, d->stride, st)
                      : d->dtype == SRCGAN_F16 ? dispatch_wgrad<_Float16, 1>(p, d->kh, d->kw, d->stride, st)
                                               : dispatch_wgrad<__bf16, 1>(p, d->kh, d->kw, d->stride, st);
    else rc = d->dtype == SRCGAN_F32 ? dispatch_wgrad<float, 2>(p, d->kh, d->kw, d->stride, st)
            : d->dtype == SRCGAN_F16 ? dispatch_wgrad<_Float16, 2>(p, d->kh, d->kw, d->stride, st)
                                     : dispatch_wgrad<__bf16, 2>(p, d->kh, d->kw, d->stride, st);
    if (rc) return rc;
    WredP q;
    q.slab = d->slab; q.grad = d->grad; q.bias_grad = d->bias_grad; q.nsplit = p.nsplit; q.ctiles = p.ctiles; q.citiles = p.citiles;
    q.ntap = d->kh * d->kw; q.kw = d->kw; q.COT = cot; q.Cout = d->Cout; q.Cin = d->Cin;
    q.sr = d->sr; q.sk = d->sk; q.sty = d->sty; q.stx = d->stx; q.off = d->off;
    q.alpha = d->alpha; q.accumulate = d->accumulate;
    const long per_split = (long)p.ctiles * p.citiles * (q.ntap + 1) * cot * 32;
    hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)cdivl(per_split, 64)), dim3(256), 0, st, q);
    SG_LAUNCH_CHECK();
    return 0;
}
