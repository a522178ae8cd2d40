// Dense-block weight gradient: ONE pass over a block's activation buffer A (nf+4gc channels) and its dense
// gradient buffer Gd = [dy5|dy4|dy3|dy2|dy1] produces the weight and bias gradients of all five 3x3 convolutions of a
// ResidualDenseBlock_5 (reference src/model/rddb.py:52-68; replaces five aten::convolution_backward wgrad calls).
//
// Why: the per-conv wgrad kernel (conv_wgrad.hip) stages 32x32 or 64x32 (Cout x Cin) tiles = 124-190 FLOP per staged
// byte and measured HBM-bound (1.1 GB fetched per launch, 6.4 TB/s, 383-533 TFLOP/s).  Here a workgroup owns a
// (up to 128 gradient channels) x (64 input channels) x 9 taps block: 346 FLOP per staged byte; 64 input channels =
// full 128-byte lines of the NHWC buffers.  Each wave owns one 32x32 (g-tile, ci-tile) pair and keeps all nine taps'
// accumulators (+ the bias pseudo-tap, B operand = ones) in registers across its pixel range; split-K over pixel
// ranges into an f32 slab; a segment-aware reduce scatters rows to the five canonical [Cout,Cin,3,3] gradients
// (rows of a tile may belong to different convolutions with different Cin; surplus columns are dropped).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

struct WdSeg { int g0, g1; float* grad; float* bias; int Cin; float alpha; };
struct WdP {
    const void* dy; const void* x; float* slab;
    int B, H, W, G, C, dycoff, xcoff;
    long dypix, dyplane, xpix, xplane;       // bytes; see conv_params.h (interleaved NHWC: plane = 64)
    int g_base;                 // first gradient channel of this launch's row group
    int cin_lim[4];             // per 32-row block of the group: input channels some segment needs (blocks above the dense connectivity's triangle are skipped)
    int nsplit, tiles_x, tiles_y, ntiles, ncit, want_bias;
    unsigned long long* trace;       // diagnostic (SG_TRACE builds): per-tile-step timestamps of one workgroup
};
struct WdRedP {
    const float* slab; int nsplit, ncit, COT, g_base, nseg; WdSeg seg[8]; int accumulate;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_;
__device__ __forceinline__ bf16x8 tr_frag2(const char* p0, const char* p1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_*)(p1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

static __device__ __attribute__((aligned(64))) unsigned int wd_zero_page[16];
template <typename T>
__device__ __forceinline__ size_t wd_chan_off(int c, long plane) {
    constexpr int KCE = DT<T>::KCE;
    return (size_t)(c / KCE) * plane + (size_t)(c % KCE) * sizeof(T);
}
typedef const __attribute__((address_space(1))) void* wd_gptr_t;
typedef __attribute__((address_space(3))) void* wd_lptr_t;

// Operands arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a double-buffered LDS image
// of 32-channel planes [pixel][32 ch] (no staging registers: the 10 accumulators already take 160 VGPRs).
template <typename T, int MT, int NT, int TH>
__global__ __launch_bounds__(MT * NT * 64) void wgrad_dense_k(const WdP p) {
    using D = DT<T>;
    constexpr int NW = MT * NT, TW = 32, NTAP = 9;
    constexpr int IHT = TH + 2, IWT = TW + 2;
    constexpr int PB = 32 * (int)sizeof(T), PPP = PB / 16, PXP = 64 / PPP;      // pixels per 1-KiB piece
    constexpr int DPX = TH * TW, XPX = IHT * IWT;                               // pixels per plane
    constexpr int DPP = (DPX + PXP - 1) / PXP, XPP = (XPX + PXP - 1) / PXP;     // pieces per plane
    constexpr int DBYTES = DPP * 1024, XBYTES = XPP * 1024;
    constexpr int NPIECE = MT * DPP + NT * XPP, SBYTES = MT * DBYTES + NT * XBYTES;
    constexpr int IPW = (NPIECE + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];               // [2 stages][MT dy planes | NT x planes]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave / NT, wn = wave % NT;        // this wave's (g-tile, ci-tile)
    const int cit = blockIdx.x, split = blockIdx.y;
    const bool do_bias = wn == 0 && cit == 0 && p.want_bias;
    const char* zp = (const char*)wd_zero_page;

    f32x16 acc[NTAP + 1];
#pragma unroll
    for (int a = 0; a <= NTAP; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

    const int t_begin = (int)((long)p.ntiles * split / p.nsplit);
    const int t_end = (int)((long)p.ntiles * (split + 1) / p.nsplit);

    auto issue = [&](int t, int stage) {
        int q = t;
        const int tx = q % p.tiles_x; q /= p.tiles_x;
        const int ty = q % p.tiles_y;
        const int b = q / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const char* dyb = (const char*)p.dy + (size_t)b * p.H * p.W * p.dypix;
        const char* xb = (const char*)p.x + (size_t)b * p.H * p.W * p.xpix;
        char* ls = smem + stage * SBYTES;
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int pi = it * NW + wave;                    // wave-uniform piece id
            if (IPW * NW != NPIECE && pi >= NPIECE) continue;
            const char* src = zp;
            const int part = lane % PPP, lpx = lane / PPP;
            if (pi < MT * DPP) {
                const int m = pi / DPP, pix = (pi - m * DPP) * PXP + lpx;
                const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
                const int ch = p.g_base + m * 32 + part * D::EPP;
                if (pix < DPX && oy < p.H && ox < p.W && ch < p.G)
                    src = dyb + (size_t)(oy * p.W + ox) * p.dypix + wd_chan_off<T>(p.dycoff + ch, p.dyplane);
            } else {
                const int pj = pi - MT * DPP, n = pj / XPP, pix = (pj - n * XPP) * PXP + lpx;
                const int iy = pix / IWT, ix = pix - iy * IWT;
                const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
                const int ch = (cit * NT + n) * 32 + part * D::EPP;
                if (pix < XPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && ch < p.C)
                    src = xb + (size_t)(gy * p.W + gx) * p.xpix + wd_chan_off<T>(p.xcoff + ch, p.xplane);
            }
            __builtin_amdgcn_global_load_lds((wd_gptr_t)src, (wd_lptr_t)(ls + pi * 1024), 16, 0, 0);
        }
    };

    if (t_begin < t_end) issue(t_begin, 0);
    int stage = 0;
    for (int t = t_begin; t < t_end; ++t, stage ^= 1) {
        __syncthreads();            // own DMAs of tile t landed (vmcnt(0)); everyone is done with the other stage
        if (t + 1 < t_end) issue(t + 1, stage ^ 1);
        const char* my_d = smem + stage * SBYTES + wm * DBYTES;
        const char* my_x = smem + stage * SBYTES + MT * DBYTES + wn * XBYTES;
        if constexpr (std::is_same<T, float>::value) {
#pragma unroll 2
            for (int kk = 0; kk < TH * TW / 2; ++kk) {
                const int pix = 2 * kk + h;
                const int py = pix / TW, px = pix % TW;
                const float a = *(const float*)(my_d + pix * PB + r * 4);
#pragma unroll
                for (int tap = 0; tap < NTAP; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    const float bv = *(const float*)(my_x + ((py + ky) * IWT + px + kx) * PB + r * 4);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[tap], 0, 0, 0);
                }
                if (do_bias) acc[NTAP] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, acc[NTAP], 0, 0, 0);
            }
        } else {
            const int gq = lane >> 4, idx = lane & 15, qq = idx >> 2, pp = idx & 3;
            const int choff = ((gq & 1) * 16 + 4 * pp) * 2;
            const bf16x8 ones = sg_ones16<T>();
#pragma unroll 2
            for (int kk = 0; kk < TH * 2; ++kk) {
                const int py = kk >> 1, xh = (kk & 1) * 16;
                const int px0 = xh + 8 * h + qq;
                const char* ab = my_d + (py * TW + px0) * PB + choff;
                const bf16x8 a = tr_frag2(ab, ab + 4 * PB);
#pragma unroll
                for (int tap = 0; tap < NTAP; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    const char* bb = my_x + ((py + ky) * IWT + px0 + kx) * PB + choff;
                    acc[tap] = sg_mfma16<T>(a, tr_frag2(bb, bb + 4 * PB), acc[tap]);
                }
                if (do_bias) acc[NTAP] = sg_mfma16<T>(a, ones, acc[NTAP]);
            }
        }
    }

    // ---- slab [split][cit][tap (10)][row (32*MT)][col (32*NT)];  acc[.][4g+i] = D[row = 8g+4h+i][col = r]
    constexpr int COT = 32 * MT, CIT = 32 * NT;
    float* sp = p.slab + ((size_t)split * p.ncit + cit) * (NTAP + 1) * COT * CIT;
#pragma unroll
    for (int tap = 0; tap <= NTAP; ++tap) {
        if (tap == NTAP && !do_bias) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wm * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            sp[((size_t)tap * COT + row) * CIT + wn * 32 + r] = acc[tap][i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 fast path (image a whole number of tiles, operands < 2 GiB): the same tiling and LDS image, but
//  * no per-piece address arithmetic: a wave's IPW per-lane offsets (relative to the tile origin) are computed once;
//    a piece is `buffer_load_dwordx4 voff, rsrc(tile origin), soff(channel plane) offen lds`.  The halo pixels outside
//    the image only occur in the first/last tile row/column: a 4-bit class per piece (8 pieces in one VGPR) ANDed with
//    the tile's border mask swaps in an out-of-range offset, which the buffer range check zero-fills (3 VALU/piece);
//  * the pieces of tile t+1 are issued from inside the MFMA loop of tile t, two per k-step, instead of in a block in
//    front of it: a DMA issue costs ~60 cycles among MFMAs but ~150 in a block, during which the co-resident wave of
//    the SIMD -- in lockstep through the barrier -- was issuing too, leaving the matrix pipe idle (the old form ran
//    at 45-50 % MFMA utilisation: 11.4k cycles per tile step against 5.1k of MFMA).
// Inline asm, not __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc (ROCm 7.2) treats the builtin as an LDS store that may
// alias every later ds_read of the wave and puts `s_waitcnt vmcnt(0)` in front of the next fragment read -- the piece
// had to LAND (~2 us) before the MFMA loop could go on (seen in the .s; the older waves of each SIMD ran at 50 %).
// The compiler does not count asm memory operations: the loop waits `vmcnt(0)` by hand before its barrier.
typedef int wd_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wd_dma16(const void* base, int voff, int soff, wd_lptr_t lds) {
    const unsigned long long a = (unsigned long long)base;
    wd_v4i rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    rs[2] = 0x7fffffff;
    rs[3] = 0x00020000;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
}

// Timing experiments (wrong results; scripts/wd_variants.sh): 1 no DMA after the first tile, 2 no MFMA, 4 no input-fragment
// reads in the loop, 8 every piece out of range (issue cost without memory traffic).  Round-1 findings at the bench shape
// (random data, so lower clocks than in training): 452 us as built; 316 us without the DMA stream; 197 us without MFMAs;
// 283 us with the DMAs issued but out of range -- the issue itself is free, the loss is the memory path pushing back on the
// issuing (MFMA) waves; a step is 4864 matrix-pipe cycles + the per-wave stall time, because the older wave of each SIMD
// pair takes the pipe first and the younger one finishes alone (trace: waves 4-7 reach the barrier ~1.2 us after 0-3).
// Letting the younger wave lead for the first K groups (SG_WD_PRIO_K, s_setprio) evens the arrivals but not the step time.
// Back-to-back launches of this kernel are power-limited: the first launch of a run takes 357 us, the sustained median is
// 464 us at ~1.75-1.9 GHz; in the training step (mixed with HBM-bound convolutions) it runs at the 360 us end.
#ifndef SG_WD_EXP
#define SG_WD_EXP 0
#endif
template <typename T, int MT, int NT, int TH>
__global__ __launch_bounds__(MT * NT * 64) void wgrad_dense_fast_k(const WdP p) {
    using D = DT<T>;
    constexpr int NW = MT * NT, TW = 32, NTAP = 9;
    constexpr int IHT = TH + 2, IWT = TW + 2;
    constexpr int PB = 64, PXP = 16;                                            // bytes per pixel of a plane; pixels per piece
    constexpr int DPX = TH * TW, XPX = IHT * IWT;
    constexpr int DPP = (DPX + PXP - 1) / PXP, XPP = (XPX + PXP - 1) / PXP;
    constexpr int DBYTES = DPP * 1024, XBYTES = XPP * 1024;
    constexpr int NPIECE = MT * DPP + NT * XPP, SBYTES = MT * DBYTES + NT * XBYTES;
    constexpr int IPW = (NPIECE + NW - 1) / NW;
    constexpr int OOB = 0x7fffffff;
#ifndef SG_WD_PPK
#define SG_WD_PPK 0
#endif
    constexpr int PPK = SG_WD_PPK ? SG_WD_PPK : 4;                              // pieces issued per group, from the first group on
    static_assert(IPW <= 16, "piece classes are packed 4 bits each into two registers");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NT, wn = wave % NT;
    // Workgroup -> (input-channel tile, pixel split).  Consecutive workgroup ids go round-robin over the 8 XCDs, so the
    // ncit workgroups that read the SAME gradient tiles (same split, different cit) are given ids 8 apart: same XCD,
    // dispatched together, and two of the three gradient reads hit that XCD's L2 (measured before: 1.4 GB fetched per
    // launch against 0.67 GB of operands).
    int cit, split;
    {
        const int L = blockIdx.x + gridDim.x * blockIdx.y, nwg = gridDim.x * gridDim.y;
        const int xcd = L & 7, j = L >> 3;
        const int full = (nwg >> 3) * 8;                       // ids below `full` fill whole rounds of 8 XCDs
        // position in the XCD-major order: XCD k owns positions [k * per, (k + 1) * per)
        const int per = nwg >> 3;
        const int pos = L < full ? xcd * per + j : L;          // the ragged tail keeps its id
        cit = pos % p.ncit; split = pos / p.ncit;
        if (split >= p.nsplit) return;
    }
    const bool do_bias = wn == 0 && cit == 0 && p.want_bias;
    // 32x32 blocks no convolution of the dense block needs (conv m reads only the first 64 + 32 (m - 1) channels): the wave still
    // issues its share of the DMA and joins the barriers, but reads no fragments and issues no MFMAs -- 6 of the 32 blocks of a
    // dense block's two launches.  The step gets no shorter for it (another SIMD pair still has two busy waves); the kernel
    // runs power-limited, and the matrix pipe's energy is what the clock is traded against.
#ifdef SG_WD_NO_SKIP
    const bool useful = true;
#else
    const bool useful = (cit * NT + wn) * 32 < p.cin_lim[wm] || do_bias;
#endif

    // MFMA shape.  S16 = 1: four 16x16x32 MFMAs per (tap, 32 pixels) instead of two 32x32x16 -- same cycles per FLOP, same LDS
    // bytes, same 160 accumulator registers; the kernel runs power-limited (1.9-2.0 GHz in the training step) and the chip
    // holds a higher clock on the 16x16x32 shape (MI355X_MICROARCH.md, DVFS give-back item 7).
#ifndef SG_WD_S16
#define SG_WD_S16 1
#endif
    constexpr bool S16 = SG_WD_S16 != 0;
    // LDS image swizzle of the 16x16x32 form: lanes l and l + 16 of a fragment read pixel blocks 8 columns apart -- 512 bytes, the
    // same banks (SQ_LDS_BANK_CONFLICT: 6.3e9 cycles per 207 launches before this).  The two 32-byte channel halves of a
    // pixel are therefore stored swapped where bit 3 of the pixel's column is set (on the DMA SOURCE side: the LDS destination
    // of a piece is lane-linear), and a fragment read selects the half by  cb ^ bit3(column): columns 8 apart -> opposite halves.
    // ---- per-wave piece table: per-lane offset from the tile origin, channel-plane offset (uniform), border class
    int voff[IPW]; long soff[IPW];          // soff: channel-plane offset, 64-bit and wave-uniform (a blocked 1024x1024 batch spans > 2^31 bytes)
    unsigned cls[2] = {0u, 0u};
    {
        const int part = lane & 3, lpx = lane >> 2;
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int pi = it * NW + wave;
            int v = OOB; long so = 0;
            if (pi < MT * DPP) {
                const int m = pi / DPP, pix = (pi - m * DPP) * PXP + lpx;
                const int psw = S16 ? part ^ ((((pix % TW) >> 3) & 1) << 1) : part;       // the global 16-byte part this LDS slot holds
                const int ch = p.g_base + m * 32 + psw * D::EPP;
                if (pix < DPX && ch < p.G) v = ((pix / TW) * p.W + pix % TW) * (int)p.dypix + psw * 16;
                so = (long)wd_chan_off<T>(p.dycoff + p.g_base + m * 32, p.dyplane);
            } else if (pi < NPIECE) {
                const int pj = pi - MT * DPP, n = pj / XPP, pix = (pj - n * XPP) * PXP + lpx;
                const int iy = pix / IWT, ix = pix - iy * IWT;
                const int psw = S16 ? part ^ (((ix >> 3) & 1) << 1) : part;
                const int ch = (cit * NT + n) * 32 + psw * D::EPP;
                if (pix < XPX && ch < p.C) v = (iy * p.W + ix) * (int)p.xpix + psw * 16;
                so = (long)wd_chan_off<T>(p.xcoff + (cit * NT + n) * 32, p.xplane);
                cls[it / 8] |= (unsigned)((iy == 0) | ((iy == IHT - 1) << 1) | ((ix == 0) << 2) | ((ix == IWT - 1) << 3)) << (4 * (it % 8));
            }
            voff[it] = v;
            soff[it] = ((long)__builtin_amdgcn_readfirstlane((int)(so >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(so & 0xffffffff));
        }
    }

    f32x16 acc[S16 ? 1 : NTAP + 1];
    f32x4 acq[S16 ? NTAP + 1 : 1][2][2];          // [tap][16-row block of g][16-column block of ci]
#pragma unroll
    for (int a = 0; a < (S16 ? 1 : NTAP + 1); ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
#pragma unroll
    for (int a = 0; a < (S16 ? NTAP + 1 : 1); ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acq[a][(i >> 3) & 1][(i >> 2) & 1][i & 3] = 0.f;

    const int t_begin = (int)((long)p.ntiles * split / p.nsplit);
    const int t_end = (int)((long)p.ntiles * (split + 1) / p.nsplit);
    // tile decode, advanced with scalar carries
    // tiles are walked column-wise (ty fastest): consecutive tiles share 2 of their TH+2 halo rows, just fetched -> L2
    int ty = t_begin % p.tiles_y, tx = (t_begin / p.tiles_y) % p.tiles_x, tb = t_begin / (p.tiles_x * p.tiles_y);
    tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty); tb = __builtin_amdgcn_readfirstlane(tb);
    const char* dyo = nullptr; const char* xo = nullptr; unsigned clm[2] = {0u, 0u};
    auto origin = [&]() {                 // operands of the tile (tb, ty, tx): uniform origins + this wave's masked classes
        const long px = ((long)tb * p.H + ty * TH) * p.W + tx * TW;
        dyo = (const char*)p.dy + px * p.dypix;
        xo = (const char*)p.x + (px - p.W - 1) * p.xpix;
        const unsigned m = (unsigned)((ty == 0) | ((ty == p.tiles_y - 1) << 1) | ((tx == 0) << 2) | ((tx == p.tiles_x - 1) << 3));
        clm[0] = cls[0] & (m * 0x11111111u); clm[1] = cls[1] & (m * 0x11111111u);
    };
    auto advance = [&]() { if (++ty == p.tiles_y) { ty = 0; if (++tx == p.tiles_x) { tx = 0; ++tb; } } };
    auto piece = [&](int it, int stage) __attribute__((always_inline)) {
        const int pi = it * NW + wave;
        if (IPW * NW != NPIECE && pi >= NPIECE) return;
        int v = ((clm[it / 8] >> (4 * (it % 8))) & 15u) ? OOB : voff[it];
        if (SG_WD_EXP & 8) v = OOB;
        wd_dma16((pi < MT * DPP ? dyo : xo) + soff[it], v, 0, (wd_lptr_t)(smem + stage * SBYTES + pi * 1024));
    };

    if (t_begin < t_end) {
        origin();
#pragma unroll
        for (int it = 0; it < IPW; ++it) piece(it, 0);
    }
    const int gq = lane >> 4, idx = lane & 15, qq = idx >> 2, pp = idx & 3;
    const int choff = ((gq & 1) * 16 + 4 * pp) * 2;
    const bf16x8 ones = sg_ones16<T>();
    int stage = 0;
    for (int t = t_begin; t < t_end; ++t, stage ^= 1) {
#ifdef SG_TRACE
        const unsigned long long t_arr = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0x0f70);
        const unsigned long long t_land = __builtin_amdgcn_s_memrealtime();
#endif
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): own (asm-issued) DMAs of tile t landed
        __syncthreads();            // everyone's landed; everyone is done with the other stage
#ifdef SG_TRACE
        if (p.trace && blockIdx.x == 1 && blockIdx.y == 3 && lane == 0 && t - t_begin >= 8 && t - t_begin < 12)
            p.trace[160 + (t - t_begin - 8) * 8 + wave] = t_arr;
        if (p.trace && blockIdx.x == 1 && blockIdx.y == 3 && wave == 0 && lane == 0 && t - t_begin < 40) {
            p.trace[(t - t_begin) * 4 + 0] = t_arr; p.trace[(t - t_begin) * 4 + 1] = t_land; p.trace[(t - t_begin) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            p.trace[(t - t_begin) * 4 + 3] = __builtin_readcyclecounter();
        }
#endif
        const bool more = t + 1 < t_end;
        if (more) { advance(); origin(); }
        if (!useful) {
            if (more && !(SG_WD_EXP & 1)) {
#pragma unroll
                for (int it = 0; it < IPW; ++it) piece(it, stage ^ 1);
            }
            continue;
        }
        const char* my_d = smem + stage * SBYTES + wm * DBYTES;
        const char* my_x = smem + stage * SBYTES + MT * DBYTES + wn * XBYTES;
        // Row-ordered: the TH*2 gradient fragments a[py][xh] of the tile are read once; an input fragment b(i, kx, xh)
        // (input row i, tap column kx, pixel half xh) is read ONCE and feeds every (output row py, ky) with py + ky == i
        // -- (TH+2)*3*2 = 36 input reads per tile instead of TH*9*2 = 72.  Reads run PD groups ahead of the MFMAs.
#ifndef SG_WD_PD
#define SG_WD_PD 4
#endif
        if constexpr (S16) {
            // Groups G = (input row i2, tap column kx); a group's fragments cover all 32 pixels of the row (k = 8 * (lane >> 4) + j)
            // for the two 16-channel halves: gradient fragments fa[py][cb] are read once per tile, input fragments fb[..][cb] once
            // per group and feed every (output row py, ky) with py + ky == i2.
            constexpr int NG = IHT * 3, PDG = 1;
            bf16x8 fa[TH][2], fb[PDG + 1][2];
            const int kq = lane >> 4;                              // k-group: pixels 8 kq .. 8 kq + 7
            // lane part of a fragment address: pixel 8 kq + qq, half (kq & 1) = bit 3 of that column, 8-byte piece pp; the other
            // channel half and a column whose bit 3 differs (second read, shifted by kx: carry out of the low three bits) are an
            // XOR with 32
            const int lb = (8 * kq + qq) * PB + (kq & 1) * 32 + 8 * pp;
            const int mk1 = qq + 1 >= 4 ? 32 : 0, mk2 = qq + 2 >= 4 ? 32 : 0;
            auto rd_b = [&](int G) {
                const int i2 = G / 3, kx = G % 3;
                const char* bb = my_x + (i2 * IWT + kx) * PB;
                const int l0 = lb, l1 = lb ^ (kx == 0 ? 0 : kx == 1 ? mk1 : mk2);
                fb[G % (PDG + 1)][0] = tr_frag2(bb + l0, bb + 4 * PB + l1);
                fb[G % (PDG + 1)][1] = tr_frag2(bb + (l0 ^ 32), bb + 4 * PB + (l1 ^ 32));
            };
            auto rd_a = [&](int py) {
                const char* ab = my_d + py * TW * PB;
                fa[py][0] = tr_frag2(ab + lb, ab + 4 * PB + lb);
                fa[py][1] = tr_frag2(ab + (lb ^ 32), ab + 4 * PB + (lb ^ 32));
            };
            rd_a(0); rd_b(0);
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                const int i2 = G / 3, kx = G % 3;
                if (kx == 1 && i2 + 1 < TH) rd_a(i2 + 1);          // row py is first used by group 3 * py
                if (G + PDG < NG) rd_b(G + PDG);
                if (more) {              // next tile's pieces ride in the shadow of this tile's MFMAs
#pragma unroll
                    for (int j = 0; j < 2 * PPK; ++j)
                        if (2 * PPK * G + j < IPW) piece(2 * PPK * G + j, stage ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int py = 0; py < TH; ++py) {
                    const int ky = i2 - py;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acq[ky * 3 + kx][mb][nb] = sg_mfma16s<T>(fa[py][mb], fb[G % (PDG + 1)][nb], acq[ky * 3 + kx][mb][nb]);
                }
                if (do_bias && kx == 0 && (i2 == 0 || i2 >= IHT - 2))
#pragma unroll
                    for (int py = (i2 == 0 ? 0 : i2 == IHT - 2 ? 1 : TH / 2); py < (i2 == 0 ? 1 : i2 == IHT - 2 ? TH / 2 : TH); ++py)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acq[NTAP][mb][0] = sg_mfma16s<T>(fa[py][mb], ones, acq[NTAP][mb][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        constexpr int NGRP = IHT * 3 * 2, PD = SG_WD_PD;
        bf16x8 fa[TH][2], fb[PD + 1];
        auto read_b = [&](int G) {
            const int i2 = G / 6, kx = (G % 6) / 2, xh = G % 2;
            const char* bb = my_x + (i2 * IWT + xh * 16 + 8 * h + qq + kx) * PB + choff;
            fb[G % (PD + 1)] = tr_frag2(bb, bb + 4 * PB);
        };
        // Gradient fragments are read lazily: row py is first used by group 6 * py, so only row 0 is read in front of the
        // loop (with the first PD input fragments) and row py follows FA_LEAD groups ahead of its first use -- the matrix
        // pipe starts after 4 LDS reads per wave instead of 20 (all 8 waves leave the barrier together and queue on the LDS
        // port).
        auto read_a = [&](int py, int xh) {
            const char* ab = my_d + (py * TW + xh * 16 + 8 * h + qq) * PB + choff;
            fa[py][xh] = tr_frag2(ab, ab + 4 * PB);
        };
        read_a(0, 0); read_b(0); read_a(0, 1);
#pragma unroll
        for (int G = 1; G < PD; ++G) read_b(G);
        constexpr int FA_LEAD = 4;
#ifndef SG_WD_PRIO_K
#define SG_WD_PRIO_K 0
#endif
#pragma unroll
        for (int G = 0; G < NGRP; ++G) {
            if (SG_WD_PRIO_K > 0 && wave >= NW / 2) {       // the younger wave of each SIMD pair leads for the first K groups
                if (G == 0) __builtin_amdgcn_s_setprio(1);
                if (G == SG_WD_PRIO_K) __builtin_amdgcn_s_setprio(0);
            }
            if ((G + FA_LEAD) % 6 < 2 && (G + FA_LEAD) / 6 >= 1 && (G + FA_LEAD) / 6 < TH) read_a((G + FA_LEAD) / 6, (G + FA_LEAD) % 6);
            if (G + PD < NGRP && !(SG_WD_EXP & 4)) read_b(G + PD);
            if (more && !(SG_WD_EXP & 1)) {              // next tile's pieces ride in the shadow of this tile's MFMAs
#pragma unroll
                for (int j = 0; j < PPK; ++j)
                    if (PPK * G + j < IPW) piece(PPK * G + j, stage ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef SG_TRACE
            if (G % 5 == 0 && p.trace && blockIdx.x == 1 && blockIdx.y == 3 && lane == 0 && t - t_begin == 10)
                p.trace[192 + wave * 8 + G / 5] = __builtin_amdgcn_s_memrealtime();
#endif
            const int i2 = G / 6, kx = (G % 6) / 2, xh = G % 2;
#pragma unroll
            for (int py = 0; py < TH; ++py) {
                const int ky = i2 - py;
                if (ky < 0 || ky > 2) continue;
                if (!(SG_WD_EXP & 2)) acc[ky * 3 + kx] = sg_mfma16<T>(fa[py][xh], fb[G % (PD + 1)], acc[ky * 3 + kx]);
                else asm volatile("" :: "v"(fa[py][xh]), "v"(fb[G % (PD + 1)]));
            }
            // bias pseudo-tap: one MFMA per gradient fragment, placed in the light groups of the first/last input rows
            // (row 0 in input row 0 -- the only gradient row read by then --, the rest in the last two input rows)
            if (do_bias && kx == 0 && (i2 == 0 || i2 >= IHT - 2))
#pragma unroll
                for (int py = (i2 == 0 ? 0 : i2 == IHT - 2 ? 1 : TH / 2); py < (i2 == 0 ? 1 : i2 == IHT - 2 ? TH / 2 : TH); ++py)
                    acc[NTAP] = sg_mfma16<T>(fa[py][xh], ones, acc[NTAP]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    }

    constexpr int COT = 32 * MT, CIT = 32 * NT;
    float* sp = p.slab + ((size_t)split * p.ncit + cit) * (NTAP + 1) * COT * CIT;
    // a block no convolution of the dense block needs (its accumulators are zero) is not written: the reduce kernel does not read
    // elements beyond a segment's Cin either (12.5 % of the 83 MB slab of the 128-row launch, a sixth of the 64-row one)
    const bool blk_needed = (cit * NT + wn) * 32 < p.cin_lim[wm];
#pragma unroll
    for (int tap = 0; tap <= NTAP; ++tap) {
        if (tap == NTAP ? !do_bias : !blk_needed) continue;
        if constexpr (S16) {
            // 16x16 result block: column = lane & 15, row = 4 * (lane >> 4) + register
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = wm * 32 + mb * 16 + 4 * (lane >> 4) + i;
                        sp[((size_t)tap * COT + row) * CIT + wn * 32 + nb * 16 + (lane & 15)] = acq[tap][mb][tap == NTAP ? 0 : nb][i];
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wm * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                sp[((size_t)tap * COT + row) * CIT + wn * 32 + r] = acc[tap][i];
            }
        }
    }
}

// 64 slab elements x 4 split lanes per block; fixed order -> deterministic.  Scatter by segment.
template <int CIT>
__global__ __launch_bounds__(256) void wgrad_dense_reduce_k(const WdRedP p) {
    __shared__ float red[4][64];
    const long per_split = (long)p.ncit * 10 * p.COT * CIT;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + el;
    float s = 0.f;
    // Only elements some segment scatters are read: the bias pseudo-tap holds one useful column (10 % of the slab otherwise), and the
    // columns beyond a segment's Cin are not written by the fast kernel.
    bool wanted = false;
    if (e < per_split) {
        long q = e;
        const int col = (int)(q % CIT); q /= CIT;
        const int row = (int)(q % p.COT); q /= p.COT;
        const int tap = (int)(q % 10);
        const int cit = (int)(q / 10);
        const int g = p.g_base + row, ci = cit * CIT + col;
        for (int k = 0; k < p.nseg; ++k) {
            const WdSeg sg = p.seg[k];
            if (g < sg.g0 || g >= sg.g1) continue;
            wanted = tap == 9 ? (sg.bias && cit == 0 && col == 0) : (sg.grad && ci < sg.Cin);
            break;
        }
    }
    if (wanted) {                             // four loads in flight per lane; the order stays fixed
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
    if (sl != 0 || !wanted) return;
    s = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    long q = e;
    const int col = (int)(q % CIT); q /= CIT;
    const int row = (int)(q % p.COT); q /= p.COT;
    const int tap = (int)(q % 10);
    const int cit = (int)(q / 10);
    const int g = p.g_base + row, ci = cit * CIT + col;
    for (int k = 0; k < p.nseg; ++k) {
        const WdSeg sg = p.seg[k];
        if (g < sg.g0 || g >= sg.g1) continue;
        const int co = g - sg.g0;
        const float v = s * sg.alpha;
        if (tap == 9) {
            if (sg.bias && cit == 0 && col == 0) sg.bias[co] = p.accumulate ? sg.bias[co] + v : v;
        } else if (sg.grad && ci < sg.Cin) {
            float* gp = sg.grad + ((size_t)co * sg.Cin + ci) * 9 + tap;
            *gp = p.accumulate ? *gp + v : v;
        }
        return;
    }
}

// ------------------------------------------------------------------ host side
template <typename T, int MT, int NT, int TH>
static int launch_wd(WdP p, hipStream_t st) {
    constexpr int PB = 32 * (int)sizeof(T), PXP = 64 / (PB / 16);
    constexpr size_t SMEM = 2 * 1024 * ((size_t)MT * ((TH * 32 + PXP - 1) / PXP) + (size_t)NT * (((TH + 2) * 34 + PXP - 1) / PXP));
    static_assert(SMEM <= 160 * 1024, "wgrad_dense tile exceeds LDS");
    void (*kern)(const WdP) = wgrad_dense_k<T, MT, NT, TH>;
    bool fast = false;
    if constexpr (!std::is_same<T, float>::value) {
        // whole tiles, 32-channel-aligned slices when planar, and every byte offset within 31 bits
        static const bool no_fast = sg_env("SRCGAN_WD_SLOW") != nullptr;
        // per-lane offsets are relative to a tile's origin (64-bit, uniform) and the channel-plane offset is 64-bit too: only a
        // tile's own extent must fit 31 bits
        const double span_dy = (double)(TH + 2) * p.W * p.dypix, span_x = (double)(TH + 4) * p.W * p.xpix;
        fast = !no_fast && p.H % TH == 0 && p.W % 32 == 0 && span_dy < 2.0e9 && span_x < 2.0e9 &&
               (p.dyplane == 64 || (p.dycoff + p.g_base) % 32 == 0) && (p.xplane == 64 || p.xcoff % 32 == 0);
        constexpr int IPWF = (MT * (TH * 2) + NT * (((TH + 2) * 34 + 15) / 16) + MT * NT - 1) / (MT * NT);      // pieces per wave
        if constexpr (IPWF <= 10) { if (fast) kern = wgrad_dense_fast_k<T, MT, NT, TH>; }   // larger tables would spill beside 160 accumulators
        else fast = false;
    }
    static bool attr_set[2] = {false, false};
    if (!attr_set[fast]) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set[fast] = true;
    }
    p.tiles_x = cdiv(p.W, 32);
    p.tiles_y = cdiv(p.H, TH);
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    if (p.nsplit > p.ntiles) p.nsplit = p.ntiles;
    char cls[96];
    snprintf(cls, sizeof(cls), "wgrad_dense<%s,MT%d,NT%d%s>", sizeof(T) == 4 ? "f32" : (__is_same(T, __bf16) ? "bf16" : "f16"), MT, NT, fast ? ",fast" : "");
    const double px = (double)p.B * p.H * p.W;
    const int rows = (p.G - p.g_base) < 32 * MT ? (p.G - p.g_base) : 32 * MT;
    // algorithmic work = the (gradient channel, input channel) pairs some convolution of the block owns (the triangle), not
    // the rectangle the workgroups cover
    double pairs = 0.0;
    for (int b = 0; b < MT; ++b) {
        const int rb = rows - 32 * b < 32 ? rows - 32 * b : 32;
        if (rb > 0) pairs += (double)rb * (p.cin_lim[b] < p.C ? p.cin_lim[b] : p.C);
    }
    const int tok = sg_prof_start(cls, 2.0 * px * 9 * pairs, px * (rows + p.C) * sizeof(T), st);
#ifdef SG_TRACE
    static unsigned long long* trace = nullptr;
    if (!trace) SG_HIP(hipMalloc(&trace, 64 * 4 * 8));
    SG_HIP(hipMemsetAsync(trace, 0, 64 * 4 * 8, st));
    p.trace = trace;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)p.ncit, (unsigned)p.nsplit), dim3(MT * NT * 64), SMEM, st, p);
#ifdef SG_TRACE
    {
        static int dumps = 0;
        if (sg_env("SRCGAN_TRACE") && dumps < 2 && fast) {
            ++dumps;
            unsigned long long h[64 * 4];
            SG_HIP(hipStreamSynchronize(st));
            SG_HIP(hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[trace] %s (10 ns ticks: arrive at barrier, own DMA landed, barrier exit)\n", cls);
            for (int k = 0; k < 4; ++k) {
                fprintf(stderr, "[trace] step %d arrivals by wave (rel. wave 0):", 8 + k);
                for (int w = 0; w < 8; ++w) fprintf(stderr, " %5lld", (long long)(h[160 + k * 8 + w] - h[160 + k * 8]));
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[trace] step 10: time of groups 0,5,..,35 by wave (rel. wave 0 group 0)\n");
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "[trace]   wave %d:", w);
                for (int g = 0; g < 8; ++g) fprintf(stderr, " %5lld", (long long)(h[192 + w * 8 + g] - h[192]));
                fprintf(stderr, "\n");
            }
            for (int k = 0; k < 40 && h[k * 4]; ++k)
                fprintf(stderr, "[trace] %2d arrive %6lld landed %6lld exit %6lld  shader-clock %5.0f MHz\n", k, (long long)(h[k * 4] - h[0]), (long long)(h[k * 4 + 1] - h[0]), (long long)(h[k * 4 + 2] - h[0]),
                        k ? (double)(h[k * 4 + 3] - h[(k - 1) * 4 + 3]) / (double)(h[k * 4 + 2] - h[(k - 1) * 4 + 2]) * 100.0 : 0.0);
        }
    }
#endif
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return p.nsplit;
}

template <typename T>
static int dispatch_wd(const WdP& p, int mt, int nt, hipStream_t st, int& nsplit_used) {
    constexpr bool F = std::is_same<T, float>::value;
    int rc = -1;
    if (nt == 4) {
        if constexpr (!F) {              // bf16 only: two f32 stages of 4 input planes exceed the LDS
            switch (mt) {
                case 1: rc = launch_wd<T, 1, 4, 4>(p, st); break;
                case 2: rc = launch_wd<T, 2, 4, 4>(p, st); break;
            }
        }
    } else if (nt == 3) {
        if constexpr (!F) {
            switch (mt) {
                case 1: rc = launch_wd<T, 1, 3, 4>(p, st); break;
                case 2: rc = launch_wd<T, 2, 3, 4>(p, st); break;
            }
        }
    } else
    switch (mt) {
        case 1: rc = launch_wd<T, 1, 2, F ? 2 : 4>(p, st); break;
        case 2: rc = launch_wd<T, 2, 2, F ? 2 : 4>(p, st); break;
        case 3: rc = launch_wd<T, 3, 2, F ? 2 : 4>(p, st); break;
        case 4: rc = launch_wd<T, 4, 2, F ? 2 : 4>(p, st); break;
    }
    if (rc < 0) SG_FAIL("srcgan_wgrad_dense: bad row-tile count %d", mt);
    nsplit_used = rc;
    return 0;
}

// Input-channel tiles per workgroup: a row group of <= 64 gradient channels takes 4 (128 input channels, 8 waves)
// instead of 2 -- the 2x2-wave form left one wave per SIMD (640 TFLOP/s against 1000 for the 4x2 form) and staged
// the gradient planes once per 64 input channels.
// (3 when 96 input channels suffice -- the (dy2, dy1) row group of a dense block: 55 instead of 68 KiB staged per tile step)
static int wd_nt(int mt, int cin_max, int dtype) { return (sg_is16(dtype) && mt <= 2 && cin_max > 64) ? (cin_max <= 96 ? 3 : 4) : 2; }

// pixel splits per row group: fill the chip once (workgroups resident per CU follow from the LDS tile)
static int wd_nsplit(int mt, int nt, int ncit, int dtype, int B, int H, int W) {
    const int th = dtype == SRCGAN_F32 ? 2 : 4;
    const int pb = dtype == SRCGAN_F32 ? 128 : 64;
    const long lds = 2 * ((long)mt * th * 32 * pb + (long)nt * (th + 2) * 34 * pb);
    int per_cu = (int)((160 * 1024) / lds); if (per_cu < 1) per_cu = 1;
    if (per_cu * mt * nt > 16) per_cu = 16 / (mt * nt) > 0 ? 16 / (mt * nt) : 1;      // <= 16 waves per CU
    const long ntiles = (long)B * cdiv(H, th) * cdiv(W, 32);
    long cus = 256;
    if (const char* e = sg_env("SRCGAN_WD_CUS")) { const int v = atoi(e); if (v > 0 && v < 256) cus = v; }     // diagnostic builds: share the chip with a concurrent kernel
    long ns = (cus * per_cu) / ncit;
    if (ns > ntiles) ns = ntiles;
    return (int)(ns < 1 ? 1 : ns);
}

extern "C" size_t srcgan_wgrad_dense_slab_bytes(int G, int C, int dtype, int B, int H, int W) {
    size_t mx = 0;
    for (int g_base = 0; g_base < G; g_base += 128) {
        const int rows = G - g_base < 128 ? G - g_base : 128, mt = cdiv(rows, 32);
        for (int nt = 2; nt <= (sg_is16(dtype) && mt <= 2 ? 4 : 2); ++nt) {                    // any column-tile form may be chosen at run time
            const int ncit = cdiv(C, 32 * nt);
            const size_t b = (size_t)wd_nsplit(mt, nt, ncit, dtype, B, H, W) * ncit * 10 * (32 * mt) * (32 * nt) * sizeof(float);
            if (b > mx) mx = b;
        }
    }
    return mx;
}

extern "C" int srcgan_wgrad_dense(const srcgan_wgrad_dense_desc* d, void* stream) {
    SG_REQUIRE(d && d->dy && d->x && d->slab, "srcgan_wgrad_dense: null pointer");
    SG_REQUIRE(sg_dtype_ok(d->dtype), "srcgan_wgrad_dense: bad dtype %d", d->dtype);
    const int esz = d->dtype == SRCGAN_F32 ? 4 : 2, epp = 16 / esz;
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->G > 0 && d->C > 0 && d->nseg > 0 && d->nseg <= 8,
               "srcgan_wgrad_dense: bad dimensions");
    SG_REQUIRE(d->G % epp == 0 && d->C % epp == 0 && d->dy_cs % epp == 0 && d->dy_coff % epp == 0 && d->x_cs % epp == 0 && d->x_coff % epp == 0,
               "srcgan_wgrad_dense: channel counts/strides/offsets must be multiples of %d", epp);
    SG_REQUIRE((d->dy_plane || d->dy_coff + d->G <= d->dy_cs) && (d->x_plane || d->x_coff + d->C <= d->x_cs), "srcgan_wgrad_dense: channel slice exceeds stride");
    SG_REQUIRE(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->dy % 16) == 0, "srcgan_wgrad_dense: tensors must be 16-byte aligned");
    for (int k = 0; k < d->nseg; ++k)
        SG_REQUIRE(d->seg[k].g0 >= 0 && d->seg[k].g1 > d->seg[k].g0 && d->seg[k].g1 <= d->G && d->seg[k].Cin > 0 && d->seg[k].Cin <= d->C,
                   "srcgan_wgrad_dense: bad segment %d", k);
    hipStream_t st = (hipStream_t)stream;
    // row groups of up to 128 gradient channels; per group only the Cin tiles some segment of the group needs
    for (int g_base = 0; g_base < d->G; g_base += 128) {
        const int rows = d->G - g_base < 128 ? d->G - g_base : 128;
        const int mt = cdiv(rows, 32);
        int cin_max = 0, want_bias = 0;
        for (int k = 0; k < d->nseg; ++k)
            if (d->seg[k].g0 < g_base + rows && d->seg[k].g1 > g_base) {
                if (d->seg[k].grad && d->seg[k].Cin > cin_max) cin_max = d->seg[k].Cin;
                if (d->seg[k].bias) want_bias = 1;
            }
        if (cin_max == 0 && !want_bias) continue;
        if (cin_max == 0) cin_max = 1;                       // bias only: one Cin tile carries the pseudo-tap
        WdP p;
        memset(&p, 0, sizeof(p));
        p.dy = d->dy; p.x = d->x; p.slab = d->slab;
        p.B = d->B; p.H = d->H; p.W = d->W; p.G = d->G; p.C = d->C;
        p.dycoff = d->dy_coff; p.xcoff = d->x_coff;
        p.dypix = (long)d->dy_cs * esz; p.dyplane = d->dy_plane ? d->dy_plane : 64; p.xpix = (long)d->x_cs * esz; p.xplane = d->x_plane ? d->x_plane : 64;
        static const bool nt2_only = sg_env("SRCGAN_WD_NT2") != nullptr;
        const int nt = nt2_only ? 2 : wd_nt(mt, cin_max, d->dtype);
        p.g_base = g_base; p.ncit = cdiv(cin_max, 32 * nt); p.want_bias = want_bias;
        for (int b = 0; b < 4; ++b) {
            p.cin_lim[b] = 0;
            for (int k = 0; k < d->nseg; ++k)
                if (d->seg[k].grad && d->seg[k].g0 < g_base + 32 * (b + 1) && d->seg[k].g1 > g_base + 32 * b && d->seg[k].Cin > p.cin_lim[b])
                    p.cin_lim[b] = d->seg[k].Cin;
        }
        p.nsplit = wd_nsplit(mt, nt, p.ncit, d->dtype, d->B, d->H, d->W);
        int ns = 0;
        if (d->dtype == SRCGAN_F32) SG_TRY(dispatch_wd<float>(p, mt, nt, st, ns));
        else if (d->dtype == SRCGAN_F16) SG_TRY(dispatch_wd<_Float16>(p, mt, nt, st, ns));
        else SG_TRY(dispatch_wd<__bf16>(p, mt, nt, st, ns));
        WdRedP q;
        memset(&q, 0, sizeof(q));
        q.slab = d->slab; q.nsplit = ns; q.ncit = p.ncit; q.COT = 32 * mt; q.g_base = g_base; q.nseg = d->nseg; q.accumulate = d->accumulate;
        for (int k = 0; k < d->nseg; ++k) { q.seg[k].g0 = d->seg[k].g0; q.seg[k].g1 = d->seg[k].g1; q.seg[k].grad = d->seg[k].grad;
                                            q.seg[k].bias = d->seg[k].bias; q.seg[k].Cin = d->seg[k].Cin; q.seg[k].alpha = d->seg[k].alpha; }
        const long per_split = (long)p.ncit * 10 * q.COT * 32 * nt;
        if (nt == 4) hipLaunchKernelGGL(wgrad_dense_reduce_k<128>, dim3((unsigned)cdivl(per_split, 64)), dim3(256), 0, st, q);
        else if (nt == 3) hipLaunchKernelGGL(wgrad_dense_reduce_k<96>, dim3((unsigned)cdivl(per_split, 64)), dim3(256), 0, st, q);
        else hipLaunchKernelGGL(wgrad_dense_reduce_k<64>, dim3((unsigned)cdivl(per_split, 64)), dim3(256), 0, st, q);
        SG_LAUNCH_CHECK();
    }
    return 0;
}
