// 3x3 stride-1 convolution, the hot kernel of the RRDB trunk (reference src/model/rddb.py:52-58, 345 of
// them per 23-block generator pass, and -- with transposed packs -- all of their dgrads).
//
// Differences from the generic conv_igemm kernel, all aimed at the LDS bottleneck its profile showed
// (LDS array ~80 % busy: ds_write_b128 staging costs 13 cycles/instruction, 3x a read):
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no VGPR staging,
//     no ds_write, ~4 LDS cycles per KiB.
//   * the LDS image is UNPADDED (64 B per pixel / per weight row) and XOR-swizzled: 16-byte slot s of
//     pixel p holds channel-part s ^ ((p >> 2) & 3).  LDS-DMA writes lane-linear, so the swizzle is applied
//     to each lane's global SOURCE address and again on the ds_read_b128 address (both sides or neither).
//     Any 16 consecutive pixels then cover all 16 slots of the 256-B bank row: conflict-free.
//   * one workgroup of 8 waves (2 per SIMD) per CU works on a 16x32-pixel tile with double-buffered
//     LDS: the DMA of chunk c+1 is in flight during the MFMAs of chunk c; one barrier per chunk.
//     (tile twice as large as the generic kernel -> weight traffic per FLOP halves.)
//   * out-of-image / past-Cin pieces are fetched from a 64-byte zero page instead of being branched on.
#include "conv_params.h"
#include <type_traits>
#include <stdlib.h>

static __device__ __attribute__((aligned(64))) unsigned int sg_zero_page[16];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// NLW > 0: wave specialisation -- NLW extra "loader" waves issue every LDS-DMA piece, the NWV MFMA waves never touch
// VMEM inside the loop.  In the DMA-bound regime the issuing wave stalls in VMEM issue until the memory pipeline
// accepts its pieces (measured: DMA-only 100 us + MFMA-only 92 us ran 154 us when the same waves did both).
template <typename T, int MT, int NWV, int NBUF, int PT, int NLW>
__global__ __launch_bounds__((NWV + NLW) * 64) void conv3x3_dma_k(const ConvP p) {
    using D = DT<T>;
    static_assert(NBUF == 2, "persistent pipeline is double-buffered");
    constexpr int TH = PT * NWV, TW = 32, IHT = TH + 2, IWT = TW + 2;
    constexpr int COT = 32 * MT, NTAP = 9;
    constexpr int NHP = IHT * IWT;                       // halo pixels
    constexpr int HPIECES = (NHP * 64 + 1023) / 1024;    // one-KiB DMA pieces (last one partly junk)
    constexpr int WPIECES = NTAP * COT * 64 / 1024;      // 36 (MT=2) / 18 (MT=1)
    constexpr int HBYTES = HPIECES * 1024, WBYTES = WPIECES * 1024, SBYTES = HBYTES + WBYTES;
    constexpr int NIW = NLW ? NLW : NWV;                 // waves that issue DMA
    constexpr int HIT = (HPIECES + NIW - 1) / NIW, WIT = (WPIECES + NIW - 1) / NIW;
    constexpr int ERS = COT * 4 + 16;                    // epilogue transpose row stride
    static_assert(NWV * 32 * ERS <= SBYTES, "epilogue transpose space must fit one stage");
    extern __shared__ __attribute__((aligned(1024))) char smem[];    // [stage0: halo | weights][stage1: halo | weights]

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: role branches stay scalar
    const bool loader = NLW > 0 && wave >= NWV;
    const int iw = NLW ? wave - NWV : wave;              // index among the issuing waves
    // Persistent workgroups.  Units (spatial tile x Cout tile) are dealt so that the workgroups of one XCD
    // (blockIdx % 8) walk one contiguous range together: neighbouring halos / the Cout tiles of a tile share an L2.
    const int nunits = p.tiles_x * p.tiles_y * p.B * p.ctiles;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, gw = (gridDim.x + 7 - xcd) >> 3;   // my index / #WGs in my XCD group
    const int u8 = (nunits + 7) >> 3;
    const int u_lo = xcd * u8, u_hi = (u_lo + u8 < nunits) ? u_lo + u8 : nunits;

    const char* zp = (const char*)sg_zero_page;
    const int wsw = (r >> 2) & 3;

    // ---- DMA descriptors of the unit being fetched.  Piece pi (1 KiB) = LDS slots [pi*64, pi*64+64); slot q -> pixel
    // q>>2, physical part q&3, which holds logical channel-part (q&3) ^ ((pixel>>2)&3).
    int h_goff[HIT], h_part[HIT], w_off[WIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int q = (it * NIW + iw) * 64 + lane, lp = q >> 2;
        h_part[it] = (q & 3) ^ ((lp >> 2) & 3);
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int q = (it * NIW + iw) * 64 + lane, wr = q >> 2;
        w_off[it] = wr * 64 + (((q & 3) ^ ((wr >> 2) & 3)) * 16);
    }
    const char* f_xb = nullptr; const char* f_wb = nullptr;      // fetch-side base pointers
    auto setup_fetch = [&](int u, int& ob, int& oct, int& ooy0, int& oox0) {
        const int ct = u % p.ctiles; int t = u / p.ctiles;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
        ob = b; oct = ct; ooy0 = ty * TH; oox0 = tx * TW;
        const int gy0 = ooy0 - p.pad_y, gx0 = oox0 - p.pad_x;
        f_xb = (const char*)p.x + (size_t)b * p.H * p.W * p.xpix;
        f_wb = (const char*)p.wp + (size_t)ct * p.nchunk * NTAP * COT * 64;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int q = (it * NIW + iw) * 64 + lane, lp = q >> 2;
            const int iy = lp / IWT, ix = lp - iy * IWT;
            const int gy = gy0 + iy, gx = gx0 + ix;
            const bool ok = lp < NHP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            h_goff[it] = ok ? (gy * p.W + gx) * (int)p.xpix : -1;
        }
    };
    auto issue = [&](int c, int stage) {
        char* lh = smem + stage * SBYTES;
        char* lw = lh + HBYTES;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int pi = it * NIW + iw;
            if (HIT * NIW == HPIECES || pi < HPIECES) {        // wave-uniform
                const int chl = c * D::KCE + h_part[it] * D::EPP;          // channel within the conv's input slice
                const bool ok = h_goff[it] >= 0 && chl < p.Cin;
                const char* src = ok ? f_xb + h_goff[it] + chan_off<T>(p.xcoff + chl, p.xplane) : zp;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lh + pi * 1024), 16, 0, 0);
            }
        }
        const char* ws = f_wb + (size_t)c * NTAP * COT * 64;
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int pi = it * NIW + iw;
            if (WIT * NIW == WPIECES || pi < WPIECES)
                __builtin_amdgcn_global_load_lds((gptr_t)(ws + w_off[it]), (lptr_t)(lw + pi * 1024), 16, 0, 0);
        }
    };

    int u = u_lo + jw;
    if (u >= u_hi) return;
    int cb, cct, coy0, cox0;            // unit being computed
    int nb_, nct, noy0, nox0;           // unit being fetched
    setup_fetch(u, nb_, nct, noy0, nox0);
    if (NLW == 0 || loader) issue(0, 0);
    int stage = 0;
    for (; u < u_hi; u += gw) {
        cb = nb_; cct = nct; coy0 = noy0; cox0 = nox0;
        f32x16 acc[MT][PT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < PT; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

        for (int c = 0; c < p.nchunk; ++c, stage ^= 1) {
            __syncthreads();        // own DMAs into `stage` landed (vmcnt(0)); everyone is done with the other stage
            if (!SG_DBG(p, 2)) {
                if (c + 1 < p.nchunk) { if (NLW == 0 || loader) issue(c + 1, stage ^ 1); }
                else if (u + gw < u_hi) { setup_fetch(u + gw, nb_, nct, noy0, nox0); if (NLW == 0 || loader) issue(0, stage ^ 1); }   // next unit's first chunk
            }
            if (loader) continue;
            const char* lh = smem + stage * SBYTES;
            const char* lw = lh + HBYTES;
            if (SG_DBG(p, 1)) continue;
            // 18 k-steps (9 taps x 2 halves of the 64-byte chunk), software-pipelined: the fragments of step s+1 are
            // requested from LDS before the MFMAs of step s issue, so LDS latency hides under the matrix pipe.
            using frag_t = typename std::conditional<std::is_same<T, float>::value, f32x4, bf16x8>::type;
            frag_t fa[2][MT], fb[2][PT];
            auto load_step = [&](int s, frag_t (&a)[MT], frag_t (&bq)[PT]) {
                const int tap = s >> 1, ks = s & 1, ky = tap / 3, kx = tap - ky * 3;
                const int kp = ks * 2 + h;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[m] = *(const frag_t*)(lw + (tap * COT + m * 32 + r) * 64 + ((kp ^ wsw) * 16));
#pragma unroll
                for (int q = 0; q < PT; ++q) {
                    const int lp = (wave * PT + q + ky) * IWT + r + kx;
                    bq[q] = *(const frag_t*)(lh + lp * 64 + ((kp ^ ((lp >> 2) & 3)) * 16));
                }
            };
            load_step(0, fa[0], fb[0]);
            const bool skip_rd = SG_DBG(p, 16), skip_mm = SG_DBG(p, 8);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s + 1 < 18 && !skip_rd) load_step(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);      // keep the next step's ds_reads ahead of this step's MFMAs
                if (skip_mm) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) asm volatile("" :: "v"(fa[s & 1][m]));
#pragma unroll
                    for (int q = 0; q < PT; ++q) asm volatile("" :: "v"(fb[s & 1][q]));
                } else if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int q = 0; q < PT; ++q)
                                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][m][j], fb[s & 1][q][j], acc[m][q], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < PT; ++q)
                            acc[m][q] = sg_mfma16<T>(fa[s & 1][m], fb[s & 1][q], acc[m][q]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SG_DBG(p, 4)) continue;
        // ---- epilogue of the finished unit; the next unit's first chunk is already in flight into `stage`
        // (the loop increment flipped it), so the transpose space is the OTHER stage = the one just computed from.
        if (p.buf16) {
            __syncthreads();      // every wave finished reading the last chunk before its stage is reused (loaders join)
            if (loader) continue;
            char* tsp = smem + (stage ^ 1) * SBYTES + wave * (32 * ERS);
#pragma unroll
            for (int q = 0; q < PT; ++q)
                conv_epilogue_lds_row<T, MT, PT>(p, acc, q, tsp, cb, cct, coy0 + wave * PT + q, cox0, lane);
        } else if (!loader) {
            conv_epilogue<T, MT, PT>(p, acc, cb, cct, coy0 + wave * PT, cox0, r, h);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Loader-specialised variant: 8 MFMA waves + 4 loader waves in DISJOINT code regions (separate loops with matching
// barrier counts), so the register allocation is max(loader, compute) instead of their union, and the DMA stream is
// issued by waves that do nothing else.  Diagnostic that motivated it: issuing the next stage's DMA *after* the
// MFMAs instead of before them did not change the run time -- the DMA issued by an MFMA wave does not overlap that
// wave's compute.
// Workgroup barriers of the loader-specialised kernel.  __syncthreads() waits for vmcnt(0) AND lgkmcnt(0) in every wave:
// for an MFMA wave that is the acknowledgement of its epilogue's global stores (~2 us, once per unit, traced), for a
// loader wave at the pre-epilogue barrier it is the landing of the NEXT unit's first stage.  Each role waits only for
// what the LDS hand-over needs: MFMA waves for their own LDS reads/writes, loader waves for their own DMA.
__device__ __forceinline__ void sg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void sg_barrier_dma() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void sg_barrier_raw() { asm volatile("s_barrier" ::: "memory"); }

// 16 B per lane, global -> LDS, through a raw buffer descriptor {base, num_records = nrec bytes}: lanes whose byte
// offset is outside [0, nrec) write zeros (hardware range check).  `base`/`nrec`/`lds` must be wave-uniform.
// AUX = cache policy bits of the instruction (gfx94x/95x: 1 = sc0, 2 = nt, 16 = sc1).
template <int AUX = 0>
__device__ __forceinline__ void dma_buf16(const void* base, int nrec, int voff, lptr_t lds, int soff = 0) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, voff, soff, 0, AUX);
}
#ifndef SG_HALO_AUX
#define SG_HALO_AUX 0
#endif

// WRES: the packed weights of ALL K chunks stay resident in LDS (one DMA per workgroup at kernel start) and a stage
// holds only the halo tile.  For Cout <= 32 the stage traffic, not the matrix pipe, bounds the chunk period (traced:
// 2.6 us per chunk against 1.2 us of MFMA), and the weights are 18 of the 57 KiB a stage moves.  Needs ctiles == 1 and
// nchunk * 18 KiB + two halo stages within 160 KiB (Cin <= 128 in bf16).
// DIR: the epilogue form, chosen on the HOST (launch_ls): 0 = generic per-element form (any alignment), 1 = direct from the accumulator
// layout with 8-byte stores, 2 = direct with 16-byte lane-pair stores, 3 = LDS-transposed rows (every operand 16-byte accessible: p.vec16).  As a run-time branch inside one kernel both epilogues shared a register allocation
// and the compiler's pending-load state of the unused one (exec-masked mask loads) put `s_waitcnt vmcnt(0)` at the head of every chunk
// loop of the gradient-slice class: on gfx9 vmcnt counts stores, so each unit's first chunk waited for the previous unit's output stores.
template <typename T, int MT, int NLW, bool WRES, int EM, int NSTG = 2, int PT = 2, int DIR = 0>
__global__ __launch_bounds__((16 / PT + NLW) * 64) void conv3x3_ls_k(const ConvP p) {
    using D = DT<T>;
    constexpr int NWV = 16 / PT;      // MFMA waves: PT output rows each, 16 rows per unit
    constexpr int TH = PT * NWV, TW = 32, IHT = TH + 2, IWT = TW + 2;
    constexpr int COT = 32 * MT, NTAP = 9;
    constexpr int NHP = IHT * IWT;
    constexpr int HPIECES = (NHP * 64 + 1023) / 1024, WPIECES = NTAP * COT * 64 / 1024;
    constexpr int HIT = (HPIECES + NLW - 1) / NLW, WIT = (WPIECES + NLW - 1) / NLW;
    // every loader wave issues HIT + WIT DMAs per chunk with no per-piece branch: the LDS regions are rounded up to
    // whole rounds of NLW pieces and the surplus pieces carry out-of-range offsets (zero fill, no memory traffic)
    constexpr int HBYTES = HIT * NLW * 1024, WBYTES = WRES ? 0 : WIT * NLW * 1024, SBYTES = HBYTES + WBYTES;
    constexpr int WCH = NTAP * COT * 64;                         // packed weight bytes per K chunk
    // NSTG stages: the loaders keep NSTG - 1 of them in flight.  With 2 stages a chunk period is one whole landing (DMA issue +
    // memory latency, 0.9 us with an idle chip, 1.5 us beside the MFMA stream and the epilogue stores), not hidden behind
    // anything: loads, MFMAs and epilogue of a unit ran back to back (49 us = 29 + 10 + 13 for 64->32 at the bench size).  A
    // third stage fits beside resident weights for Cin = 64 (3 x 40 + 36 + 4 KiB = 160 KiB) and overlaps two landings.
    static_assert(NSTG == 2 || NSTG == 3, "stage count");
    const int wres_bytes = WRES ? p.nchunk * WCH : 0;            // resident weights sit behind the stages
    const int bias_off = NSTG * SBYTES + wres_bytes;
    constexpr int ERS = COT * 4 + 16;
    // dense-block convolutions: epilogue straight from the accumulator layout (conv_epilogue_direct32), no scratch, no barrier
    constexpr bool DIRECT = DIR == 1 || DIR == 2;
    static_assert(!DIRECT || (MT == 1 && sizeof(T) == 2 && (EM == 0 || EM == 8 || EM == 16)), "direct epilogue: 16-bit dense-block convolutions");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nunits = p.tiles_x * p.tiles_y * p.B * p.ctiles;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, gw = (gridDim.x + 7 - xcd) >> 3;
    const int u8 = (nunits + 7) >> 3;
    const int u_lo = xcd * u8, u_hi = (u_lo + u8 < nunits) ? u_lo + u8 : nunits;
    const int u0 = u_lo + jw;
    if (u0 >= u_hi) return;
    // bias (zero-padded to ctiles * COT) staged behind the two stages; first read after >= 1 workgroup barrier
    for (int i = tid; i < p.ctiles * COT; i += (NWV + NLW) * 64)
        ((float*)(smem + bias_off))[i] = (p.bias && i < p.Cout) ? p.bias[i] : 0.f;

    if (wave >= NWV) {
        // ================================================================== loader waves
        // Steady state = scalar ALU + DMA only.  Measured (scripts/hip/overlap_test.hip): beside two MFMA waves per
        // SIMD a loader wave that needs VALU for its addresses is starved to 1/16 of its stream rate, whatever its
        // s_setprio; a loader issuing `buffer_load_dwordx4 voff, rsrc, 0 offen lds` from precomputed per-lane offsets
        // keeps the full 6.4 TB/s.  So:
        //  * per-lane byte offsets relative to the tile origin are computed ONCE, for the 3x3 tile classes
        //    {first, interior, last} row x column of tiles; lanes outside the image (or past Cin) hold an offset beyond
        //    num_records and the buffer range check zero-fills them;
        //  * the tile origin, the K-chunk advance and the unit decode (ct, tx, ty, b advance by a fixed step with
        //    carries) are scalar.
        const int iw = wave - NWV;
        constexpr int OOB = 0x7fffffff;
        const long cstride = p.xplane ? p.xplane : 64;                   // bytes between consecutive K chunks
        const int nrec = (IHT * p.W + IWT) * (int)p.xpix;                 // bound on a halo tile's offsets from its origin
        const int tail = p.Cin - (p.nchunk - 1) * D::KCE;                 // channels in the last chunk
        int wv[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int pi = it * NLW + iw, q = pi * 64 + lane, wr = q >> 2;
            wv[it] = pi < WPIECES ? wr * 64 + (((q & 3) ^ ((wr >> 2) & 3)) * 16) : OOB;
        }
        // nine separately named tables (an array of tables gets indexed dynamically by the tile class -> scratch)
        int tab0[HIT], tab1[HIT], tab2[HIT], tab3[HIT], tab4[HIT], tab5[HIT], tab6[HIT], tab7[HIT], tab8[HIT];
        auto fill = [&](int (&tab)[HIT], int cy, int cx) __attribute__((always_inline)) {
            const int ty = cy == 0 ? 0 : cy == 2 ? p.tiles_y - 1 : 1, tx = cx == 0 ? 0 : cx == 2 ? p.tiles_x - 1 : 1;
            const int gy0 = ty * TH - p.pad_y, gx0 = tx * TW - p.pad_x;
#pragma unroll
            for (int it = 0; it < HIT; ++it) {
                const int q = (it * NLW + iw) * 64 + lane, lp = q >> 2;
                const int iy = lp / IWT, ix = lp - iy * IWT;
                const int sw = (q & 3) ^ ((lp >> 2) & 3);
                const bool ok = lp < NHP && (unsigned)(gy0 + iy) < (unsigned)p.H && (unsigned)(gx0 + ix) < (unsigned)p.W && sw * D::EPP < tail;
                tab[it] = ok ? (iy * p.W + ix) * (int)p.xpix + sw * 16 : OOB;
            }
        };
        fill(tab0, 0, 0); fill(tab1, 0, 1); fill(tab2, 0, 2); fill(tab3, 1, 0); fill(tab4, 1, 1);
        fill(tab5, 1, 2); fill(tab6, 2, 0); fill(tab7, 2, 1); fill(tab8, 2, 2);
        // unit decode, advanced by gw units at a time with scalar carries
        int uct = u0 % p.ctiles, utx, uty, ub;
        { int t = u0 / p.ctiles; utx = t % p.tiles_x; t /= p.tiles_x; uty = t % p.tiles_y; ub = t / p.tiles_y; }
        int dct = gw % p.ctiles, dtx, dty, db;
        { int t = gw / p.ctiles; dtx = t % p.tiles_x; t /= p.tiles_x; dty = t % p.tiles_y; db = t / p.tiles_y; }
        uct = __builtin_amdgcn_readfirstlane(uct); utx = __builtin_amdgcn_readfirstlane(utx);
        uty = __builtin_amdgcn_readfirstlane(uty); ub = __builtin_amdgcn_readfirstlane(ub);
        dct = __builtin_amdgcn_readfirstlane(dct); dtx = __builtin_amdgcn_readfirstlane(dtx);
        dty = __builtin_amdgcn_readfirstlane(dty); db = __builtin_amdgcn_readfirstlane(db);
        auto advance_st = [&](int& ct_, int& tx_, int& ty_, int& b_) __attribute__((always_inline)) {
            ct_ += dct; if (ct_ >= p.ctiles) { ct_ -= p.ctiles; ++tx_; }
            tx_ += dtx; if (tx_ >= p.tiles_x) { tx_ -= p.tiles_x; ++ty_; }
            ty_ += dty; if (ty_ >= p.tiles_y) { ty_ -= p.tiles_y; ++b_; }
            b_ += db;
        };
        auto advance = [&]() { advance_st(uct, utx, uty, ub); };
        // L2 prefetch pointer (SG_PF): one chunk ahead of the issue pointer.  A stage's landing takes 2.3-2.7 us beside the MFMA
        // waves (HBM latency under load) and only NSTG - 1 stages can be in flight per CU -- 40 KiB / 2.5 us = 16 GB/s per CU =
        // 4.1 TB/s chip-wide, the rate the Cout = 32 convolutions run at.  A 4-byte-per-lane LDS-DMA of the SAME per-lane offsets
        // into a 256-byte dummy region touches every 64-byte line of the chunk after the one being issued: its HBM latency
        // overlaps the current landings, and the real DMA of that chunk then reads L2.
        // MEASURED (round 2, scripts/microbench_conv.py, blocked layout): with the prefetch 64->32 50.8 -> 53.8 us, 128->32 83.5 -> 88.4,
        // 160->32 111 -> 118, 192->64 239 -> 246: 3-7 % SLOWER.  More bytes in flight towards HBM do not help: the kernel's skeleton
        // (scripts/hip/ingest_test.hip) overlaps DMA, fragment reads and MFMAs fully with two stages -- what costs is the part's
        // power limit and the output stores, not memory latency.  Off by default.
#ifndef SG_PF
#define SG_PF 0
#endif
        int pct = uct, ptx = utx, pty = uty, pb = ub;
        const char* xb0 = (const char*)p.x + chan_off<T>(p.xcoff, p.xplane);
        char* const pf_dummy = smem + bias_off + ((p.ctiles * COT * 4 + 255) & ~255);
        auto prefetch = [&](int c) __attribute__((always_inline)) {                       // chunk c of the unit (pct, ptx, pty, pb): lines -> L2
            const long org = ((long)(p.rev ? p.B - 1 - pb : pb) * p.H + (pty * TH - p.pad_y)) * p.W + (ptx * TW - p.pad_x);
            const char* bx = xb0 + org * p.xpix + c * cstride;
            const int cls = (pty == 0 ? 0 : pty == p.tiles_y - 1 ? 2 : 1) * 3 + (ptx == 0 ? 0 : ptx == p.tiles_x - 1 ? 2 : 1);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)bx, 0, nrec, 0x00020000);
#define SG_PF_CLASS(K)                                                                                \
            case K:                                                                                   \
                _Pragma("unroll") for (int it = 0; it < HIT; ++it)                                     \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)pf_dummy, 4, tab##K[it], 0, 0, 0); \
                asm volatile("; prefetch class " #K);                                                 \
                break;
            switch (cls) {
                SG_PF_CLASS(0) SG_PF_CLASS(1) SG_PF_CLASS(2) SG_PF_CLASS(3) SG_PF_CLASS(4)
                SG_PF_CLASS(5) SG_PF_CLASS(6) SG_PF_CLASS(7) SG_PF_CLASS(8)
            }
#undef SG_PF_CLASS
        };
        auto issue = [&](int c, int stage) __attribute__((always_inline)) {             // chunk c of the unit (uct, utx, uty, ub)
            char* lh = smem + stage * SBYTES;
            char* lw = lh + HBYTES;
            const long org = ((long)(p.rev ? p.B - 1 - ub : ub) * p.H + (uty * TH - p.pad_y)) * p.W + (utx * TW - p.pad_x);     // tile origin, pixels
            const char* bx = xb0 + org * p.xpix + c * cstride;
            const char* bw = (const char*)p.wp + ((size_t)uct * p.nchunk + c) * NTAP * COT * 64;
            const int cls = (uty == 0 ? 0 : uty == p.tiles_y - 1 ? 2 : 1) * 3 + (utx == 0 ? 0 : utx == p.tiles_x - 1 ? 2 : 1);
#ifndef SG_EXP_NO_HALO_DMA       // timing experiment: skip the halo stream
#define SG_ISSUE_CLASS(K)                                                                             \
            case K:                                                                                   \
                _Pragma("unroll") for (int it = 0; it < HIT; ++it)                                     \
                    dma_buf16<SG_HALO_AUX>(bx, nrec, tab##K[it], (lptr_t)(lh + (it * NLW + iw) * 1024));            \
                asm volatile("; tile class " #K);   /* a distinct tail: cases merged by code sinking index the tables in scratch */ \
                break;
            switch (cls) {
                SG_ISSUE_CLASS(0) SG_ISSUE_CLASS(1) SG_ISSUE_CLASS(2) SG_ISSUE_CLASS(3) SG_ISSUE_CLASS(4)
                SG_ISSUE_CLASS(5) SG_ISSUE_CLASS(6) SG_ISSUE_CLASS(7) SG_ISSUE_CLASS(8)
            }
#undef SG_ISSUE_CLASS
#endif
#ifndef SG_EXP_NO_WEIGHT_DMA     // timing experiment: skip the per-chunk weight stream
            if constexpr (!WRES) {
#pragma unroll
                for (int it = 0; it < WIT; ++it)
                    dma_buf16(bw, NTAP * COT * 64, wv[it], (lptr_t)(lw + (it * NLW + iw) * 1024));
            }
#endif
        };
        if constexpr (WRES) {
            // all chunks' weights, once: piece j = 1 KiB = 16 packed rows; lane offset is piece-independent (16 | rows per piece)
            const int lo = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
            for (int j = iw; j < p.nchunk * (WCH / 1024); j += NLW)
                dma_buf16(p.wp, wres_bytes, lo, (lptr_t)(smem + NSTG * SBYTES + j * 1024), j * 1024);
        }
        // (ac, au): the next chunk to issue, NSTG - 1 chunks ahead of the one the MFMA waves are about to consume; the unit
        // decode (uct, utx, uty, ub) belongs to it
        int ac = 0, au = u0, astage = 0;
        auto issue_next = [&]() __attribute__((always_inline)) {        // false when the workgroup's chunks are exhausted
            if (au >= u_hi) return false;
            issue(ac, astage);
            astage = astage + 1 == NSTG ? 0 : astage + 1;
            if (++ac == p.nchunk) { ac = 0; au += gw; if (au < u_hi) advance(); }
            return true;
        };
        // prefetch pointer: (pc, pu) with its own unit decode, always one chunk ahead of (ac, au)
        int pc = 0, pu = u0;
        auto pf_step = [&]() __attribute__((always_inline)) { if (++pc == p.nchunk) { pc = 0; pu += gw; if (pu < u_hi) advance_st(pct, ptx, pty, pb); } };
        auto pf_next = [&]() __attribute__((always_inline)) {          // false when nothing is left to prefetch
            if (!SG_PF || pu >= u_hi) return false;
            prefetch(pc);
            pf_step();
            return true;
        };
        // prologue in the steady-state order -- DMA(0), PF(1), [DMA(1), PF(2)] -- so that the counted wait below holds from the
        // first chunk on: behind the pieces of chunk c come HIT prefetch pieces per later stage and the later stages' own pieces
        bool ahead = issue_next(), pfd = false;
        pf_step();
#pragma unroll
        for (int k = 1; k < NSTG - 1; ++k) { pfd = pf_next(); ahead = issue_next(); }
        pfd = pf_next();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my part of the LDS bias copy
        [[maybe_unused]] int trk = 0;
        for (int u = u0; u < u_hi; u += gw) {
            for (int c = 0; c < p.nchunk; ++c) {
#ifdef SG_TRACE
                if (NSTG == 3 && ahead) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HIT) : "memory");
                else __builtin_amdgcn_s_waitcnt(0x0f70);     // my DMA pieces of this chunk landed
                const unsigned long long t_land = __builtin_amdgcn_s_memrealtime();
#endif
                // my pieces of this chunk's stage landed; MFMA waves left the stage that is issued next.  With 3 stages the
                // newest stage (HIT pieces per wave, issued last, completing in order) may still be in flight.
                // (vmcnt retires in order: what may stay in flight is everything issued AFTER this chunk's pieces -- per later
                //  stage its HIT(+WIT) pieces and the HIT prefetch pieces that followed it)
                if (SG_PF) {
                    // pfd: the prefetch issued last exists, hence every piece of the steady-state pattern behind chunk c does
                    if (pfd) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NSTG == 3 ? 3 * HIT + WIT * (WRES ? 0 : 1) : HIT) : "memory");
                    else sg_barrier_dma();                               // tail of the stream: wait for everything
                } else if (NSTG == 3 && ahead) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(HIT) : "memory");
                else sg_barrier_dma();
#ifdef SG_TRACE
                const unsigned long long t_bar = __builtin_amdgcn_s_memrealtime();
#endif
                if (SG_DBG(p, 2)) continue;
                ahead = issue_next();
                pfd = pf_next();
#ifdef SG_TRACE
                if (p.trace && blockIdx.x == 8 && iw == 0 && lane == 0 && trk < 60) {
                    p.trace[trk * 8 + 0] = t_land; p.trace[trk * 8 + 1] = t_bar; p.trace[trk * 8 + 2] = __builtin_amdgcn_s_memrealtime(); ++trk;
                }
#endif
            }
            if constexpr (DIR == 3) sg_barrier_raw();       // matches the MFMA waves' pre-epilogue barrier; the next stage keeps flying
        }
        return;
    }

    // ====================================================================== MFMA waves
    // Row-ordered schedule.  A wave owns PT output rows x 32 pixels x COT channels.  Per K chunk and k-half `ks` it walks
    // the PT+2 input rows i and the 3 horizontal taps kx: ONE pixel fragment B(i,kx) feeds every (output row q, ky)
    // pair with q + ky == i, so the pixel operand is read (PT+2)*3 times per k-half instead of PT*9 -- the LDS read
    // stream, not the matrix pipe, bounds this loop (measured: dropping 1 of 6 reads speeds it up by the same 1/6).
    // Weight fragments A(ky,kx) live in a 6-slot register ring (each is used by two groups three apart).
    // LDS->register prefetch runs PD groups ahead, across k-halves, and is drained only at the chunk barrier.
    const int r = lane & 31, h = lane >> 5;
    using frag_t = typename std::conditional<std::is_same<T, float>::value, f32x4, bf16x8>::type;
#ifndef SG_PD
#define SG_PD 2
#endif
    constexpr int PD = SG_PD, NRB = PD + 1, NRA = 3 * (PT - 1) + 1 + PD;      // a tap's fragment is live from its first group to 3 (PT - 1) groups later
    constexpr int NGRP = (PT + 2) * 3, NG2 = 2 * NGRP;               // groups per k-half / per chunk
    static_assert((PT == 2 || PT == 4) && PD >= 1 && PD <= 5, "A-ring sizing");
    // byte offsets inside a stage (stage 0), k-half 0; k-half 1 is the same address with bit 5 flipped (slot ^ 2)
    int pb[NGRP], pa;
    int L0 = wave * PT * IWT + r;
    auto calc_pb = [&]() {
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int lp = L0 + (g / 3) * IWT + (g % 3);
            pb[g] = lp * 64 + ((h ^ ((lp >> 2) & 3)) * 16);
        }
    };
    // 64-row tiles with residual operands: the epilogue holds those operands in registers from the pre-epilogue barrier on (64 VGPRs
    // beside the 64 accumulators, 168 in all), so the 12 fragment offsets are rebuilt per chunk (~40 VALU beside 216 MFMAs) instead
    // of living across the epilogue
    constexpr bool PB_PER_CHUNK = MT == 2 && DIR == 3 && (EM & 2) != 0;      // (one operand: 164 VGPRs without)
    if constexpr (!PB_PER_CHUNK) calc_pb();
    pa = (WRES ? NSTG * SBYTES : HBYTES) + r * 64 + ((h ^ ((r >> 2) & 3)) * 16);      // resident weights: + c * WCH, stage-independent
    int stage = 0;
    [[maybe_unused]] int trk = 0;
    for (int u = u0; u < u_hi; u += gw) {
        const int cct = u % p.ctiles; int t = u / p.ctiles;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; const int cb = p.rev ? p.B - 1 - t / p.tiles_y : t / p.tiles_y;
        const int coy0 = ty * TH, cox0 = tx * TW;
        f32x16 acc[MT][PT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < PT; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
#ifndef SG_SIGN_EARLY
#define SG_SIGN_EARLY 1
#endif
        [[maybe_unused]] unsigned sgpre[PT];
        if constexpr (DIRECT && (EM & 8) != 0 && SG_SIGN_EARLY) conv_direct32_sign_request<PT>(p, sgpre, cb, coy0 + wave * PT, cox0, lane);
        for (int c = 0; c < p.nchunk; ++c, stage = (stage + 1 == NSTG ? 0 : stage + 1)) {
#ifdef SG_TRACE
            const unsigned long long t_arr = __builtin_amdgcn_s_memrealtime();
#endif
            sg_barrier_lds();
#ifdef SG_TRACE
            if (p.trace && blockIdx.x == 8 && wave == 0 && lane == 0 && trk < 60) {
                p.trace[trk * 8 + 4] = t_arr; p.trace[trk * 8 + 5] = __builtin_amdgcn_s_memrealtime(); ++trk;
            }
#endif
            if (SG_DBG(p, 1)) continue;
            if constexpr (PB_PER_CHUNK) { asm volatile("" : "+v"(L0)); calc_pb(); }
            const char* ls = smem + stage * SBYTES;
            const char* lsw = WRES ? smem + c * WCH : ls;
            frag_t fa[NRA][MT], fb[NRB];
            // flat group index G = ks * NGRP + i * 3 + kx
            auto read_b = [&](int G) {
                const int ks = G / NGRP, g = G % NGRP;
#ifdef SG_EXP_B_KX0_ONLY       // timing experiment (wrong results): only the kx = 0 pixel fragments are read -- what cross-lane
                if (g % 3 != 0) return;      // shifts instead of the kx = 1, 2 reads could gain at most
#endif
#ifdef SG_EXP_NO_FRAG_READS    // timing experiment: MFMAs on stale registers, no LDS fragment reads at all
                return;
#endif
                fb[G % NRB] = *(const frag_t*)(ls + (pb[g] ^ (ks * 32)));
            };
            auto read_a = [&](int G) {                   // the weight tap first used by group G (none for the last input row)
                const int ks = G / NGRP, g = G % NGRP;
#ifdef SG_EXP_NO_FRAG_READS
                return;
#endif
                if (g < 9) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        fa[g % NRA][m] = *(const frag_t*)(lsw + (pa ^ (ks * 32)) + (g * COT + m * 32) * 64);
                }
            };
#pragma unroll
            for (int G = 0; G < PD; ++G) { read_a(G); read_b(G); }
#pragma unroll
            for (int G = 0; G < NG2; ++G) {
                if (G + PD < NG2) { read_a(G + PD); read_b(G + PD); }
                __builtin_amdgcn_sched_barrier(0);
                const int g = G % NGRP, i = g / 3, kx = g % 3;
#pragma unroll
                for (int q = 0; q < PT; ++q) {
                    const int ky = i - q;
                    if (ky < 0 || ky > 2) continue;
                    const int ta = (ky * 3 + kx) % NRA;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ta][m][jj], fb[G % NRB][jj], acc[m][q], 0, 0, 0);
                        } else {
#ifdef SG_EXP_NO_MFMA_INSTR    // timing experiment: fragment reads kept alive, no MFMA instructions
                            asm volatile("" :: "v"(fa[ta][m]), "v"(fb[G % NRB]));
#else
                            acc[m][q] = sg_mfma16<T>(fa[ta][m], fb[G % NRB], acc[m][q]);
#endif
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (DIRECT) {
            if (!SG_DBG(p, 4)) {
                constexpr bool PRE = (EM & 8) != 0 && SG_SIGN_EARLY;
                conv_epilogue_direct32<T, PT, EM, DIR == 2, PRE>(p, acc, smem + bias_off, cb, coy0 + wave * PT, cox0, lane, sgpre);
            }
        } else if constexpr (DIR == 3) {
#ifdef SG_TRACE
            const unsigned long long t_pre = __builtin_amdgcn_s_memrealtime();
#endif
            ConvRowsPre<T, MT, PT, EM> pre;
            if (!SG_DBG(p, 4)) conv_lds_rows_request<T, MT, PT, EM>(p, pre, cb, cct, coy0 + wave * PT, cox0, lane);      // in flight across the barrier
            sg_barrier_lds();
#ifdef SG_TRACE
            const unsigned long long t_eb = __builtin_amdgcn_s_memrealtime();
#endif
            char* tsp = smem + (stage == 0 ? NSTG - 1 : stage - 1) * SBYTES + wave * (32 * ERS);      // the stage just consumed: refilled only after the next chunk barrier
            if (!SG_DBG(p, 4))
                conv_epilogue_lds_rows<T, MT, PT, EM>(p, acc, tsp, smem + bias_off, cb, cct, coy0 + wave * PT, cox0, lane, pre);
#ifdef SG_TRACE
            if (p.trace && blockIdx.x == 8 && wave == 0 && lane == 0 && trk <= 60) {
                p.trace[(trk - 1) * 8 + 3] = t_pre; p.trace[(trk - 1) * 8 + 6] = t_eb; p.trace[(trk - 1) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            }
#endif
        } else {
            conv_epilogue<T, MT, PT>(p, acc, cb, cct, coy0 + wave * PT, cox0, r, h);
        }
    }
}

template <typename T, int MT, int NLW, bool WRES, int EM, int NSTG, int PT, int DIR>
static int launch_ls_dir(const ConvP& p, int ctiles, hipStream_t st);

#ifndef SG_DIRECT_X16
#define SG_DIRECT_X16 1
#endif
// Dense-block convolutions (16-bit, 32 output channels, unscaled unit-stride output, no residual operands) take the direct epilogue.
template <typename T, int MT, int NLW, bool WRES = false, int EM = 7, int NSTG = 2, int PT = 2>
static int launch_ls(const ConvP& p, int ctiles, hipStream_t st) {
#ifndef SG_NO_DIRECT_EPI
    if constexpr (MT == 1 && sizeof(T) == 2 && (EM == 0 || EM == 8 || EM == 16)) {
        const bool direct_ok = p.Cout == 32 && p.os == 1 && p.oa == 0 && p.ob == 0 && p.YH == p.OH && p.YW == p.OW && (p.yplane == 64 ? p.ycoff % 4 == 0 : p.ycoff % 32 == 0) &&
                               (!p.act || (p.slope >= 0.f && p.slope <= 1.f));           // LeakyReLU as max(v, slope v)
        if (direct_ok) return (SG_DIRECT_X16 && p.vec16) ? launch_ls_dir<T, MT, NLW, WRES, EM, NSTG, PT, 2>(p, ctiles, st) : launch_ls_dir<T, MT, NLW, WRES, EM, NSTG, PT, 1>(p, ctiles, st);
    }
#endif
    if (p.vec16) return launch_ls_dir<T, MT, NLW, WRES, EM, NSTG, PT, 3>(p, ctiles, st);
    if constexpr (EM == 8 || EM == 16) {
        SG_REQUIRE(false, "conv3x3: sign masks need 16-byte accessible operands");
        return 1;
    } else
        return launch_ls_dir<T, MT, NLW, WRES, EM, NSTG, PT, 0>(p, ctiles, st);
}

template <typename T, int MT, int NLW, bool WRES, int EM, int NSTG, int PT, int DIR>
static int launch_ls_dir(const ConvP& p, int ctiles, hipStream_t st) {
    constexpr int HB = (((18 * 34 * 64 + 1023) / 1024 + NLW - 1) / NLW) * NLW * 1024, WB = WRES ? 0 : ((9 * 32 * MT * 64 / 1024 + NLW - 1) / NLW) * NLW * 1024;
    constexpr size_t SMEM = WRES ? 160 * 1024 : 2 * ((size_t)HB + (size_t)WB) + 4096 + 256;  // + bias copy (<= 1024 output channels) + prefetch dummy
    static_assert(NSTG == 2 || WRES, "three stages only beside resident weights");
    auto kern = conv3x3_ls_k<T, MT, NLW, WRES, EM, NSTG, PT, DIR>;
    static bool attr_set = false;
    static int ncu = 0;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        int dev = 0; hipDeviceProp_t prop;
        SG_HIP(hipGetDevice(&dev)); SG_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount;
        attr_set = true;
    }
    if constexpr (DIR == 3 && ConvRowsPre<T, MT, PT, EM>::ON) {       // 32-bit per-lane offsets of the residual operands' buffer loads (conv_lds_rows_request)
        const long lim = (1L << 31) - (1L << 20);
        SG_REQUIRE((!p.r1 || p.r1plane < lim) && (!p.r2 || p.r2plane < lim) && p.r1pix < (1 << 16) && p.r2pix < (1 << 16),
                   "conv3x3: a residual operand's channel plane must be below 2 GiB");
    }
    ConvP q = p;
    q.tiles_x = cdiv(p.OW, 32); q.tiles_y = cdiv(p.OH, 16); q.ctiles = ctiles;
    { static const char* e = sg_env("SRCGAN_DBG"); q.dbg = e ? atoi(e) : 0; }
    const size_t nunits = (size_t)q.tiles_x * q.tiles_y * p.B * ctiles;
    size_t nwg = (size_t)ncu; if (nwg > nunits) nwg = nunits;
    if (const char* e = sg_env("SRCGAN_CONV_CUS")) { const int v = atoi(e); if (v > 0 && (size_t)v < nwg) nwg = (size_t)v; }     // diagnostic builds: share the chip with a concurrent kernel
    char cls[96];
    snprintf(cls, sizeof(cls), "conv3x3_ls<%s,MT%d,W%d+%d%s%s,e%d%s>", sizeof(T) == 4 ? "f32" : (__is_same(T, __bf16) ? "bf16" : "f16"), MT, 16 / PT, NLW, WRES ? ",wres" : "", NSTG == 3 ? ",s3" : "", EM,
             DIR == 2 ? ",d16" : DIR == 1 ? ",d8" : DIR == 0 ? ",gen" : "");
    const double px = (double)p.B * p.OH * p.OW;
    const int tok = sg_prof_start(cls, 2.0 * px * 9 * p.Cin * p.Cout, ((double)p.B * p.H * p.W * p.Cin + px * p.Cout) * sizeof(T), st);
#ifdef SG_TRACE
    static unsigned long long* trace = nullptr;
    if (!trace) { SG_HIP(hipMalloc(&trace, 60 * 8 * 8)); }
    SG_HIP(hipMemsetAsync(trace, 0, 60 * 8 * 8, st));
    q.trace = trace;
#else
    q.trace = nullptr;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3((16 / PT + NLW) * 64), SMEM, st, q);
#ifdef SG_TRACE
    {
        static int dumps = 0;
        if (sg_env("SRCGAN_TRACE") && dumps < 3) {
            ++dumps;
            unsigned long long h[60 * 8];
            SG_HIP(hipStreamSynchronize(st));
            SG_HIP(hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[trace] %s Cin=%d Cout=%d nchunk=%d  (10 ns ticks rel. to first barrier: loader land, loader barrier-exit, loader issue-done | mfma arrive, mfma barrier-exit)\n", cls, p.Cin, p.Cout, p.nchunk);
            const unsigned long long t0 = h[1];
            for (int k = 0; k < 60 && h[k * 8 + 1]; ++k)
            {
                fprintf(stderr, "[trace] %2d  L: land %6lld bar %6lld issued %6lld | M: arrive %6lld bar %6lld", k, (long long)(h[k * 8] - t0), (long long)(h[k * 8 + 1] - t0),
                        (long long)(h[k * 8 + 2] - t0), (long long)(h[k * 8 + 4] - t0), (long long)(h[k * 8 + 5] - t0));
                if (h[k * 8 + 3]) fprintf(stderr, " | last-chunk done %6lld epi-barrier %6lld epi-done %6lld", (long long)(h[k * 8 + 3] - t0), (long long)(h[k * 8 + 6] - t0), (long long)(h[k * 8 + 7] - t0));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return 0;
}

template <typename T, int MT, int NWV, int NBUF, int PT, int NLW = 0>
static int launch_dma(const ConvP& p, int ctiles, hipStream_t st) {
    constexpr int TH = PT * NWV;
    constexpr int HB = (((TH + 2) * 34 * 64 + 1023) / 1024) * 1024, WB = 9 * 32 * MT * 64;
    constexpr size_t SMEM = NBUF * ((size_t)HB + (size_t)WB);
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_dma_k<T, MT, NWV, NBUF, PT, NLW>;
    static bool attr_set = false;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set = true;
    }
    ConvP q = p;
    { static const char* e = sg_env("SRCGAN_DBG"); q.dbg = e ? atoi(e) : 0; }
    q.tiles_x = cdiv(p.OW, 32);
    q.tiles_y = cdiv(p.OH, TH);
    q.ctiles = ctiles;
    const size_t nunits = (size_t)q.tiles_x * q.tiles_y * p.B * ctiles;
    static int wg_per_cu = 0, ncu = 0;
    if (!wg_per_cu) {
        int dev = 0; hipDeviceProp_t prop;
        SG_HIP(hipGetDevice(&dev)); SG_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount;
        wg_per_cu = (int)((160 * 1024) / SMEM); if (wg_per_cu < 1) wg_per_cu = 1;
        if (wg_per_cu * (NWV + NLW) > 16) wg_per_cu = 16 / (NWV + NLW) > 0 ? 16 / (NWV + NLW) : 1;
    }
    size_t nwg = (size_t)ncu * wg_per_cu;
    if (nwg > nunits) nwg = nunits;
    dim3 grid((unsigned)nwg, 1, 1);
    char cls[96];
    snprintf(cls, sizeof(cls), "conv3x3_dma<%s,MT%d,W%d+%d,PT%d>", sizeof(T) == 4 ? "f32" : (__is_same(T, __bf16) ? "bf16" : "f16"), MT, NWV, NLW, PT);
    const double px = (double)p.B * p.OH * p.OW;
    const int tok = sg_prof_start(cls, 2.0 * px * 9 * p.Cin * p.Cout, ((double)p.B * p.H * p.W * p.Cin + px * p.Cout) * sizeof(T), st);
    hipLaunchKernelGGL(kern, grid, dim3((NWV + NLW) * 64), SMEM, st, q);
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int dispatch_dma(const ConvP& p, hipStream_t st) {
    static const char* cfg_env = sg_env("SRCGAN_DMA_CFG");
    char cfg = cfg_env ? cfg_env[0] : 'l';     // 'l': 8 MFMA + 4 loader waves (default); 'a': 8 self-loading waves
    // the loader-specialised kernel needs a uniform K-chunk stride and 31-bit per-image offsets
    const bool ls_ok = (p.xplane == 0 || p.xcoff % DT<T>::KCE == 0) && (p.Cin % DT<T>::KCE == 0 || p.nchunk == 1) &&
                       p.pad_y == 1 && p.pad_x == 1 && p.Cout <= 1024 && (18.0 * p.W + 34.0) * (double)p.xpix < 2147483647.0;
    if (p.sgn_in || p.sgn_out)
        SG_REQUIRE(ls_ok && cfg != 'a' && sizeof(T) == 2 && (p.Cout == 32 || (p.Cout == 64 && !p.sgn_out && !p.r1 && !p.r2 && !p.mz && p.vec16)),
                   "conv3x3: sign masks are only handled by the loader-specialised 16-bit kernel (Cout == 32, or Cout == 64 read-only without other operands)");
    if (!ls_ok) cfg = 'a';
    if (p.Cout <= 32) {
        if (cfg == 'a') return launch_dma<T, 1, 8, 2, 2>(p, 1, st);
#ifndef SG_NLW1
#define SG_NLW1 8
#endif
        if constexpr (sizeof(T) == 4) return launch_ls<T, 1, 4>(p, 1, st);      // fp32: 128 VGPRs (16 waves) would spill
        else {
            static const bool no_wres = sg_env("SRCGAN_NO_WRES") != nullptr;
            // resident weights: two 40 KiB halo stages + nchunk * 18 KiB + bias within 160 KiB
            // epilogue operand set: Cout <= 32 convs are the dense-block forward (none) and its gradient slices (mz)
            const int em = (p.r1 ? 1 : 0) | (p.r2 ? 2 : 0) | (p.mz ? 4 : 0) | (p.sgn_in ? 8 : 0) | (p.sgn_out ? 16 : 0);
            SG_REQUIRE(em < 8 || em == 8 || em == 16, "conv3x3: a sign mask cannot be combined with other epilogue operands");
            static const bool no_s3 = sg_env("SRCGAN_NO_S3") != nullptr;
            if (!no_wres && !no_s3 && p.nchunk == 2) {           // Cin = 64: three 40 KiB stages + 36 KiB of weights + bias = 160 KiB
                if (em == 0) return launch_ls<T, 1, SG_NLW1, true, 0, 3>(p, 1, st);
                if (em == 4) return launch_ls<T, 1, SG_NLW1, true, 4, 3>(p, 1, st);
                if (em == 8) return launch_ls<T, 1, SG_NLW1, true, 8, 3>(p, 1, st);
                if (em == 16) return launch_ls<T, 1, SG_NLW1, true, 16, 3>(p, 1, st);
            }
            // (PT = 4: four MFMA waves with 4 rows each read 36 % fewer fragments per MFMA, but one MFMA wave per SIMD does not
            //  keep the matrix pipe fed: 96->32 68 -> 74 us, 128->32 90 -> 98 us.  launch_ls<T, 1, SG_NLW1, true, EM, 2, 4> to retry.)
            if (!no_wres && 2 * 40 * 1024 + p.nchunk * 9 * 32 * 64 + 4096 <= 160 * 1024) {
                if (em == 0) return launch_ls<T, 1, SG_NLW1, true, 0>(p, 1, st);
                if (em == 4) return launch_ls<T, 1, SG_NLW1, true, 4>(p, 1, st);
                if (em == 8) return launch_ls<T, 1, SG_NLW1, true, 8>(p, 1, st);
                if (em == 16) return launch_ls<T, 1, SG_NLW1, true, 16>(p, 1, st);
                return launch_ls<T, 1, SG_NLW1, true, 7>(p, 1, st);
            }
            if (em == 0) return launch_ls<T, 1, SG_NLW1, false, 0>(p, 1, st);
            if (em == 4) return launch_ls<T, 1, SG_NLW1, false, 4>(p, 1, st);
            if (em == 8) return launch_ls<T, 1, SG_NLW1, false, 8>(p, 1, st);
            if (em == 16) return launch_ls<T, 1, SG_NLW1, false, 16>(p, 1, st);
            return launch_ls<T, 1, SG_NLW1, false, 7>(p, 1, st);
        }
    }
    const int ctiles = cdiv(p.Cout, 64);
    if (cfg == 'a') return launch_dma<T, 2, 8, 2, 2>(p, ctiles, st);
    if constexpr (sizeof(T) == 2) {
        const int em = (p.r1 ? 1 : 0) | (p.r2 ? 2 : 0) | (p.mz ? 4 : 0);
        if (p.sgn_in) return launch_ls<T, 2, 4, false, 8>(p, ctiles, st);       // conv_last's input gradient: 8 mask bytes per pixel instead of the 64-channel activation
        if (em == 0) return launch_ls<T, 2, 4, false, 0>(p, ctiles, st);
        if (em == 1) return launch_ls<T, 2, 4, false, 1>(p, ctiles, st);
        if (em == 3) return launch_ls<T, 2, 4, false, 3>(p, ctiles, st);
        if (em == 4) return launch_ls<T, 2, 4, false, 4>(p, ctiles, st);
    }
    return launch_ls<T, 2, 4>(p, ctiles, st);
}

// entry used by srcgan_conv_igemm for kh == kw == 3, stride 1
int sg_conv3x3_dma(const ConvP& p, int dtype, hipStream_t st) {
    if (dtype == SRCGAN_F32) return dispatch_dma<float>(p, st);
    if (dtype == SRCGAN_F16) return dispatch_dma<_Float16>(p, st);
    return dispatch_dma<__bf16>(p, st);
}
