// Implicit-GEMM convolution for gfx950 (MI355X): NHWC, halo tile staged once in LDS and
// reused by every filter tap, MFMA contraction over (tap, channel-chunk), fused epilogue
// (bias, scale, two residuals, LeakyReLU, LeakyReLU' mask, pixel-shuffle store).
//
// Replaces aten::convolution / the dgrad half of aten::convolution_backward behind
//   reference src/model/rddb.py:52-58,63-68 (RDB 3x3 convs + "cat" + 0.2*x5+x),
//   rddb.py:28-38,93-97 (ConvTranspose2d k2 s2 == 4 x [1x1 conv -> strided store]),
//   src/model/model.py:612-634 (PatchGAN 4x4 s2 / s1 convs).
//
// GEMM orientation: D[M = Cout][N = pixels] so that each lane of the 32x32 MFMA result holds
// 4 *consecutive output channels* of one pixel -> 8/16-byte epilogue loads and stores in NHWC.
//   A operand (M x K): packed weights, row = cout, k = cin within the chunk
//   B operand (K x N): LDS halo tile, col = 32 consecutive output pixels of one row
// LDS image: one pixel (or one weight row) = 64 B of channels padded to 80 B -> ds_read_b128
// of 32 consecutive pixels is bank-conflict free (stride 5 slots of 16 B, coprime with 16).
// Both dtypes use the same byte layout: 64 B = 32 bf16 (2 x mfma_32x32x16_bf16 k-steps)
//                                            = 16 f32  (8 x mfma_32x32x2_f32).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "conv_params.h"

// Cost-removal switches of this kernel's hot loop (scripts/microbench_generic.py): only in a -DSG_DIAG -DSG_IG_DIAG variant --
// as run-time branches they stop the compiler from hoisting fragment reads over MFMAs, so even other SG_DIAG variants leave them out.
#if defined(SG_DIAG) && defined(SG_IG_DIAG)
#define IG_DBG(p, bit) ((p).dbg & (bit))
#else
#define IG_DBG(p, bit) 0
#endif


// Occupancy: hipcc takes 188 registers for the 64-accumulator forms (2 workgroups per CU).  Capping them at 168 (3 per CU,
// __launch_bounds__(256, 3)) changed neither the kernels' times nor the training step (same-box A/B, round 2).
template <typename T, int KH, int KW, int S, int MT, int PT, bool WPK>
__global__ __launch_bounds__(256) void conv_igemm_k(const ConvP p) {
    using D = DT<T>;
    constexpr int TH = 4 * PT, TW = 32;
    constexpr int IHT = (TH - 1) * S + KH, IWT = (TW - 1) * S + KW;
    constexpr int COT = 32 * MT, NTAP = KH * KW, PIXB = 80;
    constexpr int WROWS = (WPK ? KW : NTAP) * COT;
    constexpr int NPH = IHT * IWT * 4;     // 16-byte pieces of the halo tile
    constexpr int NPW = WROWS * 4;         // 16-byte pieces of one weight stage
    constexpr int HIT = (NPH + 255) / 256, WIT = (NPW + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_h = smem;
    char* lds_w = smem + IHT * IWT * PIXB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    // XCD-aware block -> tile map: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a contiguous
    // run of tiles so neighbouring tiles' halos and the Cout tiles of one spatial tile hit the same L2.
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    // npar == 4 (ConvTranspose2d k2 s2 as four 1x1 convolutions, rddb.py:28-38, in ONE launch): the ctiles channel tiles are
    // (parity, tile within the parity); the four parities of a spatial tile are consecutive blocks of one XCD, so the input
    // tile comes from HBM once and from that L2 three times (four launches read the input four times)
    // npar == 2 (16-bit, 64 channels): a 128-row tile = both COLUMN parities of an output pixel pair, the channel-tile index is
    // the ROW parity; a lane group then writes 256 contiguous bytes and a workgroup whole 8-KiB row segments -- the four-parity
    // form's 128-byte records 256 bytes apart ran at 2.5 TB/s where the same kernel writing contiguously reaches 4.3
    int ct = L % p.ctiles, par = 0;
    if (p.npar) { const int cpp = p.ctiles / p.npar; par = ct / cpp; ct -= par * cpp; }
    int t = L / p.ctiles;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int gy0 = oy0 * S - p.pad_y, gx0 = ox0 * S - p.pad_x;

    f32x16 acc[MT][PT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < PT; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

    const char* xb = (const char*)p.x + (size_t)b * p.H * p.W * p.xpix + (size_t)p.xcoff * sizeof(T);
    const char* wb = (const char*)p.wp + (size_t)par * p.wpar + (size_t)ct * p.nchunk * NTAP * COT * 64;

    // ---- per-thread staging descriptors (independent of the channel chunk).  Loads are issued
    // unconditionally from a clamped address and zeroed by a select afterwards: no branches, so the
    // compiler keeps all of a stage's loads in flight together.
    // (Round 3, measured: for stride 2 the 32 lanes of a pixel-fragment read are 160 bytes apart -- 8 distinct 16-byte slots, a 2-way
    //  bank conflict, SQ_LDS_BANK_CONFLICT 104 k cycles/us on the 4x4 stride-2 layers.  Keeping the even halo columns of a row in
    //  front of the odd ones makes those reads consecutive and conflict-free; the layers' time did not move (0.841 -> 0.843 ms per
    //  discriminator pass, same-box A/B): with 2 chunks x 16 MFMAs between barrier pairs they are bound by the per-tile prologue
    //  and the barrier chain, not by LDS reads.  Not kept.)
    int h_goff[HIT];                  // byte offset of this thread's halo piece in image b (chunk 0), -1 = outside
    const int part = tid & 3;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int pc = it * 256 + tid, pix = pc >> 2;
        const int iy = pix / IWT, ix = pix - iy * IWT;
        const int gy = gy0 + iy, gx = gx0 + ix;
        const bool ok = pc < NPH && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        h_goff[it] = ok ? ((gy * p.W + gx) * (int)p.xpix + part * 16) : -1;
    }
    // Two register stages (NRS == 2, kernels with few taps): the loads of chunk c + 2 are issued while chunk c computes.  With one
    // stage a chunk's loads have only that chunk's MFMA phase to land in -- 32 MFMAs per wave for a 2x2 kernel, ~1000 cycles
    // against 1500-2500 of load latency -- and the stride-2 parity gradients ran at 0.4 PFLOP/s (4x4 s1, 128 MFMAs: 1.05).
    // (measured on the stride-2 parity gradients, 2x2 taps: 1.445 -> 1.408 ms per discriminator pass with NRS = 2 -- the loads'
    //  latency is not what bounds them.  Each parity launch moves ~445 MB for 69 GFLOP (154 FLOP/B): they are close to memory-
    //  bound as four separate launches; one kernel producing all four parities from one staged dy tile would need a third of the
    //  traffic -- DESIGN.md section 8.)  Kept switchable; the default is one stage (36 fewer VGPRs).
    constexpr int NRS = 1;
    u32x4 hreg[NRS][HIT];
    u32x4 wreg[NRS][WPK ? 1 : WIT];
    auto issue_halo = [&](int c, auto rs) {
        constexpr int R = decltype(rs)::value;
        const bool cok = c * D::KCE + part * D::EPP < p.Cin;
        if (IG_DBG(p, 2) && c > 0) return;            // diagnostic builds: operand traffic of the first chunk only
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int off = (h_goff[it] >= 0 && cok) ? h_goff[it] + c * 64 : 0;
            hreg[R][it] = *(const u32x4*)(xb + off);
        }
    };
    auto write_halo = [&](int c, auto rs) {
        constexpr int R = decltype(rs)::value;
        const bool cok = c * D::KCE + part * D::EPP < p.Cin;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int pc = it * 256 + tid;
            u32x4 v = hreg[R][it];
            if (!(h_goff[it] >= 0 && cok)) v = u32x4{0u, 0u, 0u, 0u};
            if (HIT * 256 == NPH || pc < NPH) *(u32x4*)(lds_h + (pc >> 2) * PIXB + part * 16) = v;
        }
    };
    auto issue_w = [&](int c, auto rs) {
        constexpr int R = decltype(rs)::value;
        if (IG_DBG(p, 8) && c > 0) return;            // diagnostic builds: weight traffic of the first chunk only
        if constexpr (!WPK) {
            if (MT == 4 && NTAP == 1 && p.wsplit) {      // rows 0..63 / 64..127 of the tile: two 64-row packs [chunk][row][64 B], wsplit bytes apart
#pragma unroll
                for (int it = 0; it < WIT; ++it) {
                    const int pc = it * 256 + tid, row = pc >> 2;
                    wreg[R][it] = *(const u32x4*)(wb + (row >= 64 ? p.wsplit : 0) + ((size_t)c * 64 + (row & 63)) * 64 + (pc & 3) * 16);
                }
                return;
            }
            const char* ws = wb + (size_t)c * NTAP * COT * 64;
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int pc = it * 256 + tid;
                wreg[R][it] = *(const u32x4*)(ws + (size_t)((WIT * 256 == NPW || pc < NPW) ? pc : 0) * 16);
            }
        }
    };
    auto write_w = [&](auto rs) {
        constexpr int R = decltype(rs)::value;
        if constexpr (!WPK) {
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int pc = it * 256 + tid;
                if (WIT * 256 == NPW || pc < NPW) *(u32x4*)(lds_w + (pc >> 2) * PIXB + part * 16) = wreg[R][it];
            }
        }
    };

    // WPK (4x4 and larger kernels): one kernel row of weights in LDS at a time (LDS budget).  The row is prefetched into
    // registers one row ahead -- while the previous row's MFMAs run -- so its L2 latency is not exposed between two
    // barriers (it was: 16 MFMAs per wave against a 1-2 us load, matrix pipe 22 % busy on the 4x4 stride-2 layers).
    // Two rows ahead (even KH): one row's MFMA phase is 16-32 MFMAs per wave, 0.25-0.5 us -- less than the L2 round trip of the
    // row's loads, which was exposed once per kernel row (4x4 stride-2 layers: 16 rows per workgroup).
    constexpr int WPD = (WPK && KH % 2 == 0) ? 2 : 1;
    u32x4 wv[WPD][WPK ? WIT : 1];
    auto issue_wrow = [&](int c, int ky, int slot) {
        if (IG_DBG(p, 8) && (c > 0 || ky > 0)) return;
        if constexpr (WPK) {
            const char* ws = wb + ((size_t)c * NTAP + (size_t)ky * KW) * COT * 64;
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int pc = it * 256 + tid;
                wv[slot][it] = *(const u32x4*)(ws + (size_t)((WIT * 256 == NPW || pc < NPW) ? pc : 0) * 16);
            }
        }
    };
    auto write_wrow = [&](int slot) {
        if constexpr (WPK) {
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int pc = it * 256 + tid;
                if (WIT * 256 == NPW || pc < NPW) *(u32x4*)(lds_w + (pc >> 2) * PIXB + part * 16) = wv[slot][it];
            }
        }
    };

    using RS0 = std::integral_constant<int, 0>;
    using RS1 = std::integral_constant<int, NRS - 1>;
    auto chunk = [&](int c, auto rs) {
        // registers (chunk c) -> LDS; the previous chunk's readers passed the barrier at the end of the previous call
        if (!(IG_DBG(p, 16) && c > 0)) {
            write_halo(c, rs);
            write_w(rs);
            write_wrow(0);            // kernel row 0 of this chunk
        }
        __syncthreads();
        if (c + NRS < p.nchunk) {     // prefetch into the registers just drained: in flight while NRS chunks' MFMAs run
            issue_halo(c + NRS, rs);
            issue_w(c + NRS, rs);
        }
        // Fragment reads run ONE step (tap column, k-half) ahead of the MFMAs, in a register double buffer.  As plain
        // "read, then multiply" code hipcc emitted `ds_read x3, s_waitcnt lgkmcnt(0), v_mfma` per step (ISA of the 4x4 stride-2
        // layers): every MFMA behind an exposed LDS latency, matrix pipe 28 % busy with 2 waves per SIMD.
        // Two steps ahead where a step has only one or two MFMAs per wave (64 cycles: less than an LDS round trip).
        using frag_t = typename std::conditional<std::is_same<T, float>::value, f32x4, bf16x8>::type;
        constexpr int NSTEP = KW * 2, NFLAT = WPK ? NSTEP : KH * NSTEP;      // WPK: the ring restarts with every kernel row (its weights are rewritten)
        constexpr int PD = (MT * PT <= 2 && NSTEP >= 4) ? 2 : 1, NB = PD + 1;
        frag_t fa[NB][MT], fb[NB][PT];
        auto rd = [&](int kyw, int g) {          // flat step g of the ring; kyw = the kernel row in LDS (WPK)
            const int ky = WPK ? kyw : g / NSTEP, s2 = g % NSTEP;
            const int kx = s2 >> 1, ks = s2 & 1;
            const int tapw = WPK ? kx : ky * KW + kx;
            const int koff = ks * 32 + h * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                fa[g % NB][m] = *(const frag_t*)(lds_w + (tapw * COT + m * 32 + r) * PIXB + koff);
#pragma unroll
            for (int q = 0; q < PT; ++q)
                fb[g % NB][q] = *(const frag_t*)(lds_h + (((wave * PT + q) * S + ky) * IWT + r * S + kx) * PIXB + koff);
        };
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            if constexpr (WPK) {
                if (ky > 0) {
                    __syncthreads();  // the previous row's readers are done
                    if (!IG_DBG(p, 16)) write_wrow(ky % WPD);
                    __syncthreads();
                }
                // the row WPD ahead, into the registers just written out
                if (ky + WPD < KH) issue_wrow(c, ky + WPD, ky % WPD);
                else if (c + 1 < p.nchunk) issue_wrow(c + 1, ky + WPD - KH, ky % WPD);
            }
            if (WPK || ky == 0) {
#pragma unroll
                for (int g = 0; g < PD; ++g) rd(ky, g);
            }
#pragma unroll
            for (int s2 = 0; s2 < NSTEP; ++s2) {
                const int g = (WPK ? 0 : ky * NSTEP) + s2;
                if (g + PD < NFLAT) rd(ky, g + PD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < PT; ++q) {
                        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g % NB][m][j], fb[g % NB][q][j], acc[m][q], 0, 0, 0);
                        } else if (IG_DBG(p, 1)) {      // keep the fragment reads alive without the matrix pipe
                            const u32x4 t = __builtin_bit_cast(u32x4, fa[g % NB][m]) ^ __builtin_bit_cast(u32x4, fb[g % NB][q]);
                            acc[0][0][0] += __builtin_bit_cast(float, t[0] ^ t[1] ^ t[2] ^ t[3]);
                        } else
                            acc[m][q] = sg_mfma16<T>(fa[g % NB][m], fb[g % NB][q], acc[m][q]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();   // all waves are done reading this chunk's LDS image
    };
    issue_halo(0, RS0{});
    issue_w(0, RS0{});
    issue_wrow(0, 0, 0);
    if constexpr (WPD == 2) issue_wrow(0, 1, 1);
    if constexpr (NRS == 2) {
        if (p.nchunk > 1) { issue_halo(1, RS1{}); issue_w(1, RS1{}); }
        for (int c = 0; c < p.nchunk; c += 2) {
            chunk(c, RS0{});
            if (c + 1 < p.nchunk) chunk(c + 1, RS1{});
        }
    } else {
        for (int c = 0; c < p.nchunk; ++c) chunk(c, RS0{});
    }

    // Epilogue.  The MFMA result holds one pixel per lane and 4 channels per register group: stored as it stands every
    // instruction scatters 8-byte pieces over 32 pixels (a strided parity store of a stride-2 gradient wrote 16 contiguous bytes
    // per pixel: those convolutions ran at 380 TFLOP/s, store-bound).  With 16-byte-aligned operands each wave transposes a
    // row of its tile through LDS (the stage buffers are free: every wave passed the loop's last barrier) and reads it back
    // with the lanes of a pixel contiguous: 128-byte segments per pixel for residual / mask loads and the store.
    if (IG_DBG(p, 4)) {                               // diagnostic builds: no epilogue (keep the accumulators alive)
        float keep = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < PT; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) keep += acc[m][q][i];
        if (keep == 1.2345e-30f) *(float*)p.y = keep;
        return;
    }
    ConvP pe = p;
    if (p.npar == 2) { pe.oa = par; pe.ob = 0; }
    else if (p.npar) { pe.oa = par >> 1; pe.ob = par & 1; }
    if (p.buf16) {
        constexpr int RS = COT * 4 + 16;
        constexpr bool OPS = !(MT == 4 && NTAP == 1);          // the pixel-pair form of the up-sampler has no residual / mask operands
        char* lw = smem + wave * 32 * RS;
#pragma unroll
        for (int q = 0; q < PT; ++q) conv_epilogue_lds_row<T, MT, PT, OPS>(pe, acc, q, lw, b, ct, oy0 + wave * PT + q, ox0, lane);
    } else
        conv_epilogue<T, MT, PT>(pe, acc, b, ct, oy0 + wave * PT, ox0, r, h);
}

// ------------------------------------------------------------------ host launcher
template <typename T, int KH, int KW, int S, int MT, int PT, bool WPK>
static int launch_igemm(const ConvP& p, int ctiles, hipStream_t st) {
    constexpr int TH = 4 * PT, TW = 32;
    constexpr int IHT = (TH - 1) * S + KH, IWT = (TW - 1) * S + KW;
    constexpr int COT = 32 * MT, NTAP = KH * KW;
    constexpr size_t STAGE = (size_t)IHT * IWT * 80 + (size_t)(WPK ? KW : NTAP) * COT * 80, EPI = (size_t)4 * 32 * (COT * 4 + 16);
    constexpr size_t SMEM = STAGE > EPI ? STAGE : EPI;       // the LDS-transposed epilogue reuses the stage buffers
    static bool attr_set = false;
    auto kern = conv_igemm_k<T, KH, KW, S, MT, PT, WPK>;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set = true;
    }
    ConvP q = p;
    q.tiles_x = cdiv(p.OW, TW);
    q.tiles_y = cdiv(p.OH, TH);
    q.ctiles = ctiles;
    { static const char* e = sg_env("SRCGAN_DBG"); q.dbg = e ? atoi(e) : 0; }
    dim3 grid((unsigned)((size_t)q.tiles_x * q.tiles_y * p.B * ctiles), 1, 1);
    char cls[96];
    snprintf(cls, sizeof(cls), "conv_igemm<%s,%dx%d,s%d,MT%d>", sizeof(T) == 4 ? "f32" : (__is_same(T, __bf16) ? "bf16" : "f16"), KH, KW, S, MT);
    // algorithmic work: 2*pixels*taps*Cin*Cout flop; bytes: input read once + output written once
    const double px = (double)p.B * p.OH * p.OW * (p.npar ? p.npar : 1);
    const int tok = sg_prof_start(cls, 2.0 * px * NTAP * p.Cin * p.Cout,
                                  ((double)p.B * p.H * p.W * p.Cin + px * p.Cout) * sizeof(T), st);
    hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, st, q);
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return 0;
}

#ifndef SG_UP_PT
#define SG_UP_PT 1      // rows per wave of the up-sampler's pixel-pair form (2: 128 accumulator registers beside the 128-row epilogue -- spills, 16 % slower)
#endif
template <typename T, int MT>
static int dispatch_shape(const ConvP& p, int kh, int kw, int s, int ctiles, hipStream_t st) {
#define SG_CASE(KH_, KW_, S_, PT_, WPK_) \
    if (kh == KH_ && kw == KW_ && s == S_) return launch_igemm<T, KH_, KW_, S_, MT, PT_, WPK_>(p, ctiles, st);
    SG_CASE(3, 3, 1, 2, false)
    SG_CASE(1, 1, 1, 2, false)
    SG_CASE(2, 2, 2, 1, false)
#ifndef SG_PT22
#define SG_PT22 2
#endif
    SG_CASE(2, 2, 1, SG_PT22, false)
    SG_CASE(1, 2, 1, 2, false)
    SG_CASE(2, 1, 1, 2, false)
    SG_CASE(4, 4, 2, 1, true)
    SG_CASE(4, 4, 1, 2, true)
    SG_CASE(3, 4, 1, 2, true)        // parity classes of the 7x7 stride-2 stem's input gradient (resdeconv.py:113): 3 or 4 taps per axis
    SG_CASE(4, 3, 1, 2, true)
    SG_CASE(3, 3, 2, 1, false)
    SG_CASE(7, 7, 2, 1, true)        // ResDeconv stem (resdeconv.py:113)
    SG_CASE(1, 1, 2, 2, false)       // ResDeconv down-sample shortcut (resdeconv.py:12-15,157-161)
    SG_CASE(5, 5, 1, 2, true)        // ESPCN conv1 (espcn.py:33), SRCNN conv3 (srcnn.py:36)
    SG_CASE(9, 9, 1, 2, true)        // SRCNN conv1 (srcnn.py:32)
#undef SG_CASE
    SG_FAIL("srcgan_conv_igemm: unsupported kernel %dx%d stride %d", kh, kw, s);
}

int sg_conv3x3_dma(const ConvP& p, int dtype, hipStream_t st);      // conv3x3_dma.hip
int sg_dgrad_s2k4(const ConvP& p, int dtype, hipStream_t st);       // conv_par4.hip
static const bool g_force_generic = sg_env("SRCGAN_GENERIC_3X3") != nullptr;   // A/B switch for benchmarking

extern "C" int srcgan_conv_igemm(const srcgan_conv_desc* d, void* stream) {
    SG_REQUIRE(d && d->x && d->wp && d->y, "srcgan_conv_igemm: null pointer");
    SG_REQUIRE(sg_dtype_ok(d->dtype), "srcgan_conv_igemm: bad dtype %d", d->dtype);
    const int esz = d->dtype == SRCGAN_F32 ? 4 : 2;
    const int epp = 16 / esz;
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0 && d->Cin > 0 && d->Cout > 0,
               "srcgan_conv_igemm: non-positive dimension");
    SG_REQUIRE(d->Cin % epp == 0 && d->x_cs % epp == 0 && d->x_coff % epp == 0,
               "srcgan_conv_igemm: input channels/stride/offset (%d,%d,%d) must be multiples of %d", d->Cin, d->x_cs, d->x_coff, epp);
    SG_REQUIRE((d->x_plane || d->x_coff + d->Cin <= d->x_cs) && (d->y_plane || d->y_coff + d->Cout <= d->y_cs),
               "srcgan_conv_igemm: channel slice exceeds stride");
    SG_REQUIRE(!d->x_plane || (d->kh == 3 && d->kw == 3 && d->stride == 1), "srcgan_conv_igemm: a blocked-layout input is supported by the 3x3 stride-1 kernel only");
    SG_REQUIRE(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->wp % 16) == 0, "srcgan_conv_igemm: x/wp must be 16-byte aligned");
    SG_REQUIRE(d->os >= 1 && d->oa >= 0 && d->ob >= 0 && d->oa < d->os && d->ob < d->os, "srcgan_conv_igemm: bad output scale/offset");
    SG_REQUIRE((d->OH - 1) * d->os + d->oa < d->YH && (d->OW - 1) * d->os + d->ob < d->YW, "srcgan_conv_igemm: output extent exceeds tensor");
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.x = d->x; p.wp = d->wp; p.bias = d->bias; p.y = d->y; p.r1 = d->r1; p.r2 = d->r2; p.mz = d->mz;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.xcoff = d->x_coff;
    p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.YH = d->YH; p.YW = d->YW; p.ycoff = d->y_coff;
    p.pad_y = d->pad_y; p.pad_x = d->pad_x; p.os = d->os; p.oa = d->oa; p.ob = d->ob;
    p.r1coff = d->r1_coff; p.r1cend = d->r1_cend;
    p.r2coff = d->r2_coff; p.r2cend = d->r2_cend;
    p.mzcoff = d->mz_coff; p.mzc0 = d->mz_c0;
    auto pl = [](long v) { return v ? v : 64L; };      // plane stride 0 = interleaved NHWC
    p.xpix = (long)d->x_cs * esz; p.xplane = pl(d->x_plane); p.ypix = (long)d->y_cs * esz; p.yplane = pl(d->y_plane);
    p.r1pix = (long)d->r1_cs * esz; p.r1plane = pl(d->r1_plane); p.r2pix = (long)d->r2_cs * esz; p.r2plane = pl(d->r2_plane);
    p.mzpix = (long)d->mz_cs * esz; p.mzplane = pl(d->mz_plane);
    p.rev = d->rev_batch;
    p.sgn_out = (unsigned char*)d->sign_out; p.sgn_in = (const unsigned char*)d->sign_in;
    if (d->sign_out || d->sign_in) {
        const bool dense32 = d->kh == 3 && d->kw == 3 && d->stride == 1 && sg_is16(d->dtype) && d->Cout == 32 && d->os == 1 && d->oa == 0 && d->ob == 0 &&
                             d->YH == d->OH && d->YW == d->OW && d->x_plane && !(d->sign_in && d->mz) && (!d->sign_out || d->act);
        // 64 channels (8 mask bytes per pixel): written by the up-sampler's 1x1 parity form, read by a 3x3 s1 convolution without other operands
        const bool up_out = d->sign_out && !d->sign_in && d->kh == 1 && d->kw == 1 && d->npar == 4 && sg_is16(d->dtype) && d->Cout == 64 && d->act;
        const bool in64 = d->sign_in && !d->sign_out && d->kh == 3 && d->kw == 3 && d->stride == 1 && sg_is16(d->dtype) && d->Cout == 64 && d->os == 1 &&
                          d->YH == d->OH && d->YW == d->OW && !d->mz && !d->r1 && !d->r2;
        SG_REQUIRE(dense32 || up_out || in64,
                   "srcgan_conv_igemm: sign masks need a 3x3 s1 16-bit conv with Cout == 32 on a blocked input, unscaled output (act for sign_out, no mz beside sign_in); "
                   "or Cout == 64: sign_out of the 1x1 four-parity form with act, sign_in of a 3x3 s1 conv without other operands");
    }
    p.alpha = d->alpha; p.beta1 = d->beta1; p.beta2 = d->beta2; p.slope = d->slope; p.mslope = d->mslope;
    p.act = d->act;
    const int kce = 64 / esz;
    p.nchunk = cdiv(d->Cin, kce);
    auto m4 = [](int v) { return (v & 3) == 0; };
    p.vec = m4(d->Cout) && m4(d->y_cs) && m4(d->y_coff) && ((uintptr_t)d->y % 16 == 0) &&
            (!d->bias || ((uintptr_t)d->bias % 16 == 0)) &&
            (!d->r1 || (m4(d->r1_cs) && m4(d->r1_coff) && m4(d->r1_cend) && (uintptr_t)d->r1 % 16 == 0)) &&
            (!d->r2 || (m4(d->r2_cs) && m4(d->r2_coff) && m4(d->r2_cend) && (uintptr_t)d->r2 % 16 == 0)) &&
            (!d->mz || (m4(d->mz_cs) && m4(d->mz_coff) && m4(d->mz_c0) && (uintptr_t)d->mz % 16 == 0));
    {
        auto me = [&](int v) { return v % epp == 0; };
        p.vec16 = p.vec && me(d->Cout) && me(d->y_cs) && me(d->y_coff) &&
                  (!d->r1 || (me(d->r1_cs) && me(d->r1_coff) && me(d->r1_cend))) &&
                  (!d->r2 || (me(d->r2_cs) && me(d->r2_coff) && me(d->r2_cend))) &&
                  (!d->mz || (me(d->mz_cs) && me(d->mz_coff) && me(d->mz_c0)));
    }
    {
        const long lim = (1L << 31) - (1L << 20);
        p.buf16 = p.vec16 && p.yplane < lim && p.ypix < (1 << 20) && (!d->r1 || (p.r1plane < lim && p.r1pix < (1 << 20))) &&
                  (!d->r2 || (p.r2plane < lim && p.r2pix < (1 << 20))) && (!d->mz || (p.mzplane < lim && p.mzpix < (1 << 20)));
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->npar && d->kh == 1 && d->kw == 1) {
        // four output parities of a stride-2 scatter in one launch: parity q = (oa, ob) = (q >> 1, q & 1) uses the pack at wp + q * wpar_stride
        SG_REQUIRE(d->npar == 4 && d->stride == 1 && d->os == 2 && d->oa == 0 && d->ob == 0 && d->wpar_stride > 0 && d->wpar_stride % 16 == 0 &&
                   !d->sign_in, "srcgan_conv_igemm: npar == 4 with a 1x1 kernel needs os == 2, oa == ob == 0 and the four packs wpar_stride bytes apart");
        SG_REQUIRE(!d->sign_out || p.buf16, "srcgan_conv_igemm: sign_out of the 1x1 four-parity form needs 16-byte accessible output channels");
        p.npar = 4; p.wpar = d->wpar_stride;
        // pixel-pair form: 64 channels of 2 bytes, dense interleaved output -> one 128-row tile per ROW parity writes both column parities
        if (sg_is16(d->dtype) && d->Cout == 64 && d->y_cs == 64 && d->y_coff == 0 && !d->y_plane && d->YW % 2 == 0 && d->YW == 2 * d->OW && p.buf16 &&
            !d->bias && !d->r1 && !d->r2 && !d->mz && d->Cin % 32 == 0) {
            p.npar = 2; p.wpar = 2 * d->wpar_stride; p.wsplit = d->wpar_stride;
            p.Cout = 128; p.ypix *= 2; p.YW = d->YW / 2; p.osx = 1;
            if (d->dtype == SRCGAN_F16) return launch_igemm<_Float16, 1, 1, 1, 4, SG_UP_PT, false>(p, 2, st);
            return launch_igemm<__bf16, 1, 1, 1, 4, SG_UP_PT, false>(p, 2, st);
        }
    } else if (d->npar) {
        SG_REQUIRE(d->npar == 4 && d->kh == 2 && d->kw == 2 && d->stride == 1 && d->wpar_stride > 0 && d->wpar_stride % 16 == 0 && !d->x_plane && !d->y_plane &&
                   !d->sign_in && !d->sign_out && !d->bias,
                   "srcgan_conv_igemm: npar must be 0 or 4 (2x2 stride-1 parity packs wpar_stride bytes apart, interleaved tensors, no bias / sign masks)");
        p.wpar = d->wpar_stride;
        return sg_dgrad_s2k4(p, d->dtype, st);
    }
    if (d->kh == 3 && d->kw == 3 && d->stride == 1 && (!g_force_generic || d->x_plane)) return sg_conv3x3_dma(p, d->dtype, st);
    // Cout <= 32 -> one 32-row M tile per workgroup, otherwise 64-row tiles.
    const int npm = p.npar ? p.npar : 1;        // channel tiles = parities x tiles of one parity
    if (d->Cout <= 32) {
        if (d->dtype == SRCGAN_F32) return dispatch_shape<float, 1>(p, d->kh, d->kw, d->stride, npm, st);
        if (d->dtype == SRCGAN_F16) return dispatch_shape<_Float16, 1>(p, d->kh, d->kw, d->stride, npm, st);
        return dispatch_shape<__bf16, 1>(p, d->kh, d->kw, d->stride, npm, st);
    }
#ifdef SG_MT4_22
    // (variant) stride-2 parity gradients with >= 128 output rows: one 128-row tile per workgroup halves the dy staging per FLOP
    if (d->dtype == SRCGAN_BF16 && d->kh == 2 && d->kw == 2 && d->stride == 1 && d->Cout % 128 == 0)       // (the variant instantiates bf16 only)
        return launch_igemm<__bf16, 2, 2, 1, 4, SG_PT22, false>(p, d->Cout / 128, st);
#endif
    const int ctiles = cdiv(d->Cout, 64) * npm;
    if (d->dtype == SRCGAN_F32) return dispatch_shape<float, 2>(p, d->kh, d->kw, d->stride, ctiles, st);
    if (d->dtype == SRCGAN_F16) return dispatch_shape<_Float16, 2>(p, d->kh, d->kw, d->stride, ctiles, st);
    return dispatch_shape<__bf16, 2>(p, d->kh, d->kw, d->stride, ctiles, st);
}
