// Input gradient of a 4x4 stride-2 pad-1 convolution (the PatchGAN's down-sampling layers, reference src/model/model.py:612-634)
// with ALL FOUR output parities in one launch.
//
//   dx[2t + a][2u + b][ci] = sum over (ty, tx) in {0,1}^2, co:  dy[t + a - 1 + ty][u + b - 1 + tx][co] * W[co][ci][ky][kx],
//                            ky = (a ? 2 : 3) - 2 ty,  kx = (b ? 2 : 3) - 2 tx
//
// As four separate 2x2 stride-1 convolutions (conv_igemm_k<2,2,1>, rounds 1-2) every launch staged the dy tile again for four
// taps' worth of MFMAs: 32 MFMAs per wave between a tile's load and its epilogue, 0.53 PFLOP/s, 445 MB moved per launch for 69
// GFLOP.  Here one staged (TH+2) x 34 window of dy feeds the 16 (parity, tap) products -- the work per staged tile of a 4x4
// stride-1 layer, which runs at 1.05 PFLOP/s in the same framework -- and a pixel fragment at window position (wy, wx) is read
// from LDS once for every parity that uses it (9 fragment reads per k-step instead of 16).
//   GEMM orientation as conv_igemm.hip: D[M = 32 input channels][N = 32 positions u] per parity; a wave owns PT rows t.
//   LDS: window (80 B per pixel: conflict-free ds_read_b128 of 32 consecutive pixels) + 16 x 32 weight rows of the K chunk.
#include "conv_params.h"
#include <type_traits>

template <typename T, int PT, bool VEC16>
__global__ __launch_bounds__(256, 2) void dgrad_s2k4_k(const ConvP p) {
    using D = DT<T>;
    constexpr int TH = 4 * PT, TW = 32, IHT = TH + 2, IWT = TW + 2;
    constexpr int COT = 32, PIXB = 80, NPAR = 4, WROWS = 16 * COT;
    constexpr int NPH = IHT * IWT * 4, NPW = WROWS * 4;              // 16-byte pieces of the window / of a chunk's weights
    constexpr int HIT = (NPH + 255) / 256, WIT = NPW / 256;
    static_assert(NPW % 256 == 0, "weight pieces per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_h = smem;
    char* lds_w = smem + IHT * IWT * PIXB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int L;
    {   // XCD-aware block -> tile map (conv_igemm.hip): the channel tiles of a spatial tile and neighbouring tiles share an L2
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int ct = L % p.ctiles;
    int t = L / p.ctiles;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int t0 = ty * TH, u0 = tx * TW;            // first (t, u) of the tile; window origin = (t0 - 1, u0 - 1) in dy

    f32x16 acc[NPAR][1][PT];
#pragma unroll
    for (int q = 0; q < NPAR; ++q)
#pragma unroll
        for (int k = 0; k < PT; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][0][k][i] = 0.f;

    const char* xb = (const char*)p.x + (size_t)b * p.H * p.W * p.xpix + (size_t)p.xcoff * sizeof(T);
    // packed weights of parity q: Wp[row tile][chunk][tap (ty,tx)][row][k], row tiles of 64 (32 if <= 32 rows in all)
    const int cotp = p.Cout <= 32 ? 32 : 64, rt = (ct * COT) / cotp, roff = (ct * COT) % cotp;
    const int part = tid & 3;
    int h_goff[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int pc = it * 256 + tid, pix = pc >> 2;
        const int iy = pix / IWT, ix = pix - iy * IWT;
        const int gy = t0 - 1 + iy, gx = u0 - 1 + ix;
        const bool ok = pc < NPH && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        h_goff[it] = ok ? ((gy * p.W + gx) * (int)p.xpix + part * 16) : -1;
    }
    // weight piece `it` of this thread: LDS row (it * 256 + tid) >> 2 = (parity * 4 + tap) * 32 + rr with parity = it >> 1,
    // tap = (it & 1) * 2 + (tid >> 7), rr = (tid >> 2) & 31: one per-thread base + a uniform offset per piece
    const char* wthr = (const char*)p.wp + ((long)rt * p.nchunk * 4 + (tid >> 7)) * cotp * 64 + (long)(roff + ((tid >> 2) & 31)) * 64 + part * 16;
    u32x4 hreg[HIT], wreg[WIT];
    auto issue = [&](int c) {
        const bool cok = c * D::KCE + part * D::EPP < p.Cin;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int off = (h_goff[it] >= 0 && cok) ? h_goff[it] + c * 64 : 0;
            hreg[it] = *(const u32x4*)(xb + off);
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) wreg[it] = *(const u32x4*)(wthr + (long)(it >> 1) * p.wpar + ((long)c * 4 + (it & 1) * 2) * cotp * 64);
    };
    auto stage = [&](int c) {
        const bool cok = c * D::KCE + part * D::EPP < p.Cin;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int pc = it * 256 + tid;
            u32x4 v = hreg[it];
            if (!(h_goff[it] >= 0 && cok)) v = u32x4{0u, 0u, 0u, 0u};
            if (HIT * 256 == NPH || pc < NPH) *(u32x4*)(lds_h + (pc >> 2) * PIXB + part * 16) = v;
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) *(u32x4*)(lds_w + ((it * 256 + tid) >> 2) * PIXB + part * 16) = wreg[it];
    };

    using frag_t = typename std::conditional<std::is_same<T, float>::value, f32x4, bf16x8>::type;
    issue(0);
    for (int c = 0; c < p.nchunk; ++c) {
        stage(c);                               // the previous chunk's readers passed the barrier at the end of the last iteration
        __syncthreads();
        if (c + 1 < p.nchunk) issue(c + 1);     // in flight while this chunk's MFMAs run
#pragma unroll
        for (int wy = 0; wy < 3; ++wy)
#pragma unroll
            for (int wx = 0; wx < 3; ++wx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int koff = ks * 32 + h * 16;
                    frag_t bb[PT];
#pragma unroll
                    for (int q = 0; q < PT; ++q)
                        bb[q] = *(const frag_t*)(lds_h + ((wave * PT + q + wy) * IWT + r + wx) * PIXB + koff);
#pragma unroll
                    for (int par = 0; par < NPAR; ++par) {
                        const int tyy = wy - (par >> 1), txx = wx - (par & 1);          // this window position as a tap of parity (a, b)
                        if (tyy < 0 || tyy > 1 || txx < 0 || txx > 1) continue;
                        const frag_t a = *(const frag_t*)(lds_w + ((par * 4 + tyy * 2 + txx) * COT + r) * PIXB + koff);
#pragma unroll
                        for (int q = 0; q < PT; ++q) {
                            if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[par][0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[q][j], acc[par][0][q], 0, 0, 0);
                            } else
                                acc[par][0][q] = sg_mfma16<T>(a, bb[q], acc[par][0][q]);
                        }
                    }
                }
        __syncthreads();                        // all waves are done reading this chunk's LDS image
    }

    // ---- epilogue: parity (a, b) -> dx[2t + a][2u + b], masked by LeakyReLU'(mz) like the per-parity form; t < ceil((YH - a) / 2)
    constexpr int RS = COT * 4 + 16;
    char* lw = smem + wave * 32 * RS;
#pragma unroll
    for (int par = 0; par < NPAR; ++par) {
        ConvP pq = p;
        pq.os = 2; pq.oa = par >> 1; pq.ob = par & 1;
        pq.OH = (p.YH - pq.oa + 1) / 2; pq.OW = (p.YW - pq.ob + 1) / 2;
        if constexpr (VEC16) {
#pragma unroll
            for (int q = 0; q < PT; ++q) conv_epilogue_lds_row<T, 1, PT>(pq, acc[par], q, lw, b, ct, t0 + wave * PT + q, u0, lane);
        } else
            conv_epilogue<T, 1, PT>(pq, acc[par], b, ct, t0 + wave * PT, u0, r, h);
    }
}

template <typename T, int PT, bool VEC16>
static int launch_par4v(const ConvP& p, hipStream_t st) {
    constexpr int TH = 4 * PT, IHT = TH + 2, IWT = 34;
    constexpr size_t STAGE = (size_t)IHT * IWT * 80 + (size_t)16 * 32 * 80, EPI = (size_t)4 * 32 * (32 * 4 + 16);
    constexpr size_t SMEM = STAGE > EPI ? STAGE : EPI;
    auto kern = dgrad_s2k4_k<T, PT, VEC16>;
    static bool attr_set = false;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set = true;
    }
    ConvP q = p;
    q.tiles_x = cdiv((p.YW + 1) / 2, 32);
    q.tiles_y = cdiv((p.YH + 1) / 2, TH);
    q.ctiles = cdiv(p.Cout, 32);
    dim3 grid((unsigned)((size_t)q.tiles_x * q.tiles_y * p.B * q.ctiles), 1, 1);
    char cls[96];
    snprintf(cls, sizeof(cls), "dgrad_s2k4<%s,4 parities>", sizeof(T) == 4 ? "f32" : (__is_same(T, __bf16) ? "bf16" : "f16"));
    const double px = (double)p.B * p.YH * p.YW;
    const int tok = sg_prof_start(cls, 2.0 * px * 4 * p.Cin * p.Cout, ((double)p.B * p.H * p.W * p.Cin + px * p.Cout) * sizeof(T), st);
    hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, st, q);
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return 0;
}

template <typename T, int PT>
static int launch_par4(const ConvP& p, hipStream_t st) {
    // only the LDS-transposed epilogue (16-byte accessible operands) is instantiated: the per-element form spills beside the 128
    // accumulator registers; callers with odd channel counts keep the four separate parity launches
    SG_REQUIRE(p.buf16, "srcgan_conv_igemm: npar == 4 needs output / mask channels, strides and offsets that are multiples of 16 bytes (and channel planes below 2 GiB)");
    return launch_par4v<T, PT, true>(p, st);
}

#ifndef SG_P4PT
#define SG_P4PT 2
#endif
// entry used by srcgan_conv_igemm for descriptors with npar == 4
int sg_dgrad_s2k4(const ConvP& p, int dtype, hipStream_t st) {
    if (dtype == SRCGAN_F32) return launch_par4<float, SG_P4PT>(p, st);
    if (dtype == SRCGAN_F16) return launch_par4<_Float16, SG_P4PT>(p, st);
    return launch_par4<__bf16, SG_P4PT>(p, st);
}
