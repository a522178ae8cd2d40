// 3x3 stride-1 convolution, deep-pipelined variant: same persistent-workgroup / LDS-DMA / swizzled-LDS design as
// conv3x3_dma.hip, but the operand stream is cut into 32-byte channel slices (16 bf16 / 8 f32 channels) held in a
// FOUR-slot LDS ring, so three DMA batches are always in flight per CU.  Motivation (round-1 measurements): with one
// batch in flight the operand ingest ran at ~5 TB/s chip-wide and did not overlap the MFMAs (DMA-only 115 us,
// MFMA+LDS-only 90 us, together 171 us for a 160->32 conv); the MFMA pipe was 32 % busy.
//
// Pipeline, per workgroup (8 waves, 16x32-pixel tile, stages numbered across units):
//   iteration g:  s_waitcnt vmcnt(2*IPW)   -> my DMA instructions of stage g have landed (g+1, g+2 still in flight)
//                 s_barrier                 -> everyone's have; everyone finished reading slot (g-1)%4
//                 issue stage g+3 -> slot (g+3)%4   (exactly IPW instructions per wave: dummies pad the count)
//                 9 taps x MT x PT MFMAs on slot g%4
// LDS image per slot: [612 halo pixels][32 B] + [9*COT weight rows][32 B], XOR-swizzled: 16-byte slot s of pixel p
// holds channel-half s ^ ((p >> 3) & 1): any 16 consecutive pixels hit 16 distinct bank groups.
#include "conv_params.h"
#include <type_traits>
#include <stdlib.h>

static __device__ __attribute__((aligned(64))) unsigned int sg_zero_page[16];   // per-TU copy (no -fgpu-rdc)

// LDS-DMA issued from inline asm so that hipcc does not track it: with the builtin the compiler drained vmcnt(0)
// before re-using the address registers of the next batch, which serialised the ring.  All completion accounting
// for these loads is done by hand (counted s_waitcnt vmcnt below); the compiler's own waits for its ordinary
// loads can only over-wait (vmcnt is in order).  M0 (LDS destination base) is saved/restored inside the statement.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_off);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

template <typename T, int MT>
__global__ __launch_bounds__(768) void conv3x3_pipe_k(const ConvP p) {
    constexpr int NWV = 8, NLW = 4, PT = 2, NSLOT = 4, DEPTH = 3;   // 8 MFMA waves + 4 loader waves (one per SIMD)
    constexpr int TH = PT * NWV, TW = 32, IHT = TH + 2, IWT = TW + 2;
    constexpr int COT = 32 * MT, NTAP = 9;
    constexpr int SLB = 32;                                  // slice bytes per pixel / weight row
    constexpr int SLE = SLB / (int)sizeof(T);                // channels per slice (16 bf16, 8 f32)
    constexpr int NHP = IHT * IWT;
    constexpr int HPIECES = (NHP * SLB + 1023) / 1024;       // 20
    constexpr int WPIECES = (NTAP * COT * SLB + 1023) / 1024;   // 9 / 18
    constexpr int HBYTES = HPIECES * 1024, WBYTES = WPIECES * 1024, SBYTES = HBYTES + WBYTES;
    constexpr int IPW = (HPIECES + WPIECES + NLW - 1) / NLW; // DMA instructions per LOADER wave per stage (8 / 10)
    constexpr int ERS = COT * 4 + 16;
    static_assert(NWV * 16 * ERS <= SBYTES, "half-row transpose space must fit one slot");
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [NSLOT][halo | weights] + dummy page per wave
    const bool loader = (threadIdx.x >> 6) >= NWV;
    const int lw = (threadIdx.x >> 6) - NWV;               // loader index 0..3
    char* dummy = smem + NSLOT * SBYTES + (loader ? lw : 0) * 1024;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int nunits = p.tiles_x * p.tiles_y * p.B * p.ctiles;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, gw = (gridDim.x + 7 - xcd) >> 3;
    const int u8 = (nunits + 7) >> 3;
    const int u_lo = xcd * u8, u_hi = (u_lo + u8 < nunits) ? u_lo + u8 : nunits;
    const int nsl = (p.Cin + SLE - 1) / SLE;                 // slices per unit
    const char* zp = (const char*)sg_zero_page;

    // ---- fetch side (loader waves).  Piece pi = it*NLW + lw is wave-uniform; its per-lane geometry is recomputed at
    // issue time (a few VALU ops) so that only the per-unit halo pixel offsets stay in registers.
    int poff[IPW];          // halo pieces: global byte offset of this lane's pixel in the current fetch unit (-1 outside)
#pragma unroll
    for (int it = 0; it < IPW; ++it) poff[it] = -1;
    const char* f_xb = nullptr; const char* f_wb = nullptr;
    auto setup_fetch = [&](int u, int& ob, int& oct, int& ooy0, int& oox0) {
        const int ct = u % p.ctiles; int t = u / p.ctiles;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
        ob = b; oct = ct; ooy0 = ty * TH; oox0 = tx * TW;
        const int gy0 = ooy0 - p.pad_y, gx0 = oox0 - p.pad_x;
        f_xb = (const char*)p.x + ((size_t)b * p.H * p.W * p.xCs + p.xcoff) * sizeof(T);
        f_wb = (const char*)p.wp + (size_t)ct * p.nchunk * NTAP * COT * 64;
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int pi = it * NLW + lw;
            if (pi < HPIECES) {
                const int q = pi * 64 + lane, lp = q >> 1;
                const int iy = lp / IWT, ix = lp - iy * IWT;
                const int gy = gy0 + iy, gx = gx0 + ix;
                const bool ok = lp < NHP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                poff[it] = ok ? (gy * p.W + gx) * p.xCs * (int)sizeof(T) : -1;
            }
        }
    };
    // slice sl of the unit: channels [sl*SLE, +SLE); packed weights keep 64-byte chunks: chunk = sl/2, half-offset (sl&1)*32
    auto issue = [&](int sl, int slot) {
        char* ls = smem + slot * SBYTES;
        const char* ws = f_wb + (size_t)(sl >> 1) * NTAP * COT * 64;
#pragma unroll
        for (int it = 0; it < IPW; ++it) {
            const int pi = it * NLW + lw;
            const char* src = zp; char* dst = dummy;
            if (pi < HPIECES) {
                const int q = pi * 64 + lane, lp = q >> 1, half = (q & 1) ^ ((lp >> 3) & 1);
                dst = ls + pi * 1024;
                if (poff[it] >= 0 && sl * SLE + half * (SLE / 2) < p.Cin) src = f_xb + poff[it] + sl * SLB + half * 16;
            } else if (pi < HPIECES + WPIECES) {
                const int q = (pi - HPIECES) * 64 + lane, wr = q >> 1, half = (q & 1) ^ ((wr >> 3) & 1);
                dst = ls + HBYTES + (pi - HPIECES) * 1024;
                if (wr < NTAP * COT) src = ws + (size_t)wr * 64 + (sl & 1) * 32 + half * 16;
            }
            dma16(src, lds_offset(dst));
        }
    };

    int u = u_lo + jw;
    if (u >= u_hi) return;
    const int my_units = (u_hi - 1 - u) / gw + 1;
    const int total = my_units * nsl;                       // stages this workgroup will compute
    int cb, cct, coy0, cox0, nb_, nct, noy0, nox0;
    setup_fetch(u, nb_, nct, noy0, nox0);
    int fu = u, fsl = 0, fg = 0;                             // fetch cursor: unit, slice, global stage index
    auto advance_fetch = [&]() {                              // issue stage fg (real or dummy) and move the cursor
        if (fg < total) {
            issue(fsl, fg % NSLOT);
            if (++fsl == nsl) {
                fsl = 0; fu += gw;
                if (fu < u_hi) setup_fetch(fu, nb_, nct, noy0, nox0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < IPW; ++it) dma16(zp, lds_offset(dummy));
        }
        ++fg;
    };
    // Wave specialisation: issuing an LDS-DMA piece costs the issuing wave ~100-180 cycles (MI355X_MICROARCH.md),
    // 8 pieces per stage stalled both MFMA waves of a SIMD at once.  The 4 loader waves (one per SIMD) issue every
    // DMA; the 8 MFMA waves never touch VMEM inside the loop.  All 12 waves meet at one s_barrier per stage.
    if (loader) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) advance_fetch();
    }

    const int wsw = (r >> 3) & 1;
    int g = 0;
    for (int un = 0; un < my_units; ++un) {
        {   // coordinates of the unit being computed (the fetch cursor may be several units ahead)
            const int uc = u + un * gw;
            const int ct = uc % p.ctiles; int t = uc / p.ctiles;
            const int tx = t % p.tiles_x; t /= p.tiles_x;
            const int ty = t % p.tiles_y;
            cb = t / p.tiles_y; cct = ct; coy0 = ty * TH; cox0 = tx * TW;
        }
        f32x16 acc[MT][PT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < PT; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

        for (int sl = 0; sl < nsl; ++sl, ++g) {
            if (loader) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * IPW) : "memory");   // my pieces of stage g landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (loader) { advance_fetch(); continue; }       // stage g + DEPTH -> the slot everyone just left
            const char* lh = smem + (g % NSLOT) * SBYTES;
            const char* lw = lh + HBYTES;
            using frag_t = typename std::conditional<std::is_same<T, float>::value, f32x4, bf16x8>::type;
            frag_t fa[2][MT], fbq[2][PT];
            auto load_tap = [&](int tap, frag_t (&a)[MT], frag_t (&bq)[PT]) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[m] = *(const frag_t*)(lw + (tap * COT + m * 32 + r) * SLB + ((h ^ wsw) * 16));
#pragma unroll
                for (int q = 0; q < PT; ++q) {
                    const int lp = (wave * PT + q + ky) * IWT + r + kx;
                    bq[q] = *(const frag_t*)(lh + lp * SLB + ((h ^ ((lp >> 3) & 1)) * 16));
                }
            };
            load_tap(0, fa[0], fbq[0]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fbq[(tap + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int q = 0; q < PT; ++q)
                                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tap & 1][m][j], fbq[tap & 1][q][j], acc[m][q], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < PT; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap & 1][m], fbq[tap & 1][q], acc[m][q], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: the slot of the last computed stage is free once every wave has left it
        if (p.vec16) {
            __builtin_amdgcn_s_barrier();                    // (loader waves join: uniform barrier count)
            asm volatile("" ::: "memory");
            if (loader) continue;
            char* tsp = smem + ((g - 1) % NSLOT) * SBYTES + wave * (16 * ERS);
#pragma unroll
            for (int q = 0; q < PT; ++q)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    conv_epilogue_lds_half<T, MT, PT>(p, acc, q, half, tsp, cb, cct, coy0 + wave * PT + q, cox0, lane);
            // the next iteration's barrier orders these LDS reads before the slot is refilled: wait for them here
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (!loader) {
            conv_epilogue<T, MT, PT>(p, acc, cb, cct, coy0 + wave * PT, cox0, r, h);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // drain the trailing dummy DMAs before the wave ends
}

template <typename T, int MT>
static int launch_pipe(const ConvP& p, int ctiles, hipStream_t st) {
    constexpr int HB = ((18 * 34 * 32 + 1023) / 1024) * 1024, WB = ((9 * 32 * MT * 32 + 1023) / 1024) * 1024;
    constexpr size_t SMEM = 4 * ((size_t)HB + WB) + 4 * 1024;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_pipe_k<T, MT>;
    static bool attr_set = false;
    static int ncu = 0;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        int dev = 0; hipDeviceProp_t prop;
        SG_HIP(hipGetDevice(&dev)); SG_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount;
        attr_set = true;
    }
    ConvP q = p;
    q.tiles_x = cdiv(p.OW, 32);
    q.tiles_y = cdiv(p.OH, 16);
    q.ctiles = ctiles;
    const size_t nunits = (size_t)q.tiles_x * q.tiles_y * p.B * ctiles;
    size_t nwg = (size_t)ncu;
    if (nwg > nunits) nwg = nunits;
    char cls[96];
    snprintf(cls, sizeof(cls), "conv3x3_pipe<%s,MT%d>", sizeof(T) == 4 ? "f32" : "bf16", MT);
    const double px = (double)p.B * p.OH * p.OW;
    const int tok = sg_prof_start(cls, 2.0 * px * 9 * p.Cin * p.Cout, ((double)p.B * p.H * p.W * p.Cin + px * p.Cout) * sizeof(T), st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(768), SMEM, st, q);
    sg_prof_stop(tok, st);
    SG_LAUNCH_CHECK();
    return 0;
}

int sg_conv3x3_pipe(const ConvP& p, int dtype, hipStream_t st) {
    if (p.Cout <= 32) return dtype == SRCGAN_F32 ? launch_pipe<float, 1>(p, 1, st) : launch_pipe<__bf16, 1>(p, 1, st);
    const int ctiles = cdiv(p.Cout, 64);
    return dtype == SRCGAN_F32 ? launch_pipe<float, 2>(p, ctiles, st) : launch_pipe<__bf16, 2>(p, ctiles, st);
}
