// Bandwidth-bound kernels of the SRCGAN hot path (gfx950): layout changes, weight packing,
// per-channel reductions (bias grad, BatchNorm statistics), BatchNorm+LeakyReLU apply / backward,
// residual-gradient joins, L1 / MSE / lsgan loss reductions and the in-step preprocessing.
// All reductions are two-stage and order-fixed (deterministic); wavefront (64-lane) shuffles do
// the in-wave part.
#include "common.h"

static thread_local char g_err[512] = "";
void srcgan_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* srcgan_last_error(void) { return g_err; }
extern "C" int srcgan_version(void) { return 100; }
extern "C" int srcgan_dtype_size(int dtype) { return dtype == SRCGAN_F32 ? 4 : (sg_is16(dtype) ? 2 : 0); }

// --------------------------------------------------------------------------- launch profiling
#include <vector>
#include <string>
#include <map>
namespace {
struct ProfRec { std::string cls; double flops, bytes; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
struct ProfAgg { std::string cls; long count; double ms, flops, bytes; };
std::vector<ProfAgg> g_agg;
}
int sg_prof_start(const char* cls, double flops, double bytes, hipStream_t st) {
    if (!g_prof_on) return -1;
    ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    (void)hipEventRecord(r.e0, st);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void sg_prof_stop(int token, hipStream_t st) {
    if (token >= 0 && token < (int)g_prof.size()) (void)hipEventRecord(g_prof[token].e1, st);
}
extern "C" int srcgan_prof_enable(int on) { g_prof_on = on != 0; return 0; }
// Synchronises, aggregates the recorded launches per kernel class and clears the log.  Returns #classes.
extern "C" int srcgan_prof_collect(void) {
    std::map<std::string, ProfAgg> m;
    for (auto& r : g_prof) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        ProfAgg& a = m[r.cls];
        a.cls = r.cls; a.count += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    g_agg.clear();
    for (auto& kv : m) g_agg.push_back(kv.second);
    return (int)g_agg.size();
}
extern "C" int srcgan_prof_get(int i, const char** cls, long* count, double* ms, double* flops, double* bytes) {
    SG_REQUIRE(i >= 0 && i < (int)g_agg.size(), "srcgan_prof_get: index out of range");
    if (cls) *cls = g_agg[i].cls.c_str();
    if (count) *count = g_agg[i].count;
    if (ms) *ms = g_agg[i].ms;
    if (flops) *flops = g_agg[i].flops;
    if (bytes) *bytes = g_agg[i].bytes;
    return 0;
}

#define DISPATCH_DTYPE(dtype, ...) \
    if ((dtype) == SRCGAN_F32) { using T = float; __VA_ARGS__; } \
    else if ((dtype) == SRCGAN_BF16) { using T = __bf16; __VA_ARGS__; } \
    else if ((dtype) == SRCGAN_F16) { using T = _Float16; __VA_ARGS__; } \
    else SG_FAIL("bad dtype %d", (int)(dtype));

static inline int ew_blocks(long n, int per_block = 256) {
    long b = cdivl(n, per_block);
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// --------------------------------------------------------------------------- layout
// NCHW f32 -> NHWC T.  A block transposes a [C][64 pixels] panel through LDS so both sides coalesce.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_k(const float* __restrict__ src, T* __restrict__ dst,
                                                      int C, long HW, int cs, long npanels) {
    __shared__ float tile[64 * 33];
    for (long pn = blockIdx.x; pn < npanels; pn += gridDim.x) {
        const long panels_per_img = (HW + 63) / 64;
        const long b = pn / panels_per_img, p0 = (pn % panels_per_img) * 64;
        for (int c0 = 0; c0 < cs; c0 += 32) {
            __syncthreads();
            for (int e = threadIdx.x; e < 32 * 64; e += 256) {
                const int c = c0 + e / 64, px = e % 64;
                float v = 0.f;
                if (c < C && p0 + px < HW) v = src[((size_t)b * C + c) * HW + p0 + px];
                tile[px * 33 + e / 64] = v;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 32 * 64; e += 256) {
                const int px = e / 32, cl = e % 32;
                if (c0 + cl < cs && p0 + px < HW)
                    dst[((size_t)b * HW + p0 + px) * cs + c0 + cl] = from_f<T>(tile[px * 33 + cl]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_k(const T* __restrict__ src, float* __restrict__ dst,
                                                      int C, long HW, int cs, int coff, long npanels) {
    __shared__ float tile[64 * 33];
    for (long pn = blockIdx.x; pn < npanels; pn += gridDim.x) {
        const long panels_per_img = (HW + 63) / 64;
        const long b = pn / panels_per_img, p0 = (pn % panels_per_img) * 64;
        for (int c0 = 0; c0 < C; c0 += 32) {
            __syncthreads();
            for (int e = threadIdx.x; e < 32 * 64; e += 256) {
                const int px = e / 32, cl = e % 32;
                float v = 0.f;
                if (c0 + cl < C && p0 + px < HW) v = to_f(src[((size_t)b * HW + p0 + px) * cs + coff + c0 + cl]);
                tile[px * 33 + cl] = v;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 32 * 64; e += 256) {
                const int c = c0 + e / 64, px = e % 64;
                if (c < C && p0 + px < HW) dst[((size_t)b * C + c) * HW + p0 + px] = tile[px * 33 + e / 64];
            }
        }
    }
}

// Image-channel fast paths (C <= 8 channels in 8-channel bf16 records, HW % 4 == 0): a thread moves 4 consecutive
// pixels -- one float4 per plane on the NCHW side, 4 x 16 B = 64 contiguous bytes on the NHWC side.
template <typename T16>
__global__ __launch_bounds__(256) void nchw_to_nhwc8_k(const float* __restrict__ src, T16* __restrict__ dst, int C, long HW, long nquad) {
    typedef __attribute__((ext_vector_type(8))) T16 rec8;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (long)gridDim.x * 256) {
        const long b = q / (HW / 4), p0 = (q % (HW / 4)) * 4;
        float4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            v[c] = c < C ? *(const float4*)(src + ((size_t)b * C + c) * HW + p0) : make_float4(0.f, 0.f, 0.f, 0.f);
        rec8 r[4];
#pragma unroll
        for (int c = 0; c < 8; ++c) { r[0][c] = from_f<T16>(v[c].x); r[1][c] = from_f<T16>(v[c].y); r[2][c] = from_f<T16>(v[c].z); r[3][c] = from_f<T16>(v[c].w); }
        rec8* o = (rec8*)(dst + ((size_t)b * HW + p0) * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = r[j];
    }
}

template <typename T16>
__global__ __launch_bounds__(256) void nhwc8_to_nchw_k(const T16* __restrict__ src, float* __restrict__ dst, int C, long HW, long nquad) {
    typedef __attribute__((ext_vector_type(8))) T16 rec8;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (long)gridDim.x * 256) {
        const long b = q / (HW / 4), p0 = (q % (HW / 4)) * 4;
        const rec8* in = (const rec8*)(src + ((size_t)b * HW + p0) * 8);
        rec8 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = in[j];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) *(float4*)(dst + ((size_t)b * C + c) * HW + p0) = make_float4(to_f(r[0][c]), to_f(r[1][c]), to_f(r[2][c]), to_f(r[3][c]));
    }
}

// Space-to-depth forms of an image-channel tensor, for a 4x4 stride-2 pad-1 first layer (model/model.py:612): block (j, i)
// of the (H/2+1) x (W/2+1) grid holds the 2x2 pixels (2j-1+dy, 2i-1+dx) as a 32-channel record [dy][dx][8] (zero outside
// the image and past C), so the layer is a 2x2 stride-1 convolution over K = 4 x 32 values with no padded K: the 8-channel
// NHWC form carries 24 zero channels per 64-byte K chunk through the matrix pipe.  A thread moves one block.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_s2d8_k(const float* __restrict__ src, T* __restrict__ dst, int C, int H, int W, long total) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int BH = H / 2 + 1, BW = W / 2 + 1;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int i = (int)(q % BW); const long t = q / BW; const int j = (int)(t % BH); const long b = t / BH;
        T rec[32];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int y = 2 * j - 1 + (s >> 1), x = 2 * i - 1 + (s & 1);
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rec[s * 8 + c] = from_f<T>((in && c < C) ? src[(((size_t)b * C + c) * H + y) * W + x] : 0.f);
        }
        vecT* o = (vecT*)(dst + (size_t)q * 32);
#pragma unroll
        for (int v = 0; v < 32 / EPP; ++v) {
            vecT t2;
#pragma unroll
            for (int e = 0; e < EPP; ++e) t2[e] = rec[v * EPP + e];
            o[v] = t2;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void s2d8_to_nchw_k(const T* __restrict__ src, float* __restrict__ dst, int C, int H, int W, long total) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int BH = H / 2 + 1, BW = W / 2 + 1;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const int i = (int)(q % BW); const long t = q / BW; const int j = (int)(t % BH); const long b = t / BH;
        const vecT* in = (const vecT*)(src + (size_t)q * 32);
        T rec[32];
#pragma unroll
        for (int v = 0; v < 32 / EPP; ++v) {
            const vecT t2 = in[v];
#pragma unroll
            for (int e = 0; e < EPP; ++e) rec[v * EPP + e] = t2[e];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int y = 2 * j - 1 + (s >> 1), x = 2 * i - 1 + (s & 1);
            if (y < 0 || y >= H || x < 0 || x >= W) continue;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) dst[(((size_t)b * C + c) * H + y) * W + x] = to_f(rec[s * 8 + c]);
        }
    }
}

// gradient of the folded first-layer weight [Cout][32 = (dy,dx,c8)][2][2] -> canonical [Cout][Cin][4][4]
__global__ __launch_bounds__(256) void s2d_wgrad_unfold_k(const float* __restrict__ gf, float* __restrict__ g, int Cout, int Cin, int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Cout * Cin * 16) return;
    const int kx = e & 3, ky = (e >> 2) & 3, c = (e >> 4) % Cin, co = (e >> 4) / Cin;
    const float v = gf[(((size_t)co * 32 + ((ky & 1) * 2 + (kx & 1)) * 8 + c) * 2 + (ky >> 1)) * 2 + (kx >> 1)];
    g[e] = accumulate ? g[e] + v : v;
}

extern "C" int srcgan_nchw_f32_to_s2d(const float* src, void* dst, int B, int C, int H, int W, int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && C <= 8 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "srcgan_nchw_f32_to_s2d: needs 1..8 channels and even H, W");
    const long total = (long)B * (H / 2 + 1) * (W / 2 + 1);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(nchw_to_s2d8_k<T>, dim3(ew_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, C, H, W, total));
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_s2d_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W, int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && C <= 8 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "srcgan_s2d_to_nchw_f32: needs 1..8 channels and even H, W");
    const long total = (long)B * (H / 2 + 1) * (W / 2 + 1);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(s2d8_to_nchw_k<T>, dim3(ew_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)src, dst, C, H, W, total));
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_s2d_wgrad_unfold(const float* gfold, float* grad, int Cout, int Cin, int accumulate, void* stream) {
    SG_REQUIRE(gfold && grad && Cout > 0 && Cin > 0 && Cin <= 8, "srcgan_s2d_wgrad_unfold: bad arguments");
    hipLaunchKernelGGL(s2d_wgrad_unfold_k, dim3(cdiv(Cout * Cin * 16, 256)), dim3(256), 0, (hipStream_t)stream, gfold, grad, Cout, Cin, accumulate);
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int cs, int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && cs >= C, "srcgan_nchw_f32_to_nhwc: bad arguments");
    const long HW = (long)H * W, np = (long)B * cdivl(HW, 64);
    if (sg_is16(dtype) && cs == 8 && HW % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
        const long nq = (long)B * (HW / 4);
        if (dtype == SRCGAN_F16) hipLaunchKernelGGL(nchw_to_nhwc8_k<_Float16>, dim3(ew_blocks(nq, 256)), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, C, HW, nq);
        else hipLaunchKernelGGL(nchw_to_nhwc8_k<__bf16>, dim3(ew_blocks(nq, 256)), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, C, HW, nq);
        SG_LAUNCH_CHECK();
        return 0;
    }
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(nchw_to_nhwc_k<T>, dim3(ew_blocks(np, 1)), dim3(256), 0, (hipStream_t)stream,
                                             src, (T*)dst, C, HW, cs, np));
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_nhwc_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W, int cs, int coff, int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && cs >= C + coff, "srcgan_nhwc_to_nchw_f32: bad arguments");
    const long HW = (long)H * W, np = (long)B * cdivl(HW, 64);
    if (sg_is16(dtype) && cs == 8 && coff == 0 && HW % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
        const long nq = (long)B * (HW / 4);
        if (dtype == SRCGAN_F16) hipLaunchKernelGGL(nhwc8_to_nchw_k<_Float16>, dim3(ew_blocks(nq, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst, C, HW, nq);
        else hipLaunchKernelGGL(nhwc8_to_nchw_k<__bf16>, dim3(ew_blocks(nq, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, dst, C, HW, nq);
        SG_LAUNCH_CHECK();
        return 0;
    }
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(nhwc_to_nchw_k<T>, dim3(ew_blocks(np, 1)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)src, dst, C, HW, cs, coff, np));
    SG_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- weight packing
// Wp[rt][chunk][tap][row (COT)][k (KCE)].  COT = rows<=32 ? 32 : 64 (must match conv_igemm).
// A call fills the k-range [k_off, k_off+kdim) of rows [0,rows) of a packed matrix whose full K is k_total
// (composite matrices -- the dense-block backward -- are assembled by several calls); with k_off == 0 &&
// kdim == k_total the whole padded buffer is written (zero padding included).
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_k(const float* __restrict__ w, T* __restrict__ wp, int rows, int kdim,
                                                     int tys, int txs, long sr, long sk, long sty, long stx, long off,
                                                     int cot, int nchunk, long total, int k_off, int k_total, float scale) {
    constexpr int KCE = DT<T>::KCE;
    const int ntap = tys * txs;
    const bool whole = (k_off == 0 && kdim == k_total);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        long q = e;
        const int kl = (int)(q % KCE); q /= KCE;
        const int rl = (int)(q % cot); q /= cot;
        const int tap = (int)(q % ntap); q /= ntap;
        const int ch = (int)(q % nchunk);
        const int rt = (int)(q / nchunk);
        const int row = rt * cot + rl, kk = ch * KCE + kl - k_off;
        const bool in = row < rows && kk >= 0 && kk < kdim;
        if (in) wp[e] = from_f<T>(scale * w[off + row * sr + kk * sk + (tap / txs) * sty + (tap % txs) * stx]);
        else if (whole) wp[e] = from_f<T>(0.f);
    }
}

// Batched packing: one launch for every weight tensor of a network pass (a 23-block generator has 345 forward and
// 1035 composite backward packs; one launch each instead of ~1400 x 4 us).  Block b serves job j with
// jobs[j].blk0 <= b < jobs[j+1].blk0 (binary search); destinations are offsets from `wp_base` so the table can be
// cached across calls while the workspace moves.
// guard: {fingerprint the pack was made from, fingerprint of the parameters now} (srcgan_params_fingerprint); equal -> the persistent
// pack is still valid and the launch does nothing.  The decision is taken on the DEVICE: no host synchronisation, and a weight update
// the host cannot see (p.data.mul_(), a raw-pointer write) is still noticed.
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_k(const SgPackJob* __restrict__ jobs, int njobs, char* wp_base, const unsigned long long* __restrict__ guard) {
    constexpr int KCE = DT<T>::KCE;
    if (guard && guard[0] == guard[1]) return;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].blk0 <= (long)blockIdx.x) lo = mid; else hi = mid - 1; }
    const SgPackJob j = jobs[lo];
    const long e = ((long)blockIdx.x - j.blk0) * 256 + threadIdx.x;
    if (e >= j.total) return;
    T* wp = (T*)(wp_base + j.wp_off);
    const int ntap = j.tys * j.txs;
    long q = e;
    const int kl = (int)(q % KCE); q /= KCE;
    const int rl = (int)(q % j.cot); q /= j.cot;
    const int tap = (int)(q % ntap); q /= ntap;
    const int ch = (int)(q % j.nchunk);
    const int rt = (int)(q / j.nchunk);
    const int row = rt * j.cot + rl, kk = ch * KCE + kl - j.k_off;
    const bool in = row < j.rows && kk >= 0 && kk < j.kdim;
    if (in) wp[e] = from_f<T>(j.scale * j.w[j.off + row * j.sr + kk * j.sk + (tap / j.txs) * j.sty + (tap % j.txs) * j.stx]);
    else if (j.k_off == 0 && j.kdim == j.k_total) wp[e] = from_f<T>(0.f);
}

void sg_pack_job_finish(SgPackJob& j, int dtype, long& blk_cursor) {
    const int esz = dtype == SRCGAN_F32 ? 4 : 2, kce = 64 / esz;
    j.cot = j.rows <= 32 ? 32 : 64;
    j.nchunk = cdiv(j.k_total, kce);
    j.total = (long)cdiv(j.rows, j.cot) * j.nchunk * j.tys * j.txs * j.cot * kce;
    j.blk0 = blk_cursor;
    blk_cursor += cdivl(j.total, 256);
}

int sg_pack_multi_launch(const SgPackJob* jobs_dev, int njobs, long nblocks, void* wp_base, int dtype, hipStream_t st, const unsigned long long* guard) {
    if (njobs <= 0) return 0;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(pack_multi_k<T>, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_dev, njobs, (char*)wp_base, guard));
    SG_LAUNCH_CHECK();
    return 0;
}

// zero fill that obeys the same guard as the pack it prepares (a plain memset would wipe a pack the guard then leaves alone)
__global__ __launch_bounds__(256) void fill_zero_guarded_k(uint4* __restrict__ p, long n16, const unsigned long long* __restrict__ guard) {
    if (guard && guard[0] == guard[1]) return;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
int sg_fill_zero_guarded(void* p, size_t bytes, const unsigned long long* guard, hipStream_t st) {
    if (!guard) { SG_HIP(hipMemsetAsync(p, 0, bytes, st)); return 0; }
    SG_REQUIRE(((uintptr_t)p % 16) == 0 && bytes % 16 == 0, "guarded fill: 16-byte granularity");
    const long n16 = (long)(bytes / 16);
    hipLaunchKernelGGL(fill_zero_guarded_k, dim3(ew_blocks(n16)), dim3(256), 0, st, (uint4*)p, n16, guard);
    SG_LAUNCH_CHECK();
    return 0;
}

// 64-bit fingerprint of a parameter list: sum over tensors t and elements i of mix(bits(x[t][i]), i, t).  The sum is order independent
// (atomic adds commute), every bit of every element takes part, and a changed element changes the sum unless 2^-64 luck intervenes.
// Two independent 32-bit avalanche mixes (murmur3's finaliser on differently keyed words) make the two halves; 64-bit multiplies
// would cost ~16 VALU each.  A block serves a 16 Ki-element slice with 8 loads in flight per thread: the first version (one
// dependent load per iteration, 64 Ki-element slices) was latency-bound at 120 us per call for 66 MB -- 1 ms of a 138 ms step.
__device__ __forceinline__ unsigned sg_fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
#define SG_FP_SLICE 16384
__global__ __launch_bounds__(256) void params_fingerprint_k(const long* __restrict__ table, int ntens, unsigned long long* __restrict__ out) {
    // block -> (tensor, slice): binary search over the table's cumulative block counts (third column)
    int lo = 0, hi = ntens - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[3 * mid + 2] <= (long)blockIdx.x) lo = mid; else hi = mid - 1; }
    const unsigned* __restrict__ p = (const unsigned*)table[3 * lo];
    const long n = table[3 * lo + 1], i0 = ((long)blockIdx.x - table[3 * lo + 2]) * SG_FP_SLICE;
    const long i1 = i0 + SG_FP_SLICE < n ? i0 + SG_FP_SLICE : n;
    const unsigned salt_a = sg_fmix32(0x9E3779B9u * (unsigned)(lo + 1)), salt_b = sg_fmix32(0x7F4A7C15u + (unsigned)lo);
    unsigned long long ha = 0, hb = 0;
    for (long base = i0 + threadIdx.x; base < i1; base += 256 * 8) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const long i = base + k * 256; v[k] = i < i1 ? p[i] : 0u; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long i = base + k * 256;
            if (i < i1) {
                const unsigned iu = (unsigned)i;
                ha += sg_fmix32((v[k] ^ salt_a) + iu * 0x9E3779B1u);
                hb += sg_fmix32((v[k] + salt_b) ^ (iu * 0x85EBCA77u + 0x165667B1u));
            }
        }
    }
    unsigned long long h = (hb << 32) + ha + (hb >> 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
extern "C" int srcgan_params_fingerprint(const void* table_dev, int ntensors, long nblocks, void* out_u64, void* stream) {
    SG_REQUIRE(table_dev && out_u64 && ntensors > 0 && nblocks > 0, "srcgan_params_fingerprint: bad arguments");
    SG_HIP(hipMemsetAsync(out_u64, 0, 8, (hipStream_t)stream));
    hipLaunchKernelGGL(params_fingerprint_k, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const long*)table_dev, ntensors, (unsigned long long*)out_u64);
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t srcgan_packed_weight_bytes(int rows, int kdim, int ntaps, int dtype) {
    const int esz = dtype == SRCGAN_F32 ? 4 : 2, kce = 64 / esz;
    const int cot = rows <= 32 ? 32 : 64;
    return (size_t)cdiv(rows, cot) * cdiv(kdim, kce) * ntaps * cot * 64;
}

extern "C" int srcgan_pack_weight_part(const float* w, void* wp, int rows, int kdim, int tys, int txs,
                                       long sr, long sk, long sty, long stx, long off, int k_off, int k_total, float scale,
                                       int dtype, void* stream) {
    SG_REQUIRE(w && wp && rows > 0 && kdim > 0 && tys > 0 && txs > 0, "srcgan_pack_weight: bad arguments");
    SG_REQUIRE(k_off >= 0 && k_off + kdim <= k_total, "srcgan_pack_weight: k range [%d,%d) exceeds k_total %d", k_off, k_off + kdim, k_total);
    const int esz = dtype == SRCGAN_F32 ? 4 : 2, kce = 64 / esz;
    const int cot = rows <= 32 ? 32 : 64, nchunk = cdiv(k_total, kce);
    const long total = (long)cdiv(rows, cot) * nchunk * tys * txs * cot * kce;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(pack_weight_k<T>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                                             w, (T*)wp, rows, kdim, tys, txs, sr, sk, sty, stx, off, cot, nchunk, total,
                                             k_off, k_total, scale));
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_pack_weight(const float* w, void* wp, int rows, int kdim, int tys, int txs,
                                  long sr, long sk, long sty, long stx, long off, int dtype, void* stream) {
    return srcgan_pack_weight_part(w, wp, rows, kdim, tys, txs, sr, sk, sty, stx, off, 0, kdim, 1.f, dtype, stream);
}

// --------------------------------------------------------------------------- column reductions
// Block = 64 channel lanes x 4 pixel lanes.  Stage 1 writes partial[which][blk][c]; stage 2 sums
// the blocks in index order.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void col_reduce_k(const T* __restrict__ a, int aCs, int acoff, const T* __restrict__ z,
                                                    int zCs, int zcoff, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, long npix, int C,
                                                    float* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const long per = cdivl(npix, gridDim.x);
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cx;
        float s0 = 0.f, s1 = 0.f;
        if (c < C) {
            const float mu = (MODE >= 1) ? mean[c] : 0.f;
            const float rs = (MODE == 2) ? rstd[c] : 0.f;
            for (long px = p0 + py; px < p1; px += 4) {
                const float v = to_f(a[(size_t)px * aCs + acoff + c]);
                if (MODE == 0) s0 += v;
                else if (MODE == 1) { const float dv = v - mu; s0 += dv * dv; }
                else { const float zz = to_f(z[(size_t)px * zCs + zcoff + c]); s0 += v; s1 += v * (zz - mu) * rs; }
            }
        }
        red[0][py][cx] = s0; red[1][py][cx] = s1;
        __syncthreads();
        if (py == 0 && c < C) {
            partial[(size_t)blockIdx.x * C + c] = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
            if (MODE == 2)
                partial[((size_t)gridDim.x + blockIdx.x) * C + c] = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
        }
        __syncthreads();
    }
}

// Vectorised variant (C % EPP == 0, (C/EPP) | 256): a thread owns EPP consecutive channels (one 16-byte load per
// pixel) and every 256/(C/EPP)-th pixel, so a wave reads whole contiguous pixel records.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void col_reduce_vec_k(const T* __restrict__ a, int aCs, int acoff, const T* __restrict__ z,
                                                        int zCs, int zcoff, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, long npix, int C,
                                                        float* __restrict__ partial) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    __shared__ float red[2][256 * EPP];
    const int G = C / EPP, PL = 256 / G;
    const int cg = threadIdx.x % G, pl = threadIdx.x / G, c0 = cg * EPP;
    const long per = cdivl(npix, gridDim.x);
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    float s0[EPP], s1[EPP], mu[EPP], rs[EPP];
#pragma unroll
    for (int i = 0; i < EPP; ++i) { s0[i] = 0.f; s1[i] = 0.f; mu[i] = MODE >= 1 ? mean[c0 + i] : 0.f; rs[i] = MODE == 2 ? rstd[c0 + i] : 0.f; }
    // four pixels' loads in flight per thread (one dependent 16-byte load per iteration ran at 1.8-3.4 TB/s: latency-bound); the
    // additions keep the pixel order, so the sums are bit-identical to the one-at-a-time loop
    auto acc1 = [&](const vecT& av, const vecT& zv) {
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < EPP; ++i) { const float v = to_f(av[i]); s0[i] += v; s1[i] += v * (to_f(zv[i]) - mu[i]) * rs[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < EPP; ++i) { const float v = to_f(av[i]) - mu[i]; s0[i] += (MODE == 0) ? v : v * v; }
        }
    };
    long px = p0 + pl;
    for (; px + 3 * (long)PL < p1; px += 4 * (long)PL) {
        vecT av[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = *(const vecT*)(a + (size_t)(px + u * (long)PL) * aCs + acoff + c0);
            if (MODE == 2) zv[u] = *(const vecT*)(z + (size_t)(px + u * (long)PL) * zCs + zcoff + c0);
            else zv[u] = av[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc1(av[u], zv[u]);
    }
    for (; px < p1; px += PL) {
        const vecT av = *(const vecT*)(a + (size_t)px * aCs + acoff + c0);
        vecT zv = av;
        if (MODE == 2) zv = *(const vecT*)(z + (size_t)px * zCs + zcoff + c0);
        acc1(av, zv);
    }
#pragma unroll
    for (int i = 0; i < EPP; ++i) { red[0][threadIdx.x * EPP + i] = s0[i]; red[1][threadIdx.x * EPP + i] = s1[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t0 = 0.f, t1 = 0.f;
        for (int q = 0; q < PL; ++q) { t0 += red[0][(q * G + c / EPP) * EPP + c % EPP]; t1 += red[1][(q * G + c / EPP) * EPP + c % EPP]; }
        partial[(size_t)blockIdx.x * C + c] = t0;
        if (MODE == 2) partial[((size_t)gridDim.x + blockIdx.x) * C + c] = t1;
    }
}

// one block per 8 channels; 32 block-lanes split the partials (4 loads in flight each), fixed order
__global__ __launch_bounds__(256) void col_finalize_k(const float* __restrict__ partial, int nblk, int C, float scale,
                                                      float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float red[2][32][8];
    const int cx = threadIdx.x & 7, py = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        int b = py;
        for (; b + 96 < nblk; b += 128) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0[u] += partial[(size_t)(b + 32 * u) * C + c];
                if (out1) s1[u] += partial[((size_t)nblk + b + 32 * u) * C + c];
            }
        }
        for (; b < nblk; b += 32) {
            s0[0] += partial[(size_t)b * C + c];
            if (out1) s1[0] += partial[((size_t)nblk + b) * C + c];
        }
    }
    red[0][py][cx] = (s0[0] + s0[1]) + (s0[2] + s0[3]); red[1][py][cx] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    __syncthreads();
    if (py == 0 && c < C) {
        float t0 = 0.f, t1 = 0.f;
        for (int q = 0; q < 32; ++q) { t0 += red[0][q][cx]; t1 += red[1][q][cx]; }
        out0[c] = t0 * scale;
        if (out1) out1[c] = t1 * scale;
    }
}

// Mean and variance in ONE pass over the tensor (BatchNorm statistics; mode 3 of srcgan_col_reduce).  The two-pass form (mean, then
// sum of squared deviations) read the convolution's output twice.  Here a thread keeps  S1 = sum (v - s), S2 = sum (v - s)^2  around
// a shift s = ITS OWN first sample, which is within a few standard deviations of the mean whatever the mean is: no cancellation in
// M2 = S2 - S1^2 / n.  Partials (n, mean, M2) are combined exactly (Chan et al.: M2 = M2a + M2b + d^2 na nb / (na + nb)) in a
// fixed order: threads of a block by pixel lane, blocks by index -- deterministic.  partial[blk][c] = mean, partial[nblk + blk][c] = M2;
// a block's count follows from (blk, npix, nblk).
template <typename T>
__global__ __launch_bounds__(256) void col_meanvar_vec_k(const T* __restrict__ a, int aCs, int acoff, long npix, int C, float* __restrict__ partial) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    __shared__ float red[2][256 * EPP];
    __shared__ float cnt[256];
    const int G = C / EPP, PL = 256 / G;
    const int cg = threadIdx.x % G, pl = threadIdx.x / G, c0 = cg * EPP;
    const long per = cdivl(npix, gridDim.x);
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    float s1[EPP], s2[EPP], sh[EPP];
#pragma unroll
    for (int i = 0; i < EPP; ++i) { s1[i] = 0.f; s2[i] = 0.f; sh[i] = 0.f; }
    long px = p0 + pl;
    float n = 0.f;
    if (px < p1) {
        const vecT av = *(const vecT*)(a + (size_t)px * aCs + acoff + c0);
#pragma unroll
        for (int i = 0; i < EPP; ++i) sh[i] = to_f(av[i]);
    }
    auto acc1 = [&](const vecT& av) {
#pragma unroll
        for (int i = 0; i < EPP; ++i) { const float d = to_f(av[i]) - sh[i]; s1[i] += d; s2[i] += d * d; }
    };
    for (; px + 3 * (long)PL < p1; px += 4 * (long)PL) {          // four pixels' loads in flight, additions in pixel order
        vecT av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = *(const vecT*)(a + (size_t)(px + u * (long)PL) * aCs + acoff + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc1(av[u]);
        n += 4.f;
    }
    for (; px < p1; px += PL) { acc1(*(const vecT*)(a + (size_t)px * aCs + acoff + c0)); n += 1.f; }
    const float inv = n > 0.f ? 1.f / n : 0.f;
#pragma unroll
    for (int i = 0; i < EPP; ++i) {
        red[0][threadIdx.x * EPP + i] = sh[i] + s1[i] * inv;                  // thread mean
        red[1][threadIdx.x * EPP + i] = s2[i] - s1[i] * s1[i] * inv;          // thread M2
    }
    cnt[threadIdx.x] = n;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float nn = 0.f, mean = 0.f, m2 = 0.f;
        for (int q = 0; q < PL; ++q) {
            const int t = q * G + c / EPP;
            chan_combine(nn, mean, m2, cnt[t], red[0][t * EPP + c % EPP], red[1][t * EPP + c % EPP]);
        }
        partial[(size_t)blockIdx.x * C + c] = mean;
        partial[((size_t)gridDim.x + blockIdx.x) * C + c] = m2;
    }
}
// one block per 8 channels; 32 block-lanes each combine every 32nd block's partial (loads issued 4 at a time), then one thread per
// channel combines the 32 lane results in lane order: a fixed order, and no 512-long chain of dependent loads and divisions
// (one thread per channel over all blocks took 200 us)
__global__ __launch_bounds__(256) void col_meanvar_finalize_k(const float* __restrict__ partial, int nblk, int C, long npix,
                                                              float* __restrict__ mean_out, float* __restrict__ var_out) {
    __shared__ float red[3][32][8];
    const int cx = threadIdx.x & 7, py = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    const long per = cdivl(npix, (long)nblk);
    auto count = [&](int b) { const long p0 = (long)b * per, p1 = (p0 + per < npix) ? p0 + per : npix; return p1 > p0 ? (float)(p1 - p0) : 0.f; };
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C) {
        int b = py;
        for (; b + 96 < nblk; b += 128) {
            float mb[4], qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { mb[u] = partial[(size_t)(b + 32 * u) * C + c]; qb[u] = partial[((size_t)nblk + b + 32 * u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) chan_combine(n, mean, m2, count(b + 32 * u), mb[u], qb[u]);
        }
        for (; b < nblk; b += 32) chan_combine(n, mean, m2, count(b), partial[(size_t)b * C + c], partial[((size_t)nblk + b) * C + c]);
    }
    red[0][py][cx] = n; red[1][py][cx] = mean; red[2][py][cx] = m2;
    __syncthreads();
    if (py == 0 && c < C) {
        float nn = 0.f, mm = 0.f, qq = 0.f;
        for (int q = 0; q < 32; ++q) chan_combine(nn, mm, qq, red[0][q][cx], red[1][q][cx], red[2][q][cx]);
        mean_out[c] = mm;
        var_out[c] = qq / (float)npix;
    }
}

extern "C" int srcgan_col_reduce_blocks(long npix) {
    long b = cdivl(npix, 256);
    return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

extern "C" int srcgan_col_reduce(int mode, const void* a, int a_cs, int a_coff, const void* z, int z_cs, int z_coff,
                                 const float* m, const float* rstd, long npix, int C, float scale,
                                 float* out0, float* out1, float* scratch, int dtype, void* stream) {
    SG_REQUIRE(a && out0 && scratch && npix > 0 && C > 0, "srcgan_col_reduce: bad arguments");
    SG_REQUIRE(mode >= 0 && mode <= 3, "srcgan_col_reduce: bad mode %d", mode);
    SG_REQUIRE(mode == 0 || mode == 3 || m, "srcgan_col_reduce: mode %d needs mean", mode);
    SG_REQUIRE(mode != 2 || (z && rstd && out1), "srcgan_col_reduce: mode 2 needs z, rstd, out1");
    SG_REQUIRE(mode != 3 || out1, "srcgan_col_reduce: mode 3 needs out1 (variance)");
    const int nblk = srcgan_col_reduce_blocks(npix);
    hipStream_t st = (hipStream_t)stream;
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    const bool vec = C % epp == 0 && 256 % (C / epp) == 0 && a_cs % epp == 0 && a_coff % epp == 0 && ((uintptr_t)a % 16) == 0 &&
                     (mode != 2 || (z_cs % epp == 0 && z_coff % epp == 0 && ((uintptr_t)z % 16) == 0));
    if (mode == 3) {        // mean (out0) and biased variance (out1) of every channel; `scale` is not used
        if (!vec) {         // odd channel counts: the two-pass form
            SG_TRY(srcgan_col_reduce(0, a, a_cs, a_coff, nullptr, 0, 0, nullptr, nullptr, npix, C, 1.f / (float)npix, out0, nullptr, scratch, dtype, stream));
            return srcgan_col_reduce(1, a, a_cs, a_coff, nullptr, 0, 0, out0, nullptr, npix, C, 1.f / (float)npix, out1, nullptr, scratch, dtype, stream);
        }
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(col_meanvar_vec_k<T>, dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, npix, C, scratch));
        SG_LAUNCH_CHECK();
        hipLaunchKernelGGL(col_meanvar_finalize_k, dim3(cdiv(C, 8)), dim3(256), 0, st, scratch, nblk, C, npix, out0, out1);
        SG_LAUNCH_CHECK();
        return 0;
    }
    if (vec) {
        DISPATCH_DTYPE(dtype, {
            if (mode == 0) hipLaunchKernelGGL((col_reduce_vec_k<T, 0>), dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, (const T*)z, z_cs, z_coff, m, rstd, npix, C, scratch);
            else if (mode == 1) hipLaunchKernelGGL((col_reduce_vec_k<T, 1>), dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, (const T*)z, z_cs, z_coff, m, rstd, npix, C, scratch);
            else hipLaunchKernelGGL((col_reduce_vec_k<T, 2>), dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, (const T*)z, z_cs, z_coff, m, rstd, npix, C, scratch);
        });
    } else
    DISPATCH_DTYPE(dtype, {
        if (mode == 0) hipLaunchKernelGGL((col_reduce_k<T, 0>), dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, (const T*)z, z_cs, z_coff, m, rstd, npix, C, scratch);
        else if (mode == 1) hipLaunchKernelGGL((col_reduce_k<T, 1>), dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, (const T*)z, z_cs, z_coff, m, rstd, npix, C, scratch);
        else hipLaunchKernelGGL((col_reduce_k<T, 2>), dim3(nblk), dim3(256), 0, st, (const T*)a, a_cs, a_coff, (const T*)z, z_cs, z_coff, m, rstd, npix, C, scratch);
    });
    SG_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_finalize_k, dim3(cdiv(C, 8)), dim3(256), 0, st, scratch, nblk, C, scale, out0, mode == 2 ? out1 : nullptr);
    SG_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- BatchNorm + LeakyReLU
__global__ void bn_finalize_k(const float* mean, const float* var, float* rstd, float* rmean, float* rvar,
                              int64_t* nbt, int C, float unbias, float momentum, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    rstd[c] = rsqrtf(var[c] + eps);
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * var[c] * unbias;
}

extern "C" int srcgan_bn_finalize(const float* mean, const float* var, float* rstd, float* running_mean, float* running_var,
                                  int64_t* nbt, int C, long count, float momentum, float eps, void* stream) {
    SG_REQUIRE(mean && var && rstd && C > 0 && count > 0, "srcgan_bn_finalize: bad arguments");
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    hipLaunchKernelGGL(bn_finalize_k, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var, rstd,
                       running_mean, running_var, nbt, C, unbias, momentum, eps);
    SG_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_eval_rstd_k(const float* rvar, float* rstd, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) rstd[c] = rsqrtf(rvar[c] + eps);
}
extern "C" int srcgan_bn_eval_rstd(const float* running_var, float* rstd, int C, float eps, void* stream) {
    SG_REQUIRE(running_var && rstd && C > 0, "srcgan_bn_eval_rstd: bad arguments");
    hipLaunchKernelGGL(bn_eval_rstd_k, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_var, rstd, C, eps);
    SG_LAUNCH_CHECK();
    return 0;
}

// A thread owns EPP consecutive channels (their BN constants live in registers) and walks pixels: 16-byte
// accesses, whole pixel records per wave, no per-element modulo or constant reloads.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_k(const T* __restrict__ z, T* __restrict__ y, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, long npix, int C, float slope) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int G = C / EPP, PL = 256 / G;
    const int c0 = (threadIdx.x % G) * EPP, pl = threadIdx.x / G;
    float sc[EPP], sh[EPP];
#pragma unroll
    for (int i = 0; i < EPP; ++i) { sc[i] = rstd[c0 + i] * gamma[c0 + i]; sh[i] = beta[c0 + i] - mean[c0 + i] * sc[i]; }
    for (long px = (long)blockIdx.x * PL + pl; px < npix; px += (long)gridDim.x * PL) {
        const vecT v = *(const vecT*)(z + (size_t)px * C + c0);
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) { const float u = to_f(v[i]) * sc[i] + sh[i]; o[i] = from_f<T>(u > 0.f ? u : u * slope); }
        *(vecT*)(y + (size_t)px * C + c0) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const T* __restrict__ g, const T* __restrict__ z, T* __restrict__ dz,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                      const float* __restrict__ sum_gx, long npix, int C, float invn) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int G = C / EPP, PL = 256 / G;
    const int c0 = (threadIdx.x % G) * EPP, pl = threadIdx.x / G;
    float ka[EPP], kb[EPP], kc[EPP], mu[EPP];          // dz = ka * (g - kb - (z - mu) * kc)
#pragma unroll
    for (int i = 0; i < EPP; ++i) {
        ka[i] = gamma[c0 + i] * rstd[c0 + i]; kb[i] = sum_g[c0 + i] * invn;
        kc[i] = rstd[c0 + i] * sum_gx[c0 + i] * invn; mu[i] = mean[c0 + i];
    }
    for (long px = (long)blockIdx.x * PL + pl; px < npix; px += (long)gridDim.x * PL) {
        const vecT gv = *(const vecT*)(g + (size_t)px * C + c0);
        const vecT zv = *(const vecT*)(z + (size_t)px * C + c0);
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(ka[i] * (to_f(gv[i]) - kb[i] - (to_f(zv[i]) - mu[i]) * kc[i]));
        *(vecT*)(dz + (size_t)px * C + c0) = o;
    }
}

extern "C" int srcgan_bn_apply_lrelu(const void* z, void* y, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, long npix, int C, int cs, float slope, int dtype, void* stream) {
    SG_REQUIRE(z && y && mean && rstd && gamma && beta && npix > 0, "srcgan_bn_apply_lrelu: bad arguments");
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(cs == C && C % epp == 0 && 256 % (C / epp) == 0,
               "srcgan_bn_apply_lrelu: needs a dense NHWC tensor with C a multiple of %d and C/%d dividing 256 (C=%d cs=%d)", epp, epp, C, cs);
    const int pl = 256 / (C / epp);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_apply_k<T>, dim3(ew_blocks(npix, pl * 4)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)z, (T*)y, mean, rstd, gamma, beta, npix, C, slope));
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_bn_bwd_apply(const void* g, const void* z, void* dz, const float* mean, const float* rstd,
                                   const float* gamma, const float* sum_g, const float* sum_gx, long npix, int C, int cs,
                                   int dtype, void* stream) {
    SG_REQUIRE(g && z && dz && mean && rstd && gamma && sum_g && sum_gx && npix > 0, "srcgan_bn_bwd_apply: bad arguments");
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(cs == C && C % epp == 0 && 256 % (C / epp) == 0,
               "srcgan_bn_bwd_apply: needs a dense NHWC tensor with C a multiple of %d and C/%d dividing 256", epp, epp);
    const int pl = 256 / (C / epp);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_bwd_apply_k<T>, dim3(ew_blocks(npix, pl * 4)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)g, (const T*)z, (T*)dz, mean, rstd, gamma, sum_g, sum_gx, npix, C,
                                             1.f / (float)npix));
    SG_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- residual-gradient join
template <typename T>
__device__ __forceinline__ size_t ew_chan_off(int c, long plane) {
    constexpr int KCE = DT<T>::KCE;
    return (size_t)(c / KCE) * plane + (size_t)(c % KCE) * sizeof(T);
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_k(char* __restrict__ y, long ypix, long yplane, int ycoff, const char* __restrict__ x,
                                                     long xpix, long xplane, int xcoff, const char* __restrict__ mz, long mzpix,
                                                     long mzplane, int mzcoff, float mslope, long npix, int C4) {
    const long total = npix * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long px = e / C4; const int c = (int)(e % C4) * 4;
        T* yp = (T*)(y + px * ypix + ew_chan_off<T>(ycoff + c, yplane));
        float a[4], b[4];
        load4<T>(yp, a);
        load4<T>((const T*)(x + px * xpix + ew_chan_off<T>(xcoff + c, xplane)), b);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += b[i];
        if (mz) {
            float z[4];
            load4<T>((const T*)(mz + px * mzpix + ew_chan_off<T>(mzcoff + c, mzplane)), z);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] *= (z[i] > 0.f ? 1.f : mslope);
        }
        store4<T>(yp, a);
    }
}

extern "C" int srcgan_add_inplace_planes(void* y, int y_cs, int y_coff, long y_plane, const void* x, int x_cs, int x_coff, long x_plane,
                                         const void* mz, int mz_cs, int mz_coff, long mz_plane, float mslope, long npix, int C,
                                         int dtype, void* stream) {
    SG_REQUIRE(y && x && npix > 0 && C > 0, "srcgan_add_inplace: bad arguments");
    SG_REQUIRE(C % 4 == 0 && y_cs % 4 == 0 && y_coff % 4 == 0 && x_cs % 4 == 0 && x_coff % 4 == 0 &&
               (!mz || (mz_cs % 4 == 0 && mz_coff % 4 == 0)),
               "srcgan_add_inplace: channel counts/strides/offsets must be multiples of 4");
    const int esz = dtype == SRCGAN_F32 ? 4 : 2;
    auto pl = [](long v) { return v ? v : 64L; };
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(add_inplace_k<T>, dim3(ew_blocks(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                                             (char*)y, (long)y_cs * esz, pl(y_plane), y_coff, (const char*)x, (long)x_cs * esz, pl(x_plane), x_coff,
                                             (const char*)mz, (long)mz_cs * esz, pl(mz_plane), mz_coff, mslope, npix, C / 4));
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_add_inplace(void* y, int y_cs, int y_coff, const void* x, int x_cs, int x_coff,
                                  const void* mz, int mz_cs, int mz_coff, float mslope, long npix, int C, int dtype, void* stream) {
    return srcgan_add_inplace_planes(y, y_cs, y_coff, 0, x, x_cs, x_coff, 0, mz, mz_cs, mz_coff, 0, mslope, npix, C, dtype, stream);
}

// --------------------------------------------------------------------------- losses
#define LOSS_BLOCKS 1024
extern "C" int srcgan_loss_scratch_floats(void) { return LOSS_BLOCKS; }

// KIND: 0 |a - b| (L1), 1 (a - b)^2 (MSE), 2 (a - label)^2 (lsgan: MSE against the expanded scalar label, train.py:86-87),
//       3 BCE-with-logits against the scalar label (GANLoss 'vanilla', train.py:88-89: max(x,0) - x t + log1p(exp(-|x|)), torch's
//         stable form), 4 label * a (GANLoss 'wgangp', train.py:121-126: -mean for real, +mean for fake; label = -1 / +1)
template <int KIND>
__device__ __forceinline__ float loss_elem(float x, float t) {
    if (KIND == 0) return fabsf(x - t);
    if (KIND == 3) return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    if (KIND == 4) return x * t;
    const float d = x - t;
    return d * d;
}
template <int KIND>
__device__ __forceinline__ float loss_grad(float x, float t) {       // d elem / d x
    if (KIND == 0) { const float d = x - t; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    if (KIND == 3) return 1.f / (1.f + expf(-x)) - t;
    if (KIND == 4) return t;
    return 2.f * (x - t);
}
template <int KIND>
__global__ __launch_bounds__(256) void loss_fwd_k(const float* __restrict__ a, const float* __restrict__ b, float label,
                                                  long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n / 4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const f32x4 av = *(const f32x4*)(a + e * 4);
        f32x4 bv = {label, label, label, label};
        if (KIND < 2) bv = *(const f32x4*)(b + e * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += loss_elem<KIND>(av[i], bv[i]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {       // tail
        const long e = n4 * 4 + threadIdx.x;
        s += loss_elem<KIND>(a[e], KIND >= 2 ? label : b[e]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void loss_final_k(const float* __restrict__ partial, int nblk, float invn, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((red[0] + red[1]) + (red[2] + red[3])) * invn;
}

extern "C" int srcgan_loss_fwd(int kind, const float* a, const float* b, float label, long n, float* out, float* scratch, void* stream) {
    SG_REQUIRE(a && out && scratch && n > 0 && kind >= 0 && kind <= 4, "srcgan_loss_fwd: bad arguments");
    SG_REQUIRE(kind >= 2 || b, "srcgan_loss_fwd: kind %d needs b", kind);
    SG_REQUIRE(((uintptr_t)a % 16) == 0 && (kind >= 2 || ((uintptr_t)b % 16) == 0), "srcgan_loss_fwd: inputs must be 16-byte aligned");
    long nb = cdivl(n / 4 + 1, 256); if (nb > LOSS_BLOCKS) nb = LOSS_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(loss_fwd_k<0>, dim3((int)nb), dim3(256), 0, st, a, b, label, n, scratch);
    else if (kind == 1) hipLaunchKernelGGL(loss_fwd_k<1>, dim3((int)nb), dim3(256), 0, st, a, b, label, n, scratch);
    else if (kind == 2) hipLaunchKernelGGL(loss_fwd_k<2>, dim3((int)nb), dim3(256), 0, st, a, b, label, n, scratch);
    else if (kind == 3) hipLaunchKernelGGL(loss_fwd_k<3>, dim3((int)nb), dim3(256), 0, st, a, b, label, n, scratch);
    else hipLaunchKernelGGL(loss_fwd_k<4>, dim3((int)nb), dim3(256), 0, st, a, b, label, n, scratch);
    SG_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_k, dim3(1), dim3(256), 0, st, scratch, (int)nb, (float)(1.0 / (double)n), out);
    SG_LAUNCH_CHECK();
    return 0;
}

template <int KIND>
__global__ __launch_bounds__(256) void loss_bwd_k(const float* __restrict__ a, const float* __restrict__ b, float label, long n,
                                                  const float* __restrict__ gout, float gscale, float* __restrict__ da) {
    const float g = gout[0] * gscale;     // upstream gradient * 1/n
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        da[e] = loss_grad<KIND>(a[e], KIND >= 2 ? label : b[e]) * g;
}

extern "C" int srcgan_loss_bwd(int kind, const float* a, const float* b, float label, long n, const float* gout,
                               float gscale, float* da, void* stream) {
    SG_REQUIRE(a && gout && da && n > 0 && kind >= 0 && kind <= 4, "srcgan_loss_bwd: bad arguments");
    SG_REQUIRE(kind >= 2 || b, "srcgan_loss_bwd: kind %d needs b", kind);
    hipStream_t st = (hipStream_t)stream;
    const float gs = gscale / (float)n;
    if (kind == 0) hipLaunchKernelGGL(loss_bwd_k<0>, dim3(ew_blocks(n)), dim3(256), 0, st, a, b, label, n, gout, gs, da);
    else if (kind == 1) hipLaunchKernelGGL(loss_bwd_k<1>, dim3(ew_blocks(n)), dim3(256), 0, st, a, b, label, n, gout, gs, da);
    else if (kind == 2) hipLaunchKernelGGL(loss_bwd_k<2>, dim3(ew_blocks(n)), dim3(256), 0, st, a, b, label, n, gout, gs, da);
    else if (kind == 3) hipLaunchKernelGGL(loss_bwd_k<3>, dim3(ew_blocks(n)), dim3(256), 0, st, a, b, label, n, gout, gs, da);
    else hipLaunchKernelGGL(loss_bwd_k<4>, dim3(ew_blocks(n)), dim3(256), 0, st, a, b, label, n, gout, gs, da);
    SG_LAUNCH_CHECK();
    return 0;
}

__global__ void psnr_k(const float* mse, float* out) { *out = 10.f * log10f(1.f / *mse); }
extern "C" int srcgan_psnr_from_mse(const float* mse, float* out, void* stream) {
    SG_REQUIRE(mse && out, "srcgan_psnr_from_mse: null pointer");
    hipLaunchKernelGGL(psnr_k, dim3(1), dim3(1), 0, (hipStream_t)stream, mse, out);
    SG_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- preprocessing
__global__ __launch_bounds__(256) void rgb_to_gray_k(const float* __restrict__ rgb, float* __restrict__ gray, long HW, long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / HW, p = e % HW;
        const float* s = rgb + (size_t)b * 3 * HW + p;
        // same association as the reference expression (trainCas.py:85-87)
        gray[e] = (0.2125f * s[0] + 0.7154f * s[HW]) + 0.0721f * s[2 * HW];
    }
}
extern "C" int srcgan_rgb_to_gray(const float* rgb, float* gray, int B, int H, int W, void* stream) {
    SG_REQUIRE(rgb && gray && B > 0 && H > 0 && W > 0, "srcgan_rgb_to_gray: bad arguments");
    const long HW = (long)H * W, total = HW * B;
    hipLaunchKernelGGL(rgb_to_gray_k, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, rgb, gray, HW, total);
    SG_LAUNCH_CHECK();
    return 0;
}

// bilinear, align_corners=False, scale 1/up (up even): source coord = (o+.5)*up-.5 = o*up + up/2 - .5
// -> mean of pixels (up/2-1, up/2) in each direction, weights exactly .5/.5.
__global__ __launch_bounds__(256) void bilinear_down_k(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                       int up, long total) {
    const int OH = H / up, OW = W / up;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % OW); const long q = e / OW;
        const int oy = (int)(q % OH); const long bc = q / OH;
        const float* s = src + (size_t)bc * H * W;
        const int y0 = oy * up + up / 2 - 1, x0 = ox * up + up / 2 - 1;
        const float top = 0.5f * s[(size_t)y0 * W + x0] + 0.5f * s[(size_t)y0 * W + x0 + 1];
        const float bot = 0.5f * s[(size_t)(y0 + 1) * W + x0] + 0.5f * s[(size_t)(y0 + 1) * W + x0 + 1];
        dst[e] = 0.5f * top + 0.5f * bot;
    }
}
extern "C" int srcgan_bilinear_down(const float* src, float* dst, int B, int C, int H, int W, int up, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "srcgan_bilinear_down: bad arguments");
    SG_REQUIRE(up >= 2 && up % 2 == 0 && H % up == 0 && W % up == 0, "srcgan_bilinear_down: up must be even and divide H, W");
    const long total = (long)B * C * (H / up) * (W / up);
    hipLaunchKernelGGL(bilinear_down_k, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, up, total);
    SG_LAUNCH_CHECK();
    return 0;
}

// ---- x2 nearest up-sampling of an NHWC feature map and its adjoint (legacy generators, model/model.py:384-386,428-433:
// F.interpolate(scale_factor=2, mode='nearest') between 3x3 convolutions).  16-byte vectors along the channels.
template <typename T>
__global__ __launch_bounds__(256) void upsample2_nhwc_k(const T* __restrict__ src, int s_cs, int s_coff, long s_plane,
                                                        T* __restrict__ dst, int d_cs, int H, int W, int C, long nvec) {
    constexpr int EPP = DT<T>::EPP, KCE = DT<T>::KCE;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int vpc = C / EPP;                                  // vectors per output pixel
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const int v = (int)(e % vpc); long q = e / vpc;
        const int ox = (int)(q % (2 * W)); q /= 2 * W;
        const int oy = (int)(q % (2 * H)); const long b = q / (2 * H);
        const long ip = (b * H + (oy >> 1)) * W + (ox >> 1);
        const int c = s_coff + v * EPP;
        const char* sp = s_plane ? (const char*)src + ip * 64 + (long)(c / KCE) * s_plane + (c % KCE) * sizeof(T)
                                 : (const char*)(src + ip * s_cs + c);
        *(vecT*)(dst + ((b * 2 * H + oy) * 2 * W + ox) * (long)d_cs + v * EPP) = *(const vecT*)sp;
    }
}
// dst[y][x] = sum of the 2x2 block of src, optionally times LeakyReLU'(mz[y][x]) (mz = the activated tensor that was up-sampled)
template <typename T>
__global__ __launch_bounds__(256) void sum2x2_nhwc_k(const T* __restrict__ src, int s_cs, T* __restrict__ dst, int d_cs,
                                                     const T* __restrict__ mz, int m_cs, float mslope, int H, int W, int C, long nvec) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int vpc = C / EPP;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const int v = (int)(e % vpc); long q = e / vpc;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H); const long b = q / H;
        float acc[EPP];
#pragma unroll
        for (int i = 0; i < EPP; ++i) acc[i] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const vecT t = *(const vecT*)(src + ((b * 2 * H + 2 * y + a) * 2 * W + 2 * x + bb) * (long)s_cs + v * EPP);
#pragma unroll
                for (int i = 0; i < EPP; ++i) acc[i] += to_f(t[i]);
            }
        const long op = (b * H + y) * W + x;
        if (mz) {
            const vecT m = *(const vecT*)(mz + op * m_cs + v * EPP);
#pragma unroll
            for (int i = 0; i < EPP; ++i) acc[i] *= to_f(m[i]) > 0.f ? 1.f : mslope;
        }
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) o[i] = from_f<T>(acc[i]);
        *(vecT*)(dst + op * d_cs + v * EPP) = o;
    }
}
extern "C" int srcgan_upsample2_nhwc(const void* src, int s_cs, int s_coff, long s_plane, void* dst, int d_cs,
                                     int B, int H, int W, int C, int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "srcgan_upsample2_nhwc: bad arguments");
    SG_REQUIRE(sg_dtype_ok(dtype), "srcgan_upsample2_nhwc: bad dtype %d", dtype);
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(C % epp == 0 && s_cs % epp == 0 && s_coff % epp == 0 && d_cs % epp == 0 && C <= d_cs, "srcgan_upsample2_nhwc: channel counts/strides must be multiples of %d", epp);
    const long nvec = (long)B * 4 * H * W * (C / epp);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(upsample2_nhwc_k<T>, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const T*)src, s_cs, s_coff, s_plane, (T*)dst, d_cs, H, W, C, nvec));
    SG_LAUNCH_CHECK();
    return 0;
}
extern "C" int srcgan_sum2x2_nhwc(const void* src, int s_cs, void* dst, int d_cs, const void* mz, int m_cs, float mslope,
                                  int B, int H, int W, int C, int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "srcgan_sum2x2_nhwc: bad arguments");
    SG_REQUIRE(sg_dtype_ok(dtype), "srcgan_sum2x2_nhwc: bad dtype %d", dtype);
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(C % epp == 0 && s_cs % epp == 0 && d_cs % epp == 0 && (!mz || m_cs % epp == 0), "srcgan_sum2x2_nhwc: channel counts/strides must be multiples of %d", epp);
    const long nvec = (long)B * H * W * (C / epp);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sum2x2_nhwc_k<T>, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const T*)src, s_cs, (T*)dst, d_cs, (const T*)mz, m_cs, mslope, H, W, C, nvec));
    SG_LAUNCH_CHECK();
    return 0;
}

// bilinear x up (integer), align_corners=False, as F.interpolate(scale_factor=up, mode="bilinear") in trainCasConst.py:91-92:
// src coordinate = (dst + 0.5) / up - 0.5 clamped at 0, neighbour index clamped at the border
__global__ __launch_bounds__(256) void bilinear_up_k(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int up, long total) {
    const int OH = H * up, OW = W * up;
    const float inv = 1.f / (float)up;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % OW); const long q = e / OW;
        const int oy = (int)(q % OH); const long bc = q / OH;
        float sy = ((float)oy + 0.5f) * inv - 0.5f, sx = ((float)ox + 0.5f) * inv - 0.5f;
        sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float* s = src + (size_t)bc * H * W;
        dst[e] = (1.f - ly) * ((1.f - lx) * s[(size_t)y0 * W + x0] + lx * s[(size_t)y0 * W + x1]) +
                 ly * ((1.f - lx) * s[(size_t)y1 * W + x0] + lx * s[(size_t)y1 * W + x1]);
    }
}
extern "C" int srcgan_bilinear_up(const float* src, float* dst, int B, int C, int H, int W, int up, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && up >= 1, "srcgan_bilinear_up: bad arguments");
    const long total = (long)B * C * H * up * W * up;
    hipLaunchKernelGGL(bilinear_up_k, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, up, total);
    SG_LAUNCH_CHECK();
    return 0;
}

// nearest resize, torch 'nearest' rule: src = floor(dst * in/out)
__global__ __launch_bounds__(256) void nearest_resize_k(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                        int OH, int OW, long total) {
    const float sy = (float)H / OH, sx = (float)W / OW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % OW); const long q = e / OW;
        const int oy = (int)(q % OH); const long bc = q / OH;
        int iy = (int)floorf(oy * sy), ix = (int)floorf(ox * sx);
        iy = iy < H - 1 ? iy : H - 1; ix = ix < W - 1 ? ix : W - 1;
        dst[e] = src[((size_t)bc * H + iy) * W + ix];
    }
}
extern "C" int srcgan_nearest_resize(const float* src, float* dst, int B, int C, int H, int W, int OH, int OW, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "srcgan_nearest_resize: bad arguments");
    const long total = (long)B * C * OH * OW;
    hipLaunchKernelGGL(nearest_resize_k, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, OH, OW, total);
    SG_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- PixelShuffle / activation mask
// nn.PixelShuffle(r) on NHWC (espcn.py:44,50; edsr.py:57-66): out[b][y*r+a][x*r+c][ch] = in[b][y][x][ch*r*r + a*r + c];
// inverse = 1 runs the adjoint (gradient) direction: in[b][y][x][ch*r*r + a*r + c] = out[b][y*r+a][x*r+c][ch].
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle_nhwc_k(const T* __restrict__ src, int s_cs, T* __restrict__ dst, int d_cs,
                                                            int H, int W, int C, int r, int inverse, long total) {
    // thread per element of the LOW-resolution tensor [B,H,W,C*r*r]
    const int Cr = C * r * r;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int cc = (int)(e % Cr); long q = e / Cr;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H); const long b = q / H;
        const int ch = cc / (r * r), ab = cc % (r * r), a = ab / r, c = ab % r;
        const long lo = ((b * H + y) * W + x), hi = ((b * H * r + (long)y * r + a) * W * r + (long)x * r + c);
        if (inverse) dst[lo * d_cs + cc] = src[hi * s_cs + ch];
        else dst[hi * d_cs + ch] = src[lo * s_cs + cc];
    }
}
extern "C" int srcgan_pixel_shuffle_nhwc(const void* src, int s_cs, void* dst, int d_cs, int B, int H, int W, int C, int r, int inverse,
                                         int dtype, void* stream) {
    SG_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && r >= 1, "srcgan_pixel_shuffle_nhwc: bad arguments");
    const long total = (long)B * H * W * C * r * r;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(pixel_shuffle_nhwc_k<T>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)src, s_cs, (T*)dst, d_cs, H, W, C, r, inverse, total));
    SG_LAUNCH_CHECK();
    return 0;
}
// g *= act > 0 ? 1 : slope  (gradient of ReLU / LeakyReLU applied to a gradient arriving from outside the network)
template <typename T>
__global__ __launch_bounds__(256) void mask_inplace_k(T* __restrict__ g, const T* __restrict__ act, float slope, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        if (!(to_f(act[e]) > 0.f)) g[e] = from_f<T>(to_f(g[e]) * slope);
}
extern "C" int srcgan_mask_inplace(void* g, const void* act, float slope, long n, int dtype, void* stream) {
    SG_REQUIRE(g && act && n > 0, "srcgan_mask_inplace: bad arguments");
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(mask_inplace_k<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (T*)g, (const T*)act, slope, n));
    SG_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------- fused multi-tensor Adam
// One launch updates every parameter of an optimiser group (697 tensors for the 23-block generator): replaces the
// foreach kernels behind torch.optim.Adam.step() (reference trainCas.py:38-41,143-150; train.py:191-192,331-340) with the
// arithmetic of torch's single-tensor path:  m += (g - m)(1 - b1)  [lerp];  v = b2 v + (1 - b2) g g;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  State tensors stay torch's (exp_avg, exp_avg_sq, step).
struct SgAdamTensor { float* p; const float* g; float* m; float* v; };
struct SgAdamChunk { int tensor, off, n, pad; };
__global__ __launch_bounds__(256) void adam_multi_k(const SgAdamTensor* __restrict__ tensors, const SgAdamChunk* __restrict__ chunks,
                                                    float w1, float b2, float omb2, float step_size, float inv_bc2_sqrt, float eps) {
    const SgAdamChunk ck = chunks[blockIdx.x];
    const SgAdamTensor t = tensors[ck.tensor];
    float* cp = t.p + ck.off; const float* cg = t.g + ck.off; float* cm = t.m + ck.off; float* cv = t.v + ck.off;
    for (int i = threadIdx.x * 4; i < ck.n; i += 1024) {
        if (i + 4 <= ck.n) {
            f32x4 p = *(f32x4*)(cp + i), m = *(f32x4*)(cm + i), v = *(f32x4*)(cv + i);
            const f32x4 g = *(const f32x4*)(cg + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // torch.lerp: weight < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
                m[j] = w1 < 0.5f ? m[j] + w1 * (g[j] - m[j]) : g[j] - (g[j] - m[j]) * (1.f - w1);
                v[j] = v[j] * b2 + omb2 * g[j] * g[j];
                p[j] = p[j] - step_size * (m[j] / (sqrtf(v[j]) * inv_bc2_sqrt + eps));
            }
            *(f32x4*)(cp + i) = p; *(f32x4*)(cm + i) = m; *(f32x4*)(cv + i) = v;
        } else {
            for (int j = i; j < ck.n; ++j) {
                const float g = cg[j];
                float m = cm[j], v = cv[j];
                m = w1 < 0.5f ? m + w1 * (g - m) : g - (g - m) * (1.f - w1);
                v = v * b2 + omb2 * g * g;
                cp[j] = cp[j] - step_size * (m / (sqrtf(v) * inv_bc2_sqrt + eps));
                cm[j] = m; cv[j] = v;
            }
        }
    }
}
// tensors_dev: ntensors records {p, g, m, v} (device f32 pointers, 16-byte aligned); chunks_dev: nchunks records
// {tensor index, element offset (multiple of 4), count <= 4096, 0}.  The chunk table depends only on the shapes; the tensor
// table is refreshed by the caller (srcgan_amd/optim.py) whenever a pointer (typically a fresh .grad) changes.
extern "C" int srcgan_adam_step(const void* tensors_dev, const void* chunks_dev, int nchunks, double lr, double beta1, double beta2, double eps,
                                long step, void* stream) {
    SG_REQUIRE(tensors_dev && chunks_dev && nchunks > 0 && step >= 1, "srcgan_adam_step: bad arguments");
    // hyper-parameters arrive as the Python doubles torch uses: 1 - beta, the bias corrections and lr / bc1 are formed in double
    // and rounded once, like the scalars torch hands to its kernels
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_multi_k, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const SgAdamTensor*)tensors_dev,
                       (const SgAdamChunk*)chunks_dev, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, inv_bc2_sqrt, (float)eps);
    SG_LAUNCH_CHECK();
    return 0;
}
