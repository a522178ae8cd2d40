// GroupNorm forward / backward on NHWC activations (reference src/model/resdeconv.py:61-76,118-121:
// nn.GroupNorm(32, C) after every convolution of the ResDeconv colouriser, affine, eps 1e-5, biased variance).
// Replaces aten::native_group_norm(+_backward), the ReLU that follows it and the residual add of BasicBlock.forward
// (resdeconv.py:78-97): y = relu?( (x - mean_bg) * rstd_bg * gamma_c + beta_c [+ res] ).
//
// HBM-bound: forward = one read pass for the statistics + one read/write pass; backward = one pass over (dy, y, x)
// for the per-(image, channel) sums + one read/write pass.  Statistics are two-stage and order-fixed (deterministic):
// partial sums per (image, pixel-range block, channel) -> one block per image folds blocks, then the channels of a group.
#include "common.h"
#include "../../include/srcgan_amd.h"

#define DISPATCH_DTYPE(dtype, ...) \
    if ((dtype) == SRCGAN_F32) { using T = float; __VA_ARGS__; } \
    else if ((dtype) == SRCGAN_BF16) { using T = __bf16; __VA_ARGS__; } \
    else if ((dtype) == SRCGAN_F16) { using T = _Float16; __VA_ARGS__; } \
    else SG_FAIL("bad dtype %d", (int)(dtype));

namespace {
constexpr int GN_MAXBLK = 32;
static inline int ew_blocks(long n, int per_block = 256) {
    long b = cdivl(n, per_block);
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

static inline int gn_nblk(long hw, int C, int epp) {
    const long vec = hw * (C / epp);
    long n = vec / (256 * 8);
    return (int)(n < 1 ? 1 : n > GN_MAXBLK ? GN_MAXBLK : n);
}

// MODE 0: sums (x, x^2).  MODE 1: g = dy [* (yact > 0)], sums (g, g * xhat), xhat from the saved statistics.
// partial[((b * nblk + blk) * C + c) * 2 + {0,1}]
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gn_partial_k(const T* __restrict__ x, int x_cs, const T* __restrict__ dy, int dy_cs,
                                                    const T* __restrict__ yact, int ya_cs, const float* __restrict__ stats,
                                                    long hw, int C, int G, float slope, float* __restrict__ partial) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    __shared__ float red[2][256 * EPP];
    const int b = blockIdx.y, nblk = gridDim.x;
    const int VG = C / EPP, PL = 256 / VG;                       // host guarantees VG | 256
    const int cg = threadIdx.x % VG, pl = threadIdx.x / VG, c0 = cg * EPP;
    const long per = cdivl(hw, nblk);
    const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < hw) ? p0 + per : hw;
    float s0[EPP], s1[EPP], mu[EPP], rs[EPP];
    const int cpg = C / G;
#pragma unroll
    for (int i = 0; i < EPP; ++i) {
        s0[i] = 0.f; s1[i] = 0.f;
        if (MODE == 1) { const int g = (c0 + i) / cpg; mu[i] = stats[((size_t)b * G + g) * 2]; rs[i] = stats[((size_t)b * G + g) * 2 + 1]; }
    }
    // MODE 0: sums around the thread's first sample (s0 = sum (v - sh), s1 = sum (v - sh)^2): the variance of ONE channel of ONE image
    // (InstanceNorm2d: G == C) whose mean is many standard deviations from zero loses its digits in  E[x^2] - mean^2
    float sh[EPP], nsmp = 0.f;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < EPP; ++i) sh[i] = 0.f;
        if (p0 + pl < p1) {
            const vecT fv = *(const vecT*)(x + ((size_t)b * hw + p0 + pl) * x_cs + c0);
#pragma unroll
            for (int i = 0; i < EPP; ++i) sh[i] = to_f(fv[i]);
        }
    }
    for (long px = p0 + pl; px < p1; px += PL) {
        const size_t q = (size_t)b * hw + px;
        const vecT xv = *(const vecT*)(x + q * x_cs + c0);
        if (MODE == 0) {
            nsmp += 1.f;
#pragma unroll
            for (int i = 0; i < EPP; ++i) { const float v = to_f(xv[i]) - sh[i]; s0[i] += v; s1[i] += v * v; }
        } else {
            const vecT gv = *(const vecT*)(dy + q * dy_cs + c0);
            vecT av;
            if (yact) av = *(const vecT*)(yact + q * ya_cs + c0);
#pragma unroll
            for (int i = 0; i < EPP; ++i) {
                float g = to_f(gv[i]);
                if (yact && !(to_f(av[i]) > 0.f)) g *= slope;
                s0[i] += g; s1[i] += g * (to_f(xv[i]) - mu[i]) * rs[i];
            }
        }
    }
    __shared__ float cnt[256];
    if (MODE == 0) {        // thread partial -> (count, mean, M2)
        const float inv = nsmp > 0.f ? 1.f / nsmp : 0.f;
#pragma unroll
        for (int i = 0; i < EPP; ++i) { const float a = s0[i]; s0[i] = sh[i] + a * inv; s1[i] = s1[i] - a * a * inv; }
        cnt[threadIdx.x] = nsmp;
    }
#pragma unroll
    for (int i = 0; i < EPP; ++i) { red[0][threadIdx.x * EPP + i] = s0[i]; red[1][threadIdx.x * EPP + i] = s1[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t0 = 0.f, t1 = 0.f, tn = 0.f;
        for (int q = 0; q < PL; ++q) {
            const int t = q * VG + c / EPP;
            if (MODE == 0) chan_combine(tn, t0, t1, cnt[t], red[0][t * EPP + c % EPP], red[1][t * EPP + c % EPP]);      // fixed order: deterministic
            else { t0 += red[0][t * EPP + c % EPP]; t1 += red[1][t * EPP + c % EPP]; }
        }
        float* o = partial + (((size_t)b * nblk + blockIdx.x) * C + c) * 2;
        o[0] = t0; o[1] = t1;           // MODE 0: the block's (mean, M2) over its p1 - p0 pixels
    }
}

// one block per image.  MODE 0: stats[b][g] = {mean, rstd}.  MODE 1: gsum[b][g] = {S1/N, S2/N} (gamma-weighted), and the
// per-(image, channel) sums go to chan[b][c] = {sum g, sum g*xhat} for the parameter gradients.
template <int MODE>
__global__ __launch_bounds__(256) void gn_fold_k(const float* __restrict__ partial, int nblk, int C, int G, long hw, float eps,
                                                 const float* __restrict__ gamma, float* __restrict__ out, float* __restrict__ chan) {
    __shared__ float cs[2][1024];
    const int b = blockIdx.x;
    const long per = cdivl(hw, (long)nblk);
    for (int c = threadIdx.x; c < C; c += 256) {
        float t0 = 0.f, t1 = 0.f, tn = 0.f;
        for (int k = 0; k < nblk; ++k) {
            const float* p = partial + (((size_t)b * nblk + k) * C + c) * 2;
            if (MODE == 0) {            // (mean, M2) of block k's pixel range
                const long p0 = (long)k * per, p1 = (p0 + per < hw) ? p0 + per : hw;
                chan_combine(tn, t0, t1, p1 > p0 ? (float)(p1 - p0) : 0.f, p[0], p[1]);
            } else { t0 += p[0]; t1 += p[1]; }
        }
        if (MODE == 1) { chan[((size_t)b * C + c) * 2] = t0; chan[((size_t)b * C + c) * 2 + 1] = t1; if (gamma) { t0 *= gamma[c]; t1 *= gamma[c]; } }
        cs[0][c] = t0; cs[1][c] = t1;
    }
    __syncthreads();
    const int cpg = C / G;
    const float invn = 1.f / ((float)hw * (float)cpg);
    for (int g = threadIdx.x; g < G; g += 256) {
        float t0 = 0.f, t1 = 0.f, tn = 0.f;
        for (int i = 0; i < cpg; ++i) {
            if (MODE == 0) chan_combine(tn, t0, t1, (float)hw, cs[0][g * cpg + i], cs[1][g * cpg + i]);      // channel (mean, M2) over hw pixels each
            else { t0 += cs[0][g * cpg + i]; t1 += cs[1][g * cpg + i]; }
        }
        float* o = out + ((size_t)b * G + g) * 2;
        if (MODE == 0) { o[0] = t0; o[1] = rsqrtf(t1 * invn + eps); }
        else { o[0] = t0 * invn; o[1] = t1 * invn; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_k(const T* __restrict__ x, int x_cs, const T* __restrict__ res, int r_cs, T* __restrict__ y, int y_cs,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ stats,
                                                  long hw, int C, int G, int relu, float slope, long nvec) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int VG = C / EPP, cpg = C / G;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const int c0 = (int)(e % VG) * EPP; const long q = e / VG; const long b = q / hw;
        const vecT xv = *(const vecT*)(x + q * x_cs + c0);
        vecT rv; if (res) rv = *(const vecT*)(res + q * r_cs + c0);
        vecT o;
#pragma unroll
        for (int i = 0; i < EPP; ++i) {
            const int c = c0 + i, g = c / cpg;
            const float mu = stats[(b * G + g) * 2], rs = stats[(b * G + g) * 2 + 1];
            float v = (to_f(xv[i]) - mu) * rs * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);      // null gamma / beta: no affine part (InstanceNorm2d)
            if (res) v += to_f(rv[i]);
            if (relu) v = v > 0.f ? v : v * slope;
            o[i] = from_f<T>(v);
        }
        *(vecT*)(y + q * y_cs + c0) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_k(const T* __restrict__ dy, int dy_cs, const T* __restrict__ yact, int ya_cs, const T* __restrict__ x, int x_cs,
                                                      const float* __restrict__ gamma, const float* __restrict__ stats, const float* __restrict__ gsum,
                                                      T* __restrict__ dx, int dx_cs, T* __restrict__ dres, int dr_cs, int dres_acc, float slope, long hw, int C, int G, long nvec) {
    constexpr int EPP = DT<T>::EPP;
    typedef __attribute__((ext_vector_type(EPP))) T vecT;
    const int VG = C / EPP, cpg = C / G;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const int c0 = (int)(e % VG) * EPP; const long q = e / VG; const long b = q / hw;
        const vecT gv = *(const vecT*)(dy + q * dy_cs + c0);
        const vecT xv = *(const vecT*)(x + q * x_cs + c0);
        vecT av; if (yact) av = *(const vecT*)(yact + q * ya_cs + c0);
        vecT o, gm;
#pragma unroll
        for (int i = 0; i < EPP; ++i) {
            const int c = c0 + i, g = c / cpg;
            const float mu = stats[(b * G + g) * 2], rs = stats[(b * G + g) * 2 + 1];
            float gg = to_f(gv[i]);
            if (yact && !(to_f(av[i]) > 0.f)) gg *= slope;
            const float xh = (to_f(xv[i]) - mu) * rs;
            o[i] = from_f<T>(rs * (gg * (gamma ? gamma[c] : 1.f) - (gsum[(b * G + g) * 2] + xh * gsum[(b * G + g) * 2 + 1])));
            gm[i] = from_f<T>(gg);
        }
        *(vecT*)(dx + q * dx_cs + c0) = o;
        if (dres) {
            if (dres_acc) {
                const vecT old = *(const vecT*)(dres + q * dr_cs + c0);
#pragma unroll
                for (int i = 0; i < EPP; ++i) gm[i] = from_f<T>(to_f(gm[i]) + to_f(old[i]));
            }
            *(vecT*)(dres + q * dr_cs + c0) = gm;
        }
    }
}

// dgamma[c] = sum_b chan[b][c][1], dbeta[c] = sum_b chan[b][c][0]
__global__ __launch_bounds__(256) void gn_param_grad_k(const float* __restrict__ chan, int B, int C, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, bq = 0.f;
    for (int b = 0; b < B; ++b) { a += chan[((size_t)b * C + c) * 2]; bq += chan[((size_t)b * C + c) * 2 + 1]; }
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + bq : bq;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + a : a;
}

static int gn_check(const char* who, int B, long hw, int C, int G, int dtype) {
    SG_REQUIRE(B > 0 && hw > 0 && C > 0 && G > 0 && C % G == 0 && C <= 1024, "%s: bad B/HW/C/G (C <= 1024, G | C)", who);
    SG_REQUIRE(sg_dtype_ok(dtype), "%s: bad dtype %d", who, dtype);
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(C % epp == 0 && 256 % (C / epp) == 0, "%s: C/%d must divide 256", who, epp);
    return 0;
}
}  // namespace

extern "C" size_t srcgan_gn_scratch_floats(int B, int C) { return (size_t)B * GN_MAXBLK * C * 2 + (size_t)B * C * 2 + (size_t)B * 1024 * 2; }

extern "C" int srcgan_gn_forward(const void* x, int x_cs, const void* res, int res_cs, void* y, int y_cs, const float* gamma, const float* beta,
                                 float* stats, int B, long hw, int C, int G, float eps, int relu, float slope, int dtype, float* scratch, void* stream) {
    SG_REQUIRE(x && y && stats && scratch, "srcgan_gn_forward: null pointer");
    SG_TRY(gn_check("srcgan_gn_forward", B, hw, C, G, dtype));
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(x_cs % epp == 0 && y_cs % epp == 0 && (!res || res_cs % epp == 0), "srcgan_gn_forward: channel strides must be multiples of %d", epp);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = gn_nblk(hw, C, epp);
    const long nvec = (long)B * hw * (C / epp);
    DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((gn_partial_k<T, 0>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, x_cs, (const T*)nullptr, 0, (const T*)nullptr, 0, (const float*)nullptr, hw, C, G, 0.f, scratch);
        hipLaunchKernelGGL(gn_fold_k<0>, dim3(B), dim3(256), 0, st, (const float*)scratch, nblk, C, G, hw, eps, (const float*)nullptr, stats, (float*)nullptr);
        hipLaunchKernelGGL(gn_apply_k<T>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const T*)x, x_cs, (const T*)res, res_cs, (T*)y, y_cs, gamma, beta, (const float*)stats, hw, C, G, relu, slope, nvec);
    });
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_gn_backward(const void* dy, int dy_cs, const void* yact, int ya_cs, const void* x, int x_cs, const float* gamma, const float* stats,
                                  void* dx, int dx_cs, void* dres, int dres_cs, int dres_accumulate, float* dgamma, float* dbeta, int accumulate,
                                  float slope, int B, long hw, int C, int G, int dtype, float* scratch, void* stream) {
    SG_REQUIRE(dy && x && stats && dx && scratch, "srcgan_gn_backward: null pointer");
    SG_TRY(gn_check("srcgan_gn_backward", B, hw, C, G, dtype));
    const int epp = dtype == SRCGAN_F32 ? 4 : 8;
    SG_REQUIRE(dy_cs % epp == 0 && x_cs % epp == 0 && dx_cs % epp == 0 && (!yact || ya_cs % epp == 0) && (!dres || dres_cs % epp == 0),
               "srcgan_gn_backward: channel strides must be multiples of %d", epp);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = gn_nblk(hw, C, epp);
    const long nvec = (long)B * hw * (C / epp);
    float* chan = scratch + (size_t)B * GN_MAXBLK * C * 2;
    float* gsum = chan + (size_t)B * C * 2;
    DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((gn_partial_k<T, 1>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, x_cs, (const T*)dy, dy_cs, (const T*)yact, ya_cs, stats, hw, C, G, slope, scratch);
        hipLaunchKernelGGL(gn_fold_k<1>, dim3(B), dim3(256), 0, st, (const float*)scratch, nblk, C, G, hw, 0.f, gamma, gsum, chan);
        hipLaunchKernelGGL(gn_bwd_apply_k<T>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const T*)dy, dy_cs, (const T*)yact, ya_cs, (const T*)x, x_cs, gamma, stats,
                           (const float*)gsum, (T*)dx, dx_cs, (T*)dres, dres_cs, dres_accumulate, slope, hw, C, G, nvec);
    });
    if (dgamma || dbeta) hipLaunchKernelGGL(gn_param_grad_k, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)chan, B, C, dgamma, dbeta, accumulate);
    SG_LAUNCH_CHECK();
    return 0;
}
