// Evaluation metrics on device (reference src/metrics.py:10-144, driven by the test loop src/testCas.py:65-90): angular error,
// SSIM, value range.  MSE / PSNR reuse the loss reductions (elementwise.hip).  NCHW f32 in (network outputs), f32 out.
// All reductions are two-stage with a fixed order (deterministic).
#include "common.h"
#include "../../include/srcgan_amd.h"

namespace {
constexpr int MT_BLK = 64;      // partial blocks per image

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    __syncthreads();
    return t;       // valid in thread 0
}

// AE (metrics.py:12-33): per pixel acos(<p,t> / (|p||t| + eps)) in degrees; partial[b][blk] = sum over the block's pixels
__global__ __launch_bounds__(256) void ae_partial_k(const float* __restrict__ p, const float* __restrict__ t, int C, long hw, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* pb = p + (size_t)b * C * hw; const float* tb = t + (size_t)b * C * hw;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        float dot = 0.f, np = 0.f, nt = 0.f;
        for (int c = 0; c < C; ++c) { const float a = pb[(size_t)c * hw + i], q = tb[(size_t)c * hw + i]; dot += a * q; np += a * a; nt += q * q; }
        s += 57.29577951308232f * acosf(dot / (sqrtf(np) * sqrtf(nt) + 1e-6f));
    }
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = tot;
}
// out[b*nout + j] = scale * sum_k partial[(b*nblk + k)*nout + j]
__global__ void fold_k(const float* __restrict__ partial, int nblk, int nout, float scale, float* __restrict__ out) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= nout) return;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += partial[((size_t)b * nblk + k) * nout + j];
    out[b * nout + j] = s * scale;
}

// min / max of a flat array (SSIM picks its dynamic range from them, metrics.py:100-107)
__global__ __launch_bounds__(256) void minmax_partial_k(const float* __restrict__ x, long n, float* __restrict__ partial) {
    __shared__ float rmin[4], rmax[4];
    float lo = 3.4e38f, hi = -3.4e38f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = x[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_down(lo, o, 64)); hi = fmaxf(hi, __shfl_down(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = lo; rmax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = rmin[0]; hi = rmax[0];
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, rmin[w]); hi = fmaxf(hi, rmax[w]); }
        partial[blockIdx.x * 2] = lo; partial[blockIdx.x * 2 + 1] = hi;
    }
}
__global__ void minmax_fold_k(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
    if (threadIdx.x) return;
    float lo = partial[0], hi = partial[1];
    for (int k = 1; k < nblk; ++k) { lo = fminf(lo, partial[2 * k]); hi = fmaxf(hi, partial[2 * k + 1]); }
    out[0] = lo; out[1] = hi;
}

// SSIM (metrics.py:72-144): 11x11 gaussian (sigma 1.5) depth-wise "valid" windows of x, y, x^2, y^2, xy; one workgroup per
// 16x16 tile of the (H-10)x(W-10) map of one (image, channel): 26x26 patches in LDS, separable passes, map value + contrast term
// summed per tile: partial[((b*C + c)*ntiles + tile)*2 + {ssim, cs}].  The dynamic range L is read from device memory
// (written by the range kernel) so the whole metric is stream-ordered without a host round trip.
__global__ __launch_bounds__(256) void ssim_tile_k(const float* __restrict__ p, const float* __restrict__ t, int H, int W, int tiles_x, int ntiles,
                                                  const float* __restrict__ range, float* __restrict__ partial) {
    __shared__ float sp[26][27], st[26][27];
    __shared__ float hz[5][26][16];
    __shared__ float red[4];
    __shared__ float gw[11];
    const int bc = blockIdx.y, tile = blockIdx.x, ty = tile / tiles_x, tx = tile % tiles_x;
    const int oy0 = ty * 16, ox0 = tx * 16, OH = H - 10, OW = W - 10;
    if (threadIdx.x < 11) {
        float s = 0.f;
        for (int i = 0; i < 11; ++i) s += expf(-(float)((i - 5) * (i - 5)) / 4.5f);
        gw[threadIdx.x] = expf(-(float)((threadIdx.x - 5) * (threadIdx.x - 5)) / 4.5f) / s;
    }
    const float* pb = p + (size_t)bc * H * W; const float* tb = t + (size_t)bc * H * W;
    for (int i = threadIdx.x; i < 26 * 26; i += 256) {
        const int y = i / 26, x = i % 26, gy = oy0 + y, gx = ox0 + x;
        const bool ok = gy < H && gx < W;
        sp[y][x] = ok ? pb[(size_t)gy * W + gx] : 0.f;
        st[y][x] = ok ? tb[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 26 * 16; i += 256) {
        const int y = i / 16, x = i % 16;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) { const float w = gw[k], u = sp[y][x + k], v = st[y][x + k]; a += w * u; b += w * v; aa += w * u * u; bb += w * v * v; ab += w * u * v; }
        hz[0][y][x] = a; hz[1][y][x] = b; hz[2][y][x] = aa; hz[3][y][x] = bb; hz[4][y][x] = ab;
    }
    __syncthreads();
    const int y = threadIdx.x / 16, x = threadIdx.x % 16;
    float ssim = 0.f, cs = 0.f;
    if (oy0 + y < OH && ox0 + x < OW) {
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k)
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += gw[k] * hz[q][y + k][x];
        const float max_val = range[1] > 128.f ? 255.f : 1.f, min_val = range[0] < -0.5f ? -1.f : 0.f, L = max_val - min_val;
        const float C1 = (0.01f * L) * (0.01f * L), C2 = (0.03f * L) * (0.03f * L);
        const float mu1sq = m[0] * m[0], mu2sq = m[1] * m[1], mu12 = m[0] * m[1];
        const float v1 = 2.f * (m[4] - mu12) + C2, v2 = (m[2] - mu1sq) + (m[3] - mu2sq) + C2;
        cs = v1 / v2;
        ssim = ((2.f * mu12 + C1) * v1) / ((mu1sq + mu2sq + C1) * v2);
    }
    const float s0 = block_sum(ssim, red);
    const float s1 = block_sum(cs, red);
    if (threadIdx.x == 0) { float* o = partial + ((size_t)bc * ntiles + tile) * 2; o[0] = s0; o[1] = s1; }
}
}  // namespace

extern "C" int srcgan_metric_scratch_floats(int B, int C, int H, int W) {
    const long tiles = (long)cdiv(H > 10 ? H - 10 : 1, 16) * cdiv(W > 10 ? W - 10 : 1, 16);
    const long a = (long)B * MT_BLK, s = (long)B * C * tiles * 2, m = 2 * 256;
    return (int)((a > s ? a : s) + m + 16);
}

// out[b] = mean angular error (degrees) of image b
extern "C" int srcgan_metric_ae(const float* pred, const float* truth, int B, int C, int H, int W, float* out, float* scratch, void* stream) {
    SG_REQUIRE(pred && truth && out && scratch && B > 0 && C > 0 && H > 0 && W > 0, "srcgan_metric_ae: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W;
    hipLaunchKernelGGL(ae_partial_k, dim3(MT_BLK, B), dim3(256), 0, st, pred, truth, C, hw, scratch);
    hipLaunchKernelGGL(fold_k, dim3(B), dim3(64), 0, st, (const float*)scratch, MT_BLK, 1, 1.f / (float)hw, out);
    SG_LAUNCH_CHECK();
    return 0;
}

// out[b][0] = mean of the SSIM map of image b (all channels), out[b][1] = mean contrast term; range_from = the tensor whose min / max
// select the dynamic range (the prediction, metrics.py:100-107)
extern "C" int srcgan_metric_ssim(const float* pred, const float* truth, int B, int C, int H, int W, float* out, float* scratch, void* stream) {
    SG_REQUIRE(pred && truth && out && scratch && B > 0 && C > 0, "srcgan_metric_ssim: bad arguments");
    SG_REQUIRE(H >= 11 && W >= 11, "srcgan_metric_ssim: images must be at least 11x11 (valid 11x11 windows)");
    hipStream_t st = (hipStream_t)stream;
    const int OH = H - 10, OW = W - 10, tiles_x = cdiv(OW, 16), ntiles = tiles_x * cdiv(OH, 16);
    float* range = scratch + (size_t)B * C * ntiles * 2;
    float* mm = range + 8;
    const long n = (long)B * C * H * W;
    hipLaunchKernelGGL(minmax_partial_k, dim3(256), dim3(256), 0, st, pred, n, mm);
    hipLaunchKernelGGL(minmax_fold_k, dim3(1), dim3(64), 0, st, (const float*)mm, 256, range);
    hipLaunchKernelGGL(ssim_tile_k, dim3(ntiles, B * C), dim3(256), 0, st, pred, truth, H, W, tiles_x, ntiles, (const float*)range, scratch);
    hipLaunchKernelGGL(fold_k, dim3(B), dim3(64), 0, st, (const float*)scratch, C * ntiles, 2, 1.f / ((float)C * OH * OW), out);
    SG_LAUNCH_CHECK();
    return 0;
}
