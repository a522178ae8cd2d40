// Input pipeline colour conversions on device (reference src/dataset.py:114-159, used by G2RGB / G2LAB.__getitem__ :179-199,
// :234-254): 8-bit interleaved RGB -> the float tensors the training step consumes, and the inverse used for visualisation
// (dataset.py:92-112).  The reference calls scikit-image (skimage.color.rgb2gray / rgb2lab / lab2rgb) per sample on the host;
// this is its published algorithm: /255, luma weights (0.2125, 0.7154, 0.0721); sRGB companding, the 3x3 sRGB->XYZ matrix
// rounded to 6 places, D65 / 2-degree white (0.95047, 1, 1.08883), CIE f(t) with the 0.008856 / 7.787 constants.
// Arithmetic in double like the host code (float64 ndarray), one rounding to float at the store.  HBM-bound: 3 B in, 4-12 B
// out per pixel; one thread per pixel, consecutive threads on consecutive pixels.
#include "common.h"
#include "../../include/srcgan_amd.h"

namespace {
__device__ __forceinline__ double srgb_to_lin(double c) { return c > 0.04045 ? pow((c + 0.055) / 1.055, 2.4) : c / 12.92; }
__device__ __forceinline__ double lin_to_srgb(double c) { return c > 0.0031308 ? 1.055 * pow(c, 1.0 / 2.4) - 0.055 : c * 12.92; }
__device__ __forceinline__ double lab_f(double t) { return t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0; }
__device__ __forceinline__ double lab_finv(double t) { return t > 0.2068966 ? t * t * t : (t - 16.0 / 116.0) / 7.787; }

// mode 0 gray [B,1,HW], 1 rgb/255 [B,3,HW], 2 (L/100, (a+128)/255, (b+128)/255) [B,3,HW], 3 the two chroma planes [B,2,HW]
__global__ __launch_bounds__(256) void u8rgb_to_planes_k(const unsigned char* __restrict__ src, float* __restrict__ dst, long hw, long total, int mode) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / hw, px = i - b * hw;
        const double r = src[3 * i] / 255.0, g = src[3 * i + 1] / 255.0, bl = src[3 * i + 2] / 255.0;
        if (mode == 0) { dst[i] = (float)(0.2125 * r + 0.7154 * g + 0.0721 * bl); continue; }
        if (mode == 1) {
            float* o = dst + (size_t)b * 3 * hw + px;
            o[0] = (float)r; o[hw] = (float)g; o[2 * hw] = (float)bl;
            continue;
        }
        const double lr = srgb_to_lin(r), lg = srgb_to_lin(g), lb = srgb_to_lin(bl);
        const double x = (0.412453 * lr + 0.357580 * lg + 0.180423 * lb) / 0.95047;
        const double y = 0.212671 * lr + 0.715160 * lg + 0.072169 * lb;
        const double z = (0.019334 * lr + 0.119193 * lg + 0.950227 * lb) / 1.08883;
        const double fx = lab_f(x), fy = lab_f(y), fz = lab_f(z);
        const double L = 116.0 * fy - 16.0, A = 500.0 * (fx - fy), Bb = 200.0 * (fy - fz);
        if (mode == 2) {
            float* o = dst + (size_t)b * 3 * hw + px;
            o[0] = (float)(L / 100.0); o[hw] = (float)((A + 128.0) / 255.0); o[2 * hw] = (float)((Bb + 128.0) / 255.0);
        } else {
            float* o = dst + (size_t)b * 2 * hw + px;
            o[0] = (float)((A + 128.0) / 255.0); o[hw] = (float)((Bb + 128.0) / 255.0);
        }
    }
}

// normalised (L, a, b) planes -> 8-bit RGB (dataset.py:92-104: L*100, ab*255-128, lab2rgb, *255, truncation to uint8)
__global__ __launch_bounds__(256) void lab_planes_to_u8rgb_k(const float* __restrict__ lab, unsigned char* __restrict__ dst, long hw, long total) {
    // inverse of the rounded sRGB->XYZ matrix above (numpy.linalg.inv in the reference)
    const double m00 = 3.240481343200527, m01 = -1.5371515162713185, m02 = -0.49853632616888777;
    const double m10 = -0.9692549499965682, m11 = 1.8759900014898907, m12 = 0.04155592655829284;
    const double m20 = 0.05564663913517715, m21 = -0.20404133836651123, m22 = 1.0573110696453443;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / hw, px = i - b * hw;
        const float* in = lab + (size_t)b * 3 * hw + px;
        const double L = (double)in[0] * 100.0, A = (double)in[hw] * 255.0 - 128.0, Bb = (double)in[2 * hw] * 255.0 - 128.0;
        const double fy = (L + 16.0) / 116.0, fx = A / 500.0 + fy;
        double fz = fy - Bb / 200.0;
        if (fz < 0.0) fz = 0.0;
        const double x = lab_finv(fx) * 0.95047, y = lab_finv(fy), z = lab_finv(fz) * 1.08883;
        double c[3] = {m00 * x + m01 * y + m02 * z, m10 * x + m11 * y + m12 * z, m20 * x + m21 * y + m22 * z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double v = lin_to_srgb(c[k]);
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            dst[3 * i + k] = (unsigned char)(v * 255.0);
        }
    }
}
}   // namespace

extern "C" int srcgan_u8rgb_to_planes(const unsigned char* rgb, float* dst, int B, long hw, int mode, void* stream) {
    SG_REQUIRE(rgb && dst && B > 0 && hw > 0, "srcgan_u8rgb_to_planes: bad arguments");
    SG_REQUIRE(mode >= 0 && mode <= 3, "srcgan_u8rgb_to_planes: mode %d (0 gray, 1 rgb, 2 lab, 3 ab)", mode);
    const long total = (long)B * hw;
    long nb = (total + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(u8rgb_to_planes_k, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, rgb, dst, hw, total, mode);
    SG_LAUNCH_CHECK();
    return 0;
}

extern "C" int srcgan_lab_planes_to_u8rgb(const float* lab, unsigned char* rgb, int B, long hw, void* stream) {
    SG_REQUIRE(lab && rgb && B > 0 && hw > 0, "srcgan_lab_planes_to_u8rgb: bad arguments");
    const long total = (long)B * hw;
    long nb = (total + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(lab_planes_to_u8rgb_k, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lab, rgb, hw, total);
    SG_LAUNCH_CHECK();
    return 0;
}
