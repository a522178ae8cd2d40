// Shared device/host helpers for the gfx950 SRCGAN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/srcgan_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Per-dtype constants.  A "piece" is 16 bytes; a K-chunk is 64 bytes of channels
// per pixel (16 f32 or 32 bf16) so both dtypes share one LDS byte layout.
template <typename T> struct DT;
template <> struct DT<float>  { static constexpr int id = SRCGAN_F32;  static constexpr int EPP = 4; static constexpr int KCE = 16; };
template <> struct DT<__bf16> { static constexpr int id = SRCGAN_BF16; static constexpr int EPP = 8; static constexpr int KCE = 32; };
template <> struct DT<_Float16> { static constexpr int id = SRCGAN_F16; static constexpr int EPP = 8; static constexpr int KCE = 32; };
static inline bool sg_is16(int dtype) { return dtype == SRCGAN_BF16 || dtype == SRCGAN_F16; }
static inline bool sg_dtype_ok(int dtype) { return dtype == SRCGAN_F32 || sg_is16(dtype); }

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f(_Float16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 from_f<_Float16>(float v) { return (_Float16)v; }

// (count, mean, M2 = sum of squared deviations) of a sample set A joined with those of B -- Chan, Golub, LeVeque: exact in exact arithmetic,
// no cancellation; used for BatchNorm / GroupNorm / InstanceNorm statistics (elementwise.hip, groupnorm.hip)
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb <= 0.f) return;
    const float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
}

// 4 consecutive channels <-> f32 registers (8-byte bf16 / 16-byte f32 accesses)
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load4<__bf16>(const __bf16* p, float (&v)[4]) {
    bf16x4 t = *(const bf16x4*)p; v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <> __device__ __forceinline__ void load4<_Float16>(const _Float16* p, float (&v)[4]) {
    f16x4 t = *(const f16x4*)p; v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = t;
}
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const float (&v)[4]) {
    bf16x4 t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; *(bf16x4*)p = t;
}

template <> __device__ __forceinline__ void store4<_Float16>(_Float16* p, const float (&v)[4]) {
    f16x4 t = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; *(f16x4*)p = t;
}
// 32x32x16 MFMA on 16-byte fragments of a 16-bit type.  Fragments travel as bf16x8 (a 16-byte container: LDS reads do not care);
// the element type selects the instruction (bf16 and f16 run at the same rate).
template <typename T> __device__ __forceinline__ f32x16 sg_mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, __bf16))
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16x16x32 form (f32x4 result: column = lane & 15, row = 4 * (lane >> 4) + register; operands: row / column lane & 15, k = 8 * (lane >> 4) + j)
template <typename T> __device__ __forceinline__ f32x4 sg_mfma16s(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, __bf16))
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// eight copies of 1.0 in T, as a fragment (bias gradients: a pseudo-tap whose B operand is all ones)
template <typename T> __device__ __forceinline__ bf16x8 sg_ones16() {
    typedef __attribute__((ext_vector_type(8))) T v8;
    v8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (T)1.0f;
    return __builtin_bit_cast(bf16x8, o);
}

// ---------------------------------------------------------------- host side
void srcgan_set_error(const char* fmt, ...);
#define SG_FAIL(...) do { srcgan_set_error(__VA_ARGS__); return 1; } while (0)
#define SG_REQUIRE(cond, ...) do { if (!(cond)) { srcgan_set_error(__VA_ARGS__); return 1; } } while (0)
#define SG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    srcgan_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define SG_LAUNCH_CHECK() SG_HIP(hipGetLastError())
#define SG_TRY(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline long cdivl(long a, long b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------- diagnostics
// The production library carries NO run-time experiment switches: SG_DBG(p, bit) is the constant 0 and sg_env() returns
// nullptr unless the file is compiled with -DSG_DIAG (scripts/build_variant.sh builds such variants side by side and
// SRCGAN_AMD_LIB selects one).  Wrong-result timing experiments (SG_EXP_*, SG_WD_EXP, SG_TRACE) are compile-time only.
#ifdef SG_DIAG
#include <stdlib.h>
#define SG_DBG(p, bit) ((p).dbg & (bit))
static inline const char* sg_env(const char* name) { return getenv(name); }
#else
#define SG_DBG(p, bit) 0
static inline const char* sg_env(const char*) { return nullptr; }
#if defined(SG_TRACE) || defined(SG_WD_EXP) || defined(SG_EXP_NO_HALO_DMA) || defined(SG_EXP_NO_WEIGHT_DMA) || defined(SG_EXP_B_KX0_ONLY) || \
    defined(SG_EXP_NO_FRAG_READS) || defined(SG_EXP_NO_MFMA_INSTR)
#error "timing experiments need -DSG_DIAG (scripts/build_variant.sh adds it)"
#endif
#endif

// ---------------------------------------------------------------- optional per-launch profiling
// (bench.py: HIP events on the launch stream around the hot kernels; off by default -> zero cost)
int sg_prof_start(const char* cls, double flops, double bytes, hipStream_t st);   // returns token or -1
void sg_prof_stop(int token, hipStream_t st);

// ---------------------------------------------------------------- batched weight packing (nets.hip -> elementwise.hip)
struct SgPackJob {
    const float* w; size_t wp_off;
    int rows, kdim, tys, txs;
    long sr, sk, sty, stx, off;
    int k_off, k_total; float scale;
    int cot, nchunk; long total, blk0;
};
void sg_pack_job_finish(SgPackJob& j, int dtype, long& blk_cursor);
int sg_pack_multi_launch(const SgPackJob* jobs_dev, int njobs, long nblocks, void* wp_base, int dtype, hipStream_t st, const unsigned long long* guard = nullptr);
int sg_fill_zero_guarded(void* p, size_t bytes, const unsigned long long* guard, hipStream_t st);     // guard == nullptr: plain hipMemsetAsync
