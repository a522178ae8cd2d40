// Skeleton of the loader-specialised 3x3 convolution (csrc/conv3x3_dma.hip, Cout = 32 form) on every CU: which of its streams
// overlap, and what does each cost?  
//   one workgroup per CU = 8 MFMA waves + NLW loader waves, NSTG stage buffers in LDS, one barrier per K chunk;
//   loaders: a stage per chunk by `buffer_load_dwordx4 ... lds` -- either flat 1-KiB pieces of a stream, or (PAT) the kernel's
//            18x34-pixel halo tile of one 64-byte channel plane of a [16, 256, 256] blocked tensor, per-lane offsets fixed;
//   MFMA waves: per chunk either generic groups of ds_read_b128 + 32x32x16 bf16 MFMAs, or (RL = 1) the kernel's own row-ordered
//            loop (swizzled pixel fragments, resident weight fragments, 42 reads + 36 MFMAs per wave and chunk, 2 accumulators);
//   EPI: a unit's output stores after its 4th chunk (8 B per lane as the accumulator layout gives them / 16 B lane pairs /
//            1 KiB contiguous per instruction / nontemporal).
// Every stream is a template switch (no run-time branches in the loops).  `ingest_test.bin const|random [all]`: the operand
// bits matter -- the part is power-limited and the matrix pipe's clock follows their toggling.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ingest_test.hip -o ingest_test.bin ; output of a run: profiles/r02_conv_skeleton.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// random bf16 values in (-0.5, 0.5): the matrix pipe's power (hence the clock) depends on the operand bits
__global__ void fill_k(unsigned short* p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const float f = ((x & 0xffff) / 65536.0f - 0.5f);
        p[i] = (unsigned short)(__builtin_bit_cast(unsigned, f) >> 16);
    }
}

template <int WIDTH>
__device__ __forceinline__ void dma(const __amdgpu_buffer_rsrc_t r, int voff, int soff, lptr_t lds) {
    if constexpr (WIDTH == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, voff, soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void bar_dma() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }

// DMA: loaders stream; RD: fragment reads; MF: MFMAs; NLW loader waves; NSTG stages; SP KiB per stage; NRD reads, NMF MFMAs per chunk
template <bool DMA, bool RD, bool MF, int NLW, int NSTG, int SP, int NRD, int NMF, int WIDTH, int PAT, int EPI, int ORD, int RL, int EPV = 0>
__global__ __launch_bounds__((8 + NLW) * 64) void k(const char* src, long bytes_per_wg, int nchunk, float* sink, char* dst, int imask, long dmask) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int SB = SP * 1024;
    constexpr int PPI = 1024 / (64 * WIDTH);        // instructions per KiB piece (1 for 16-byte lanes)
    if (wave >= 8) {
        const int iw = wave - 8;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)blockIdx.x * bytes_per_wg), 0, (int)bytes_per_wg, 0x00020000);
        const int voff = lane * WIDTH;
        // PAT 1: the convolution's operand pattern.  Source = 4 channel planes of [16][256][256] pixels x 64 B (67 MB each, "blocked"
        // layout); a unit = 16x32-pixel tile, 4 chunks = its 18x34-pixel halo in each plane; 8 units per workgroup, the workgroups
        // of an XCD walk a contiguous unit range (tiles along a row first).  Piece p of a stage = halo pixels 16p .. 16p+15.
        // PAT 2: the same with tiles of 8x64 pixels (halo 10x66: rows twice as long).
        constexpr int TW_ = PAT == 2 ? 64 : 32, TH_ = PAT == 2 ? 8 : 16, IW_ = TW_ + 2, IH_ = TH_ + 2, NPIX = IW_ * IH_;
        constexpr int HP = (NPIX + 15) / 16;                 // KiB pieces of a halo stage (39 for 18x34, 42 for 10x66)
        int tab[(HP + NLW - 1) / NLW];
        if (PAT) {
#pragma unroll
            for (int j = 0; j < (HP + NLW - 1) / NLW; ++j) {
                const int pix = (j * NLW + iw) * 16 + (lane >> 2);
                const int iy = pix / IW_, ix = pix - iy * IW_;
                tab[j] = pix < NPIX ? (iy * 256 + ix) * 64 + (lane & 3) * 16 : 0x7fffffff;      // beyond num_records: zero fill
            }
        }
        const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7ffffff0, 0x00020000);
        const int ubase = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (nchunk >> 2);
        const int ulo = (blockIdx.x & 7) * (gridDim.x >> 3) * (nchunk >> 2) + (blockIdx.x >> 3), ugw = gridDim.x >> 3;
        auto issue = [&](int c) {
            if (!DMA) return;
            char* st = smem + (c % NSTG) * SB;
            if (PAT) {
                const int u = ORD ? ulo + (c >> 2) * ugw : ubase + (c >> 2), plane = c & 3;
                constexpr int TX = 256 / TW_, TY = 256 / TH_;
                const int tx = u % TX, ty = (u / TX) % TY, img = (u / (TX * TY)) & imask;
                long org = (((long)img * 256 + ty * TH_ - 1) * 256 + tx * TW_ - 1) * 64;
                if (org < 0) org = 0;
                const char* b = src + (long)plane * (16L * 256 * 256 * 64) + org;
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, IH_ * 256 * 64, 0x00020000);
#pragma unroll
                for (int j = 0; j < (HP + NLW - 1) / NLW; ++j)
                    if (j * NLW + iw < HP) dma<16>(rr, tab[j], 0, (lptr_t)(st + (j * NLW + iw) * 1024));
                return;
            }
            const int soff = (int)(((long)c * SB) % (bytes_per_wg - SB));
#pragma unroll
            for (int p = iw; p < SP; p += NLW)
#pragma unroll
                for (int q = 0; q < PPI; ++q)
                    dma<WIDTH>(r, voff, soff + p * 1024 + q * 64 * WIDTH, (lptr_t)(st + p * 1024 + q * 64 * WIDTH));
        };
        for (int c = 0; c < NSTG - 1; ++c) issue(c);
        for (int c = 0; c < nchunk; ++c) {
            // stage c must have landed: with NSTG - 1 stages in flight wait for all but the NSTG - 2 youngest
            if (NSTG == 2) bar_dma();
            else { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(((SP + NLW - 1) / NLW) * PPI * (NSTG - 2)) : "memory"); }
            if (c + NSTG - 1 < nchunk) issue(c + NSTG - 1);     // into the stage the MFMA waves left at this barrier
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                      // nothing in flight towards LDS when the workgroup ends
    } else {
        f32x16 acc[4] = {};
        f32x4 acc4[8] = {};
        bf16x8 fr[6];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) fr[i][j] = (__bf16)(float)(lane + i + j);
        const char* rd0 = smem + lane * 16 + wave * 1024;
        // EPI 5 / 6 (round 3): the finished unit's accumulators are copied to a second set and its four 16-byte stores (with EPV
        // dummy conversion instructions each, standing in for bias / LeakyReLU / packing) are issued INSIDE the next unit's MFMA
        // groups -- EPI 5: all four in the next unit's first chunk (groups 3, 9, 15, 21); EPI 6: one per chunk of the next unit.
        f32x16 accp[2] = {};
        auto store_piece = [&](int uc, int k) __attribute__((always_inline)) {      // piece k = (row q, half g) of the unit that ended with chunk uc
            const int ulo = (blockIdx.x & 7) * (gridDim.x >> 3) * (nchunk >> 2) + (blockIdx.x >> 3), ugw = gridDim.x >> 3;
            const int ubase = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (nchunk >> 2);
            const int u = ORD ? ulo + (uc >> 2) * ugw : ubase + (uc >> 2);
            const int tx = u % 8, ty = (u / 8) % 16, img = (u / 128) & imask;
            const int q = k >> 1, g = k & 1;
            char* row = dst + (((((long)img * 256 + ty * 16 + wave * 2 + q) * 256 + tx * 32) * 64) & dmask);
            char* o = row + (lane & 31) * 64 + (lane >> 5) * 16;
            float4 v = {accp[q][8 * g], accp[q][8 * g + 1], accp[q][8 * g + 2], accp[q][8 * g + 3]};
            if (EPV) {          // stand-in for the production epilogue's arithmetic on the 8 values a 16-byte store carries
                float w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { float t = accp[q][8 * g + i] + 0.25f; t = t > 0.f ? t : t * 0.2f; w[i] = t; }
                v.x = w[0] + w[4]; v.y = w[1] + w[5]; v.z = w[2] + w[6]; v.w = w[3] + w[7];
            }
            *(float4*)(o + 32 * g) = v;
        };
        for (int c = 0; c < nchunk; ++c) {
            bar_lds();                                           // stage c has landed, everyone left stage c - 1
            const char* st = rd0 + (c % NSTG) * SB;
            constexpr int G = NMF > 0 ? NMF / 6 : 1;             // 6 groups per chunk
            if constexpr (RL == 1) {
                // the convolution kernel's own loop (conv3x3_ls_k, MT = 1, PT = 2): row-ordered groups (input row i, tap kx), one pixel
                // fragment per group from the swizzled 18x34 halo image, weight fragments from the resident region behind the stages
                // (chunk c & 3), 6-slot weight ring, reads 2 groups ahead, 2 accumulators
                constexpr int IWT = 34, PT_ = 2, NGRP = 12, NG2 = 24, PD = 2, NRB = 3, NRA = 6;
                const int r = lane & 31, h = lane >> 5;
                const char* ls = smem + (c % NSTG) * SB;
                const char* lsw = smem + NSTG * SB + (c & 3) * (9 * 32 * 64);
                const int L0 = wave * PT_ * IWT + r;
                const int pa = r * 64 + ((h ^ ((r >> 2) & 3)) * 16);
                bf16x8 fa[NRA], fb[NRB];
                auto read_b = [&](int GG) {
                    const int ks = GG / NGRP, g = GG % NGRP;
                    const int lp = L0 + (g / 3) * IWT + (g % 3);
                    const int pb = lp * 64 + ((h ^ ((lp >> 2) & 3)) * 16);
                    fb[GG % NRB] = *(const bf16x8*)(ls + (pb ^ (ks * 32)));
                };
                auto read_a = [&](int GG) {
                    const int ks = GG / NGRP, g = GG % NGRP;
                    if (g < 9) fa[g % NRA] = *(const bf16x8*)(lsw + (pa ^ (ks * 32)) + (g * 32) * 64);
                };
#pragma unroll
                for (int GG = 0; GG < PD; ++GG) { read_a(GG); read_b(GG); }
#pragma unroll
                for (int GG = 0; GG < NG2; ++GG) {
                    if (GG + PD < NG2) { read_a(GG + PD); read_b(GG + PD); }
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = GG % NGRP, i = g / 3, kx = g % 3;
#pragma unroll
                    for (int q = 0; q < PT_; ++q) {
                        const int ky = i - q;
                        if (ky < 0 || ky > 2) continue;
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(ky * 3 + kx) % NRA], fb[GG % NRB], acc[q], 0, 0, 0);
                    }
                    if constexpr (EPI == 5) { if (GG % 6 == 3 && c >= 4 && (c & 3) == 0) store_piece(c - 1, GG / 6); }
                    if constexpr (EPI == 6) { if (GG == 9 && c >= 4) store_piece((c & ~3) - 1, c & 3); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (RD) {
#pragma unroll
                    for (int i = 0; i < NRD / 6; ++i) {
                        // conflict-free: 64 lanes x 16 B contiguous
                        const bf16x8 v = *(const bf16x8*)(st + ((g * (NRD / 6) + i) * 1024) % (SB - 8 * 1024));
                        fr[i % 6] = v;
                    }
                }
                if (MF && RL == 2) {          // the same FLOPs as 16x16x32 MFMAs (two per 32x32x16)
#pragma unroll
                    for (int i = 0; i < 2 * G; ++i) acc4[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i % 6], fr[(i + 1) % 6], acc4[i & 7], 0, 0, 0);
                } else if (MF) {
#pragma unroll
                    for (int i = 0; i < G; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % 6], fr[(i + 1) % 6], acc[i & 3], 0, 0, 0);
                } else if (RD) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(fr[i]));
                }
            }
            if ((EPI == 5 || EPI == 6) && (c & 3) == 3) { accp[0] = acc[0]; accp[1] = acc[1]; }
            else if (EPI && (c & 3) == 3) {
                // a unit's output: 16x32 pixels x 32 channels bf16 in one plane of the same blocked tensor; this wave's 2 rows, one pixel
                // per lane pair, four 8-byte pieces per lane (the accumulator layout's store pattern)
                const int ulo = (blockIdx.x & 7) * (gridDim.x >> 3) * (nchunk >> 2) + (blockIdx.x >> 3), ugw = gridDim.x >> 3;
                const int ubase = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (nchunk >> 2);
                const int u = ORD ? ulo + (c >> 2) * ugw : ubase + (c >> 2);
                const int tx = u % 8, ty = (u / 8) % 16, img = (u / 128) & imask;
                // EPI 1: the accumulator layout's store (8 B per lane, lanes r / r+32 fill a 16-byte piece, 4 instructions per row)
                // EPI 2: 16 B per lane, lanes r / r+32 fill 32 contiguous bytes of pixel r (after a half-wave exchange), 2 per row
                // EPI 3: 16 B per lane, 4 consecutive lanes = one pixel's 64 B (after a transposition), 2 instructions per row
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    char* row = dst + (((((long)img * 256 + ty * 16 + wave * 2 + q) * 256 + tx * 32) * 64) & dmask);
                    if (EPI == 1) {
                        char* o = row + (lane & 31) * 64 + (lane >> 5) * 8;
#pragma unroll
                        for (int g = 0; g < 4; ++g) { float2 v = {acc[q][4 * g], acc[q][4 * g + 1]}; *(float2*)(o + 16 * g) = v; }
                    } else if (EPI == 4) {
                        char* o = row + (lane & 31) * 64 + (lane >> 5) * 16;
#pragma unroll
                        for (int g = 0; g < 2; ++g) { f32x4 v = {acc[q][8 * g], acc[q][8 * g + 1], acc[q][8 * g + 2], acc[q][8 * g + 3]}; __builtin_nontemporal_store(v, (f32x4*)(o + 32 * g)); }
                    } else if (EPI == 2) {
                        char* o = row + (lane & 31) * 64 + (lane >> 5) * 16;
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            float4 v = {acc[q][8 * g], acc[q][8 * g + 1], acc[q][8 * g + 2], acc[q][8 * g + 3]};
                            if (EPV) {
                                float w[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) { float t = acc[q][8 * g + i] + 0.25f; t = t > 0.f ? t : t * 0.2f; w[i] = t; }
                                v.x = w[0] + w[4]; v.y = w[1] + w[5]; v.z = w[2] + w[6]; v.w = w[3] + w[7];
                            }
                            *(float4*)(o + 32 * g) = v;
                        }
                    } else {
                        char* o = row + lane * 16;
#pragma unroll
                        for (int g = 0; g < 2; ++g) { float4 v = {acc[q][8 * g], acc[q][8 * g + 1], acc[q][8 * g + 2], acc[q][8 * g + 3]}; *(float4*)(o + 1024 * g) = v; }
                    }
                }
            }
        }
        if (EPI == 5 || EPI == 6) for (int k = 0; k < 4; ++k) store_piece(nchunk - 1, k);
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
        for (int i = 0; i < 8; ++i) s += acc4[i][0] + acc4[i][3];
        if (s == 123.456f) sink[0] = s;
    }
}

template <bool DMA, bool RD, bool MF, int NLW, int NSTG, int SP, int NRD, int NMF, int WIDTH = 16, int PAT = 0, int EPI = 0, int ORD = 0, int RL = 0, int EPV = 0>
static void run(const char* name, const char* buf, long total, float* sink, int nchunk, long span_per_wg, int imask = 15, long dmask = -1L) {
    auto kern = k<DMA, RD, MF, NLW, NSTG, SP, NRD, NMF, WIDTH, PAT, EPI, ORD, RL, EPV>;
    const size_t smem = (size_t)NSTG * SP * 1024 + (RL == 1 ? 72 * 1024 : 8 * 1024);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3((8 + NLW) * 64), smem, 0, buf, span_per_wg, nchunk, sink, (char*)buf + 4L * 16 * 256 * 256 * 64, imask, dmask);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = DMA ? 256.0 * nchunk * (PAT == 1 ? 18 * 34 * 64 : PAT == 2 ? 10 * 66 * 64 : SP * 1024) : 0, flops = MF ? 256.0 * 8 * nchunk * (NMF / 6 * 6) * 32768.0 : 0;
    printf("%-46s %8.1f us  %5.2f TB/s  %5.2f PFLOP/s  (%.2f us per chunk)\n", name, best * 1e3, bytes / best / 1e9, flops / best / 1e12, best * 1e3 / nchunk);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
}

int main(int argc, char** argv) {
    // usage: ingest_test.bin [const|random] [all]   -- operand bits; "all" adds the exploratory variants
    const bool constant = argc > 1 && argv[1][0] == 'c', all = argc > 2 && argv[2][0] == 'a';
    const long total = 1L << 30;
    char* buf; float* sink;
    hipMalloc(&buf, total + 4096);
    if (constant) hipMemset(buf, 1, total);
    else hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, (unsigned short*)buf, total / 2);
    hipDeviceSynchronize(); hipMalloc(&sink, 64);
    const long span = total / 256;
    printf("operand bits: %s.  Shape of the 128 -> 32 dense-block convolution at the bench size: 2048 units x 4 chunks, 268 MB in, 67 MB out, 77 GFLOP\n",
           constant ? "constant (every byte 0x01)" : "random bf16 in (-0.5, 0.5)");
    for (int rep = 0; rep < 2; ++rep) {
        run<true,  false, false, 8, 2, 40, 42, 36, 16, 1, 0, 1, 1>("DMA only (halo stages)", buf, total, sink, 32, span);
        run<false, true,  true,  8, 2, 40, 42, 36, 16, 1, 0, 1, 1>("MFMA loop only (kernel's reads + MFMAs)", buf, total, sink, 32, span);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 0, 1, 1>("DMA + MFMA loop", buf, total, sink, 32, span);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 1, 1, 1>("DMA + MFMA loop + stores, 8 B per lane", buf, total, sink, 32, span);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 2, 1, 1>("DMA + MFMA loop + stores, 16 B lane pairs", buf, total, sink, 32, span);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 4, 1, 1>("  the same, nontemporal", buf, total, sink, 32, span);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 3, 1, 1>("DMA + MFMA loop + stores, 1 KiB contiguous", buf, total, sink, 32, span);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 2, 1, 1>("16 B stores into a 4 MiB window", buf, total, sink, 32, span, 15, (4L << 20) - 1);
        run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 2, 1, 1>("16 B stores, 4 images (cache-resident)", buf, total, sink, 32, span, 3);
        run<false, true,  true,  8, 2, 40, 42, 36, 16, 1, 2, 1, 1>("MFMA loop + 16 B stores (no DMA)", buf, total, sink, 32, span);
    }
    if (argc > 2 && argv[2][0] == 'e') {
        // round 3: the unit's stores issued inside the NEXT unit's MFMA groups (second accumulator set), interleaved A/B, 3 rounds
        for (int rep = 0; rep < 3; ++rep) {
            run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 0, 1, 1>("8 loaders: DMA + MFMA loop (no stores)", buf, total, sink, 32, span);
            run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 2, 1, 1, 1>("8 loaders: stores + arithmetic at unit end", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 0, 1, 1>("4 loaders: DMA + MFMA loop (no stores)", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 2, 1, 1>("4 loaders: stores at unit end", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 2, 1, 1, 1>("4 loaders: stores + arithmetic at unit end", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 5, 1, 1>("4 loaders: stores deferred into next chunk 0", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 5, 1, 1, 1>("4 loaders: stores + arithmetic deferred, chunk 0", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 6, 1, 1>("4 loaders: stores deferred, one per chunk", buf, total, sink, 32, span);
            run<true,  true,  true,  4, 2, 40, 42, 36, 16, 1, 6, 1, 1, 1>("4 loaders: stores + arithmetic deferred, 1/chunk", buf, total, sink, 32, span);
            run<false, true,  true,  4, 2, 40, 42, 36, 16, 1, 6, 1, 1, 1>("  the same without DMA", buf, total, sink, 32, span);
            run<false, true,  true,  4, 2, 40, 42, 36, 16, 1, 2, 1, 1, 1>("  unit-end stores + arithmetic without DMA", buf, total, sink, 32, span);
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 's') {
        // round 3: the MFMA shape in the Cout = 64 regime (72 MFMA-equivalents and 42 fragment reads per wave and chunk, DMA running):
        // 32x32x16 against 16x16x32 (two per 32x32x16: same FLOPs, same LDS bytes), interleaved, 4 rounds
        for (int rep = 0; rep < 4; ++rep) {
            run<true,  true,  true,  8, 2, 40, 42, 72>("Cout=64 ratio, 32x32x16", buf, total, sink, 96, span);
            run<true,  true,  true,  8, 2, 40, 42, 72, 16, 0, 0, 0, 2>("Cout=64 ratio, 16x16x32", buf, total, sink, 96, span);
            run<false, true,  true,  8, 2, 40, 42, 72>("  no DMA, 32x32x16", buf, total, sink, 96, span);
            run<false, true,  true,  8, 2, 40, 42, 72, 16, 0, 0, 0, 2>("  no DMA, 16x16x32", buf, total, sink, 96, span);
        }
        return 0;
    }
    if (!all) return 0;
    const int NC = 96;
    printf("-- flat 1-KiB pieces streamed from 1 GiB, generic read/MFMA groups\n");
    run<true,  false, false, 8, 2, 40, 42, 36>("DMA only", buf, total, sink, NC, span);
    run<false, true,  true,  8, 2, 40, 42, 36>("reads + MFMA only", buf, total, sink, NC, span);
    run<true,  true,  true,  8, 2, 40, 42, 36>("DMA + reads + MFMA", buf, total, sink, NC, span);
    run<true,  false, true,  8, 2, 40, 42, 36>("DMA + MFMA (no reads)", buf, total, sink, NC, span);
    run<true,  true,  false, 8, 2, 40, 42, 36>("DMA + reads (no MFMA)", buf, total, sink, NC, span);
    run<true,  true,  true,  8, 3, 40, 42, 36>("all, 3 stages", buf, total, sink, NC, span);
    run<true,  true,  true,  4, 2, 40, 42, 36>("all, 4 loader waves", buf, total, sink, NC, span);
    run<true,  true,  true,  8, 2, 40, 42, 72>("all, 72 MFMAs per chunk (Cout = 64 ratio)", buf, total, sink, NC, span);
    run<true,  true,  true,  8, 2, 40, 42, 36, 4>("all, 4-byte DMA lanes", buf, total, sink, NC, span);
    run<true,  true,  true,  8, 2, 40, 42, 36, 16, 1, 0, 1, 2>("halo stages, 16x16x32 MFMAs", buf, total, sink, 32, span);
    run<true,  true,  true,  8, 2, 42, 42, 36, 16, 2>("8x64-pixel tiles: DMA + reads + MFMA", buf, total, sink, 32, span);
    return 0;
}
