/*
 * srcgan_amd.h -- C ABI of the MI355X (gfx950) native SRCGAN training hot path.
 *
 * Drop-in boundary.  The reference (huster-wgm/SRCGAN) has no native code: its
 * hot path is torch.nn modules executed by ATen.  Each
 * entry point below replaces the ATen work behind one reference call site; the
 * Python host (srcgan_amd/) mirrors the reference's nn.Module / loss classes and
 * reaches these symbols through ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless said
 *     otherwise; `stream` is a hipStream_t passed as void*.
 *   - activations are NHWC ("channels-last") with an explicit channel stride
 *     (`cs`, elements per pixel) and channel offset (`coff`) so a convolution
 *     can read a channel prefix of a dense-block buffer and write its own
 *     channel slice -- this is what removes torch.cat (rddb.py:64-67).
 *   - dtype: SRCGAN_F32 (exact f32 MFMA, parity mode), SRCGAN_BF16 (bf16
 *     storage + bf16 MFMA, f32 accumulate; perf mode) or SRCGAN_F16 (the same
 *     kernels on IEEE half).
 *   - return value: 0 on success, non-zero on error; srcgan_last_error() gives
 *     the message (thread-local).  Shape/alignment violations are rejected on
 *     the host before any launch.
 */
#ifndef SRCGAN_AMD_H
#define SRCGAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRCGAN_F32 0
#define SRCGAN_BF16 1
#define SRCGAN_F16 2        /* IEEE half storage + f16 MFMA (same rate as bf16), f32 accumulate: 3 more mantissa bits than bf16, 5-bit exponent
                               (BASELINE.json configs[4]); gradients below 2^-24 vanish, so training harnesses scale the loss */

int srcgan_version(void);
const char* srcgan_last_error(void);
/* bytes of one element of dtype */
int srcgan_dtype_size(int dtype);

/* ---------------------------------------------------------------------------
 * Layout: NCHW f32 (the reference's tensor format, dataset.py:131) <-> NHWC.
 * to_nhwc writes channels [0,C) and zero-fills [C,cs).
 * ------------------------------------------------------------------------- */
int srcgan_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W,
                            int cs, int dtype, void* stream);
int srcgan_nhwc_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W,
                            int cs, int coff, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Weight packing.  Canonical torch weights stay f32 [Cout,Cin,kh,kw] (Conv2d) or
 * [Cin,Cout,kh,kw] (ConvTranspose2d) so state_dict / Adam are untouched; kernels
 * read a packed copy  Wp[row_tile][k_chunk][tap][row][k]  (dtype, zero padded).
 * packed(row r, k, tap (ty,tx)) = w[off + r*sr + k*sk + ty*sty + tx*stx].
 * The same routine produces forward, flipped/transposed dgrad and stride-2
 * parity-class packs by choice of strides.
 * ------------------------------------------------------------------------- */
size_t srcgan_packed_weight_bytes(int rows, int kdim, int ntaps, int dtype);
int srcgan_pack_weight(const float* w, void* wp, int rows, int kdim, int tys, int txs,
                       long sr, long sk, long sty, long stx, long off, int dtype, void* stream);
/* Fill only the k-range [k_off, k_off+kdim) of a packed matrix whose full K is k_total, values scaled by
 * `scale` (the caller zeroes the buffer first).  Used to assemble the composite transposed weights of the
 * dense-block backward: K = [conv5 co | conv4 co | ...] for one input-channel slice. */
int srcgan_pack_weight_part(const float* w, void* wp, int rows, int kdim, int tys, int txs,
                            long sr, long sk, long sty, long stx, long off, int k_off, int k_total, float scale,
                            int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Implicit-GEMM convolution (forward form).  Replaces aten::convolution for
 *   rddb.py:52-58 (3x3 s1), rddb.py:28-38 (k2 s2 deconv = 4 x 1x1 + pixel
 *   shuffle store), model/model.py:612-634 (4x4 s2 / s1), and -- with packed
 *   transposed weights -- every dgrad of aten::convolution_backward.
 *
 *   v   = alpha * (conv(x)[co] + bias[co])
 *       + (co < r1_cend ? beta1 * r1 : 0) + (co < r2_cend ? beta2 * r2 : 0)
 *   v   = act ? leaky_relu(v, slope) : v
 *   v  *= (co >= mz_c0 && mz) ? (mz > 0 ? 1 : mslope) : 1      (LeakyReLU' mask)
 *   y[b, oy*os+oa, ox*os+ob, ycoff+co] = v
 * r1/r2/mz are indexed like y.  r1 may alias y (in-place accumulate).
 * ------------------------------------------------------------------------- */
typedef struct srcgan_conv_desc {
    const void* x; const void* wp; const float* bias; void* y;
    const void* r1; const void* r2; const void* mz;
    int dtype;
    int kh, kw, stride;
    int B, H, W, Cin, x_cs, x_coff;         /* input tensor; Cin = channels read (multiple of 16B piece) */
    int OH, OW, Cout;                       /* conv output extent and true Cout */
    int YH, YW, y_cs, y_coff;               /* output tensor extent (after os scaling) */
    int pad_y, pad_x, os, oa, ob;
    int r1_cs, r1_coff, r1_cend;
    int r2_cs, r2_coff, r2_cend;
    int mz_cs, mz_coff, mz_c0;
    float alpha, beta1, beta2, slope, mslope;
    int act;
    /* plane strides in BYTES for the blocked layout (0 = interleaved NHWC): channel c of pixel q lives at
     * q*cs*esz + (c/KCE)*plane + (c%KCE)*esz with KCE = 64/esz channels; blocked tensors use cs = KCE. */
    long x_plane, y_plane, r1_plane, r2_plane, mz_plane;
    int rev_batch;     /* walk the images in reverse order (3x3 s1 kernel): consecutive layers alternate so that a layer starts on
                          the data its predecessor touched last (Infinity Cache reuse when a layer's footprint exceeds 256 MB) */
    /* LeakyReLU sign masks, one bit per channel instead of re-reading the 2-byte activation in the backward pass (3x3 s1 bf16,
     * Cout == 32, os == 1, blocked input: the dense-block convs).  u32 per output pixel (b, oy, ox), bit c = output channel c.
     *   sign_out: written by a forward conv with act != 0 (bit = activation output > 0)
     *   sign_in : read instead of mz:  v *= bit ? 1 : mslope
     * 64 channels (round 3; Cout / 8 = 8 bytes per output pixel, byte c / 8, bit c % 8): sign_out of the 1x1 four-parity form with act
     * (the up-sampler's last stage, rddb.py:93-97), sign_in of a 3x3 s1 convolution with Cout == 64 and no other epilogue operand
     * (conv_last's input gradient, rddb.py:98,113).
     * A descriptor that sets either and does not meet the conditions is refused (no silent fallback). */
    void* sign_out; const void* sign_in;
    /* npar == 4: the input gradient of a 4x4 stride-2 pad-1 convolution (model/model.py:612-634) with all four output parities in
     * ONE launch.  x = dy [B,H,W,Cin = the layer's Cout], y = dx [B,YH,YW,..] (Cout = the layer's Cin), kh = kw = 2, stride = 1:
     *   dx[2t+a][2u+b] = sum_{ty,tx in {0,1}} dy[t+a-1+ty][u+b-1+tx] * pack_q[tap (ty,tx)],   q = 2a + b,
     * pack_q = wp + q * wpar_stride bytes (rows = the layer's Cin, k = its Cout, taps ky = (a?2:3) - 2ty, kx = (b?2:3) - 2tx).
     * OH / OW / pad / os / oa / ob are derived from YH, YW; epilogue operands (mz, r1, ...) are indexed like y.  npar == 0: plain.
     * npar == 4 with kh = kw = 1 (ConvTranspose2d k2 s2 as four 1x1 convolutions, rddb.py:28-38, in one launch): os = 2, oa = ob = 0;
     *   y[2oy+a][2ox+b] = epilogue(x[oy][ox] * pack_q),  q = 2a + b,  pack_q = wp + q * wpar_stride bytes (Cout rows each). */
    int npar; long wpar_stride;
} srcgan_conv_desc;
int srcgan_conv_igemm(const srcgan_conv_desc* d, void* stream);

/* ---------------------------------------------------------------------------
 * Weight gradient (the wgrad third of aten::convolution_backward):
 *   dW[co, tap, ci] = alpha * sum_{b,oy,ox} dy[b,oy,ox,co] * x[b, oy*s+ky-pad, ox*s+kx-pad, ci]
 * computed split-K over pixel ranges into an f32 slab, then reduced in a fixed
 * order (deterministic) and scattered to the canonical f32 gradient:
 *   grad[off + co*sr + ci*sk + ky*sty + kx*stx] (=|+=) value
 * ------------------------------------------------------------------------- */
typedef struct srcgan_wgrad_desc {
    const void* dy; const void* x; float* slab; float* grad;
    float* bias_grad;                       /* optional: alpha * sum_p dy[p,co], fused (3x3 kernels only) */
    int dtype;
    int kh, kw, stride;
    int B, H, W, Cin, x_cs, x_coff;         /* x tensor; Cin = true input channels */
    int OH, OW, Cout, dy_cs, dy_coff;       /* dy tensor; Cout = true output channels */
    int pad_y, pad_x;
    int nsplit;
    long sr, sk, sty, stx, off;
    float alpha;
    int accumulate;
} srcgan_wgrad_desc;
/* slab bytes needed for (Cout,Cin,kh,kw,nsplit) */
size_t srcgan_conv_wgrad_slab_bytes(int Cout, int Cin, int kh, int kw, int nsplit);
/* a good nsplit for this problem (fills the chip, bounded slab) */
int srcgan_conv_wgrad_nsplit(int B, int OH, int OW, int Cout, int Cin, int stride);
int srcgan_conv_wgrad(const srcgan_wgrad_desc* d, void* stream);

/* ---------------------------------------------------------------------------
 * Dense-block weight gradient (3x3, stride 1, pad 1): ONE pass over a ResidualDenseBlock_5's activation buffer x
 * (C channels) and its dense gradient buffer dy (G channels = [dy5|dy4|dy3|dy2|dy1]) yields the weight and bias
 * gradients of all five convolutions (rddb.py:52-58).  Rows [g0,g1) of dy belong to a Conv2d weight
 * grad[g1-g0][Cin][3][3] (canonical f32) with bias gradient bias[g1-g0]; values are scaled by alpha.
 * Big (128 x 64 channel) tiles: 346 FLOP per staged byte vs 124-190 for srcgan_conv_wgrad.
 * ------------------------------------------------------------------------- */
typedef struct srcgan_wgrad_seg { int g0, g1; float* grad; float* bias; int Cin; float alpha; } srcgan_wgrad_seg;
typedef struct srcgan_wgrad_dense_desc {
    const void* dy; const void* x; float* slab;
    int dtype;
    int B, H, W;
    int G, dy_cs, dy_coff;
    int C, x_cs, x_coff;
    int nseg;
    srcgan_wgrad_seg seg[8];
    int accumulate;
    long dy_plane, x_plane;                 /* blocked-layout plane strides in bytes (0 = interleaved NHWC) */
} srcgan_wgrad_dense_desc;
size_t srcgan_wgrad_dense_slab_bytes(int G, int C, int dtype, int B, int H, int W);
int srcgan_wgrad_dense(const srcgan_wgrad_dense_desc* d, void* stream);

/* ---------------------------------------------------------------------------
 * Column reductions over pixels (deterministic two-stage):
 *   mode 0: out0[c] = scale * sum a[p,c]                         (bias grad; BN mean)
 *   mode 1: out0[c] = scale * sum (a[p,c]-m[c])^2                (BN variance)
 *   mode 2: out0[c] = sum g[p,c] ; out1[c] = sum g[p,c]*(z[p,c]-m[c])*rstd[c]   (BN backward)
 *   mode 3: out0[c] = mean of a[p,c] ; out1[c] = its biased variance -- ONE pass over a (per-thread shifted sums, partials
 *           combined exactly in a fixed order); `scale` and `m` are not used                       (BN statistics, training)
 * scratch: 2*nblk*C floats, nblk = srcgan_col_reduce_blocks(npix).
 * ------------------------------------------------------------------------- */
int srcgan_col_reduce_blocks(long npix);
int srcgan_col_reduce(int mode, const void* a, int a_cs, int a_coff, const void* z, int z_cs, int z_coff,
                      const float* m, const float* rstd, long npix, int C, float scale,
                      float* out0, float* out1, float* scratch, int dtype, void* stream);

/* BatchNorm2d(train)+LeakyReLU (model/model.py:622-623,630-631), NHWC.
 * bn_finalize: mean/var -> rstd, running-stat update (momentum .1, unbiased var), nbt++.
 * bn_apply:    y = lrelu(gamma*(z-mean)*rstd+beta)
 * bn_bwd_apply:dz = gamma*rstd*(g - sum_g/N - xhat*sum_gx/N)   (g already holds dy*lrelu'(y)) */
int srcgan_bn_finalize(const float* mean, const float* var, float* rstd, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, int C, long count,
                       float momentum, float eps, void* stream);
int srcgan_bn_eval_rstd(const float* running_var, float* rstd, int C, float eps, void* stream);
int srcgan_bn_apply_lrelu(const void* z, void* y, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, long npix, int C, int cs, float slope, int dtype, void* stream);
int srcgan_bn_bwd_apply(const void* g, const void* z, void* dz, const float* mean, const float* rstd,
                        const float* gamma, const float* sum_g, const float* sum_gx, long npix, int C, int cs,
                        int dtype, void* stream);

/* y[p, ycoff+c] = (y + x[p, xcoff+c]) * (mz ? (mz[p, mzcoff+c] > 0 ? 1 : mslope) : 1) for c < C
 * (residual gradient joins; optional LeakyReLU' of the tensor the gradient belongs to) */
int srcgan_add_inplace(void* y, int y_cs, int y_coff, const void* x, int x_cs, int x_coff,
                       const void* mz, int mz_cs, int mz_coff, float mslope, long npix, int C, int dtype, void* stream);
/* same with blocked-layout plane strides (bytes, 0 = interleaved NHWC) for each tensor */
int srcgan_add_inplace_planes(void* y, int y_cs, int y_coff, long y_plane, const void* x, int x_cs, int x_coff, long x_plane,
                              const void* mz, int mz_cs, int mz_coff, long mz_plane, float mslope, long npix, int C,
                              int dtype, void* stream);

/* GroupNorm (+ residual add + ReLU) on NHWC activations (resdeconv.py:61-97,118-121; nn.GroupNorm(G, C), affine, biased
 * variance): y = act?((x - mean[b][g]) * rstd[b][g] * gamma[c] + beta[c] [+ res]), act = (Leaky)ReLU with `slope` (0 = ReLU;
 * edsr.py:43-49 uses 0.2) when relu != 0.  stats: device f32 [B][G][2] = {mean, rstd}
 * (written by forward, read by backward).  backward: g = dy [* (yact > 0)] (yact = the forward output when ReLU was applied);
 * dx = GroupNorm backward of g; dres (optional) = g (gradient of the residual branch; += when dres_accumulate); dgamma / dbeta
 * (optional) f32 [C], += when accumulate (a GroupNorm module applied twice, edsr.py:41,47,49).
 * scratch: srcgan_gn_scratch_floats(B, C) floats.  C/epp must divide 256 (epp = 16 bytes of channels).
 * gamma and / or beta may be null: no affine part -- with G == C that is nn.InstanceNorm2d(C) (model/model.py:598-631 with
 * norm_layer = InstanceNorm2d, basicModel.py:24-25). */
size_t srcgan_gn_scratch_floats(int B, int C);
int srcgan_gn_forward(const void* x, int x_cs, const void* res, int res_cs, void* y, int y_cs, const float* gamma, const float* beta,
                      float* stats, int B, long hw, int C, int G, float eps, int relu, float slope, int dtype, float* scratch, void* stream);
int srcgan_gn_backward(const void* dy, int dy_cs, const void* yact, int ya_cs, const void* x, int x_cs, const float* gamma, const float* stats,
                       void* dx, int dx_cs, void* dres, int dres_cs, int dres_accumulate, float* dgamma, float* dbeta, int accumulate,
                       float slope, int B, long hw, int C, int G, int dtype, float* scratch, void* stream);

/* x2 nearest up-sampling of an NHWC feature map (src may be a channel slice of a blocked buffer: s_plane != 0) and its
 * adjoint: dst[y][x] = sum of the 2x2 block of src, times LeakyReLU'(mz[y][x]) when mz is given.  Replaces
 * F.interpolate(scale_factor=2, mode='nearest') and its backward in the legacy generators (model/model.py:384-386,428-433). */
int srcgan_upsample2_nhwc(const void* src, int s_cs, int s_coff, long s_plane, void* dst, int d_cs,
                          int B, int H, int W, int C, int dtype, void* stream);
int srcgan_sum2x2_nhwc(const void* src, int s_cs, void* dst, int d_cs, const void* mz, int m_cs, float mslope,
                       int B, int H, int W, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Loss reductions on flat f32 arrays (replace aten::l1_loss / mse_loss and their
 * backward; losses.py:95-147, train.py:67-128).  out: device f32 scalar.
 *   kind 0: mean |a-b|          kind 1: mean (a-b)^2      kind 2: mean (a-label)^2
 * bwd: da[i] = gscale * d/da of the mean (gscale = upstream grad, device scalar * host scale)
 * scratch: srcgan_loss_scratch_floats() floats.
 * ------------------------------------------------------------------------- */
int srcgan_loss_scratch_floats(void);
int srcgan_loss_fwd(int kind, const float* a, const float* b, float label, long n, float* out,
                    float* scratch, void* stream);
int srcgan_loss_bwd(int kind, const float* a, const float* b, float label, long n, const float* gout,
                    float gscale, float* da, void* stream);
/* psnr = 10*log10(1/mse) from a device mse scalar (losses.py:144-147) */
int srcgan_psnr_from_mse(const float* mse, float* out, void* stream);

/* In-step preprocessing (trainCas.py:85-90): gray = .2125R+.7154G+.0721B (NCHW f32 in/out),
 * bilinear x(1/up) with align_corners=False == mean of the centre 2x2 of each up x up block. */
int srcgan_rgb_to_gray(const float* rgb, float* gray, int B, int H, int W, void* stream);
int srcgan_bilinear_down(const float* src, float* dst, int B, int C, int H, int W, int up, void* stream);
int srcgan_nearest_resize(const float* src, float* dst, int B, int C, int H, int W, int OH, int OW, void* stream);
/* bilinear x up (integer, align_corners=False): the second half of the blur of trainCasConst.py:89-92 */
int srcgan_bilinear_up(const float* src, float* dst, int B, int C, int H, int W, int up, void* stream);

/* ---------------------------------------------------------------------------
 * Whole-network passes (C++ sequencing of the kernels above; one call per
 * nn.Module.forward / autograd backward).
 *
 * RDDBNet (rddb.py:85-114).  params/grads: arrays of device f32 pointers in
 * state_dict order (conv_first.weight, conv_first.bias, RRDB_trunk.0.RDB1.conv1.weight, ...).
 * ------------------------------------------------------------------------- */
typedef struct srcgan_rddbnet_cfg {
    int in_ch, out_ch, up, nf, nb, gc;
    int B, H, W;
    int dtype;
    int down;          /* 0: RDDBNet (LR->HR, deconv up-sampler).  >0: HR->LR mirror
                          ("RDDBNetA", build-defined): `down` = /2^k factor, strided 3x3 convs */
    int legacy;        /* 0: rddb.py RDDBNet.  1: model/model.py:394-440 RDDBNetB (G_A of train.py:172): nearest x2 +
                          upconv1/upconv2, HRconv applied 8 times, conv_last with bias; `up` = 2 ('x2': upconv1 twice) or
                          4 ('x4').  2: model/model.py:347-391 legacy RDDBNet (its trunk result is discarded by the
                          reference's forward: not computed, no gradients); `up` = 1, 2 or 4.
                          3: srdn.py:56-74 SRDN (conv_first, RRDB_encoder, + skip, RRDB_decoder, + skip, conv_last; its trunk_conv
                          is never applied: pass it, it gets no gradient); `up` = 1, `nb` = RRDBs per stack.
                          params/grads follow the respective state_dict order. */
} srcgan_rddbnet_cfg;
/* Per-call options of the whole-network entry points (the plain forms pass NULL):
 *   wpack / pack    persistent packed-weight buffer (srcgan_*_wpack_bytes(cfg), 256-byte aligned).  NULL: the weights are packed
 *                   into the call's workspace every call.  Given: packed into it only when pack != 0, so a caller that tracks
 *                   weight updates packs ONCE per optimiser step although a cycle step runs each generator three times
 *                   (train.py:228-260).  Forward and backward packs are separate regions: pack the forward set in a forward
 *                   call, the backward set in a backward call.  The layout depends on the cfg's channel counts, dtype, up /
 *                   down / legacy (and for the discriminator on H, W parity), not on B, H, W.
 *                   pack == 2 ("verify"): `guard` points at two 64-bit device words {fingerprint the pack was made from,
 *                   fingerprint of the parameters now} (srcgan_params_fingerprint); the pack kernels run but do nothing when the
 *                   words are equal -- the decision is taken on the device, with no host synchronisation, so weight updates the
 *                   host cannot see (p.data.mul_(), raw-pointer writes) are still honoured.  The caller copies word 1 to word 0
 *                   after the call.
 *   rrdb_lo/rrdb_hi rddbnet backward only.  hi <= 0: the whole backward.  Otherwise this call handles the RRDBs [lo, hi), last
 *                   to first; the call with hi == number of RRDBs also runs everything behind the trunk (conv_last, up-sampler,
 *                   trunk_conv), the call with lo == 0 everything in front of it (conv_first, dx).  Calls come in descending,
 *                   gap-free order on one stream with the same ws / scratch / grads; when a call returns, the gradients of the
 *                   parameters it covers are final (data parallel: their all-reduce starts while earlier blocks still compute). */
typedef struct srcgan_net_opts {
    void* wpack;
    int pack;
    int rrdb_lo, rrdb_hi;
    const void* guard;
} srcgan_net_opts;
/* 64-bit order-independent fingerprint of a list of f32 tensors (every bit of every element).  table_dev: device int64
 * [ntensors][3] = {pointer, element count, first block}; tensor t is served by the blocks [first block(t), first block(t+1)) of 16384
 * elements each; nblocks = their total.  out_u64: one device word (zeroed and accumulated on `stream`). */
int srcgan_params_fingerprint(const void* table_dev, int ntensors, long nblocks, void* out_u64, void* stream);
int srcgan_rddbnet_num_params(const srcgan_rddbnet_cfg* c);
size_t srcgan_rddbnet_ws_bytes(const srcgan_rddbnet_cfg* c);        /* forward workspace (kept for backward) */
size_t srcgan_rddbnet_bwd_scratch_bytes(const srcgan_rddbnet_cfg* c);
int srcgan_rddbnet_forward(const srcgan_rddbnet_cfg* c, const float* x_nchw, const float* const* params,
                           void* ws, float* y_nchw, void* stream);
/* grads[i] may be NULL (parameter frozen); dx_nchw may be NULL */
int srcgan_rddbnet_backward(const srcgan_rddbnet_cfg* c, const float* dy_nchw, const float* const* params,
                            void* ws, void* scratch, float* const* grads, float* dx_nchw, void* stream);
size_t srcgan_rddbnet_wpack_bytes(const srcgan_rddbnet_cfg* c);
int srcgan_rddbnet_forward_ex(const srcgan_rddbnet_cfg* c, const float* x_nchw, const float* const* params,
                              void* ws, float* y_nchw, const srcgan_net_opts* opt, void* stream);
int srcgan_rddbnet_backward_ex(const srcgan_rddbnet_cfg* c, const float* dy_nchw, const float* const* params,
                               void* ws, void* scratch, float* const* grads, float* dx_nchw, const srcgan_net_opts* opt, void* stream);

/* NLayerDiscriminator (model/model.py:595-639).  params in state_dict order of the
 * learnable tensors: conv0.w, conv0.b, [conv_l.w, bn_l.gamma, bn_l.beta]*, conv_last.w, conv_last.b.
 * bn_state: per BN layer running_mean, running_var (f32) ; nbt: int64 counters.
 * norm = 1: norm_layer = nn.InstanceNorm2d (model/model.py:607-610: the normalised convolutions then HAVE a bias; InstanceNorm2d
 * as basicModel.py:24-25 builds it -- no affine parameters, no running statistics, instance statistics in train and eval mode):
 * learnable tensors conv0.w, conv0.b, [conv_l.w, conv_l.b]*, conv_last.w, conv_last.b; bn_running / bn_nbt are not read. */
typedef struct srcgan_nlayerd_cfg {
    int in_ch, ndf, n_layers;
    int B, H, W;
    int dtype;
    int training;
    int norm;              /* 0 = BatchNorm2d (the reference's default), 1 = InstanceNorm2d */
} srcgan_nlayerd_cfg;
int srcgan_nlayerd_num_params(const srcgan_nlayerd_cfg* c);
int srcgan_nlayerd_out_hw(const srcgan_nlayerd_cfg* c, int* oh, int* ow);
size_t srcgan_nlayerd_ws_bytes(const srcgan_nlayerd_cfg* c);
size_t srcgan_nlayerd_bwd_scratch_bytes(const srcgan_nlayerd_cfg* c);
int srcgan_nlayerd_forward(const srcgan_nlayerd_cfg* c, const float* x_nchw, const float* const* params,
                           float* const* bn_running, int64_t* const* bn_nbt, void* ws, float* y_nchw, void* stream);
int srcgan_nlayerd_backward(const srcgan_nlayerd_cfg* c, const float* dy_nchw, const float* const* params,
                            void* ws, void* scratch, float* const* grads, float* dx_nchw, void* stream);
size_t srcgan_nlayerd_wpack_bytes(const srcgan_nlayerd_cfg* c);
int srcgan_nlayerd_forward_ex(const srcgan_nlayerd_cfg* c, const float* x_nchw, const float* const* params,
                              float* const* bn_running, int64_t* const* bn_nbt, void* ws, float* y_nchw,
                              const srcgan_net_opts* opt, void* stream);
int srcgan_nlayerd_backward_ex(const srcgan_nlayerd_cfg* c, const float* dy_nchw, const float* const* params,
                               void* ws, void* scratch, float* const* grads, float* dx_nchw, const srcgan_net_opts* opt, void* stream);

/* ResDeconv colouriser (resdeconv.py:99-195; C network of trainCas.py:31,99-100): [B,3,H,W] f32 NCHW -> [B,tar_ch,H,W];
 * H, W multiples of 16.  params/grads in state_dict order (conv1.weight, bn1.{weight,bias}, layer1.0.conv1.weight, ...,
 * pred.weight); a null grads[i] skips that gradient.  The input gradient is optional (the reference harnesses feed data). */
typedef struct srcgan_resdeconv_cfg {
    int in_ch, out_ch;     /* in_ch must be 3 (a 1-channel source is replicated by the caller, resdeconv.py:166-167) */
    int B, H, W;
    int dtype;
    int layers[4];         /* BasicBlocks per stage, resdeconv.py:107,123-137 ([2,2,2,2] = ResNet-18 layout, [3,4,6,3] = ResNet-34);
                            * all zero = [2,2,2,2].  The up path uses layers[2], layers[1], layers[0]. */
    int norm;              /* 0 = BN='GN': nn.GroupNorm(32, C) (the reference's default); 1 = BN='IN': nn.InstanceNorm2d(C), which has
                            * no parameters -- the state_dict then holds the convolution weights only */
} srcgan_resdeconv_cfg;
int srcgan_resdeconv_num_params(const srcgan_resdeconv_cfg* c);
size_t srcgan_resdeconv_ws_bytes(const srcgan_resdeconv_cfg* c);
size_t srcgan_resdeconv_bwd_scratch_bytes(const srcgan_resdeconv_cfg* c);
int srcgan_resdeconv_forward(const srcgan_resdeconv_cfg* c, const float* x_nchw, const float* const* params, void* ws,
                             float* y_nchw, void* stream);
/* dx_nchw: gradient w.r.t. the input [B,3,H,W] f32, or NULL */
int srcgan_resdeconv_backward(const srcgan_resdeconv_cfg* c, const float* dy_nchw, const float* const* params, void* ws,
                              void* scratch, float* const* grads, float* dx_nchw, void* stream);

/* SR networks selectable as --SRModel (trainCas.py:169): kind 0 = ESPCN (espcn.py:18-51; the CLI default), kind 1 = SRCNN
 * (srcnn.py:17-42), kind 2 = EDSR (edsr.py:37-110: GroupNorm residual blocks, [B,out_ch,H*up,W*up]).  [B,in_ch,H,W] f32 NCHW -> ESPCN [B,out_ch,H*up,W*up] / SRCNN [B,out_ch,H,W].  params/grads in state_dict
 * order (conv1.weight, conv1.bias, ...).  dx_nchw: gradient w.r.t. the input [B,in_ch,H,W] f32, or NULL. */
typedef struct srcgan_srnet_cfg {
    int kind, in_ch, out_ch, up, base;     /* base = base_kernel / base_channel (64) */
    int B, H, W;
    int dtype;
    int nres;                              /* EDSR: num_residuals (50) */
} srcgan_srnet_cfg;
int srcgan_srnet_num_params(const srcgan_srnet_cfg* c);
size_t srcgan_srnet_ws_bytes(const srcgan_srnet_cfg* c);
size_t srcgan_srnet_bwd_scratch_bytes(const srcgan_srnet_cfg* c);
int srcgan_srnet_forward(const srcgan_srnet_cfg* c, const float* x_nchw, const float* const* params, void* ws, float* y_nchw, void* stream);
int srcgan_srnet_backward(const srcgan_srnet_cfg* c, const float* dy_nchw, const float* const* params, void* ws, void* scratch,
                          float* const* grads, float* dx_nchw, void* stream);

/* nn.PixelShuffle(r) on NHWC (espcn.py:44,50): src [B,H,W,C*r*r] -> dst [B,H*r,W*r,C]; inverse = 1: the adjoint, src [B,H*r,W*r,C]
 * -> dst [B,H,W,C*r*r].  srcgan_mask_inplace: g *= (act > 0 ? 1 : slope) over n elements (ReLU' / LeakyReLU' on an incoming gradient). */
int srcgan_pixel_shuffle_nhwc(const void* src, int s_cs, void* dst, int d_cs, int B, int H, int W, int C, int r, int inverse,
                              int dtype, void* stream);
int srcgan_mask_inplace(void* g, const void* act, float slope, long n, int dtype, void* stream);

/* Evaluation metrics on device (metrics.py:10-144; test loop testCas.py:65-90).  pred / truth: [B,C,H,W] f32 NCHW.
 *   srcgan_metric_ae:   out[b] = mean over pixels of acos(<p,t>/(|p||t| + 1e-6)) in degrees
 *   srcgan_metric_ssim: out[b][0] = mean SSIM (11x11 gaussian sigma 1.5 "valid" windows, dynamic range from the prediction's
 *                       min / max as metrics.py:100-107), out[b][1] = mean contrast term
 * MSE / PSNR: srcgan_loss_fwd(kind 1) + srcgan_psnr_from_mse.  scratch: srcgan_metric_scratch_floats(B, C, H, W) floats. */
int srcgan_metric_scratch_floats(int B, int C, int H, int W);
int srcgan_metric_ae(const float* pred, const float* truth, int B, int C, int H, int W, float* out, float* scratch, void* stream);
int srcgan_metric_ssim(const float* pred, const float* truth, int B, int C, int H, int W, float* out, float* scratch, void* stream);

/* Space-to-depth forms of an image-channel tensor for a 4x4 stride-2 pad-1 first layer (NLayerDiscriminator, model/model.py:612):
 * block (j,i) of the (H/2+1) x (W/2+1) grid = the 2x2 pixels (2j-1+dy, 2i-1+dx) as a 32-channel record [dy][dx][8], zero
 * outside the image / past C; the layer becomes a 2x2 stride-1 convolution with K = 4 x 32 and no padded K.
 * srcgan_s2d_wgrad_unfold maps the gradient of the folded weight [Cout][32][2][2] to the canonical [Cout][Cin][4][4]. */
int srcgan_nchw_f32_to_s2d(const float* src, void* dst, int B, int C, int H, int W, int dtype, void* stream);
int srcgan_s2d_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W, int dtype, void* stream);
int srcgan_s2d_wgrad_unfold(const float* gfold, float* grad, int Cout, int Cin, int accumulate, void* stream);

/* Input pipeline colour conversions on device (dataset.py:114-159 behind G2RGB / G2LAB.__getitem__ :179-199,:234-254; the
 * reference calls skimage.color per sample on the host).  rgb: [B,H*W,3] interleaved 8-bit; dst: f32 planes [B,C,H*W].
 *   mode 0: gray = rgb2gray(rgb)                               C = 1   (_arr2gray)
 *   mode 1: rgb / 255                                          C = 3   (_arr2rgb)
 *   mode 2: (L/100, (a+128)/255, (b+128)/255) of rgb2lab(rgb)  C = 3   (_arr2lab)
 *   mode 3: ((a+128)/255, (b+128)/255)                         C = 2   (_arr2ab)
 * srcgan_lab_planes_to_u8rgb is the inverse used for visualisation (_lab2img, dataset.py:92-104; truncating store). */
int srcgan_u8rgb_to_planes(const unsigned char* rgb, float* dst, int B, long hw, int mode, void* stream);
int srcgan_lab_planes_to_u8rgb(const float* lab, unsigned char* rgb, int B, long hw, void* stream);

/* Fused multi-tensor Adam (torch.optim.Adam.step() of trainCas.py:38-41,143-150 / train.py:191-192,331-340; torch's
 * single-tensor arithmetic, default flags: no weight decay, no amsgrad).  tensors_dev: device array of records
 * {float* p; const float* g; float* m; float* v;} (32 bytes); chunks_dev: device array of nchunks records
 * {int tensor; int off; int n; int pad;} (16 bytes; off a multiple of 4, n <= 4096).  `step` = the 1-based step count of
 * every tensor in the launch. */
int srcgan_adam_step(const void* tensors_dev, const void* chunks_dev, int nchunks, double lr, double beta1, double beta2, double eps,
                     long step, void* stream);

/* ---------------------------------------------------------------------------
 * Launch profiling for bench.py: when enabled, every conv_igemm / conv_wgrad launch is bracketed by
 * HIP events on its launch stream.  collect() synchronises, aggregates per kernel class
 * (template instance) and returns the number of classes; get(i) reads one aggregate:
 * launches, total ms, total algorithmic flop and bytes.
 * ------------------------------------------------------------------------- */
int srcgan_prof_enable(int on);
int srcgan_prof_collect(void);
int srcgan_prof_get(int i, const char** cls, long* count, double* ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SRCGAN_AMD_H */
