// Whole-network sequencing for the SRCGAN hot path: one C call per nn.Module.forward and one per
// autograd backward.  Pure host code that chains the gfx950 kernels of conv_igemm.hip,
// conv_wgrad.hip and elementwise.hip on the caller's stream (no allocation, no synchronisation:
// graph-capturable).
//
//   RDDBNet            reference src/model/rddb.py:48-114
//   NLayerDiscriminator reference src/model/model.py:595-639
//
// Memory design (HBM): activations live in NHWC.  Each ResidualDenseBlock_5 owns ONE dense buffer of
// nf+4*gc channels; conv k reads the channel prefix [0, nf+(k-1)gc) and writes its own slice, so the
// four torch.cat copies of rddb.py:64-67 (and CatBackward) never exist.  conv5's epilogue fuses
// "x5*0.2 + x" (rddb.py:68), and for RDB3 also the RRDB residual (rddb.py:82), writing straight into
// channels [0,nf) of the next block's dense buffer.  Backward mirrors this with a dense *gradient*
// buffer per block that the dgrads of conv5..conv1 accumulate into in place.
#include "common.h"
#include <vector>
#include <map>
#include <string>
#include <mutex>

namespace {

struct TRef { void* p; int cs; int coff; long plane; };     // plane: blocked-layout plane stride in bytes (0 = NHWC)
static inline TRef tref(void* p, int cs, int coff = 0, long plane = 0) { return TRef{p, cs, coff, plane}; }
static inline TRef sl(TRef t, int coff) { return TRef{t.p, t.cs, t.coff + coff, t.plane}; }
static const TRef TNULL = {nullptr, 0, 0, 0};

static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
static inline int img_cs(int c) { return round_up(c, 8); }     // image-like tensors: channels padded to 8

struct Conv {
    srcgan_conv_desc d;
    Conv(int dtype, int kh, int kw, int stride) {
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.kh = kh; d.kw = kw; d.stride = stride;
        d.os = 1; d.alpha = 1.f; d.slope = 0.2f; d.mslope = 0.2f;
    }
    Conv& in(TRef x, int B, int H, int W, int Cin) { d.x = x.p; d.x_cs = x.cs; d.x_coff = x.coff; d.x_plane = x.plane; d.B = B; d.H = H; d.W = W; d.Cin = Cin; return *this; }
    Conv& w(const void* wp, const float* bias = nullptr) { d.wp = wp; d.bias = bias; return *this; }
    Conv& out(TRef y, int OH, int OW, int Cout) { d.y = y.p; d.y_cs = y.cs; d.y_coff = y.coff; d.y_plane = y.plane; d.OH = OH; d.OW = OW; d.Cout = Cout; d.YH = OH; d.YW = OW; return *this; }
    Conv& pad(int py, int px) { d.pad_y = py; d.pad_x = px; return *this; }
    Conv& scatter(int os, int oa, int ob, int YH, int YW) { d.os = os; d.oa = oa; d.ob = ob; d.YH = YH; d.YW = YW; return *this; }
    Conv& alpha(float a) { d.alpha = a; return *this; }
    Conv& res1(TRef r, int cend, float beta) { d.r1 = r.p; d.r1_cs = r.cs; d.r1_coff = r.coff; d.r1_plane = r.plane; d.r1_cend = cend; d.beta1 = beta; return *this; }
    Conv& res2(TRef r, int cend, float beta) { d.r2 = r.p; d.r2_cs = r.cs; d.r2_coff = r.coff; d.r2_plane = r.plane; d.r2_cend = cend; d.beta2 = beta; return *this; }
    Conv& lrelu() { d.act = 1; return *this; }
    Conv& sign_out(void* m) { d.sign_out = m; return *this; }                  // write / read the LeakyReLU sign mask (u32 per pixel)
    Conv& sign_in(const void* m) { d.sign_in = m; return *this; }
    Conv& mask(TRef z, int c0) { d.mz = z.p; d.mz_cs = z.cs; d.mz_coff = z.coff; d.mz_plane = z.plane; d.mz_c0 = c0; return *this; }
    // consecutive 3x3 s1 launches walk the batch in alternating directions (SRCGAN_NO_ZIGZAG=1 disables): see srcgan_conv_desc.rev_batch
    int run(void* st) {
        static const bool zig = sg_env("SRCGAN_NO_ZIGZAG") == nullptr;
        static int flip = 0;
        if (zig && d.kh == 3 && d.kw == 3 && d.stride == 1) { d.rev_batch = flip; flip ^= 1; }
        return srcgan_conv_igemm(&d, st);
    }
};

// canonical weight layouts
struct WLayout { long sr, sk, sty, stx, off; };
// Conv2d weight [co][ci][kh][kw]: forward pack rows = co
static inline WLayout lay_fwd(int cin, int kh, int kw) { return {(long)cin * kh * kw, (long)kh * kw, kw, 1, 0}; }
// Conv2d dgrad for stride 1: rows = ci, k = co, taps flipped
static inline WLayout lay_dgrad_s1(int cin, int kh, int kw) { return {(long)kh * kw, (long)cin * kh * kw, -kw, -1, (long)kh * kw - 1}; }

struct Bump {   // workspace bump allocator (256-byte aligned)
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; }
};

static int wgrad_call(int dtype, TRef dy, int OH, int OW, int Cout, TRef x, int B, int H, int W, int Cin, int kh, int kw,
                      int stride, int pad_y, int pad_x, WLayout lay, float alpha, float* slab, float* grad, void* st,
                      float* bias_grad = nullptr, int accumulate = 0) {
    srcgan_wgrad_desc d;
    memset(&d, 0, sizeof(d));
    d.dy = dy.p; d.dy_cs = dy.cs; d.dy_coff = dy.coff; d.x = x.p; d.x_cs = x.cs; d.x_coff = x.coff;
    d.slab = slab; d.grad = grad; d.bias_grad = bias_grad; d.dtype = dtype; d.kh = kh; d.kw = kw; d.stride = stride;
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.OH = OH; d.OW = OW; d.Cout = Cout; d.pad_y = pad_y; d.pad_x = pad_x;
    d.nsplit = srcgan_conv_wgrad_nsplit(B, OH, OW, Cout, Cin, stride);
    d.sr = lay.sr; d.sk = lay.sk; d.sty = lay.sty; d.stx = lay.stx; d.off = lay.off;
    d.alpha = alpha; d.accumulate = accumulate;
    return srcgan_conv_wgrad(&d, st);
}
static size_t wgrad_slab(int B, int OH, int OW, int Cout, int Cin, int kh, int kw, int stride) {
    return srcgan_conv_wgrad_slab_bytes(Cout, Cin, kh, kw, srcgan_conv_wgrad_nsplit(B, OH, OW, Cout, Cin, stride));
}
static int bias_grad(int dtype, TRef dy, long npix, int C, float scale, float* out, float* scratch, void* st) {
    return srcgan_col_reduce(0, dy.p, dy.cs, dy.coff, nullptr, 0, 0, nullptr, nullptr, npix, C, scale, out, nullptr, scratch, dtype, st);
}

// ---- batched weight packing with a cached device-side job table
struct PackCache { std::vector<SgPackJob> host; SgPackJob* dev = nullptr; SgPackJob* pinned = nullptr; size_t cap = 0; hipEvent_t staged = nullptr; };
static std::map<std::string, PackCache> g_pack_cache;
static std::mutex g_pack_mutex;

struct PackList {
    std::vector<SgPackJob> jobs; long nblk = 0; int dtype; char* base;
    PackList(int dt, void* wp_base) : dtype(dt), base((char*)wp_base) {}
    void add(const float* w, void* wp, int rows, int kdim, int tys, int txs, long sr, long sk, long sty, long stx, long off,
             int k_off = 0, int k_total = -1, float scale = 1.f) {
        SgPackJob j;
        memset(&j, 0, sizeof(j));
        j.w = w; j.wp_off = (size_t)((char*)wp - base); j.rows = rows; j.kdim = kdim; j.tys = tys; j.txs = txs;
        j.sr = sr; j.sk = sk; j.sty = sty; j.stx = stx; j.off = off; j.k_off = k_off; j.k_total = k_total < 0 ? kdim : k_total; j.scale = scale;
        sg_pack_job_finish(j, dtype, nblk);
        jobs.push_back(j);
    }
    int run(const char* tag, const void* key_ptr, void* st, const unsigned long long* guard = nullptr) {
        if (jobs.empty()) return 0;
        int dev = 0;
        SG_HIP(hipGetDevice(&dev));
        char key[112];
        snprintf(key, sizeof(key), "%s:%d:%p:%d:%zu", tag, dev, key_ptr, dtype, jobs.size());
        std::lock_guard<std::mutex> lock(g_pack_mutex);
        PackCache& c = g_pack_cache[key];
        const size_t bytes = jobs.size() * sizeof(SgPackJob);
        if (c.host.size() != jobs.size() || memcmp(c.host.data(), jobs.data(), bytes) != 0) {
            // cold path (first call / parameters or buffers moved).  The table is staged in pinned memory and copied on the
            // call's stream: stream order puts the copy behind every earlier pack kernel that still reads the old table and in
            // front of this call's -- no host synchronisation.  Only a table that outgrows its buffers allocates (first call,
            // behind a stream drain: the old device table may still be in use), and a second re-staging waits for the first
            // one's copy to have left the pinned buffer.
            if (c.cap < bytes) {
                SG_HIP(hipStreamSynchronize((hipStream_t)st));
                if (c.dev) SG_HIP(hipFree(c.dev));
                if (c.pinned) SG_HIP(hipHostFree(c.pinned));
                SG_HIP(hipMalloc((void**)&c.dev, bytes));
                SG_HIP(hipHostMalloc((void**)&c.pinned, bytes, hipHostMallocDefault));
                c.cap = bytes;
            }
            if (!c.staged) SG_HIP(hipEventCreateWithFlags(&c.staged, hipEventDisableTiming));
            else SG_HIP(hipEventSynchronize(c.staged));
            memcpy(c.pinned, jobs.data(), bytes);
            SG_HIP(hipMemcpyAsync(c.dev, c.pinned, bytes, hipMemcpyHostToDevice, (hipStream_t)st));
            SG_HIP(hipEventRecord(c.staged, (hipStream_t)st));
            c.host = jobs;
        }
        return sg_pack_multi_launch(c.dev, (int)jobs.size(), nblk, base, dtype, (hipStream_t)st, guard);
    }
};

// opts.pack == 2: the persistent pack is re-made only if the device-side guard words differ (srcgan_net_opts)
static inline const unsigned long long* pack_guard(const srcgan_net_opts* o) {
    return (o && o->wpack && o->pack == 2) ? (const unsigned long long*)o->guard : nullptr;
}

// ======================================================================================== RDDBNet
struct RddbPlan {
    int dtype, esz, nf, gc, nb, C, nst, ndn, kce, nplane; long plane_bytes;
    int B, H, W;           // input
    size_t bm; long bm_bytes;   // sign masks of a dense buffer's four LeakyReLU slices: offset inside the buffer, bytes per slice (0 = not used)
    size_t um; long um_bytes;   // sign mask of the last up-sampler stage's output (8 bytes per HR pixel; 0 = not used): conv_last's input gradient
                                // reads it instead of the 64-channel activation (2.1 GB at the bench size)
    int Ht, Wt;            // trunk resolution
    int HO, WO;            // output resolution
    int in_cs, out_cs;
    int nparams;
    size_t xin, fea0, dn[5], A, szA, T, U[6], out, wpk, total;
    size_t w_rdb_d0, w_rdb_dsz;                    // contiguous region of the composite dense-block dgrad packs
    // packed weights (offsets relative to wpk): per conv
    size_t w_first_f, w_first_d, w_trunk_f, w_trunk_d, w_last_f, w_last_d;
    std::vector<size_t> w_rdb_f, w_rdb_d;          // [nb*15]
    size_t w_up_f[5][4], w_up_d[5];
    size_t w_dn_f[5], w_dn_d[5][4];
    // parameter indices
    int p_first_w, p_first_b, p_rdb0, p_trunk_w, p_trunk_b, p_up0, p_dn0, p_last_w;
    // legacy generators (model/model.py:347-440): tail = nearest x2 / shared 3x3 convs + LeakyReLU, conv_last with bias
    int legacy, ntail, nlw, p_lg[3], p_last_b;
    int nrr;                  // RRDBs in the trunk (legacy 3 = SRDN: encoder + decoder = 2 nb, srdn.py:56-74)
    int prdb(int r) const { return p_rdb0 + r * 10 + ((legacy == 3 && r >= 3 * nb) ? 2 : 0); }     // SRDN: trunk_conv's two parameters sit between the stacks
    struct TailOp { int conv, w, hin, win, hout, wout; size_t out; } tail[16];     // conv: 1 = 3x3 conv w + LReLU, 0 = nearest x2
    size_t lw_f[3], lw_d[3];
};

static int log2i(int v) { int n = 0; while ((1 << n) < v) ++n; return n; }

static int rddb_plan(const srcgan_rddbnet_cfg* c, RddbPlan& P) {
    SG_REQUIRE(c, "rddbnet: null cfg");
    SG_REQUIRE(sg_dtype_ok(c->dtype), "rddbnet: bad dtype %d", c->dtype);
    SG_REQUIRE(c->in_ch > 0 && c->in_ch <= 8 && c->out_ch > 0 && c->out_ch <= 8, "rddbnet: in/out channels must be in 1..8");
    SG_REQUIRE(c->nf > 0 && c->gc > 0 && c->nf % 8 == 0 && c->gc % 8 == 0, "rddbnet: nf and gc must be multiples of 8 (nf=%d gc=%d)", c->nf, c->gc);
    SG_REQUIRE(c->nb >= 1 && c->B > 0 && c->H > 0 && c->W > 0, "rddbnet: bad nb/B/H/W");
    SG_REQUIRE(c->up >= 1 && (c->up & (c->up - 1)) == 0 && c->up <= 16, "rddbnet: upscale_factor must be a power of two <= 16");
    SG_REQUIRE(c->down >= 0 && (c->down == 0 || ((c->down & (c->down - 1)) == 0 && c->down <= 16)), "rddbnet: bad down factor");
    SG_REQUIRE(!(c->down > 1 && c->up > 1), "rddbnet: up and down are exclusive");
    SG_REQUIRE(c->legacy >= 0 && c->legacy <= 3, "rddbnet: legacy must be 0..3");
    SG_REQUIRE(c->legacy == 0 || c->legacy == 3 || (c->down == 0 && (c->up == 2 || c->up == 4 || (c->legacy == 2 && c->up == 1))),
               "rddbnet: legacy generators take mode x2 / x4 (legacy RDDBNet also x1) and no down factor");
    SG_REQUIRE(c->legacy != 3 || (c->up == 1 && c->down == 0), "rddbnet: SRDN keeps the resolution (its upscale_factor is unused, srdn.py:67-74): pass up = 1");
    P.legacy = c->legacy == 3 ? 3 : c->legacy; P.ntail = 0; P.nlw = 0;
    P.nrr = c->legacy == 3 ? 2 * c->nb : (c->legacy == 2 ? 0 : c->nb);
    P.dtype = c->dtype; P.esz = c->dtype == SRCGAN_F32 ? 4 : 2;
    P.nf = c->nf; P.gc = c->gc; P.nb = c->nb; P.C = c->nf + 4 * c->gc;
    P.B = c->B; P.H = c->H; P.W = c->W;
    P.nst = c->down > 0 ? 0 : log2i(c->up);
    P.ndn = c->down > 1 ? log2i(c->down) : 0;
    if (P.ndn) SG_REQUIRE(c->H % c->down == 0 && c->W % c->down == 0, "rddbnet: H, W must be divisible by the down factor");
    P.Ht = c->H >> P.ndn; P.Wt = c->W >> P.ndn;
    P.HO = P.Ht << P.nst; P.WO = P.Wt << P.nst;
    P.in_cs = img_cs(c->in_ch); P.out_cs = img_cs(c->out_ch);
    const size_t e = P.esz, B = c->B;
    Bump b;
    P.xin = b.take(B * c->H * c->W * P.in_cs * e);
    P.fea0 = P.ndn ? b.take(B * c->H * c->W * c->nf * e) : 0;     // conv_first output at HR (HR->LR variant only)
    for (int s = 0; s < P.ndn; ++s) P.dn[s] = b.take(B * (c->H >> (s + 1)) * (c->W >> (s + 1)) * c->nf * e);
    // dense buffers use the blocked layout [plane = 64-byte channel chunk][pixel][64 B]: every operand fetch of the 3x3 kernel
    // and of the dense wgrad is then >= 1 KiB contiguous (64-byte pieces at a 384-byte pixel stride ran at half rate)
    P.kce = 64 / P.esz; P.nplane = (P.C + P.kce - 1) / P.kce; P.plane_bytes = (long)B * P.Ht * P.Wt * 64;
    // one bit per element for LeakyReLU' (instead of re-reading the activation in the backward pass): bf16, 32-channel slices
    static const bool no_sign = sg_env("SRCGAN_NO_SIGNMASK") != nullptr || sg_env("SRCGAN_DMA_CFG") != nullptr;
    P.bm_bytes = (!no_sign && sg_is16(c->dtype) && c->gc == 32 && c->nf % 32 == 0 && c->legacy != 2) ? (long)B * P.Ht * P.Wt * 4 : 0;
    P.bm = align_up((size_t)P.nplane * P.plane_bytes, 256);
    P.szA = align_up(P.bm + 4 * (size_t)P.bm_bytes, 256);
    P.A = b.take(P.szA * (c->legacy == 2 ? 1 : 3 * P.nrr));        // legacy RDDBNet discards its trunk: one buffer holds conv_first's output
    P.T = b.take(B * P.Ht * P.Wt * c->nf * e);
    for (int s = 0; s <= ((c->legacy == 1 || c->legacy == 2) ? 0 : P.nst); ++s) P.U[s] = b.take(B * (P.Ht << s) * (P.Wt << s) * c->nf * e);
    if (c->legacy == 3) P.U[0] = P.T;                  // SRDN: conv_last reads trunk output + skip, accumulated in place in T
    if (c->legacy == 1 || c->legacy == 2) {
        int h = P.Ht, w = P.Wt;
        auto op = [&](int conv, int wi) {
            RddbPlan::TailOp& o = P.tail[P.ntail++];
            o.conv = conv; o.w = wi; o.hin = h; o.win = w;
            if (!conv) { h *= 2; w *= 2; }
            o.hout = h; o.wout = w; o.out = b.take(B * h * w * c->nf * e);
        };
        if (c->legacy == 1) {           // RDDBNetB (model.py:427-439): weights 0 = upconv1, 1 = upconv2, 2 = HRconv
            if (c->up == 4) { op(0, 0); op(1, 0); op(0, 0); op(1, 1); }
            else { op(0, 0); op(1, 0); op(1, 0); }
            for (int k = 0; k < 8; ++k) op(1, 2);
            P.nlw = 3;
        } else {                        // legacy RDDBNet (model.py:381-391): weights 0 = upconv, 1 = HRconv
            for (int t = 1; t < c->up; t *= 2) { op(0, 0); op(1, 0); }
            if (c->up == 1) op(1, 0);
            op(1, 1); op(1, 1);
            P.nlw = 2;
        }
    }
    P.out = b.take(B * P.HO * P.WO * P.out_cs * e);
    P.um_bytes = (!no_sign && sg_is16(c->dtype) && c->nf == 64 && P.nst > 0 && c->legacy == 0) ? (long)B * P.HO * P.WO * 8 : 0;
    P.um = P.um_bytes ? b.take((size_t)P.um_bytes) : 0;
    P.wpk = b.off;
    Bump wb;
    auto pk = [&](int rows, int k, int taps) { return wb.take(srcgan_packed_weight_bytes(rows, k, taps, c->dtype)); };
    P.w_first_f = pk(c->nf, P.in_cs, 9); P.w_first_d = pk(c->in_ch, c->nf, 9);
    for (int s = 0; s < P.ndn; ++s) { P.w_dn_f[s] = pk(c->nf, c->nf, 9); for (int q = 0; q < 4; ++q) P.w_dn_d[s][q] = pk(c->nf, c->nf, 4); }
    const int nrdb = (P.nrr ? P.nrr : c->nb) * 3;
    P.w_rdb_f.resize(nrdb * 5); P.w_rdb_d.resize(nrdb * 5);
    for (int i = 0; i < nrdb; ++i)
        for (int k = 0; k < 5; ++k) {
            const int cin = c->nf + k * c->gc, cout = k < 4 ? c->gc : c->nf;
            P.w_rdb_f[i * 5 + k] = pk(cout, cin, 9);
        }
    P.w_trunk_f = pk(c->nf, c->nf, 9); P.w_trunk_d = pk(c->nf, c->nf, 9);
    for (int s = 0; s < P.nst; ++s) { for (int q = 0; q < 4; ++q) P.w_up_f[s][q] = pk(c->nf, c->nf, 1); P.w_up_d[s] = pk(c->nf, c->nf, 4); }
    P.w_last_f = pk(c->out_ch, c->nf, 9); P.w_last_d = pk(c->nf, P.out_cs, 9);
    for (int k = 0; k < P.nlw; ++k) { P.lw_f[k] = pk(c->nf, c->nf, 9); P.lw_d[k] = pk(c->nf, c->nf, 9); }
    // dense-block backward: slice j of the block input gets its gradient from ONE conv over the concatenated
    // output-gradients [dy5 | dy4 | ... | dy_{j+1}] (rows = slice channels, K = nf + (4-j)*gc)
    P.w_rdb_d0 = wb.off;
    for (int i = 0; i < nrdb; ++i)
        for (int j = 0; j < 5; ++j) P.w_rdb_d[i * 5 + j] = pk(j == 0 ? c->nf : c->gc, c->nf + (4 - j) * c->gc, 9);
    P.w_rdb_dsz = wb.off - P.w_rdb_d0;
    P.total = align_up(P.wpk + wb.off + 256, 256);
    // parameter indices (state_dict order)
    int n = 0;
    P.p_first_w = n++; P.p_first_b = n++;
    P.p_dn0 = n; n += 2 * P.ndn;
    P.p_rdb0 = n; n += c->nb * 30;
    P.p_trunk_w = n++; P.p_trunk_b = n++;
    if (c->legacy == 3) n += c->nb * 30;          // RRDB_decoder
    if (c->legacy == 1 || c->legacy == 2) { P.p_up0 = n; for (int k = 0; k < P.nlw; ++k) { P.p_lg[k] = n; n += 2; } }
    else { P.p_up0 = n; n += P.nst; }
    P.p_last_w = n++;
    P.p_last_b = (c->legacy == 1 || c->legacy == 2) ? n++ : -1;
    P.nparams = n;
    return 0;
}

}  // namespace

extern "C" int srcgan_rddbnet_num_params(const srcgan_rddbnet_cfg* c) { RddbPlan P; if (rddb_plan(c, P)) return -1; return P.nparams; }
extern "C" size_t srcgan_rddbnet_ws_bytes(const srcgan_rddbnet_cfg* c) { RddbPlan P; if (rddb_plan(c, P)) return 0; return P.total; }

namespace {
struct RddbBwdPlan {
    size_t dout, dU[6], dT, Pg[3], szP, dfea_dn[5], dxin, slab, colscr, biasred, total;
};
static void rddb_bwd_plan(const srcgan_rddbnet_cfg* c, const RddbPlan& P, RddbBwdPlan& Q) {
    const size_t e = P.esz, B = c->B;
    Bump b;
    Q.dout = b.take(B * P.HO * P.WO * P.out_cs * e);
    for (int s = 0; s <= ((P.legacy == 1 || P.legacy == 2) ? 0 : P.nst); ++s) Q.dU[s] = b.take(B * (P.Ht << s) * (P.Wt << s) * c->nf * e);
    if (P.legacy == 1 || P.legacy == 2) { Q.dU[1] = b.take(B * P.HO * P.WO * c->nf * e); Q.dU[2] = b.take(B * P.HO * P.WO * c->nf * e); }   // ping-pong
    Q.dT = b.take(B * P.Ht * P.Wt * c->nf * e);
    Q.szP = align_up((size_t)P.nplane * P.plane_bytes, 256);
    for (int i = 0; i < 3; ++i) Q.Pg[i] = b.take(Q.szP);
    for (int s = 0; s <= P.ndn; ++s) Q.dfea_dn[s] = b.take(B * (c->H >> s) * (c->W >> s) * c->nf * e);
    Q.dxin = b.take(B * c->H * c->W * P.in_cs * e);
    size_t slab = 0;
    auto mx = [&](size_t v) { if (v > slab) slab = v; };
    mx(wgrad_slab(c->B, c->H, c->W, c->nf, P.in_cs, 3, 3, 1));
    for (int k = 0; k < 5; ++k) mx(wgrad_slab(c->B, P.Ht, P.Wt, k < 4 ? c->gc : c->nf, c->nf + k * c->gc, 3, 3, 1));
    mx(wgrad_slab(c->B, P.Ht, P.Wt, c->nf, c->nf, 3, 3, 1));
    for (int s = 0; s < ((P.legacy == 1 || P.legacy == 2) ? 0 : P.nst); ++s) mx(wgrad_slab(c->B, P.Ht << s, P.Wt << s, c->nf, c->nf, 2, 2, 2));
    for (int k = 0; k < P.ntail; ++k) if (P.tail[k].conv) mx(wgrad_slab(c->B, P.tail[k].hout, P.tail[k].wout, c->nf, c->nf, 3, 3, 1));
    for (int s = 0; s < P.ndn; ++s) mx(wgrad_slab(c->B, c->H >> (s + 1), c->W >> (s + 1), c->nf, c->nf, 3, 3, 2));
    mx(wgrad_slab(c->B, P.HO, P.WO, c->out_ch, c->nf, 3, 3, 1));
    mx(srcgan_wgrad_dense_slab_bytes(P.C, P.C, c->dtype, c->B, P.Ht, P.Wt));
    Q.slab = b.take(slab);
    const long maxpix = (long)B * (P.HO > c->H ? P.HO : c->H) * (P.WO > c->W ? P.WO : c->W);
    Q.colscr = b.take((size_t)2 * srcgan_col_reduce_blocks(maxpix) * P.C * sizeof(float));
    Q.total = b.off + 256;
}
}  // namespace

extern "C" size_t srcgan_rddbnet_bwd_scratch_bytes(const srcgan_rddbnet_cfg* c) {
    RddbPlan P; if (rddb_plan(c, P)) return 0;
    RddbBwdPlan Q; rddb_bwd_plan(c, P, Q); return Q.total;
}

extern "C" size_t srcgan_rddbnet_wpack_bytes(const srcgan_rddbnet_cfg* c) { RddbPlan P; if (rddb_plan(c, P)) return 0; return P.total - P.wpk; }

// the four parity packs of up-sampler stage s are equally spaced: the stage runs as ONE launch (conv_igemm.hip, npar), which is also
// the form that writes the LeakyReLU sign mask of the last stage's output
static inline bool up_packs_spaced(const RddbPlan& P, int s) {
    const long wstep = (long)(P.w_up_f[s][1] - P.w_up_f[s][0]);
    return wstep > 0 && P.w_up_f[s][2] == P.w_up_f[s][0] + 2 * (size_t)wstep && P.w_up_f[s][3] == P.w_up_f[s][0] + 3 * (size_t)wstep;
}
static inline bool up_mask_written(const RddbPlan& P) { return P.um_bytes && P.nst > 0 && up_packs_spaced(P, P.nst - 1); }

extern "C" int srcgan_rddbnet_forward_ex(const srcgan_rddbnet_cfg* c, const float* x_nchw, const float* const* params,
                                         void* ws, float* y_nchw, const srcgan_net_opts* opt, void* st) {
    RddbPlan P;
    SG_TRY(rddb_plan(c, P));
    SG_REQUIRE(x_nchw && params && ws && y_nchw, "srcgan_rddbnet_forward: null pointer");
    SG_REQUIRE(((uintptr_t)ws % 256) == 0, "srcgan_rddbnet_forward: workspace must be 256-byte aligned");
    const int dt = c->dtype, nf = c->nf, gc = c->gc, B = c->B;
    char* w8 = (char*)ws;
    // packed weights: a persistent buffer of the caller's (packed once per optimiser step) or a region of this call's workspace
    char* wp = (opt && opt->wpack) ? (char*)opt->wpack : w8 + P.wpk;
    SG_REQUIRE(((uintptr_t)wp % 256) == 0, "srcgan_rddbnet_forward: wpack must be 256-byte aligned");
    const bool do_pack = !(opt && opt->wpack) || opt->pack;
    auto T_ = [&](size_t off, int cs) { return tref(w8 + off, cs); };
    auto Abuf = [&](int r) { return tref(w8 + P.A + (size_t)r * P.szA, P.kce, 0, P.plane_bytes); };

    // ---- pack weights for this call (f32 canonical -> dtype, MFMA-friendly): ONE batched launch
    PackList packs(dt, wp);
    packs.add(params[P.p_first_w], wp + P.w_first_f, nf, c->in_ch, 3, 3, (long)c->in_ch * 9, 9, 3, 1, 0);
    for (int s = 0; s < P.ndn; ++s)
        packs.add(params[P.p_dn0 + 2 * s], wp + P.w_dn_f[s], nf, nf, 3, 3, (long)nf * 9, 9, 3, 1, 0);
    for (int i = 0; i < P.nrr * 3; ++i)
        for (int k = 0; k < 5; ++k) {
            const int cin = nf + k * gc, cout = k < 4 ? gc : nf;
            packs.add(params[P.prdb(i) + k * 2], wp + P.w_rdb_f[i * 5 + k], cout, cin, 3, 3, (long)cin * 9, 9, 3, 1, 0);
        }
    if (P.legacy != 3) packs.add(params[P.p_trunk_w], wp + P.w_trunk_f, nf, nf, 3, 3, (long)nf * 9, 9, 3, 1, 0);
    for (int s = 0; s < ((P.legacy == 1 || P.legacy == 2) ? 0 : P.nst); ++s)
        for (int q = 0; q < 4; ++q)   // ConvTranspose2d weight [ci][co][2][2]; parity (a,b) = q: rows = co, k = ci
            packs.add(params[P.p_up0 + s], wp + P.w_up_f[s][q], nf, nf, 1, 1, 4, (long)nf * 4, 0, 0, q);
    for (int k = 0; k < P.nlw; ++k)
        packs.add(params[P.p_lg[k]], wp + P.lw_f[k], nf, nf, 3, 3, (long)nf * 9, 9, 3, 1, 0);
    packs.add(params[P.p_last_w], wp + P.w_last_f, c->out_ch, nf, 3, 3, (long)nf * 9, 9, 3, 1, 0);

    SG_REQUIRE(!(opt && opt->pack == 2) || (opt->wpack && opt->guard), "srcgan_rddbnet_forward: pack == 2 needs wpack and guard");
    if (do_pack) SG_TRY(packs.run(P.legacy == 1 ? "rddbB_fwd" : P.legacy == 2 ? "rddbL_fwd" : P.legacy == 3 ? "srdn_fwd" : "rddb_fwd", params[0], st, pack_guard(opt)));

    // ---- input: NCHW f32 -> NHWC (channels zero-padded to 8)
    SG_TRY(srcgan_nchw_f32_to_nhwc(x_nchw, w8 + P.xin, B, c->in_ch, c->H, c->W, P.in_cs, dt, st));
    // conv_first (rddb.py:89,108).  The trunk input lives in channels [0,nf) of the first dense buffer so the
    // first RDB reads it in place and the global skip (rddb.py:110) reads it back later.
    const int H = P.Ht, W = P.Wt;
    TRef trunk_in = Abuf(0);
    TRef fea = P.ndn ? T_(P.fea0, nf) : trunk_in;
    SG_TRY(Conv(dt, 3, 3, 1).in(T_(P.xin, P.in_cs), B, c->H, c->W, P.in_cs).w(wp + P.w_first_f, params[P.p_first_b])
               .out(fea, c->H, c->W, nf).pad(1, 1).run(st));
    // optional HR->LR stages (build-defined RDDBNetA): 3x3 s2 p1 + bias + LeakyReLU
    for (int s = 0; s < P.ndn; ++s) {
        TRef o = (s == P.ndn - 1) ? trunk_in : T_(P.dn[s], nf);
        SG_TRY(Conv(dt, 3, 3, 2).in(fea, B, c->H >> s, c->W >> s, nf).w(wp + P.w_dn_f[s], params[P.p_dn0 + 2 * s + 1])
                   .out(o, c->H >> (s + 1), c->W >> (s + 1), nf).pad(1, 1).lrelu().run(st));
        fea = o;
    }
    // RRDB trunk (rddb.py:62-68,78-82).  The legacy RDDBNet computes it and throws it away (model.py:382-383): skipped.
    for (int i = 0; i < P.nrr; ++i) {
        for (int j = 0; j < 3; ++j) {
            const int r = i * 3 + j;
            TRef A = Abuf(r);
            for (int k = 0; k < 4; ++k) {
                const int cin = nf + k * gc;
                Conv cv(dt, 3, 3, 1);
                cv.in(A, B, H, W, cin).w(wp + P.w_rdb_f[r * 5 + k], params[P.prdb(r) + k * 2 + 1]).out(sl(A, cin), H, W, gc).pad(1, 1).lrelu();
                if (P.bm_bytes) cv.sign_out((char*)A.p + P.bm + (size_t)k * P.bm_bytes);
                SG_TRY(cv.run(st));
            }
            // conv5 + residual(s) -> channels [0,nf) of the next dense buffer (or the trunk output)
            const bool last = (r == P.nrr * 3 - 1);
            TRef dst = last ? T_(P.T, nf) : Abuf(r + 1);
            Conv cv(dt, 3, 3, 1);
            cv.in(A, B, H, W, P.C).w(wp + P.w_rdb_f[r * 5 + 4], params[P.prdb(r) + 4 * 2 + 1]).out(dst, H, W, nf).pad(1, 1);
            if (j < 2) cv.alpha(0.2f).res1(A, nf, 1.f);
            else cv.alpha(0.04f).res1(A, nf, 0.2f).res2(Abuf(i * 3), nf, 1.f);   // RRDB: 0.2*(0.2*x5 + x_rdb3) + x_rrdb
            SG_TRY(cv.run(st));
        }
        if (P.legacy == 3 && i == c->nb - 1) {       // SRDN: fea = fea + RRDB_encoder(fea) (srdn.py:70-71), in the decoder's first buffer
            TRef d0 = Abuf((i + 1) * 3);
            SG_TRY(srcgan_add_inplace_planes(d0.p, d0.cs, 0, d0.plane, trunk_in.p, trunk_in.cs, 0, trunk_in.plane, nullptr, 0, 0, 0, 0.f,
                                             (long)B * H * W, nf, dt, st));
        }
    }
    if (P.legacy == 3) {       // fea = fea + RRDB_decoder(fea) (srdn.py:72-73): T += fea1, then conv_last reads T (= U[0])
        TRef f1 = Abuf(c->nb * 3), Tt = T_(P.T, nf);
        SG_TRY(srcgan_add_inplace_planes(Tt.p, Tt.cs, 0, 0, f1.p, f1.cs, 0, f1.plane, nullptr, 0, 0, 0, 0.f, (long)B * H * W, nf, dt, st));
    }
    // trunk_conv + global skip (rddb.py:109-110)
    if (P.legacy != 2 && P.legacy != 3)
        SG_TRY(Conv(dt, 3, 3, 1).in(T_(P.T, nf), B, H, W, nf).w(wp + P.w_trunk_f, params[P.p_trunk_b]).out(T_(P.U[0], nf), H, W, nf)
                   .pad(1, 1).res1(trunk_in, nf, 1.f).run(st));
    if (P.legacy == 1 || P.legacy == 2) {
        // legacy tail (model.py:384-390, 427-439): [nearest x2 -> 3x3 conv -> LeakyReLU] stages, HRconv applied repeatedly
        TRef cur = P.legacy == 2 ? trunk_in : T_(P.U[0], nf);
        for (int k = 0; k < P.ntail; ++k) {
            const RddbPlan::TailOp& o = P.tail[k];
            TRef dst = T_(o.out, nf);
            if (o.conv)
                SG_TRY(Conv(dt, 3, 3, 1).in(cur, B, o.hin, o.win, nf).w(wp + P.lw_f[o.w], params[P.p_lg[o.w] + 1]).out(dst, o.hout, o.wout, nf)
                           .pad(1, 1).lrelu().run(st));
            else
                SG_TRY(srcgan_upsample2_nhwc(cur.p, cur.cs, cur.coff, cur.plane, dst.p, dst.cs, B, o.hin, o.win, nf, dt, st));
            cur = dst;
        }
        SG_HIP(hipMemsetAsync(w8 + P.out, 0, (size_t)B * P.HO * P.WO * P.out_cs * P.esz, (hipStream_t)st));
        SG_TRY(Conv(dt, 3, 3, 1).in(cur, B, P.HO, P.WO, nf).w(wp + P.w_last_f, params[P.p_last_b]).out(T_(P.out, P.out_cs), P.HO, P.WO, c->out_ch)
                   .pad(1, 1).run(st));
        SG_TRY(srcgan_nhwc_to_nchw_f32(w8 + P.out, y_nchw, B, c->out_ch, P.HO, P.WO, P.out_cs, 0, dt, st));
        return 0;
    }
    // up-sampler: ConvTranspose2d(k2,s2) + LeakyReLU == 4 x (1x1 conv -> stride-2 scatter) (rddb.py:93-97,111-112)
    for (int s = 0; s < P.nst; ++s) {
        const int h = H << s, w = W << s;
        const long wstep = (long)(P.w_up_f[s][1] - P.w_up_f[s][0]);
        if (up_packs_spaced(P, s)) {
            // all four output parities in one launch (the input is read from HBM once: conv_igemm.hip, npar)
            Conv cv(dt, 1, 1, 1);
            cv.in(T_(P.U[s], nf), B, h, w, nf).w(wp + P.w_up_f[s][0]).out(T_(P.U[s + 1], nf), h, w, nf).scatter(2, 0, 0, 2 * h, 2 * w).lrelu();
            cv.d.npar = 4; cv.d.wpar_stride = wstep;
            if (P.um_bytes && s == P.nst - 1) cv.sign_out(w8 + P.um);       // LeakyReLU sign of the tensor conv_last reads
            SG_TRY(cv.run(st));
        } else
        for (int q = 0; q < 4; ++q)
            SG_TRY(Conv(dt, 1, 1, 1).in(T_(P.U[s], nf), B, h, w, nf).w(wp + P.w_up_f[s][q]).out(T_(P.U[s + 1], nf), h, w, nf)
                       .scatter(2, q >> 1, q & 1, 2 * h, 2 * w).lrelu().run(st));
    }
    // conv_last (no bias, rddb.py:98,113) as a convolution to all out_cs = 8 padded channels: the packed weight rows beyond out_ch
    // are zero, so channels out_ch.. come out as the zeros the padding wants -- and the output is 16 bytes per pixel through the
    // vectorised epilogue (with Cout = 3 it took the per-element form: three 2-byte stores per pixel of a 1024x1024 image, and a
    // 268 MB memset in front; 0.67 ms + 0.06 ms per step at the bench size)
    SG_TRY(Conv(dt, 3, 3, 1).in(T_(P.U[P.nst], nf), B, P.HO, P.WO, nf).w(wp + P.w_last_f).out(T_(P.out, P.out_cs), P.HO, P.WO, P.out_cs)
               .pad(1, 1).run(st));
    SG_TRY(srcgan_nhwc_to_nchw_f32(w8 + P.out, y_nchw, B, c->out_ch, P.HO, P.WO, P.out_cs, 0, dt, st));
    return 0;
}

extern "C" int srcgan_rddbnet_forward(const srcgan_rddbnet_cfg* c, const float* x_nchw, const float* const* params,
                                      void* ws, float* y_nchw, void* st) {
    return srcgan_rddbnet_forward_ex(c, x_nchw, params, ws, y_nchw, nullptr, st);
}

extern "C" int srcgan_rddbnet_backward_ex(const srcgan_rddbnet_cfg* c, const float* dy_nchw, const float* const* params,
                                          void* ws, void* scratch, float* const* grads, float* dx_nchw, const srcgan_net_opts* opt, void* st) {
    RddbPlan P;
    SG_TRY(rddb_plan(c, P));
    RddbBwdPlan Q;
    rddb_bwd_plan(c, P, Q);
    SG_REQUIRE(dy_nchw && params && ws && scratch && grads, "srcgan_rddbnet_backward: null pointer");
    SG_REQUIRE(((uintptr_t)ws % 256) == 0 && ((uintptr_t)scratch % 256) == 0, "srcgan_rddbnet_backward: buffers must be 256-byte aligned");
    const int dt = c->dtype, nf = c->nf, gc = c->gc, B = c->B, H = P.Ht, W = P.Wt;
    char* w8 = (char*)ws; char* s8 = (char*)scratch;
    char* wp = (opt && opt->wpack) ? (char*)opt->wpack : w8 + P.wpk;
    const bool do_pack = !(opt && opt->wpack) || opt->pack;
    // Phased backward (data parallel: the gradients of the RRDBs a phase covers are final when it returns, so their all-reduce
    // starts while earlier blocks still compute).  A call handles the RRDBs [lo, hi), last to first; the call with hi == nrr also
    // runs everything behind the trunk (conv_last, up-sampler, trunk_conv), the call with lo == 0 everything in front of it
    // (conv_first / down-sampling stages, dx).  Calls must come in descending, gap-free order on one stream with the same
    // scratch: the running gradient sits in the scratch between them.
    int r_lo = 0, r_hi = P.nrr;
    if (opt && opt->rrdb_hi > 0) { r_lo = opt->rrdb_lo; r_hi = opt->rrdb_hi; }
    SG_REQUIRE(r_lo >= 0 && r_lo <= r_hi && r_hi <= P.nrr, "srcgan_rddbnet_backward: RRDB range [%d,%d) outside [0,%d)", r_lo, r_hi, P.nrr);
    SG_REQUIRE(P.nrr > 0 || (r_lo == 0), "srcgan_rddbnet_backward: a network without a trunk has one phase");
    const bool first_phase = r_hi == P.nrr, last_phase = r_lo == 0;
    float* slab = (float*)(s8 + Q.slab); float* colscr = (float*)(s8 + Q.colscr);
    auto T_ = [&](size_t off, int cs) { return tref(w8 + off, cs); };
    auto S_ = [&](size_t off, int cs) { return tref(s8 + off, cs); };
    auto Abuf = [&](int r) { return tref(w8 + P.A + (size_t)r * P.szA, P.kce, 0, P.plane_bytes); };
    auto G = [&](int idx) { return grads[idx]; };
    const long npix_t = (long)B * H * W;

    // ---- packed dgrad weights (flipped / transposed views of the canonical tensors): ONE batched launch
    const bool pack_dx = dx_nchw || (opt && opt->wpack);       // a persistent pack serves later calls that may want dx
    if (first_phase && do_pack) {
        PackList packs(dt, wp);
        const WLayout L = lay_dgrad_s1(nf, 3, 3);
        packs.add(params[P.p_last_w], wp + P.w_last_d, nf, c->out_ch, 3, 3, L.sr, L.sk, L.sty, L.stx, L.off);
        for (int s = 0; s < ((P.legacy == 1 || P.legacy == 2) ? 0 : P.nst); ++s)   // deconv dgrad = 2x2 s2 conv over dy: rows = ci, k = co, tap = (a,b)
            packs.add(params[P.p_up0 + s], wp + P.w_up_d[s], nf, nf, 2, 2, (long)nf * 4, 4, 2, 1, 0);
        for (int k = 0; k < P.nlw; ++k)
            packs.add(params[P.p_lg[k]], wp + P.lw_d[k], nf, nf, 3, 3, L.sr, L.sk, L.sty, L.stx, L.off);
        if (P.legacy != 3) packs.add(params[P.p_trunk_w], wp + P.w_trunk_d, nf, nf, 3, 3, L.sr, L.sk, L.sty, L.stx, L.off);
        SG_TRY(sg_fill_zero_guarded(wp + P.w_rdb_d0, P.w_rdb_dsz, pack_guard(opt), (hipStream_t)st));
        for (int r = 0; r < P.nrr * 3; ++r) {
            const float a5 = (r % 3 == 2) ? 0.04f : 0.2f;       // d(x5)/d(block out), RDB3 carries the RRDB 0.2 too
            for (int j = 0; j < 5; ++j) {
                const int rows = j == 0 ? nf : gc, ss = j == 0 ? 0 : nf + (j - 1) * gc, ktot = nf + (4 - j) * gc;
                for (int m = 5; m > j; --m) {                   // block of K coming from forward conv m
                    const int cin_m = nf + (m - 1) * gc, cout_m = m == 5 ? nf : gc;
                    const int k_off = m == 5 ? 0 : nf + (4 - m) * gc;
                    packs.add(params[P.prdb(r) + (m - 1) * 2], wp + P.w_rdb_d[r * 5 + j], rows, cout_m, 3, 3,
                                                   9, (long)cin_m * 9, -3, -1, (long)ss * 9 + 8, k_off, ktot, m == 5 ? a5 : 1.f);
                }
            }
        }
        for (int s = 0; s < P.ndn; ++s)
            for (int q = 0; q < 4; ++q) {
                // 3x3 s2 p1 dgrad, output parity (a,b): rows with ky = a+1 (mod 2).  a=0: ky=1 (1 tap); a=1: ky=2,0 (2 taps)
                const int a = q >> 1, bb = q & 1;
                const int ty = a ? 2 : 1, tx = bb ? 2 : 1;
                const long off = (a ? 2 : 1) * 3 + (bb ? 2 : 1);
                packs.add(params[P.p_dn0 + 2 * s], wp + P.w_dn_d[s][q], nf, nf, ty, tx, 9, (long)nf * 9, -6, -2, off);
            }
        if (pack_dx) {
            const WLayout L0 = lay_dgrad_s1(c->in_ch, 3, 3);
            packs.add(params[P.p_first_w], wp + P.w_first_d, c->in_ch, nf, 3, 3, L0.sr, L0.sk, L0.sty, L0.stx, L0.off);
        }
        SG_TRY(packs.run(P.legacy == 1 ? (pack_dx ? "rddbB_bwd_dx" : "rddbB_bwd") : P.legacy == 2 ? (pack_dx ? "rddbL_bwd_dx" : "rddbL_bwd")
                                       : P.legacy == 3 ? (pack_dx ? "srdn_bwd_dx" : "srdn_bwd")
                                       : (pack_dx ? "rddb_bwd_dx" : "rddb_bwd"), params[0], st, pack_guard(opt)));
    }

    // ---- dy: NCHW f32 -> NHWC
    TRef dout = S_(Q.dout, P.out_cs);
    TRef dU0 = S_(Q.dU[0], nf);
    if (first_phase) {
    SG_TRY(srcgan_nchw_f32_to_nhwc(dy_nchw, dout.p, B, c->out_ch, P.HO, P.WO, P.out_cs, dt, st));
    if (P.legacy == 1 || P.legacy == 2) {
        // ---- legacy tail backward.  dcur = gradient w.r.t. an op's output, already times LeakyReLU' of that output.
        TRef tin = P.legacy == 2 ? Abuf(0) : T_(P.U[0], nf);          // tail input (not an activation output)
        auto obuf = [&](int k) { return k < 0 ? tin : T_(P.tail[k].out, nf); };
        TRef Fl = obuf(P.ntail - 1);
        if (G(P.p_last_w))
            SG_TRY(wgrad_call(dt, dout, P.HO, P.WO, c->out_ch, Fl, B, P.HO, P.WO, nf, 3, 3, 1, 1, 1, lay_fwd(nf, 3, 3), 1.f, slab, G(P.p_last_w), st, G(P.p_last_b)));
        else if (G(P.p_last_b)) SG_TRY(bias_grad(dt, dout, (long)B * P.HO * P.WO, c->out_ch, 1.f, G(P.p_last_b), colscr, st));
        int pp = 0;
        auto nextbuf = [&](int k_in) { if (k_in < 0) return dU0; pp ^= 1; return S_(Q.dU[1 + pp], nf); };   // k_in: index of the op whose output gets this gradient
        TRef dcur = nextbuf(P.ntail - 1);
        {
            Conv cv(dt, 3, 3, 1);
            cv.in(dout, B, P.HO, P.WO, P.out_cs).w(wp + P.w_last_d).out(dcur, P.HO, P.WO, nf).pad(1, 1);
            if (P.ntail > 0 && P.tail[P.ntail - 1].conv) cv.mask(Fl, 0);
            SG_TRY(cv.run(st));
        }
        bool seen[3] = {false, false, false};
        for (int k = P.ntail - 1; k >= 0; --k) {
            const RddbPlan::TailOp& o = P.tail[k];
            TRef xin_k = obuf(k - 1);
            const bool in_act = k > 0 && P.tail[k - 1].conv;          // the op's input is a LeakyReLU output
            TRef dst = nextbuf(k - 1);
            if (o.conv) {
                const int pw = P.p_lg[o.w];
                if (G(pw)) SG_TRY(wgrad_call(dt, dcur, o.hout, o.wout, nf, xin_k, B, o.hin, o.win, nf, 3, 3, 1, 1, 1, lay_fwd(nf, 3, 3), 1.f, slab, G(pw), st, G(pw + 1), seen[o.w] ? 1 : 0));
                else if (G(pw + 1)) SG_REQUIRE(false, "rddbnet (legacy): a shared convolution's bias gradient needs its weight gradient too");
                seen[o.w] = true;
                Conv cv(dt, 3, 3, 1);
                cv.in(dcur, B, o.hout, o.wout, nf).w(wp + P.lw_d[o.w]).out(dst, o.hin, o.win, nf).pad(1, 1);
                if (in_act) cv.mask(xin_k, 0);
                SG_TRY(cv.run(st));
            } else {
                // nearest x2 backward = 2x2 block sums; the up-sampled tensor's own LeakyReLU' (if any) is applied at its resolution
                SG_REQUIRE(xin_k.plane == 0 || !in_act, "rddbnet (legacy): unexpected blocked activation");
                SG_TRY(srcgan_sum2x2_nhwc(dcur.p, dcur.cs, dst.p, dst.cs, in_act ? xin_k.p : nullptr, xin_k.cs, 0.2f, B, o.hin, o.win, nf, dt, st));
            }
            dcur = dst;
        }
        // unused parameters of the forward (upconv2 in mode x2): their gradient is zero here, None in the reference
        for (int k = 0; k < P.nlw; ++k)
            if (!seen[k]) {
                if (G(P.p_lg[k])) SG_HIP(hipMemsetAsync(G(P.p_lg[k]), 0, (size_t)nf * nf * 9 * sizeof(float), (hipStream_t)st));
                if (G(P.p_lg[k] + 1)) SG_HIP(hipMemsetAsync(G(P.p_lg[k] + 1), 0, (size_t)nf * sizeof(float), (hipStream_t)st));
            }
    } else {
    // conv_last
    TRef Ul = T_(P.U[P.nst], nf);
    if (G(P.p_last_w))
        SG_TRY(wgrad_call(dt, dout, P.HO, P.WO, c->out_ch, Ul, B, P.HO, P.WO, nf, 3, 3, 1, 1, 1, lay_fwd(nf, 3, 3), 1.f, slab, G(P.p_last_w), st));
    {
        Conv cv(dt, 3, 3, 1);
        cv.in(dout, B, P.HO, P.WO, P.out_cs).w(wp + P.w_last_d).out(S_(Q.dU[P.nst], nf), P.HO, P.WO, nf).pad(1, 1);
        if (P.nst > 0) {                      // LeakyReLU after the last deconv
            if (P.um_bytes && up_mask_written(P)) { cv.sign_in(w8 + P.um); cv.d.mslope = 0.2f; }
            else cv.mask(Ul, 0);
        }
        SG_TRY(cv.run(st));
    }
    // up-sampler stages, last to first
    for (int s = P.nst - 1; s >= 0; --s) {
        const int h = H << s, w = W << s;
        TRef dHR = S_(Q.dU[s + 1], nf), Us = T_(P.U[s], nf);
        if (G(P.p_up0 + s))   // dW[ci][co][a][b] = sum x[y,x,ci] * dy[2y+a,2x+b,co]: wgrad with roles (dy := x, x := dy), k2 s2
            SG_TRY(wgrad_call(dt, Us, h, w, nf, dHR, B, 2 * h, 2 * w, nf, 2, 2, 2, 0, 0, WLayout{(long)nf * 4, 4, 2, 1, 0}, 1.f, slab, G(P.p_up0 + s), st));
        Conv cv(dt, 2, 2, 2);
        cv.in(dHR, B, 2 * h, 2 * w, nf).w(wp + P.w_up_d[s]).out(S_(Q.dU[s], nf), h, w, nf).pad(0, 0);
        if (s > 0) cv.mask(Us, 0);
        SG_TRY(cv.run(st));
    }
    }
    }                            // first_phase: everything behind the trunk
    TRef dfea = dU0;
    if (P.legacy != 2) {
    // U0 = fea + trunk_conv(T): d(trunk_conv out) = dU0, d(fea) += dU0 (joined at the end)
    TRef Tt = T_(P.T, nf), dT = S_(Q.dT, nf);
    if (first_phase && P.legacy != 3) {
    if (G(P.p_trunk_w))
        SG_TRY(wgrad_call(dt, dU0, H, W, nf, Tt, B, H, W, nf, 3, 3, 1, 1, 1, lay_fwd(nf, 3, 3), 1.f, slab, G(P.p_trunk_w), st, G(P.p_trunk_b)));
    else if (G(P.p_trunk_b)) SG_TRY(bias_grad(dt, dU0, npix_t, nf, 1.f, G(P.p_trunk_b), colscr, st));
    }
    // Dense gradient buffers (3, rotating): Gd = [dy5 (nf) | dy4 | dy3 | dy2 | dy1] -- the mirror image of the forward
    // dense buffer.  Slice j of the block input gets its gradient from ONE conv over the channel prefix holding
    // dy5..dy_{j+1} (composite transposed weights), so every gradient element is written exactly once: no
    // read-modify-write accumulation, and the same prefix-read / slice-write pattern as forward.
    auto Pg = [&](int g) { return tref(s8 + Q.Pg[g], P.kce, 0, P.plane_bytes); };
    if (!first_phase) {
    } else if (P.legacy == 3) {
        // SRDN: d(decoder output) = d(fea2) = dU0 itself (no trunk_conv): into the first gradient buffer's channels [0,nf)
        TRef g0 = Pg(0);
        SG_HIP(hipMemsetAsync(g0.p, 0, (size_t)cdiv(nf, P.kce) * P.plane_bytes, (hipStream_t)st));
        SG_TRY(srcgan_add_inplace_planes(g0.p, g0.cs, 0, g0.plane, dU0.p, dU0.cs, 0, 0, nullptr, 0, 0, 0, 0.f, npix_t, nf, dt, st));
    } else
    SG_TRY(Conv(dt, 3, 3, 1).in(dU0, B, H, W, nf).w(wp + P.w_trunk_d).out(Pg(0), H, W, nf).pad(1, 1).run(st));
    (void)dT;
    for (int i = r_hi - 1; i >= r_lo; --i) {
        if (P.legacy == 3 && i == c->nb - 1) {
            // between the stacks: d(fea1) = d(fea2) + d(decoder input); it feeds the encoder's output AND the skip around it
            TRef g0 = Pg(0);
            SG_TRY(srcgan_add_inplace_planes(g0.p, g0.cs, 0, g0.plane, dU0.p, dU0.cs, 0, 0, nullptr, 0, 0, 0, 0.f, npix_t, nf, dt, st));
            SG_HIP(hipMemsetAsync(dU0.p, 0, (size_t)npix_t * nf * P.esz, (hipStream_t)st));
            SG_TRY(srcgan_add_inplace_planes(dU0.p, dU0.cs, 0, 0, g0.p, g0.cs, 0, g0.plane, nullptr, 0, 0, 0, 0.f, npix_t, nf, dt, st));
        }
        for (int j3 = 2; j3 >= 0; --j3) {                  // RDB3, RDB2, RDB1 use Pg(0), Pg(1), Pg(2)
            const int r = i * 3 + j3, g = 2 - j3;
            TRef A = Abuf(r), Gd = Pg(g), nxt = Pg((g + 1) % 3);
            const float a5 = (j3 == 2) ? 0.04f : 0.2f;     // folded into the packed conv5 block; wgrad/bias use it as alpha
            const float bres = (j3 == 2) ? 0.2f : 1.f;     // block-input residual: d(in) += bres * d(out)
            const int pbase = P.prdb(r);
            auto slice_grad = [&](int m) -> int {
                // gradient of input slice j = m-1 from [dy5 .. dy_m]
                const int j = m - 1, ktot = nf + (4 - j) * gc;
                Conv cv(dt, 3, 3, 1);
                cv.in(Gd, B, H, W, ktot).w(wp + P.w_rdb_d[r * 5 + j]).pad(1, 1);
                if (j > 0) {
                    // slice x_j is a LeakyReLU output: multiply by its derivative -> this IS dy_j
                    cv.out(sl(Gd, nf + (4 - j) * gc), H, W, gc);
                    if (P.bm_bytes) { cv.sign_in((const char*)A.p + P.bm + (size_t)(j - 1) * P.bm_bytes); cv.d.mslope = 0.2f; }
                    else cv.mask(sl(A, nf + (j - 1) * gc), 0);
                } else {
                    cv.out(nxt, H, W, nf).res1(Gd, nf, bres);
                    if (j3 == 0) cv.res2(nxt, nf, 1.f);     // RRDB skip; nxt == Pg(0) still holds d(out_rrdb): in-place, same element
                }
                return cv.run(st);
            };
            for (int m = 5; m >= 2; --m) SG_TRY(slice_grad(m));
            // weight + bias gradients of the block's five convs in ONE pass over (Gd, A) -- wgrad_dense.hip.  They need dy5..dy1, not the
            // block-input gradient: that convolution (m = 1) runs AFTER them, so that the next block's first gradient-slice convolution
            // (64 -> 32, memory-bound) follows a convolution instead of the weight-gradient kernels: 65 -> 57 us per launch, -0.4 ms per
            // step (same-box A/B of the two orders)
            {
                srcgan_wgrad_dense_desc wd;
                memset(&wd, 0, sizeof(wd));
                wd.dy = Gd.p; wd.dy_cs = Gd.cs; wd.dy_coff = Gd.coff; wd.dy_plane = Gd.plane; wd.G = P.C;
                wd.x = A.p; wd.x_cs = A.cs; wd.x_coff = A.coff; wd.x_plane = A.plane; wd.C = P.C;
                wd.slab = slab; wd.dtype = dt; wd.B = B; wd.H = H; wd.W = W;
                bool any = false;
                for (int m = 5; m >= 1; --m) {
                    srcgan_wgrad_seg& sg = wd.seg[wd.nseg++];
                    sg.g0 = m == 5 ? 0 : nf + (4 - m) * gc; sg.g1 = sg.g0 + (m == 5 ? nf : gc);
                    sg.grad = G(pbase + 2 * (m - 1)); sg.bias = G(pbase + 2 * (m - 1) + 1);
                    sg.Cin = nf + (m - 1) * gc; sg.alpha = m == 5 ? a5 : 1.f;
                    any = any || sg.grad || sg.bias;
                }
                if (any) SG_TRY(srcgan_wgrad_dense(&wd, st));
            }
            SG_TRY(slice_grad(1));
        }
    }
    if (!last_phase) return 0;
    TRef dcur = Pg(0);
    // gradient w.r.t. the trunk input feature = dcur + dU0 (global skip); for the HR->LR variant the trunk input
    // is a LeakyReLU output, so its derivative is applied in the same pass.
    TRef trunk_in = Abuf(0);
    SG_TRY(srcgan_add_inplace_planes(dU0.p, dU0.cs, dU0.coff, dU0.plane, dcur.p, dcur.cs, dcur.coff, dcur.plane,
                                     P.ndn ? trunk_in.p : nullptr, trunk_in.cs, 0, trunk_in.plane, 0.2f, npix_t, nf, dt, st));
    }                            // (dU0 is not needed any more: the join is accumulated into it, NHWC)
    for (int s = P.ndn - 1; s >= 0; --s) {     // 3x3 s2 p1 stages of RDDBNetA, last to first
        const int hi = c->H >> s, wi = c->W >> s, ho = hi / 2, wo = wi / 2;
        TRef xin_s = s == 0 ? T_(P.fea0, nf) : T_(P.dn[s - 1], nf);
        const int pw = P.p_dn0 + 2 * s;
        if (G(pw)) SG_TRY(wgrad_call(dt, dfea, ho, wo, nf, xin_s, B, hi, wi, nf, 3, 3, 2, 1, 1, lay_fwd(nf, 3, 3), 1.f, slab, G(pw), st, G(pw + 1)));
        else if (G(pw + 1)) SG_TRY(bias_grad(dt, dfea, (long)B * ho * wo, nf, 1.f, G(pw + 1), colscr, st));
        TRef dst = S_(Q.dfea_dn[s], nf);
        for (int q = 0; q < 4; ++q) {           // dgrad by output parity (a,b): sub-kernel of 1 or 2 taps per axis
            const int a = q >> 1, bb = q & 1;
            const int mh = (hi - a + 1) / 2, mw = (wi - bb + 1) / 2;
            Conv cv(dt, a ? 2 : 1, bb ? 2 : 1, 1);
            cv.in(dfea, B, ho, wo, nf).w(wp + P.w_dn_d[s][q]).out(dst, mh, mw, nf).pad(0, 0).scatter(2, a, bb, hi, wi);
            if (s > 0) cv.mask(xin_s, 0);
            SG_TRY(cv.run(st));
        }
        dfea = dst;
    }
    // conv_first
    TRef xin = T_(P.xin, P.in_cs);
    if (G(P.p_first_w))
        SG_TRY(wgrad_call(dt, dfea, c->H, c->W, nf, xin, B, c->H, c->W, c->in_ch, 3, 3, 1, 1, 1, lay_fwd(c->in_ch, 3, 3), 1.f, slab, G(P.p_first_w), st, G(P.p_first_b)));
    else if (G(P.p_first_b)) SG_TRY(bias_grad(dt, dfea, (long)B * c->H * c->W, nf, 1.f, G(P.p_first_b), colscr, st));
    if (dx_nchw) {
        TRef dxin = S_(Q.dxin, P.in_cs);
        SG_HIP(hipMemsetAsync(dxin.p, 0, (size_t)B * c->H * c->W * P.in_cs * P.esz, (hipStream_t)st));
        SG_TRY(Conv(dt, 3, 3, 1).in(dfea, B, c->H, c->W, nf).w(wp + P.w_first_d).out(dxin, c->H, c->W, c->in_ch).pad(1, 1).run(st));
        SG_TRY(srcgan_nhwc_to_nchw_f32(dxin.p, dx_nchw, B, c->in_ch, c->H, c->W, P.in_cs, 0, dt, st));
    }
    return 0;
}

extern "C" int srcgan_rddbnet_backward(const srcgan_rddbnet_cfg* c, const float* dy_nchw, const float* const* params,
                                       void* ws, void* scratch, float* const* grads, float* dx_nchw, void* st) {
    return srcgan_rddbnet_backward_ex(c, dy_nchw, params, ws, scratch, grads, dx_nchw, nullptr, st);
}

// ======================================================================================== NLayerDiscriminator
namespace {
struct DPlan {
    int dtype, esz, L;                 // L convs
    int ch[8], hh[8], ww[8], st[8];    // ch[l] -> ch[l+1]; spatial dims of activation l (0 = input)
    int in_cs, out_cs;
    int nparams;
    size_t xin, Y[8], Z[8], stat[8], out, colscr, wpk, total;      // stat: mean[C], var[C], rstd[C]
    size_t wf[8], wd[8][4];
    int pw[8], pb[8], pg[8], pbeta[8];   // parameter indices (-1 if absent)
    int bn_idx[8];                       // index among BN layers (-1 if none)
    int inorm;                           // norm_layer = InstanceNorm2d: the normalised layers are GroupNorm(G = C) without affine part
    int nrm[8];                          // layer l is followed by a normalisation layer
    size_t gnscr;                        // GroupNorm scratch (instance norm)
    int s2d;                             // first layer in space-to-depth form (even H, W): 2x2 s1 over 32-channel blocks, K not padded
};

static int d_plan(const srcgan_nlayerd_cfg* c, DPlan& P) {
    SG_REQUIRE(c, "nlayerd: null cfg");
    SG_REQUIRE(sg_dtype_ok(c->dtype), "nlayerd: bad dtype");
    SG_REQUIRE(c->in_ch > 0 && c->in_ch <= 8, "nlayerd: input_nc must be in 1..8");
    SG_REQUIRE(c->ndf > 0 && c->ndf % 8 == 0, "nlayerd: ndf must be a multiple of 8");
    SG_REQUIRE(c->n_layers >= 1 && c->n_layers <= 5, "nlayerd: n_layers must be in 1..5");
    SG_REQUIRE(c->B > 0 && c->H > 0 && c->W > 0, "nlayerd: bad B/H/W");
    SG_REQUIRE(c->norm == 0 || c->norm == 1, "nlayerd: norm must be 0 (BatchNorm2d) or 1 (InstanceNorm2d)");
    P.inorm = c->norm == 1;
    P.dtype = c->dtype; P.esz = c->dtype == SRCGAN_F32 ? 4 : 2;
    P.L = c->n_layers + 2;
    P.ch[0] = c->in_ch; P.ch[1] = c->ndf;
    for (int n = 1; n < c->n_layers; ++n) { int m = 1 << n; if (m > 8) m = 8; P.ch[n + 1] = c->ndf * m; }
    { int m = 1 << c->n_layers; if (m > 8) m = 8; P.ch[c->n_layers + 1] = c->ndf * m; }
    P.ch[P.L] = 1;
    P.hh[0] = c->H; P.ww[0] = c->W;
    for (int l = 0; l < P.L; ++l) {
        P.st[l] = l < c->n_layers ? 2 : 1;
        P.hh[l + 1] = (P.hh[l] + 2 - 4) / P.st[l] + 1;
        P.ww[l + 1] = (P.ww[l] + 2 - 4) / P.st[l] + 1;
        SG_REQUIRE(P.hh[l + 1] > 0 && P.ww[l + 1] > 0, "nlayerd: input %dx%d too small for %d layers", c->H, c->W, c->n_layers);
    }
    P.in_cs = img_cs(c->in_ch); P.out_cs = 8;
    const size_t e = P.esz, B = c->B;
    static const bool no_s2d = sg_env("SRCGAN_NO_S2D") != nullptr;
    P.s2d = !no_s2d && c->H % 2 == 0 && c->W % 2 == 0;
    Bump b;
    P.xin = b.take(P.s2d ? B * (c->H / 2 + 1) * (c->W / 2 + 1) * 32 * e : B * c->H * c->W * P.in_cs * e);
    int n = 0, nbn = 0;
    for (int l = 0; l < P.L; ++l) {
        const bool nrm = (l >= 1 && l <= P.L - 2), bn = nrm && !P.inorm, bias = (l == 0 || l == P.L - 1 || (nrm && P.inorm));
        P.nrm[l] = nrm;
        P.pw[l] = n++;
        P.pb[l] = bias ? n++ : -1;
        P.pg[l] = bn ? n++ : -1;
        P.pbeta[l] = bn ? n++ : -1;
        P.bn_idx[l] = bn ? nbn++ : -1;
        const size_t sz = B * P.hh[l + 1] * P.ww[l + 1] * (l == P.L - 1 ? P.out_cs : P.ch[l + 1]) * e;
        if (l == P.L - 1) P.out = b.take(sz);
        else {
            P.Y[l] = b.take(sz);
            if (nrm) {      // BatchNorm: mean[C], var[C], rstd[C]; instance norm: {mean, rstd}[B][C]
                P.Z[l] = b.take(sz);
                P.stat[l] = b.take((P.inorm ? (size_t)2 * B * P.ch[l + 1] : (size_t)3 * P.ch[l + 1]) * sizeof(float));
            }
        }
    }
    P.nparams = n;
    P.gnscr = 0;
    if (P.inorm) {
        int cmax = 8; for (int l = 1; l < P.L; ++l) if (P.ch[l] > cmax) cmax = P.ch[l];
        SG_REQUIRE(cmax <= 1024 && cmax % (16 / (int)e) == 0 && 256 % (cmax / (16 / (int)e)) == 0, "nlayerd: InstanceNorm2d needs channel counts that are powers of two up to 1024");
        P.gnscr = b.take(srcgan_gn_scratch_floats(c->B, cmax) * sizeof(float));
    }
    {
        int cmax = 8; for (int l = 1; l < P.L; ++l) if (P.ch[l] > cmax) cmax = P.ch[l];
        P.colscr = b.take((size_t)2 * srcgan_col_reduce_blocks((long)B * P.hh[1] * P.ww[1]) * cmax * sizeof(float));
    }
    P.wpk = b.off;
    Bump wb;
    for (int l = 0; l < P.L; ++l) {
        const int cin_r = l == 0 ? P.in_cs : P.ch[l];
        const int k_r = l == P.L - 1 ? P.out_cs : P.ch[l + 1];
        if (l == 0 && P.s2d) {             // folded first layer: forward rows = Cout, k = 32, 4 taps; dgrad rows = 32, k = Cout
            P.wf[l] = wb.take(srcgan_packed_weight_bytes(P.ch[1], 32, 4, c->dtype));
            P.wd[l][0] = wb.take(srcgan_packed_weight_bytes(32, k_r, 4, c->dtype));
            continue;
        }
        P.wf[l] = wb.take(srcgan_packed_weight_bytes(P.ch[l + 1], cin_r, 16, c->dtype));
        if (P.st[l] == 1) P.wd[l][0] = wb.take(srcgan_packed_weight_bytes(P.ch[l], k_r, 16, c->dtype));
        else for (int q = 0; q < 4; ++q) P.wd[l][q] = wb.take(srcgan_packed_weight_bytes(P.ch[l], k_r, 4, c->dtype));
    }
    P.total = align_up(P.wpk + wb.off + 256, 256);
    return 0;
}

struct DBwdPlan { size_t dO, g[2], dxin, slab, colscr, sums, gfold, gnscr, total; };
static void d_bwd_plan(const srcgan_nlayerd_cfg* c, const DPlan& P, DBwdPlan& Q) {
    const size_t e = P.esz, B = c->B;
    Bump b;
    Q.dO = b.take(B * P.hh[P.L] * P.ww[P.L] * P.out_cs * e);
    size_t mx = 0;
    for (int l = 0; l < P.L - 1; ++l) { size_t s = B * P.hh[l + 1] * P.ww[l + 1] * P.ch[l + 1] * e; if (s > mx) mx = s; }
    Q.g[0] = b.take(mx); Q.g[1] = b.take(mx);
    Q.dxin = b.take(P.s2d ? B * (c->H / 2 + 1) * (c->W / 2 + 1) * 32 * e : B * c->H * c->W * P.in_cs * e);
    Q.gfold = b.take((size_t)P.ch[1] * 32 * 4 * sizeof(float));
    size_t slab = 0;
    for (int l = 0; l < P.L; ++l) {
        size_t s = (l == 0 && P.s2d) ? wgrad_slab(c->B, P.hh[1], P.ww[1], P.ch[1], 32, 2, 2, 1)
                                     : wgrad_slab(c->B, P.hh[l + 1], P.ww[l + 1], P.ch[l + 1], P.ch[l], 4, 4, P.st[l]);
        if (s > slab) slab = s;
    }
    Q.slab = b.take(slab);
    int cmax = 8; for (int l = 1; l <= P.L; ++l) if (P.ch[l] > cmax) cmax = P.ch[l];
    Q.colscr = b.take((size_t)2 * srcgan_col_reduce_blocks((long)B * P.hh[1] * P.ww[1]) * cmax * sizeof(float));
    Q.sums = b.take((size_t)2 * cmax * sizeof(float));
    Q.gnscr = P.inorm ? b.take(srcgan_gn_scratch_floats(c->B, cmax) * sizeof(float)) : 0;
    Q.total = b.off + 256;
}
}  // namespace

extern "C" int srcgan_nlayerd_num_params(const srcgan_nlayerd_cfg* c) { DPlan P; if (d_plan(c, P)) return -1; return P.nparams; }
extern "C" int srcgan_nlayerd_out_hw(const srcgan_nlayerd_cfg* c, int* oh, int* ow) {
    DPlan P; SG_TRY(d_plan(c, P)); if (oh) *oh = P.hh[P.L]; if (ow) *ow = P.ww[P.L]; return 0;
}
extern "C" size_t srcgan_nlayerd_ws_bytes(const srcgan_nlayerd_cfg* c) { DPlan P; if (d_plan(c, P)) return 0; return P.total; }
extern "C" size_t srcgan_nlayerd_bwd_scratch_bytes(const srcgan_nlayerd_cfg* c) {
    DPlan P; if (d_plan(c, P)) return 0; DBwdPlan Q; d_bwd_plan(c, P, Q); return Q.total;
}

extern "C" size_t srcgan_nlayerd_wpack_bytes(const srcgan_nlayerd_cfg* c) { DPlan P; if (d_plan(c, P)) return 0; return P.total - P.wpk; }

extern "C" int srcgan_nlayerd_forward(const srcgan_nlayerd_cfg* c, const float* x_nchw, const float* const* params,
                                      float* const* bn_running, int64_t* const* bn_nbt, void* ws, float* y_nchw, void* st) {
    return srcgan_nlayerd_forward_ex(c, x_nchw, params, bn_running, bn_nbt, ws, y_nchw, nullptr, st);
}
extern "C" int srcgan_nlayerd_backward(const srcgan_nlayerd_cfg* c, const float* dy_nchw, const float* const* params,
                                       void* ws, void* scratch, float* const* grads, float* dx_nchw, void* st) {
    return srcgan_nlayerd_backward_ex(c, dy_nchw, params, ws, scratch, grads, dx_nchw, nullptr, st);
}

extern "C" int srcgan_nlayerd_forward_ex(const srcgan_nlayerd_cfg* c, const float* x_nchw, const float* const* params,
                                         float* const* bn_running, int64_t* const* bn_nbt, void* ws, float* y_nchw,
                                         const srcgan_net_opts* opt, void* st) {
    DPlan P;
    SG_TRY(d_plan(c, P));
    SG_REQUIRE(x_nchw && params && ws && y_nchw, "srcgan_nlayerd_forward: null pointer");
    SG_REQUIRE(((uintptr_t)ws % 256) == 0, "srcgan_nlayerd_forward: workspace must be 256-byte aligned");
    const int dt = c->dtype, B = c->B;
    char* w8 = (char*)ws; char* wp = (opt && opt->wpack) ? (char*)opt->wpack : w8 + P.wpk;
    SG_REQUIRE(((uintptr_t)wp % 256) == 0, "srcgan_nlayerd_forward: wpack must be 256-byte aligned");
    if (!(opt && opt->wpack) || opt->pack) {
        PackList packs(dt, wp);
        for (int l = 0; l < P.L; ++l) {
            if (l == 0 && P.s2d) {
                // W[co][c][2ty+dy][2tx+dx] -> k = (dy,dx,c8), taps (ty,tx): one part per (dy,dx); channels c >= Cin stay zero
                SG_TRY(sg_fill_zero_guarded(wp + P.wf[0], srcgan_packed_weight_bytes(P.ch[1], 32, 4, dt), pack_guard(opt), (hipStream_t)st));
                for (int q = 0; q < 4; ++q)
                    packs.add(params[P.pw[0]], wp + P.wf[0], P.ch[1], P.ch[0], 2, 2, (long)P.ch[0] * 16, 16, 8, 2, (q >> 1) * 4 + (q & 1), q * 8, 32);
                continue;
            }
            packs.add(params[P.pw[l]], wp + P.wf[l], P.ch[l + 1], P.ch[l], 4, 4, (long)P.ch[l] * 16, 16, 4, 1, 0);
        }
        SG_TRY(packs.run(P.s2d ? "d_fwd_s2d" : "d_fwd", params[0], st, pack_guard(opt)));
    }
    if (P.s2d) SG_TRY(srcgan_nchw_f32_to_s2d(x_nchw, w8 + P.xin, B, c->in_ch, c->H, c->W, dt, st));
    else SG_TRY(srcgan_nchw_f32_to_nhwc(x_nchw, w8 + P.xin, B, c->in_ch, c->H, c->W, P.in_cs, dt, st));
    TRef cur = tref(w8 + P.xin, P.s2d ? 32 : P.in_cs);
    for (int l = 0; l < P.L; ++l) {
        const int cin_r = l == 0 ? P.in_cs : P.ch[l], cout = P.ch[l + 1];
        const int oh = P.hh[l + 1], ow = P.ww[l + 1];
        const long npix = (long)B * oh * ow;
        Conv cv(dt, (l == 0 && P.s2d) ? 2 : 4, (l == 0 && P.s2d) ? 2 : 4, (l == 0 && P.s2d) ? 1 : P.st[l]);
        if (l == 0 && P.s2d) cv.in(cur, B, c->H / 2 + 1, c->W / 2 + 1, 32).w(wp + P.wf[l], params[P.pb[l]]).pad(0, 0);
        else cv.in(cur, B, P.hh[l], P.ww[l], cin_r).w(wp + P.wf[l], P.pb[l] >= 0 ? params[P.pb[l]] : nullptr).pad(1, 1);
        if (l == 0) {                       // conv + bias + LeakyReLU (model/model.py:612)
            TRef y = tref(w8 + P.Y[l], cout);
            SG_TRY(cv.out(y, oh, ow, cout).lrelu().run(st));
            cur = y;
        } else if (l == P.L - 1) {          // final 1-channel prediction map (model/model.py:634)
            SG_HIP(hipMemsetAsync(w8 + P.out, 0, (size_t)npix * P.out_cs * P.esz, (hipStream_t)st));
            SG_TRY(cv.out(tref(w8 + P.out, P.out_cs), oh, ow, 1).run(st));
        } else {                            // conv -> BatchNorm2d -> LeakyReLU (model/model.py:620-631)
            TRef z = tref(w8 + P.Z[l], cout), y = tref(w8 + P.Y[l], cout);
            SG_TRY(cv.out(z, oh, ow, cout).run(st));
            if (P.inorm) {                  // InstanceNorm2d (no affine part, instance statistics in either mode) + LeakyReLU
                SG_TRY(srcgan_gn_forward(z.p, cout, nullptr, 0, y.p, cout, nullptr, nullptr, (float*)(w8 + P.stat[l]), B, (long)oh * ow, cout, cout,
                                         1e-5f, 1, 0.2f, dt, (float*)(w8 + P.gnscr), st));
                cur = y;
                continue;
            }
            float* mean = (float*)(w8 + P.stat[l]); float* var = mean + cout; float* rstd = var + cout;
            const int bi = P.bn_idx[l];
            const float* gamma = params[P.pg[l]]; const float* beta = params[P.pbeta[l]];
            if (c->training) {
                float* scr = (float*)(w8 + P.colscr);
                SG_TRY(srcgan_col_reduce(3, z.p, cout, 0, nullptr, 0, 0, nullptr, nullptr, npix, cout, 1.f, mean, var, scr, dt, st));      // one pass: mean and variance
                SG_TRY(srcgan_bn_finalize(mean, var, rstd, bn_running ? bn_running[2 * bi] : nullptr, bn_running ? bn_running[2 * bi + 1] : nullptr,
                                          bn_nbt ? bn_nbt[bi] : nullptr, cout, npix, 0.1f, 1e-5f, st));
                SG_TRY(srcgan_bn_apply_lrelu(z.p, y.p, mean, rstd, gamma, beta, npix, cout, cout, 0.2f, dt, st));
            } else {
                SG_REQUIRE(bn_running, "srcgan_nlayerd_forward: eval mode needs running statistics");
                SG_TRY(srcgan_bn_eval_rstd(bn_running[2 * bi + 1], rstd, cout, 1e-5f, st));
                SG_TRY(srcgan_bn_apply_lrelu(z.p, y.p, bn_running[2 * bi], rstd, gamma, beta, npix, cout, cout, 0.2f, dt, st));
            }
            cur = y;
        }
    }
    SG_TRY(srcgan_nhwc_to_nchw_f32(w8 + P.out, y_nchw, B, 1, P.hh[P.L], P.ww[P.L], P.out_cs, 0, dt, st));
    return 0;
}

extern "C" int srcgan_nlayerd_backward_ex(const srcgan_nlayerd_cfg* c, const float* dy_nchw, const float* const* params,
                                          void* ws, void* scratch, float* const* grads, float* dx_nchw, const srcgan_net_opts* opt, void* st) {
    DPlan P;
    SG_TRY(d_plan(c, P));
    SG_REQUIRE(c->training || P.inorm, "srcgan_nlayerd_backward: backward through eval-mode BatchNorm is not supported");
    SG_REQUIRE(dy_nchw && params && ws && scratch && grads, "srcgan_nlayerd_backward: null pointer");
    DBwdPlan Q;
    d_bwd_plan(c, P, Q);
    const int dt = c->dtype, B = c->B;
    char* w8 = (char*)ws; char* s8 = (char*)scratch; char* wp = (opt && opt->wpack) ? (char*)opt->wpack : w8 + P.wpk;
    float* slab = (float*)(s8 + Q.slab); float* colscr = (float*)(s8 + Q.colscr); float* sums = (float*)(s8 + Q.sums);
    // ---- packed dgrad weights
    const bool pack_dx = dx_nchw || (opt && opt->wpack);       // a persistent pack serves later calls that may want dx
    PackList packs(dt, wp);
    for (int l = 0; l < P.L && (!(opt && opt->wpack) || opt->pack); ++l) {
        if (l == 0 && !pack_dx) continue;
        const int cin = P.ch[l], cout = P.ch[l + 1];
        if (l == 0 && P.s2d) {
            // dX'[j,i,(dy,dx,c)] = sum_{u,v,co} dY[j-1+u, i-1+v, co] * W[co][c][2(1-u)+dy][2(1-v)+dx]: rows (dy,dx,c8), k = co, 2x2 taps.
            // The rows of one (dy,dx) are a strided slice of the weight: one part per (dy,dx) and per half of k (so that no part
            // is a whole matrix, which would zero-fill all 32 rows of its view), written at row offset (dy*2+dx)*8.
            const int kce = 64 / P.esz;
            SG_TRY(sg_fill_zero_guarded(wp + P.wd[0][0], srcgan_packed_weight_bytes(32, cout, 4, dt), pack_guard(opt), (hipStream_t)st));
            for (int q = 0; q < 4; ++q)
                for (int hk = 0; hk < 2; ++hk)
                    packs.add(params[P.pw[0]], wp + P.wd[0][0] + (size_t)q * 8 * kce * P.esz, cin, cout / 2, 2, 2, 16, (long)cin * 16, -8, -2,
                              10 + (q >> 1) * 4 + (q & 1) + (long)hk * (cout / 2) * cin * 16, hk * (cout / 2), cout);
            continue;
        }
        if (P.st[l] == 1)
            packs.add(params[P.pw[l]], wp + P.wd[l][0], cin, cout, 4, 4, 16, (long)cin * 16, -4, -1, 15);
        else
            for (int q = 0; q < 4; ++q) {   // stride-2 dgrad by output parity (a,b): 2x2 sub-kernel, ky = (a?2:3) - 2*ty
                const int a = q >> 1, bb = q & 1;
                const long off = (a ? 2 : 3) * 4 + (bb ? 2 : 3);
                packs.add(params[P.pw[l]], wp + P.wd[l][q], cin, cout, 2, 2, 16, (long)cin * 16, -8, -2, off);
            }
    }
    SG_TRY(packs.run(P.s2d ? (pack_dx ? "d_bwd_dx_s2d" : "d_bwd_s2d") : (pack_dx ? "d_bwd_dx" : "d_bwd"), params[0], st, pack_guard(opt)));
    // ---- dy -> NHWC (1 channel, padded with zeros to 8)
    const int Lh = P.hh[P.L], Lw = P.ww[P.L];
    TRef dcur = tref(s8 + Q.dO, P.out_cs);
    SG_TRY(srcgan_nchw_f32_to_nhwc(dy_nchw, dcur.p, B, 1, Lh, Lw, P.out_cs, dt, st));
    int dcur_c = P.out_cs;      // channels to read from dcur (padded)
    for (int l = P.L - 1; l >= 0; --l) {
        const int cin = P.ch[l], cout = P.ch[l + 1];
        const int ih = P.hh[l], iw = P.ww[l], oh = P.hh[l + 1], ow = P.ww[l + 1];
        const long npix = (long)B * oh * ow;
        const bool bn = P.bn_idx[l] >= 0;
        if (P.nrm[l] && P.inorm) {
            // dcur = dL/dy * lrelu'(y) (mask fused in the producer); instance-norm backward per (image, channel), in place
            TRef z = tref(w8 + P.Z[l], cout);
            SG_TRY(srcgan_gn_backward(dcur.p, dcur.cs, nullptr, 0, z.p, cout, nullptr, (const float*)(w8 + P.stat[l]), dcur.p, dcur.cs, nullptr, 0, 0,
                                      nullptr, nullptr, 0, 0.2f, B, (long)oh * ow, cout, cout, dt, (float*)(s8 + Q.gnscr), st));
        }
        if (bn) {
            // dcur = dL/dy * lrelu'(y) (mask fused in the producer).  BN backward (train): needs sum g, sum g*xhat
            float* mean = (float*)(w8 + P.stat[l]); float* rstd = mean + 2 * cout;
            TRef z = tref(w8 + P.Z[l], cout);
            SG_TRY(srcgan_col_reduce(2, dcur.p, dcur.cs, 0, z.p, cout, 0, mean, rstd, npix, cout, 1.f, sums, sums + cout, colscr, dt, st));
            if (grads[P.pbeta[l]]) SG_HIP(hipMemcpyAsync(grads[P.pbeta[l]], sums, cout * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)st));
            if (grads[P.pg[l]]) SG_HIP(hipMemcpyAsync(grads[P.pg[l]], sums + cout, cout * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)st));
            SG_TRY(srcgan_bn_bwd_apply(dcur.p, z.p, dcur.p, mean, rstd, params[P.pg[l]], sums, sums + cout, npix, cout, cout, dt, st));
        }
        TRef xin_l = l == 0 ? tref(w8 + P.xin, P.s2d ? 32 : P.in_cs) : tref(w8 + P.Y[l - 1], cin);
        if (grads[P.pw[l]]) {
            if (l == 0 && P.s2d) {          // gradient of the folded weight, then back to [Cout][Cin][4][4]
                float* gfold = (float*)(s8 + Q.gfold);
                SG_TRY(wgrad_call(dt, dcur, oh, ow, cout, xin_l, B, c->H / 2 + 1, c->W / 2 + 1, 32, 2, 2, 1, 0, 0, lay_fwd(32, 2, 2), 1.f, slab, gfold, st));
                SG_TRY(srcgan_s2d_wgrad_unfold(gfold, grads[P.pw[l]], cout, cin, 0, st));
            } else
            SG_TRY(wgrad_call(dt, dcur, oh, ow, cout, xin_l, B, ih, iw, cin, 4, 4, P.st[l], 1, 1, lay_fwd(cin, 4, 4), 1.f, slab, grads[P.pw[l]], st));
        }
        if (P.pb[l] >= 0 && grads[P.pb[l]]) SG_TRY(bias_grad(dt, dcur, npix, cout, 1.f, grads[P.pb[l]], colscr, st));
        if (l == 0 && !dx_nchw) break;
        // dgrad -> gradient of the layer input, times LeakyReLU' of that input (it is some layer's post-activation)
        if (l == 0 && P.s2d) {              // one 2x2 "full" convolution into the space-to-depth gradient, then back to NCHW f32
            TRef dst = tref(s8 + Q.dxin, 32);
            SG_TRY(Conv(dt, 2, 2, 1).in(dcur, B, oh, ow, dcur_c).w(wp + P.wd[0][0]).out(dst, c->H / 2 + 1, c->W / 2 + 1, 32).pad(1, 1).run(st));
            SG_TRY(srcgan_s2d_to_nchw_f32(dst.p, dx_nchw, B, c->in_ch, c->H, c->W, dt, st));
            return 0;
        }
        TRef dst = l == 0 ? tref(s8 + Q.dxin, P.in_cs) : tref(s8 + Q.g[l & 1], cin);
        if (l == 0) SG_HIP(hipMemsetAsync(dst.p, 0, (size_t)B * ih * iw * P.in_cs * P.esz, (hipStream_t)st));
        TRef mz = l == 0 ? TNULL : tref(w8 + P.Y[l - 1], cin);
        if (P.st[l] == 1) {
            Conv cv(dt, 4, 4, 1);
            cv.in(dcur, B, oh, ow, dcur_c).w(wp + P.wd[l][0]).out(dst, ih, iw, cin).pad(2, 2);
            if (mz.p) cv.mask(mz, 0);
            SG_TRY(cv.run(st));
        } else {
            // stride-2 layers: all four output parities of the input gradient from one staged dy tile (conv_par4.hip); the four
            // parity packs are equally spaced.  (SRCGAN_NO_PAR4, diagnostic builds: the four separate 2x2 launches of rounds 1-2.)
            static const bool no_par4 = sg_env("SRCGAN_NO_PAR4") != nullptr;
            const long wstep = (long)(P.wd[l][1] - P.wd[l][0]);
            const bool even = P.wd[l][2] - P.wd[l][1] == (size_t)wstep && P.wd[l][3] - P.wd[l][2] == (size_t)wstep;
            if (!no_par4 && even && ih >= 2 && iw >= 2 && cin % (16 / P.esz) == 0 && dst.cs % (16 / P.esz) == 0) {
                Conv cv(dt, 2, 2, 1);
                cv.in(dcur, B, oh, ow, dcur_c).w(wp + P.wd[l][0]).out(dst, (ih + 1) / 2, (iw + 1) / 2, cin).scatter(2, 0, 0, ih, iw);
                cv.d.npar = 4; cv.d.wpar_stride = wstep;
                if (mz.p) cv.mask(mz, 0);
                SG_TRY(cv.run(st));
            } else
            for (int q = 0; q < 4; ++q) {
                const int a = q >> 1, bb = q & 1;
                const int mh = (ih - a + 1) / 2, mw = (iw - bb + 1) / 2;
                if (mh <= 0 || mw <= 0) continue;
                Conv cv(dt, 2, 2, 1);
                cv.in(dcur, B, oh, ow, dcur_c).w(wp + P.wd[l][q]).out(dst, mh, mw, cin).pad(a ? 0 : 1, bb ? 0 : 1).scatter(2, a, bb, ih, iw);
                if (mz.p) cv.mask(mz, 0);
                SG_TRY(cv.run(st));
            }
        }
        dcur = dst; dcur_c = cin;
    }
    if (dx_nchw) SG_TRY(srcgan_nhwc_to_nchw_f32(s8 + Q.dxin, dx_nchw, B, c->in_ch, c->H, c->W, P.in_cs, 0, dt, st));
    return 0;
}

// ======================================================================================== op-list networks
// ResDeconv colouriser -- reference src/model/resdeconv.py:99-195 (the second network of every trainCas step,
// trainCas.py:31,99-100): ResNet-18-style encoder (7x7 s2 stem, BasicBlocks 64-128-256-512 with 3x3 s2 + 1x1 s2 shortcut at
// each widening) and a mirrored decoder (ConvTranspose2d k2 s2 + two BasicBlocks per scale), GroupNorm(32) + ReLU, no biases.
// ESPCN -- src/model/espcn.py:18-51 (the CLI default --SRModel, trainCas.py:169): 5x5, 3x3, 3x3 convs + ReLU, 3x3 conv to
// 64 r^2 channels, PixelShuffle(r), 3x3 conv.   SRCNN -- src/model/srcnn.py:17-42: 9x9, 1x1, 5x5 convs, each + ReLU.
//
// Each network is a short op list built once per call; forward walks it, backward walks it in reverse.  A tensor with two
// consumers (a block's input: first convolution + shortcut) gets its first gradient contribution by a plain store and the
// second through the convolution epilogue's in-place residual operand.  The gradient stored for the output of a
// convolution with a fused ReLU is the gradient w.r.t. its pre-activation: whoever writes it applies the mask (the consumer's
// dgrad epilogue, or srcgan_mask_inplace for the gradient arriving from the loss).
namespace {
// Input gradient of a k x k stride-2 convolution with padding `pad`, by output parity a (0 / 1) along one axis:
//   dx[2 j + a] = sum over the kernel rows ky == (a + pad) mod 2 of dy[j + (a + pad - ky) / 2] * w[ky]
// -> a stride-1 sub-convolution with n taps; tap t (ascending dy row) uses ky = ky_max - 2 t and needs `lead` rows above row j.
struct Par2 { int n, ky_max, lead; };
static inline Par2 par2(int k, int pad, int a) {
    int ky_max = k - 1;
    if (((ky_max ^ (a + pad)) & 1) != 0) --ky_max;
    Par2 r; r.ky_max = ky_max; r.n = ky_max >= 0 ? ky_max / 2 + 1 : 0; r.lead = (ky_max - a - pad) / 2;
    return r;
}

struct RdT { int C, cs, H, W, act; size_t off; };       // act: produced by a convolution with a fused ReLU
struct RdOp {
    int type;                 // 0 conv (+bias)(+ReLU), 1 GroupNorm(+res)(+ReLU), 2 ConvTranspose2d k2 s2, 3 PixelShuffle(r)
    int in, out, res, relu;
    int k, s, pad, w, bias;   // w: parameter index of the weight (GroupNorm: gamma, beta = w + 1; -1 = no affine part); bias: parameter index or -1
    int ngrp;                 // GroupNorm: groups (InstanceNorm2d: = channels)
    float slope;              // GroupNorm activation: 0 = ReLU, 0.2 = LeakyReLU (edsr.py:42)
    size_t wf[4], wd[4], stats;
};
struct RdPlan {
    int dtype, esz, B, H, W, in_ch, out_ch, in_cs, out_cs, nparams, maxC;
    std::vector<RdT> T; std::vector<RdOp> ops;
    size_t xin, gnfwd, wpk, total, act_bytes;
    std::vector<size_t> g;    // backward: gradient buffer offsets (scratch), same shapes as T
    size_t slab, gnscr, colscr, bwd_total;
};

struct RdBuilder {
    RdPlan& P; Bump b; int np = 0; int B;
    int inorm = 0;            // normalisation layers are InstanceNorm2d (no parameters) instead of GroupNorm(32, C)
    RdBuilder(RdPlan& p, int B_) : P(p), B(B_) {}
    int tensor(int C, int cs, int H, int W, int act = 0) { P.T.push_back(RdT{C, cs, H, W, act, b.take((size_t)B * H * W * cs * P.esz)}); return (int)P.T.size() - 1; }
    int conv(int in, int cout, int k, int s, int pad, bool bias = false, bool relu = false) {
        const RdT ti = P.T[in];
        const int oh = (ti.H + 2 * pad - k) / s + 1, ow = (ti.W + 2 * pad - k) / s + 1;
        RdOp o; memset(&o, 0, sizeof(o));
        o.type = 0; o.in = in; o.res = -1; o.k = k; o.s = s; o.pad = pad; o.relu = relu; o.w = np++; o.bias = bias ? np++ : -1;
        o.out = tensor(cout, cout < 8 ? 8 : cout, oh, ow, relu);
        P.ops.push_back(o); return o.out;
    }
    int gn(int in, int res, int relu) {
        const RdT ti = P.T[in];
        RdOp o; memset(&o, 0, sizeof(o));
        o.type = 1; o.in = in; o.out = tensor(ti.C, ti.cs, ti.H, ti.W); o.res = res; o.relu = relu; o.bias = -1;
        if (inorm) { o.w = -1; o.ngrp = ti.C; } else { o.w = np; np += 2; o.ngrp = 32; }
        o.stats = b.take((size_t)B * o.ngrp * 2 * sizeof(float));
        P.ops.push_back(o); return o.out;
    }
    int deconv(int in, int cout) {
        const RdT ti = P.T[in];
        RdOp o; memset(&o, 0, sizeof(o));
        o.type = 2; o.in = in; o.out = tensor(cout, cout, 2 * ti.H, 2 * ti.W); o.res = -1; o.k = 2; o.s = 2; o.w = np++; o.bias = -1;
        P.ops.push_back(o); return o.out;
    }
    int shuffle(int in, int r) {
        const RdT ti = P.T[in];
        RdOp o; memset(&o, 0, sizeof(o));
        o.type = 3; o.in = in; o.res = -1; o.k = r; o.w = -1; o.bias = -1;
        o.out = tensor(ti.C / (r * r), ti.C / (r * r), ti.H * r, ti.W * r);
        P.ops.push_back(o); return o.out;
    }
    // BasicBlock (resdeconv.py:56-97).  state_dict order inside a block: conv1, bn1, conv2, bn2, downsample.{0,1}
    int block(int x, int planes, int stride) {
        const bool ds = stride != 1 || P.T[x].C != planes;
        int t = conv(x, planes, 3, stride, 1);
        t = gn(t, -1, 1);
        t = conv(t, planes, 3, 1, 1);
        const int gn2_param = np; if (!inorm) np += 2;      // bn2's parameters precede the shortcut's in the state_dict
        int idn = x;
        if (ds) { idn = conv(x, planes, 1, stride, 0); idn = gn(idn, -1, 0); }
        const int out = gn(t, idn, 1);
        if (!inorm) { P.ops.back().w = gn2_param; np -= 2; }
        return out;
    }
    void input(int in_ch, int H, int W) {
        P.in_ch = in_ch; P.in_cs = img_cs(in_ch); P.H = H; P.W = W;
        P.xin = b.take((size_t)B * H * W * P.in_cs * P.esz);
        P.T.push_back(RdT{in_ch, P.in_cs, H, W, 0, P.xin});
    }
    void finish(int dtype) {
        P.nparams = np;
        P.maxC = 8;
        for (const RdT& t : P.T) if (t.C > P.maxC) P.maxC = t.C;
        P.out_ch = P.T.back().C; P.out_cs = P.T.back().cs;
        P.gnfwd = b.take(srcgan_gn_scratch_floats(B, P.maxC > 1024 ? 1024 : P.maxC) * sizeof(float));
        P.act_bytes = b.off;
        P.wpk = b.off;
        Bump wb;
        auto pk = [&](int rows, int k, int taps) { return wb.take(srcgan_packed_weight_bytes(rows, k, taps, dtype)); };
        for (RdOp& o : P.ops) {
            const RdT ti = P.T[o.in], to = P.T[o.out];
            if (o.type == 0) {
                o.wf[0] = pk(to.C, ti.C, o.k * o.k);
                {       // (the input tensor's own gradient is produced on request: dx_nchw of the backward entry points)
                    if (o.s == 1) o.wd[0] = pk(ti.C, to.C, o.k * o.k);
                    else if (o.k == 1) o.wd[0] = pk(ti.C, to.C, 1);
                    else for (int q = 0; q < 4; ++q) o.wd[q] = pk(ti.C, to.C, par2(o.k, o.pad, q >> 1).n * par2(o.k, o.pad, q & 1).n);
                }
            } else if (o.type == 2) {
                for (int q = 0; q < 4; ++q) o.wf[q] = pk(to.C, ti.C, 1);
                o.wd[0] = pk(ti.C, to.C, 4);
            }
        }
        P.total = align_up(P.wpk + wb.off + 256, 256);
        // backward scratch
        Bump s;
        P.g.resize(P.T.size());
        long maxpix = 1;
        for (size_t i = 0; i < P.T.size(); ++i) {
            P.g[i] = s.take((size_t)B * P.T[i].H * P.T[i].W * P.T[i].cs * P.esz);
            if ((long)B * P.T[i].H * P.T[i].W > maxpix) maxpix = (long)B * P.T[i].H * P.T[i].W;
        }
        size_t slab = 0;
        for (const RdOp& o : P.ops) {
            const RdT ti = P.T[o.in], to = P.T[o.out];
            size_t v = 0;
            if (o.type == 0) v = wgrad_slab(B, to.H, to.W, to.C, ti.C, o.k, o.k, o.s);
            else if (o.type == 2) v = wgrad_slab(B, ti.H, ti.W, ti.C, to.C, 2, 2, 2);
            if (v > slab) slab = v;
        }
        P.slab = s.take(slab);
        P.gnscr = s.take(srcgan_gn_scratch_floats(B, P.maxC > 1024 ? 1024 : P.maxC) * sizeof(float));
        P.colscr = s.take((size_t)2 * srcgan_col_reduce_blocks(maxpix) * P.maxC * sizeof(float));
        P.bwd_total = s.off + 256;
    }
};

static int rd_common(int dtype, int B, int H, int W, RdPlan& P, const char* who) {
    SG_REQUIRE(sg_dtype_ok(dtype), "%s: bad dtype %d", who, dtype);
    SG_REQUIRE(B > 0 && H > 0 && W > 0, "%s: bad B/H/W", who);
    P.dtype = dtype; P.esz = dtype == SRCGAN_F32 ? 4 : 2; P.B = B;
    return 0;
}

static int rd_plan(const srcgan_resdeconv_cfg* c, RdPlan& P) {
    SG_REQUIRE(c, "resdeconv: null cfg");
    SG_TRY(rd_common(c->dtype, c->B, c->H, c->W, P, "resdeconv"));
    SG_REQUIRE(c->in_ch == 3 && c->out_ch > 0 && c->out_ch <= 8, "resdeconv: the stem takes 3 channels (resdeconv.py:113), tar_ch must be in 1..8");
    SG_REQUIRE(c->H % 16 == 0 && c->W % 16 == 0, "resdeconv: H and W must be multiples of 16 (four stride-2 stages mirrored by four x2 deconvolutions)");
    int layers[4];
    const bool dflt = c->layers[0] == 0 && c->layers[1] == 0 && c->layers[2] == 0 && c->layers[3] == 0;
    for (int l = 0; l < 4; ++l) {
        layers[l] = dflt ? 2 : c->layers[l];
        SG_REQUIRE(layers[l] >= 1 && layers[l] <= 64, "resdeconv: layers[%d] = %d (1..64 BasicBlocks per stage)", l, layers[l]);
    }
    SG_REQUIRE(c->norm == 0 || c->norm == 1, "resdeconv: norm must be 0 (GroupNorm(32, C)) or 1 (InstanceNorm2d)");
    RdBuilder nb(P, c->B);
    nb.inorm = c->norm == 1;
    nb.input(c->in_ch, c->H, c->W);
    int t = nb.conv(0, 64, 7, 2, 3);
    t = nb.gn(t, -1, 1);
    const int widths[4] = {64, 128, 256, 512};
    for (int l = 0; l < 4; ++l)                 // _make_layer (resdeconv.py:148-163): the first block carries the stride / shortcut
        for (int k = 0; k < layers[l]; ++k) t = nb.block(t, widths[l], (k == 0 && l > 0) ? 2 : 1);
    const int up_w[3] = {256, 128, 64};
    for (int l = 0; l < 3; ++l) {               // upRes1..3 use layers[2], layers[1], layers[0] (resdeconv.py:131-137)
        t = nb.deconv(t, up_w[l]);
        for (int k = 0; k < layers[2 - l]; ++k) t = nb.block(t, up_w[l], 1);
    }
    t = nb.deconv(t, 64);
    nb.conv(t, c->out_ch, 3, 1, 1);
    nb.finish(c->dtype);
    return 0;
}

// kind 0: ESPCN(in_ch, ou_ch, upscale_factor, base_kernel) espcn.py:18-51;  kind 1: SRCNN(in_ch, ou_ch, ., base_kernel) srcnn.py:17-42
static int sr_plan(const srcgan_srnet_cfg* c, RdPlan& P) {
    SG_REQUIRE(c, "srnet: null cfg");
    SG_TRY(rd_common(c->dtype, c->B, c->H, c->W, P, "srnet"));
    SG_REQUIRE(c->kind >= 0 && c->kind <= 2, "srnet: kind must be 0 (ESPCN), 1 (SRCNN) or 2 (EDSR)");
    SG_REQUIRE(c->kind != 2 || (c->nres >= 1 && c->base % 32 == 0 && (c->up & (c->up - 1)) == 0), "srnet: EDSR needs num_residuals >= 1, base_channel % 32 == 0 and a power-of-two upscale factor");
    SG_REQUIRE(c->in_ch > 0 && c->in_ch <= 8 && c->out_ch > 0 && c->out_ch <= 8, "srnet: in/out channels must be in 1..8");
    SG_REQUIRE(c->base > 0 && c->base % 16 == 0, "srnet: base_kernel must be a multiple of 16");
    SG_REQUIRE(c->up >= 1 && c->up <= 8, "srnet: upscale_factor must be in 1..8");
    RdBuilder nb(P, c->B);
    nb.input(c->in_ch, c->H, c->W);
    if (c->kind == 0) {
        int t = nb.conv(0, c->base, 5, 1, 2, true, true);
        t = nb.conv(t, c->base, 3, 1, 1, true, true);
        t = nb.conv(t, c->base / 2, 3, 1, 1, true, true);
        t = nb.conv(t, c->base * c->up * c->up, 3, 1, 1, true, false);
        t = nb.shuffle(t, c->up);
        nb.conv(t, c->out_ch, 3, 1, 1, true, false);
    } else if (c->kind == 1) {
        int t = nb.conv(0, c->base, 9, 1, 4, true, true);
        t = nb.conv(t, c->base / 2, 1, 1, 0, true, true);
        nb.conv(t, c->out_ch, 5, 1, 2, true, true);
    } else {
        // EDSR (edsr.py:37-110): input_conv; num_residuals x [conv1 -> gn -> LeakyReLU(0.2) -> conv2 -> gn (the SAME GroupNorm
        // module) -> + x]; mid_conv + input_conv's output; ConvTranspose2d k2 s2 per x2 stage; output_conv.  state_dict order
        // inside a block: conv1.{w,b}, conv2.{w,b}, gn.{w,b}.
        const int feat = nb.conv(0, c->base, 3, 1, 1, true, false);
        int t = feat;
        for (int i = 0; i < c->nres; ++i) {
            const int base = nb.np, x = t;
            t = nb.conv(x, c->base, 3, 1, 1, true, false);
            P.ops.back().w = base; P.ops.back().bias = base + 1;
            t = nb.gn(t, -1, 1);
            P.ops.back().w = base + 4; P.ops.back().slope = 0.2f;
            t = nb.conv(t, c->base, 3, 1, 1, true, false);
            P.ops.back().w = base + 2; P.ops.back().bias = base + 3;
            t = nb.gn(t, x, 0);
            P.ops.back().w = base + 4;
            nb.np = base + 6;
        }
        t = nb.conv(t, c->base, 3, 1, 1, true, false);
        P.ops.back().res = feat;
        for (int f = 1; f < c->up; f *= 2) t = nb.deconv(t, c->base);
        nb.conv(t, c->out_ch, 3, 1, 1, true, false);
    }
    nb.finish(c->dtype);
    return 0;
}
static inline TRef rd_t(char* base, const RdT& t) { return tref(base + t.off, t.cs); }

static int rd_forward(const RdPlan& P, const float* x_nchw, const float* const* params, void* ws, float* y_nchw, const char* tag, void* st) {
    SG_REQUIRE(x_nchw && params && ws && y_nchw, "%s forward: null pointer", tag);
    SG_REQUIRE(((uintptr_t)ws % 256) == 0, "%s forward: workspace must be 256-byte aligned", tag);
    const int dt = P.dtype, B = P.B;
    char* w8 = (char*)ws; char* wp = w8 + P.wpk;
    PackList packs(dt, wp);
    for (const RdOp& o : P.ops) {
        const RdT ti = P.T[o.in], to = P.T[o.out];
        if (o.type == 0) packs.add(params[o.w], wp + o.wf[0], to.C, ti.C, o.k, o.k, (long)ti.C * o.k * o.k, (long)o.k * o.k, o.k, 1, 0);
        else if (o.type == 2)
            for (int q = 0; q < 4; ++q) packs.add(params[o.w], wp + o.wf[q], to.C, ti.C, 1, 1, 4, (long)to.C * 4, 0, 0, q);
    }
    char key[64]; snprintf(key, sizeof(key), "%s_fwd", tag);
    SG_TRY(packs.run(key, params[0], st));
    SG_TRY(srcgan_nchw_f32_to_nhwc(x_nchw, w8 + P.xin, B, P.in_ch, P.H, P.W, P.in_cs, dt, st));
    float* gnscr = (float*)(w8 + P.gnfwd);
    for (const RdOp& o : P.ops) {
        const RdT ti = P.T[o.in], to = P.T[o.out];
        TRef xin = rd_t(w8, ti), out = rd_t(w8, to);
        if (o.type == 0) {
            if (to.C < to.cs) SG_HIP(hipMemsetAsync(out.p, 0, (size_t)B * to.H * to.W * to.cs * P.esz, (hipStream_t)st));
            Conv cv(dt, o.k, o.k, o.s);
            cv.in(xin, B, ti.H, ti.W, ti.C < 8 ? ti.cs : ti.C).w(wp + o.wf[0], o.bias >= 0 ? params[o.bias] : nullptr).out(out, to.H, to.W, to.C).pad(o.pad, o.pad);
            if (o.relu) { cv.lrelu(); cv.d.slope = 0.f; }
            if (o.res >= 0) cv.res1(rd_t(w8, P.T[o.res]), to.C, 1.f);          // y = conv(x) + res (edsr.py:97-98)
            SG_TRY(cv.run(st));
        } else if (o.type == 1) {
            const void* res = o.res >= 0 ? (w8 + P.T[o.res].off) : nullptr;
            SG_TRY(srcgan_gn_forward(xin.p, ti.cs, res, o.res >= 0 ? P.T[o.res].cs : 0, out.p, to.cs, o.w >= 0 ? params[o.w] : nullptr,
                                     o.w >= 0 ? params[o.w + 1] : nullptr, (float*)(w8 + o.stats), B, (long)ti.H * ti.W, ti.C, o.ngrp, 1e-5f, o.relu,
                                     o.slope, dt, gnscr, st));
        } else if (o.type == 2) {
            for (int q = 0; q < 4; ++q)
                SG_TRY(Conv(dt, 1, 1, 1).in(xin, B, ti.H, ti.W, ti.C).w(wp + o.wf[q]).out(out, ti.H, ti.W, to.C)
                           .scatter(2, q >> 1, q & 1, to.H, to.W).run(st));
        } else {
            SG_TRY(srcgan_pixel_shuffle_nhwc(xin.p, ti.cs, out.p, to.cs, B, ti.H, ti.W, to.C, o.k, 0, dt, st));
        }
    }
    const RdT& last = P.T.back();
    SG_TRY(srcgan_nhwc_to_nchw_f32(w8 + last.off, y_nchw, B, P.out_ch, last.H, last.W, last.cs, 0, dt, st));
    return 0;
}

static int rd_backward(const RdPlan& P, const float* dy_nchw, float* dx_nchw, const float* const* params, void* ws, void* scratch, float* const* grads,
                       const char* tag, void* st) {
    SG_REQUIRE(dy_nchw && params && ws && scratch && grads, "%s backward: null pointer", tag);
    SG_REQUIRE(((uintptr_t)ws % 256) == 0 && ((uintptr_t)scratch % 256) == 0, "%s backward: buffers must be 256-byte aligned", tag);
    const int dt = P.dtype, B = P.B;
    char* w8 = (char*)ws; char* s8 = (char*)scratch; char* wp = w8 + P.wpk;
    float* slab = (float*)(s8 + P.slab); float* gnscr = (float*)(s8 + P.gnscr); float* colscr = (float*)(s8 + P.colscr);
    auto G = [&](int idx) { return idx >= 0 ? grads[idx] : nullptr; };
    {   // dgrad weight packs
        PackList packs(dt, wp);
        for (const RdOp& o : P.ops) {
            const RdT ti = P.T[o.in], to = P.T[o.out];
            if (o.type == 0 && (o.in != 0 || dx_nchw)) {
                if (o.s == 1) { const WLayout L = lay_dgrad_s1(ti.C, o.k, o.k); packs.add(params[o.w], wp + o.wd[0], ti.C, to.C, o.k, o.k, L.sr, L.sk, L.sty, L.stx, L.off); }
                else if (o.k == 1) packs.add(params[o.w], wp + o.wd[0], ti.C, to.C, 1, 1, 1, (long)ti.C, 0, 0, 0);
                else
                    for (int q = 0; q < 4; ++q) {       // stride-2 dgrad by output parity (a,b): par2() taps per axis (3x3 p1: 1 or 2; 7x7 p3: 3 or 4)
                        const Par2 py = par2(o.k, o.pad, q >> 1), px = par2(o.k, o.pad, q & 1);
                        const long kk = (long)o.k * o.k;
                        packs.add(params[o.w], wp + o.wd[q], ti.C, to.C, py.n, px.n, kk, (long)ti.C * kk, -2 * o.k, -2, (long)py.ky_max * o.k + px.ky_max);
                    }
            } else if (o.type == 2) {
                packs.add(params[o.w], wp + o.wd[0], ti.C, to.C, 2, 2, (long)to.C * 4, 4, 2, 1, 0);
            }
        }
        char key[64]; snprintf(key, sizeof(key), "%s_bwd", tag);
        SG_TRY(packs.run(key, params[0], st));
    }
    std::vector<char> written(P.T.size(), 0), seen_param(P.nparams + 2, 0);
    auto gt = [&](int id) { return tref(s8 + P.g[id], P.T[id].cs); };
    {
        const RdT& last = P.T.back();
        const int id = (int)P.T.size() - 1;
        SG_TRY(srcgan_nchw_f32_to_nhwc(dy_nchw, s8 + P.g[id], B, P.out_ch, last.H, last.W, last.cs, dt, st));
        if (last.act) SG_TRY(srcgan_mask_inplace(s8 + P.g[id], w8 + last.off, 0.f, (long)B * last.H * last.W * last.cs, dt, st));
        written[id] = 1;
    }
    for (int k = (int)P.ops.size() - 1; k >= 0; --k) {
        const RdOp& o = P.ops[k];
        const RdT ti = P.T[o.in], to = P.T[o.out];
        SG_REQUIRE(written[o.out], "%s backward: internal error (gradient of tensor %d missing)", tag, o.out);
        TRef xin = rd_t(w8, ti), dy = gt(o.out);
        const bool need_dx = o.in != 0 || dx_nchw;
        TRef dx = need_dx ? gt(o.in) : TNULL;
        bool acc = need_dx && written[o.in];
        if (o.type == 0) {
            if (o.res >= 0) {       // y = conv(x) + res: the residual's gradient is dy itself
                const RdT tr = P.T[o.res];
                TRef dr = gt(o.res);
                if (!written[o.res]) SG_HIP(hipMemcpyAsync(dr.p, dy.p, (size_t)B * tr.H * tr.W * tr.cs * P.esz, hipMemcpyDeviceToDevice, (hipStream_t)st));
                else SG_TRY(srcgan_add_inplace(dr.p, tr.cs, 0, dy.p, to.cs, 0, nullptr, 0, 0, 0.f, (long)B * tr.H * tr.W, tr.C, dt, st));
                written[o.res] = 1;
                acc = need_dx && written[o.in];
            }
            const bool fused_bias = o.bias >= 0 && o.k == 3 && G(o.w);
            if (G(o.w))
                SG_TRY(wgrad_call(dt, dy, to.H, to.W, to.C, xin, B, ti.H, ti.W, ti.C, o.k, o.k, o.s, o.pad, o.pad, lay_fwd(ti.C, o.k, o.k), 1.f, slab, G(o.w), st,
                                  fused_bias ? G(o.bias) : nullptr));
            if (o.bias >= 0 && G(o.bias) && !fused_bias) SG_TRY(bias_grad(dt, dy, (long)B * to.H * to.W, to.C, 1.f, G(o.bias), colscr, st));
            if (need_dx) {
                SG_REQUIRE(!(acc && ti.act), "%s backward: internal error (activated tensor with two consumers)", tag);
                if (o.in == 0 && !acc) SG_HIP(hipMemsetAsync(dx.p, 0, (size_t)B * ti.H * ti.W * ti.cs * P.esz, (hipStream_t)st));    // padded image channels
                if (o.s == 1) {
                    Conv cv(dt, o.k, o.k, 1);
                    cv.in(dy, B, to.H, to.W, to.C < 8 ? to.cs : to.C).w(wp + o.wd[0]).out(dx, ti.H, ti.W, ti.C).pad(o.k - 1 - o.pad, o.k - 1 - o.pad);
                    if (acc) cv.res1(dx, ti.C, 1.f);
                    if (ti.act) { cv.mask(xin, 0); cv.d.mslope = 0.f; }
                    SG_TRY(cv.run(st));
                } else if (o.k == 1) {
                    if (!acc) SG_HIP(hipMemsetAsync(dx.p, 0, (size_t)B * ti.H * ti.W * ti.cs * P.esz, (hipStream_t)st));
                    Conv cv(dt, 1, 1, 1);
                    cv.in(dy, B, to.H, to.W, to.C).w(wp + o.wd[0]).out(dx, to.H, to.W, ti.C).pad(0, 0).scatter(2, 0, 0, ti.H, ti.W);
                    if (acc) cv.res1(dx, ti.C, 1.f);
                    SG_TRY(cv.run(st));
                } else {
                    for (int q = 0; q < 4; ++q) {
                        const int a = q >> 1, bb = q & 1;
                        const int mh = (ti.H - a + 1) / 2, mw = (ti.W - bb + 1) / 2;
                        const Par2 py = par2(o.k, o.pad, a), px = par2(o.k, o.pad, bb);
                        Conv cv(dt, py.n, px.n, 1);
                        cv.in(dy, B, to.H, to.W, to.C).w(wp + o.wd[q]).out(dx, mh, mw, ti.C).pad(py.lead, px.lead).scatter(2, a, bb, ti.H, ti.W);
                        if (acc) cv.res1(dx, ti.C, 1.f);
                        SG_TRY(cv.run(st));
                    }
                }
                written[o.in] = 1;
            }
        } else if (o.type == 1) {
            void* dres = nullptr; int dres_cs = 0, dres_acc = 0;
            if (o.res >= 0) {       // first contribution to the shortcut's source: plain store, later ones add
                dres = s8 + P.g[o.res]; dres_cs = P.T[o.res].cs; dres_acc = written[o.res]; written[o.res] = 1;
            }
            SG_REQUIRE(!acc && !ti.act, "%s backward: internal error (GroupNorm input has two consumers or a fused activation)", tag);
            const int pacc = o.w >= 0 ? seen_param[o.w] : 0;       // a GroupNorm module applied more than once (edsr.py:41,47,49): gradients add
            if (o.w >= 0) seen_param[o.w] = 1;
            SG_TRY(srcgan_gn_backward(dy.p, to.cs, o.relu ? (w8 + to.off) : nullptr, to.cs, xin.p, ti.cs, o.w >= 0 ? params[o.w] : nullptr,
                                      (const float*)(w8 + o.stats), dx.p, ti.cs, dres, dres_cs, dres_acc, o.w >= 0 ? G(o.w) : nullptr,
                                      o.w >= 0 ? G(o.w + 1) : nullptr, pacc, o.slope, B, (long)ti.H * ti.W, ti.C, o.ngrp, dt, gnscr, st));
            written[o.in] = 1;
        } else if (o.type == 2) {
            if (G(o.w))     // dW[ci][co][a][b] = sum x[y,x,ci] * dy[2y+a,2x+b,co]: wgrad with roles (dy := x, x := dy), k2 s2
                SG_TRY(wgrad_call(dt, xin, ti.H, ti.W, ti.C, dy, B, to.H, to.W, to.C, 2, 2, 2, 0, 0, WLayout{(long)to.C * 4, 4, 2, 1, 0}, 1.f, slab, G(o.w), st));
            SG_REQUIRE(!acc && !ti.act, "%s backward: internal error (deconvolution input)", tag);
            SG_TRY(Conv(dt, 2, 2, 2).in(dy, B, to.H, to.W, to.C).w(wp + o.wd[0]).out(dx, ti.H, ti.W, ti.C).pad(0, 0).run(st));
            written[o.in] = 1;
        } else {
            SG_REQUIRE(!acc && !ti.act, "%s backward: internal error (PixelShuffle input)", tag);
            SG_TRY(srcgan_pixel_shuffle_nhwc(dy.p, to.cs, dx.p, ti.cs, B, ti.H, ti.W, to.C, o.k, 1, dt, st));
            written[o.in] = 1;
        }
    }
    if (dx_nchw) {       // gradient w.r.t. the network input (an end-to-end cascade: trainCas.py:108 feeds one network's output to the next)
        SG_REQUIRE(written[0], "%s backward: internal error (no input gradient was produced)", tag);
        SG_TRY(srcgan_nhwc_to_nchw_f32(s8 + P.g[0], dx_nchw, B, P.in_ch, P.H, P.W, P.in_cs, 0, dt, st));
    }
    return 0;
}
}  // namespace

extern "C" int srcgan_resdeconv_num_params(const srcgan_resdeconv_cfg* c) { RdPlan P; if (rd_plan(c, P)) return -1; return P.nparams; }
extern "C" size_t srcgan_resdeconv_ws_bytes(const srcgan_resdeconv_cfg* c) { RdPlan P; if (rd_plan(c, P)) return 0; return P.total; }
extern "C" size_t srcgan_resdeconv_bwd_scratch_bytes(const srcgan_resdeconv_cfg* c) { RdPlan P; if (rd_plan(c, P)) return 0; return P.bwd_total; }
extern "C" int srcgan_resdeconv_forward(const srcgan_resdeconv_cfg* c, const float* x_nchw, const float* const* params, void* ws, float* y_nchw, void* st) {
    RdPlan P;
    SG_TRY(rd_plan(c, P));
    return rd_forward(P, x_nchw, params, ws, y_nchw, "resdeconv", st);
}
extern "C" int srcgan_resdeconv_backward(const srcgan_resdeconv_cfg* c, const float* dy_nchw, const float* const* params, void* ws, void* scratch,
                                         float* const* grads, float* dx_nchw, void* st) {
    RdPlan P;
    SG_TRY(rd_plan(c, P));
    return rd_backward(P, dy_nchw, dx_nchw, params, ws, scratch, grads, "resdeconv", st);
}

extern "C" int srcgan_srnet_num_params(const srcgan_srnet_cfg* c) { RdPlan P; if (sr_plan(c, P)) return -1; return P.nparams; }
extern "C" size_t srcgan_srnet_ws_bytes(const srcgan_srnet_cfg* c) { RdPlan P; if (sr_plan(c, P)) return 0; return P.total; }
extern "C" size_t srcgan_srnet_bwd_scratch_bytes(const srcgan_srnet_cfg* c) { RdPlan P; if (sr_plan(c, P)) return 0; return P.bwd_total; }
extern "C" int srcgan_srnet_forward(const srcgan_srnet_cfg* c, const float* x_nchw, const float* const* params, void* ws, float* y_nchw, void* st) {
    RdPlan P;
    SG_TRY(sr_plan(c, P));
    return rd_forward(P, x_nchw, params, ws, y_nchw, c->kind == 0 ? "espcn" : c->kind == 1 ? "srcnn" : "edsr", st);
}
extern "C" int srcgan_srnet_backward(const srcgan_srnet_cfg* c, const float* dy_nchw, const float* const* params, void* ws, void* scratch,
                                     float* const* grads, float* dx_nchw, void* st) {
    RdPlan P;
    SG_TRY(sr_plan(c, P));
    return rd_backward(P, dy_nchw, dx_nchw, params, ws, scratch, grads, c->kind == 0 ? "espcn" : c->kind == 1 ? "srcnn" : "edsr", st);
}
