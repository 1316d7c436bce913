// C ABI of libsfamd.so (declared in include/sfamd.h): argument validation, tile selection, launches.
#include "../../include/sfamd.h"
#include "sf_bn.h"
#include "sf_common.h"
#include "sf_igemm.h"
#include "sf_pool.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}
#define REQUIRE(cond, ...) \
    do { if (!(cond)) return fail(__VA_ARGS__); } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

extern "C" int sf_abi_version(void) { return SF_ABI_VERSION; }
// "gfx950" for the hipcc build; the host functional simulator used by the CPU tests reports itself.
extern "C" const char* sf_backend(void) {
#ifdef SF_HOSTSIM
    return "hostsim";
#else
    return "gfx950";
#endif
}
extern "C" const char* sf_last_error(void) { return g_err; }

static int check_desc(const sf_conv_desc* d) {
    REQUIRE(d != nullptr, "conv: null descriptor");
    REQUIRE(d->N > 0 && d->Ci > 0 && d->Co > 0, "conv: empty tensor");
    REQUIRE(d->Ci % 8 == 0 && d->Co % 8 == 0, "conv: channel counts must be multiples of 8 (Ci=%d Co=%d)", d->Ci, d->Co);
    REQUIRE(d->ldx >= d->Ci && d->ldx % 8 == 0 && d->ldy >= d->Co && d->ldy % 8 == 0, "conv: bad row pitch");
    REQUIRE(d->Cw > 0 && d->Cw <= d->Ci, "conv: Cw must be in (0, Ci]");
    REQUIRE(d->sT > 0 && d->sH > 0 && d->sW > 0 && d->dT > 0 && d->dH > 0 && d->dW > 0, "conv: stride/dilation");
    int To = (d->Ti + 2 * d->pT - d->dT * (d->kT - 1) - 1) / d->sT + 1;
    int Ho = (d->Hi + 2 * d->pH - d->dH * (d->kH - 1) - 1) / d->sH + 1;
    int Wo = (d->Wi + 2 * d->pW - d->dW * (d->kW - 1) - 1) / d->sW + 1;
    // fewer outputs than the formula gives = trailing output positions dropped (asymmetric end padding; used by
    // the W-pair-folded stem convolutions); more would read outside the padded input
    REQUIRE(d->To >= 1 && d->Ho >= 1 && d->Wo >= 1 && d->To <= To && d->Ho <= Ho && d->Wo <= Wo,
            "conv: output dims (%d,%d,%d) exceed the geometry's (%d,%d,%d)", d->To, d->Ho, d->Wo, To, Ho, Wo);
    int64_t Mo = (int64_t)d->N * d->To * d->Ho * d->Wo, Mi = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    REQUIRE(Mo < (1ll << 31) && Mi < (1ll << 31), "conv: more than 2^31 positions");
    return 0;
}

static void fill_gather_common(GatherSide& g, const sf_conv_desc* d) {
    g.kT = d->kT; g.kH = d->kH; g.kW = d->kW;
    g.strT = d->sT; g.strH = d->sH; g.strW = d->sW;
    g.padT = d->pT; g.padH = d->pH; g.padW = d->pW;
    g.dilT = d->dT; g.dilH = d->dH; g.dilW = d->dW;
    g.fdkW = make_fastdiv(d->kW);
    g.fdkH = make_fastdiv(d->kH);
    g.fdsT = make_fastdiv(d->sT);
    g.fdsH = make_fastdiv(d->sH);
    g.fdsW = make_fastdiv(d->sW);
    g.scale = nullptr; g.shift = nullptr; g.relu = 0;
    g.fdHW = make_fastdiv((uint32_t)d->Hi * (uint32_t)d->Wi);
}
// 1x1x1 / (kT,1,1) convolutions with unit strides: input and output share (H, W)
static bool is_pointwise(const sf_conv_desc* d) {
    return d->kH == 1 && d->kW == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pH == 0 && d->pW == 0 &&
           d->Ho == d->Hi && d->Wo == d->Wi;
}

// gathered forward input: rows = output positions
static GatherSide gather_fwd(const sf_conv_desc* d, const void* x, const float* sc, const float* sh, int relu) {
    GatherSide g;
    memset(&g, 0, sizeof(g));
    fill_gather_common(g, d);
    g.src = (const f16*)x; g.ld = d->ldx; g.C = d->Ci;
    g.sT = d->Ti; g.sH = d->Hi; g.sW = d->Wi;
    g.mode = 0;
    g.Ktot = d->kT * d->kH * d->kW * d->Ci;
    g.fdC = make_fastdiv(d->Ci);
    g.fdrW = make_fastdiv(d->Wo); g.fdrH = make_fastdiv(d->Ho); g.fdrT = make_fastdiv(d->To);
    g.rowT = d->To;
    g.scale = sc; g.shift = sh; g.relu = relu;
    return g;
}
// gathered output gradient: rows = input positions
static GatherSide gather_dgrad(const sf_conv_desc* d, const void* dy) {
    GatherSide g;
    memset(&g, 0, sizeof(g));
    fill_gather_common(g, d);
    g.src = (const f16*)dy; g.ld = d->ldy; g.C = d->Co;
    g.sT = d->To; g.sH = d->Ho; g.sW = d->Wo;
    g.mode = 1;
    g.Ktot = d->kT * d->kH * d->kW * d->Co;
    g.fdC = make_fastdiv(d->Co);
    g.fdrW = make_fastdiv(d->Wi); g.fdrH = make_fastdiv(d->Hi); g.fdrT = make_fastdiv(d->Ti);
    g.rowT = d->Ti;
    return g;
}

template <int BN, int WM, int WN>
static void launch_igemm(const IgemmParams& p, bool pw, hipStream_t s) {
    int mt = cdiv(p.M, 128);
    dim3 grid((unsigned)(mt * p.ntiles_n));
    if (pw) hipLaunchKernelGGL((sf_igemm_kernel<BN, WM, WN, true>), grid, dim3(SF_THREADS), 0, s, p);
    else hipLaunchKernelGGL((sf_igemm_kernel<BN, WM, WN, false>), grid, dim3(SF_THREADS), 0, s, p);
}

static int run_igemm(IgemmParams& p, bool pw, hipStream_t s) {
    if (p.Nout > 64) { p.ntiles_n = cdiv(p.Nout, 128); launch_igemm<128, 64, 64>(p, pw, s); }
    else if (p.Nout > 32) { p.ntiles_n = 1; launch_igemm<64, 32, 64>(p, pw, s); }
    else if (p.Nout > 16) { p.ntiles_n = 1; launch_igemm<32, 32, 32>(p, pw, s); }
    else { p.ntiles_n = 1; launch_igemm<16, 32, 16>(p, pw, s); }
    return check_launch("igemm");
}

extern "C" int sf_conv_weight_ld(const sf_conv_desc* d, int32_t* ldf, int32_t* ldd) {
    REQUIRE(d && ldf && ldd, "sf_conv_weight_ld: null argument");
    int taps = d->kT * d->kH * d->kW;
    *ldf = roundup(taps * d->Ci, 32);
    *ldd = roundup(taps * d->Co, 32);
    return 0;
}

extern "C" int sf_prep_weights(const sf_conv_desc* d, const float* w, void* wf, void* wd, sf_stream_t stream) {
    if (check_desc(d)) return -1;
    REQUIRE(w && wf, "sf_prep_weights: null pointer");
    PrepParams p;
    p.w = w; p.Co = d->Co; p.Cw = d->Cw; p.Cp = d->Ci; p.taps = d->kT * d->kH * d->kW;
    int32_t ldf, ldd;
    sf_conv_weight_ld(d, &ldf, &ldd);
    p.wf = (f16*)wf; p.ldf = ldf; p.wd = (f16*)wd; p.ldd = ldd;
    int64_t total = (int64_t)p.Co * ldf + (wd ? (int64_t)p.Cp * ldd : 0);
    int blocks = (int)((total + SF_THREADS - 1) / SF_THREADS);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sf_prep_weights_kernel, dim3(blocks), dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("prep_weights");
}

extern "C" int sf_conv_fwd_mtiles(const sf_conv_desc* d) {
    if (!d) return fail("sf_conv_fwd_mtiles: null descriptor");
    return cdiv((int64_t)d->N * d->To * d->Ho * d->Wo, 128);
}

extern "C" int sf_conv_fwd(const sf_conv_desc* d, const void* x, const void* wf, const float* in_scale,
                           const float* in_shift, int in_relu, const float* bias, void* y, float* stat_part,
                           sf_stream_t stream) {
    if (check_desc(d)) return -1;
    REQUIRE(x && wf && y, "sf_conv_fwd: null pointer");
    REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sf_conv_fwd: in_scale/in_shift must come together");
    REQUIRE(!in_scale || d->Ci <= 512, "sf_conv_fwd: fused input BatchNorm supports Ci <= 512 (got %d)", d->Ci);
    REQUIRE(!(bias && stat_part), "sf_conv_fwd: bias together with BatchNorm statistics is not supported");
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.g = gather_fwd(d, x, in_scale, in_shift, in_relu);
    p.M = d->N * d->To * d->Ho * d->Wo;
    int32_t ldf, ldd;
    sf_conv_weight_ld(d, &ldf, &ldd);
    p.wmat = (const f16*)wf; p.ldw = ldf; p.Nout = d->Co;
    p.ksteps = cdiv(p.g.Ktot, 32);
    p.y = (f16*)y; p.ldy = d->ldy;
    p.bias = bias; p.resid = nullptr; p.ldr = 0;
    p.stat_part = stat_part;
    return run_igemm(p, is_pointwise(d), (hipStream_t)stream);
}

extern "C" int sf_conv_dgrad(const sf_conv_desc* d, const void* dy, const void* wd, const void* resid, int32_t ldr,
                             void* dx, sf_stream_t stream) {
    if (check_desc(d)) return -1;
    REQUIRE(dy && wd && dx, "sf_conv_dgrad: null pointer");
    REQUIRE(!resid || (ldr >= d->Ci && ldr % 8 == 0), "sf_conv_dgrad: bad residual pitch");
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.g = gather_dgrad(d, dy);
    p.M = d->N * d->Ti * d->Hi * d->Wi;
    int32_t ldf, ldd;
    sf_conv_weight_ld(d, &ldf, &ldd);
    p.wmat = (const f16*)wd; p.ldw = ldd; p.Nout = d->Ci;
    p.ksteps = cdiv(p.g.Ktot, 32);
    p.y = (f16*)dx; p.ldy = d->ldx;
    p.resid = (const f16*)resid; p.ldr = ldr;
    return run_igemm(p, is_pointwise(d), (hipStream_t)stream);
}

template <int BMW, int WM, int WN, int KS>
static void launch_wgrad(const WgradParams& p, dim3 grid, bool scalar, hipStream_t s) {
    if (scalar) hipLaunchKernelGGL((sf_wgrad_kernel<BMW, WM, WN, KS, false>), grid, dim3(SF_THREADS), 0, s, p);
    else hipLaunchKernelGGL((sf_wgrad_kernel<BMW, WM, WN, KS, true>), grid, dim3(SF_THREADS), 0, s, p);
}

// Split-K plan of the weight gradient: the reduction over the M = N*To*Ho*Wo positions is cut into `splits`
// slabs so that ~4 workgroups per CU are in flight; each split stores its [Co_pad][Kpad] fp32 partial tile
// set with plain stores and sf_wgrad_reduce_kernel sums them (deterministic, no atomics).
struct WgradPlan {
    int BMW, KS, tiles_k, tiles_c, Co_pad, Kpad, nchunks, chunks_per_split, splits;
    size_t ws_bytes;
};
static WgradPlan plan_wgrad(const sf_conv_desc* d) {
    WgradPlan w;
    const int taps = d->kT * d->kH * d->kW;
    const int Ktot = taps * d->Ci;
    const int64_t M = (int64_t)d->N * d->To * d->Ho * d->Wo;
    w.BMW = d->Co >= 128 ? 128 : d->Co >= 64 ? 64 : d->Co >= 32 ? 32 : 16;
    w.KS = w.BMW <= 32 ? 4 : 1;                          // 32-position chunks per pipeline stage
    w.tiles_k = cdiv(Ktot, 128);
    w.tiles_c = cdiv(d->Co, w.BMW);
    w.Kpad = w.tiles_k * 128;
    w.Co_pad = w.tiles_c * w.BMW;
    w.nchunks = cdiv(M, 32);
    const int64_t slab = (int64_t)w.Co_pad * w.Kpad * 4;
    int splits = cdiv(1024, (int64_t)w.tiles_k * w.tiles_c);
    const int64_t cap = (256ll << 20) / slab;            // keep the workspace <= 256 MiB
    if (splits > cap) splits = (int)(cap < 1 ? 1 : cap);
    const int nstages = cdiv(w.nchunks, w.KS);
    if (splits > nstages) splits = nstages;
    if (splits < 1) splits = 1;
    w.chunks_per_split = cdiv(nstages, splits) * w.KS;   // whole stages per split
    w.splits = cdiv(w.nchunks, w.chunks_per_split);
    w.ws_bytes = (size_t)slab * w.splits;
    return w;
}

extern "C" int64_t sf_conv_wgrad_workspace(const sf_conv_desc* d) {
    if (check_desc(d)) return -1;
    return (int64_t)plan_wgrad(d).ws_bytes;
}

extern "C" int sf_conv_wgrad(const sf_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                             int in_relu, const void* dy, float* dw, float out_scale, int zero_first,
                             void* workspace, int64_t workspace_bytes, sf_stream_t stream) {
    if (check_desc(d)) return -1;
    REQUIRE(x && dy && dw && workspace, "sf_conv_wgrad: null pointer");
    REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sf_conv_wgrad: in_scale/in_shift must come together");
    REQUIRE(!in_scale || d->Ci <= 512, "sf_conv_wgrad: fused input BatchNorm supports Ci <= 512 (got %d)", d->Ci);
    static const bool scalar = getenv("SF_WGRAD_SCALAR") && atoi(getenv("SF_WGRAD_SCALAR")) != 0;
    hipStream_t s = (hipStream_t)stream;
    const WgradPlan w = plan_wgrad(d);
    REQUIRE(workspace_bytes >= (int64_t)w.ws_bytes, "sf_conv_wgrad: workspace too small (%lld < %lld bytes)",
            (long long)workspace_bytes, (long long)w.ws_bytes);
    REQUIRE(w.splits <= 65535, "sf_conv_wgrad: too many splits");
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.g = gather_fwd(d, x, in_scale, in_shift, in_relu);
    p.dy = (const f16*)dy; p.ldy = d->ldy; p.Co = d->Co;
    p.M = d->N * d->To * d->Ho * d->Wo;
    p.ws = (float*)workspace; p.Co_pad = w.Co_pad; p.Kpad = w.Kpad;
    p.nchunks = w.nchunks; p.chunks_per_split = w.chunks_per_split;
    dim3 grid(w.tiles_k, w.tiles_c, w.splits);
    switch (w.BMW) {
        case 128: launch_wgrad<128, 64, 64, 1>(p, grid, scalar, s); break;
        case 64: launch_wgrad<64, 32, 64, 1>(p, grid, scalar, s); break;
        case 32: launch_wgrad<32, 32, 32, 4>(p, grid, scalar, s); break;
        default: launch_wgrad<16, 16, 32, 4>(p, grid, scalar, s); break;
    }
    if (check_launch("wgrad")) return -1;
    WgradReduceParams r;
    r.ws = (const float*)workspace; r.splits = w.splits; r.Co = d->Co; r.Co_pad = w.Co_pad; r.Kpad = w.Kpad;
    r.Ktot = p.g.Ktot; r.fdC = p.g.fdC; r.dw = dw; r.Cw = d->Cw; r.taps = d->kT * d->kH * d->kW;
    r.out_scale = out_scale; r.accumulate = zero_first ? 0 : 1;
    int64_t total = (int64_t)d->Co * w.Kpad;
    int lanes = 1;
    while (lanes < 32 && lanes * 4 <= w.splits) lanes *= 2;   // ~>= 4 splits per lane, 8..256 elements per block
    r.lanes = lanes;
    const int per_block = SF_THREADS / lanes;
    hipLaunchKernelGGL(sf_wgrad_reduce_kernel, dim3(cdiv(total, per_block)), dim3(SF_THREADS), 0, s, r);
    return check_launch("wgrad_reduce");
}

// ------------------------------------------------------------------------------------------------
static RowTile make_rowtile(int64_t M, int C, int max_blocks, dim3& grid) {
    RowTile rt;
    rt.M = (int)M; rt.C = C;
    const int G = C / 8;
    const int TG = G < SF_THREADS ? G : SF_THREADS;
    const int rpi = SF_THREADS / TG;
    int passes = cdiv(M, (int64_t)rpi * max_blocks);
    if (passes < 1) passes = 1;
    rt.rows_per_block = rpi * passes;
    grid = dim3(cdiv(M, rt.rows_per_block), cdiv(G, SF_THREADS));
    return rt;
}
static int check_rows(const char* who, int64_t M, int C) {
    REQUIRE(M > 0 && M < (1ll << 31), "%s: bad row count", who);
    REQUIRE(C > 0 && C % 8 == 0, "%s: C must be a positive multiple of 8 (got %d)", who, C);
    return 0;
}

// Folds a long partial table in place (sf_part_fold_kernel) so that the single-workgroup-per-32-channels
// finalize kernels never walk more than a few hundred rows; returns the row stride of the surviving rows.
static int fold_partials(float* part, int& nblk, int C, hipStream_t s) {
    if (nblk <= 256) return 1;
    const int group = nblk <= 2048 ? 16 : nblk <= 8192 ? 32 : 64;
    const int cols = 2 * C;
    dim3 grid(cdiv(cols, SF_THREADS), cdiv(nblk, group));
    hipLaunchKernelGGL(sf_part_fold_kernel, grid, dim3(SF_THREADS), 0, s, part, nblk, cols, group);
    nblk = cdiv(nblk, group);
    return group;
}

extern "C" int sf_bn_finalize(float* part, int32_t nblk, int32_t C, float count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              float* scale, float* shift, float* save_mean, float* save_rstd, sf_stream_t stream) {
    REQUIRE(gamma && beta && scale && shift, "sf_bn_finalize: null pointer");
    REQUIRE(nblk > 0 ? part != nullptr : (running_mean && running_var), "sf_bn_finalize: missing statistics source");
    BnFinalizeParams p;
    p.row_stride = nblk > 0 ? fold_partials(part, nblk, C, (hipStream_t)stream) : 1;
    p.part = part; p.nblk = nblk; p.C = C; p.count = count; p.gamma = gamma; p.beta = beta;
    p.running_mean = running_mean; p.running_var = running_var; p.momentum = momentum; p.eps = eps;
    p.scale = scale; p.shift = shift; p.save_mean = save_mean; p.save_rstd = save_rstd;
    hipLaunchKernelGGL(sf_bn_finalize_kernel, dim3(cdiv(C, 32)), dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("bn_finalize");
}

extern "C" int sf_bn_act(int64_t M, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                         const void* r, int32_t ldr, const float* rscale, const float* rshift, int relu, void* out,
                         int32_t ldo, sf_stream_t stream) {
    if (check_rows("sf_bn_act", M, C)) return -1;
    REQUIRE(y && out, "sf_bn_act: null pointer");
    BnActParams p;
    dim3 grid;
    p.rt = make_rowtile(M, C, 8192, grid);
    p.y = (const f16*)y; p.ldy = ldy; p.scale = scale; p.shift = shift;
    p.r = (const f16*)r; p.ldr = ldr; p.rscale = rscale; p.rshift = rshift;
    p.relu = relu; p.out = (f16*)out; p.ldo = ldo;
    hipLaunchKernelGGL(sf_bn_act_kernel, grid, dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("bn_act");
}

static const int kBwdBlocks = 1024;
extern "C" int sf_bn_bwd_blocks(int64_t M, int32_t C) {
    if (check_rows("sf_bn_bwd_blocks", M, C)) return -1;
    dim3 grid;
    make_rowtile(M, C, kBwdBlocks, grid);
    return (int)grid.x;
}

extern "C" int sf_bn_bwd_reduce(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* zmask, int32_t ldm,
                                const void* y, int32_t ldy, const float* scale, const float* shift, int relu_self,
                                float* part, sf_stream_t stream) {
    if (check_rows("sf_bn_bwd_reduce", M, C)) return -1;
    REQUIRE(dz && y && part, "sf_bn_bwd_reduce: null pointer");
    REQUIRE(!relu_self || (scale && shift), "sf_bn_bwd_reduce: relu_self needs scale/shift");
    BnBwdReduceParams p;
    dim3 grid;
    p.rt = make_rowtile(M, C, kBwdBlocks, grid);
    p.dz = (const f16*)dz; p.lddz = lddz; p.zmask = (const f16*)zmask; p.ldm = ldm;
    p.y = (const f16*)y; p.ldy = ldy; p.scale = scale; p.shift = shift; p.relu_self = relu_self; p.part = part;
    hipLaunchKernelGGL(sf_bn_bwd_reduce_kernel, grid, dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("bn_bwd_reduce");
}

extern "C" int sf_bn_bwd_finalize(float* part, int32_t nblk, int32_t C, float count, const float* gamma,
                                  const float* mean, const float* rstd, float inv_loss_scale, float* dgamma,
                                  float* dbeta, int accumulate, float* coef, sf_stream_t stream) {
    REQUIRE(part && gamma && mean && rstd && dgamma && dbeta && coef, "sf_bn_bwd_finalize: null pointer");
    BnBwdFinalizeParams p;
    p.row_stride = fold_partials(part, nblk, C, (hipStream_t)stream);
    p.part = part; p.nblk = nblk; p.C = C; p.count = count; p.gamma = gamma; p.mean = mean; p.rstd = rstd;
    p.inv_loss_scale = inv_loss_scale; p.dgamma = dgamma; p.dbeta = dbeta; p.accumulate = accumulate; p.coef = coef;
    hipLaunchKernelGGL(sf_bn_bwd_finalize_kernel, dim3(cdiv(C, 32)), dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("bn_bwd_finalize");
}

extern "C" int sf_bn_bwd_apply(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* zmask, int32_t ldm,
                               const void* y, int32_t ldy, const float* scale, const float* shift, int relu_self,
                               const float* coef, void* dy, int32_t lddy, void* gout, int32_t ldg,
                               sf_stream_t stream) {
    if (check_rows("sf_bn_bwd_apply", M, C)) return -1;
    REQUIRE(dz && y && coef && dy, "sf_bn_bwd_apply: null pointer");
    REQUIRE(!relu_self || (scale && shift), "sf_bn_bwd_apply: relu_self needs scale/shift");
    BnBwdApplyParams p;
    dim3 grid;
    p.rt = make_rowtile(M, C, 8192, grid);
    p.dz = (const f16*)dz; p.lddz = lddz; p.zmask = (const f16*)zmask; p.ldm = ldm;
    p.y = (const f16*)y; p.ldy = ldy; p.scale = scale; p.shift = shift; p.relu_self = relu_self;
    p.coef = coef; p.dy = (f16*)dy; p.lddy = lddy; p.gout = (f16*)gout; p.ldg = ldg;
    hipLaunchKernelGGL(sf_bn_bwd_apply_kernel, grid, dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("bn_bwd_apply");
}

// ------------------------------------------------------------------------------------------------
static int fill_pool(PoolParams& p, int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW,
                     int32_t sH, int32_t sW, int32_t pH, int32_t pW, const void* y, int32_t ldy, const float* scale,
                     const float* shift, int relu) {
    REQUIRE(C > 0 && C % 8 == 0, "pool: C must be a multiple of 8");
    REQUIRE((scale == nullptr) == (shift == nullptr), "pool: scale/shift must come together");
    memset(&p, 0, sizeof(p));
    p.y = (const f16*)y; p.ldy = ldy; p.scale = scale; p.shift = shift; p.relu = relu;
    p.N = N; p.T = T; p.H = H; p.W = W; p.C = C;
    p.kH = kH; p.kW = kW; p.sH = sH; p.sW = sW; p.pH = pH; p.pW = pW;
    p.Ho = (H + 2 * pH - kH) / sH + 1;
    p.Wo = (W + 2 * pW - kW) / sW + 1;
    p.fdG = make_fastdiv(C / 8);
    return 0;
}
static int pool_grid(int64_t total) {
    int64_t b = (total + SF_THREADS - 1) / SF_THREADS;
    return (int)(b > 65536 ? 65536 : b);
}

extern "C" int sf_pool_fwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW, int32_t sH,
                           int32_t sW, int32_t pH, int32_t pW, const void* y, int32_t ldy, const float* scale,
                           const float* shift, int relu, void* out, int32_t ldo, void* argmax, sf_stream_t stream) {
    PoolParams p;
    if (fill_pool(p, N, T, H, W, C, kH, kW, sH, sW, pH, pW, y, ldy, scale, shift, relu)) return -1;
    REQUIRE(y && out, "sf_pool_fwd: null pointer");
    REQUIRE(kH * kW <= 255, "sf_pool_fwd: window too large for the byte argmax");
    p.out = (f16*)out; p.ldo = ldo; p.argmax = (uint8_t*)argmax;
    p.fdW = make_fastdiv(p.Wo); p.fdH = make_fastdiv(p.Ho);
    p.total = (int64_t)N * T * p.Ho * p.Wo * (C / 8);
    REQUIRE(p.total < (1ll << 31), "sf_pool_fwd: too many elements");
    hipLaunchKernelGGL(sf_pool_fwd_kernel, dim3(pool_grid(p.total)), dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("pool_fwd");
}

extern "C" int sf_pool_bwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW, int32_t sH,
                           int32_t sW, int32_t pH, int32_t pW, const void* pooled, int32_t ldp, const void* argmax,
                           int relu, const void* dout, int32_t lddo, void* g, int32_t ldg, sf_stream_t stream) {
    PoolParams p;
    if (fill_pool(p, N, T, H, W, C, kH, kW, sH, sW, pH, pW, nullptr, 0, nullptr, nullptr, relu)) return -1;
    REQUIRE(pooled && argmax && dout && g, "sf_pool_bwd: null pointer");
    p.out = (f16*)g; p.ldo = ldg; p.dout = (const f16*)dout; p.lddo = lddo;
    p.pooled = (const f16*)pooled; p.ldp = ldp; p.argmax = (uint8_t*)argmax;
    p.fdW = make_fastdiv(W); p.fdH = make_fastdiv(H);
    p.total = (int64_t)N * T * H * W * (C / 8);
    REQUIRE(p.total < (1ll << 31), "sf_pool_bwd: too many elements");
    hipLaunchKernelGGL(sf_pool_bwd_kernel, dim3(pool_grid(p.total)), dim3(SF_THREADS), 0, (hipStream_t)stream, p);
    return check_launch("pool_bwd");
}

extern "C" int sf_ncthw_to_cl(const float* x, int32_t N, int32_t C, int64_t S, int32_t Cp, void* out,
                              sf_stream_t stream) {
    REQUIRE(x && out && (Cp % 8 == 0 || Cp == 4) && Cp >= C, "sf_ncthw_to_cl: bad arguments");
    hipLaunchKernelGGL(sf_ncthw_to_cl_kernel, dim3(pool_grid((int64_t)N * S)), dim3(SF_THREADS), 0,
                       (hipStream_t)stream, x, (f16*)out, N, C, S, Cp);
    return check_launch("ncthw_to_cl");
}

extern "C" int sf_cl_to_ncthw(const void* x, int32_t ld, int32_t N, int32_t C, int64_t S, float* out,
                              sf_stream_t stream) {
    REQUIRE(x && out && ld >= C, "sf_cl_to_ncthw: bad arguments");
    hipLaunchKernelGGL(sf_cl_to_ncthw_kernel, dim3(pool_grid((int64_t)N * C * S)), dim3(SF_THREADS), 0,
                       (hipStream_t)stream, (const f16*)x, ld, out, N, C, S);
    return check_launch("cl_to_ncthw");
}
