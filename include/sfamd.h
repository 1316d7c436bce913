/* sfamd.h -- C ABI of libsfamd.so, the MI355X (gfx950) video-backbone forward/backward engine.
 *
 * The reference (facebookresearch/SlowFast) has no native layer: every FLOP of its hot path is a
 * torch.nn call inside the slowfast/models package.  Each entry point below therefore cites the torch
 * call site(s) it replaces.  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted;
 *   - activations are fp16, channels-last: a logical (N,C,T,H,W) tensor stored N,T,H,W,C with a row
 *     pitch `ld` (elements) >= C so channel-slice views need no copy; C % 8 == 0, 16-byte aligned;
 *   - statistics, affine parameters, weights and weight gradients are fp32;
 *   - the callee allocates nothing and never synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*); return 0 on success, negative on error (sf_last_error()).
 */
#ifndef SFAMD_H
#define SFAMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SF_ABI_VERSION 23
typedef void* sf_stream_t;

/* Geometry of one nn.Conv3d (groups == 1).  Ci is the channel count of the activation buffer
 * (the stem's 3-channel clip is zero-padded to 8), Cw the channel count of the fp32 weight. */
typedef struct sf_conv_desc {
    int32_t N, Ci, Ti, Hi, Wi;
    int32_t Co, To, Ho, Wo; /* To/Ho/Wo may be smaller than the conv formula: trailing outputs are dropped */
    int32_t kT, kH, kW;
    int32_t sT, sH, sW;
    int32_t pT, pH, pW;
    int32_t dT, dH, dW;
    int32_t Cw;
    int32_t ldx, ldy; /* row pitch of the input / output activation buffers, in elements */
    int32_t Cow;      /* channel count of the fp32 weight on the OUTPUT side (0 = Co): activation buffers hold Co =
                         Cow rounded up to 8 channels, the pad channels are produced as exact zeros (X3D widths 54, 108) */
} sf_conv_desc;

int sf_abi_version(void);
const char* sf_backend(void);     /* "gfx950" (product) or "hostsim" (CPU test build of the same sources) */
/* The 16-bit storage type this build of the library keeps activations / packed weights in and feeds the MFMAs with (every
 * `void*` activation or packed-weight argument below points at elements of THAT type; "fp16" in the comments of this header
 * means it).  The reference runs its mixed-precision path under torch.cuda.amp.autocast (tools/train_net.py:101-118), which
 * admits float16 and bfloat16; libsfamd.so is the float16 build, libsfamd_bf16.so (same sources, -DSF_ACT_BF16) the bfloat16
 * one.  Accumulation, statistics and parameter gradients are fp32 in both. */
#define SF_ACT_FP16 0
#define SF_ACT_BF16_ID 1
int sf_act_dtype(void);
const char* sf_last_error(void); /* host string, thread-local */
/* Build provenance: the first 16 hex digits of the sha256 over the kernel sources and this header the binary was compiled
 * from (slowfast_amd/build_ext.py:source_id passes it as -DSF_BUILD_ID).  The Python binding recomputes it from the sources
 * beside the binary and refuses a library built from anything else. */
const char* sf_build_id(void);

/* ---- Conv3d -- replaces nn.Conv3d at resnet_helper.py:331-369 (BottleneckTransform a/b/c),
 * resnet_helper.py:485-493 (ResBlock.branch1), stem_helper.py:182-189 (ResNetBasicStem.conv),
 * video_model_builder.py:147-154 (FuseFastToSlow.conv_f2s), and their autograd backward. */
int sf_conv_weight_ld(const sf_conv_desc* d, int32_t* ldf, int32_t* ldd);
/* fp32 [Co][Cw][kT][kH][kW] -> fp16 forward operand wf[Co][ldf] and (optional) dgrad operand wd[Ci][ldd] */
int sf_prep_weights(const sf_conv_desc* d, const float* w, void* wf, void* wd, sf_stream_t stream);
/* The same packing for MANY weights in one launch (a training step re-packs every layer after the optimizer update: one
 * launch instead of one per nn.Conv3d / nn.Linear).  `items` is a DEVICE array the caller uploads once; sf_prep_item_fill
 * writes the host copy of one entry from a descriptor (an nn.Linear [N][K] is the 1x1x1 case Co=N, Ci=Cw=K: wf = fp16
 * weight, wd = its transpose).  An item is cut into sf_prep_item_blocks(item) bricks; blk_item / blk_off (device, int32):
 * workgroup b packs brick blk_off[b] (0 .. blocks-1) of items[blk_item[b]]. */
typedef struct sf_prep_item {
    const float* w;   /* fp32 [Cow][Cw][taps] */
    void* wf;         /* fp16 [Co][ldf] */
    void* wd;         /* fp16 [Cp][ldd], or NULL */
    int32_t Co, Cow, Cw, Cp, taps, ldf, ldd;
    int32_t pad;      /* brick shape, written by sf_prep_item_fill */
} sf_prep_item;
int sf_prep_item_fill(const sf_conv_desc* d, const float* w, void* wf, void* wd, sf_prep_item* item);
int64_t sf_prep_item_blocks(const sf_prep_item* item);
int sf_prep_weights_batch(const sf_prep_item* items, const int32_t* blk_item, const int32_t* blk_off, int32_t nblocks,
                          sf_stream_t stream);
/* number of 128-row tiles = rows of `stat_part` */
int sf_conv_fwd_mtiles(const sf_conv_desc* d);
/* y = conv(act(x)), act(x) = x or relu?(x*in_scale + in_shift) applied on the fly (zero padding is
 * applied AFTER act, as in the reference where the padded tensor is the post-ReLU activation).
 * stat_part (optional) receives per-tile per-channel sum / sum of squares of y (bias included): [mtiles][2][Co] fp32. */
int sf_conv_fwd(const sf_conv_desc* d, const void* x, const void* wf, const float* in_scale, const float* in_shift,
                int in_relu, const float* bias, void* y, float* stat_part, sf_stream_t stream);
/* Inference-fused convolution (the eval / multi-view test path, tools/test_net.py:25-151):
 * eval-mode nn.BatchNorm3d is an affine map of its input, so the caller folds it into the weights (wf = pack(w * scale))
 * and a bias (shift); the nn.ReLU and the residual addition of resnet_helper.py:377-392, 512-521 run in the epilogue:
 *   y = act(conv(x) + bias [+ resid]),   act = ReLU when out_relu.   bias [Co] fp32 (may be NULL), resid [M][ldr] fp16. */
int sf_conv_fwd_fused(const sf_conv_desc* d, const void* x, const void* wf, const float* bias, const void* resid,
                      int32_t ldr, int out_relu, void* y, sf_stream_t stream);
/* dx = conv_transpose(dy, w) (+ resid), dx pitch = d->ldx, dy pitch = d->ldy.  resid_bits (optional, with resid): the
 * 1-bit ReLU mask sf_bn_act wrote for the block input ([positions][Ci/8] bytes): only residual elements whose bit is set
 * are added -- the masked block-output gradient of the identity shortcut (resnet_helper.py:512-521 backward) without
 * materialising it. */
int sf_conv_dgrad(const sf_conv_desc* d, const void* dy, const void* wd, const void* resid, int32_t ldr,
                  const void* resid_bits, void* dx, sf_stream_t stream);
/* sf_conv_dgrad + the reduction pass of the BatchNorm backward(s) that consume dx (round 3).  The gradient dx this call
 * stores is dz of a BatchNorm-ReLU whose backward first needs the per-channel sums of g and g * y over the positions
 * (sf_bn_bwd_reduce), g = dz under the ReLU mask.  The epilogue takes them from the fp16 tile it stores -- one read of the y
 * tile instead of a pass over dz and y, one launch less:
 *   - inner activations of a block (BottleneckTransform b / c reading relu(a_bn(a(x))) / relu(b_bn(b(.))),
 *     resnet_helper.py:377-392): mask = (bn_y * mask_scale + mask_shift > 0), mask_bits NULL;
 *   - a block INPUT that is the previous block's output relu(bn_c(yc) + shortcut) (resnet_helper.py:512-521): mask_bits = the
 *     1-bit image sf_bn_act wrote for that output ([positions][Ci/8]), bn_y = yc (a projection shortcut's own BatchNorm keeps
 *     its separate reduction pass).
 * bn_part[row][2][Ci] fp32, one row per M tile of the kernel that ran (fixed order: deterministic); *bn_rows = rows written,
 * to be passed to sf_bn_bwd_finalize as nblk -- or 0 when the geometry keeps the separate pass (strided data gradients): the
 * table is then untouched and the caller runs sf_bn_bwd_reduce.  The table holds ceil(positions / 128) rows (bn_part_rows);
 * bn_rows is a HOST pointer. */
int sf_conv_dgrad_bn(const sf_conv_desc* d, const void* dy, const void* wd, const void* resid, int32_t ldr,
                     const void* resid_bits, void* dx, const float* mask_scale, const float* mask_shift,
                     const void* mask_bits, const void* bn_y, int32_t bn_ldy, float* bn_part, int32_t bn_part_rows,
                     int32_t* bn_rows, sf_stream_t stream);
/* dw[Co][Cw][taps] = (zero_first ? 0 : dw) + out_scale * sum_m dy[m] (x) act(x)[m].  The reduction over positions
 * is split; `workspace` (>= sf_conv_wgrad_workspace(d) bytes, caller-owned) holds the per-split partials, which a
 * second kernel sums in a fixed order (no atomics: results are run-to-run reproducible). */
int64_t sf_conv_wgrad_workspace(const sf_conv_desc* d);
/* `rowtab` (optional): the per-output-row table {first input position, tap-validity mask} of the large-K kernel.  It
 * depends on the geometry only: build it once with sf_conv_wgrad_rowtab into sf_conv_wgrad_rowtab_bytes(d) bytes (0: this
 * geometry takes another kernel, pass NULL) and hand it to every call; NULL rebuilds it in the workspace per call. */
int64_t sf_conv_wgrad_rowtab_bytes(const sf_conv_desc* d);
int sf_conv_wgrad_rowtab(const sf_conv_desc* d, void* tab, sf_stream_t stream);
int sf_conv_wgrad(const sf_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, int in_relu,
                  const void* dy, float* dw, float out_scale, int zero_first, void* workspace, int64_t workspace_bytes,
                  const void* rowtab, sf_stream_t stream);

/* ---- BatchNorm3d -- replaces nn.BatchNorm3d built by batchnorm_helper.py:16-37 (get_norm) at every
 * *_bn call site of resnet_helper.py / stem_helper.py / video_model_builder.py:155-159, plus the
 * nn.ReLU and the residual add of resnet_helper.py:512-521. */
/* Creal <= C: number of real channels (length of gamma/beta/running stats); channels [Creal, C) are zero padding of the
 * activation buffer and get scale = shift = 0.
 * `part` is scratch: long tables are folded in place before the final reduction (contents are destroyed).
 * nblk > 0: training (partials -> batch statistics, running stats updated when non-null);
 * nblk == 0: eval (running statistics).  Outputs scale = gamma*rstd, shift = beta - mean*scale. */
int sf_bn_finalize(float* part, int32_t nblk, int32_t C, int32_t Creal, float count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                   float* save_mean, float* save_rstd, sf_stream_t stream);
/* out = relu?( y*scale+shift [+ r*rscale+rshift | + r] ); scale == NULL means identity.  mask_out (optional): [M][C/8]
 * bytes, bit e of byte (m, c/8) = out[m][c+e] > 0 -- the ReLU mask the backward kernels read instead of `out` (1/16 of
 * the bytes). */
int sf_bn_act(int64_t M, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift, const void* r,
              int32_t ldr, const float* rscale, const float* rshift, int relu, void* out, int32_t ldo, void* mask_out,
              sf_stream_t stream);
int sf_bn_bwd_blocks(int64_t M, int32_t C); /* rows of `part` for the two calls below */
/* part[blk][0][c] = sum g, part[blk][1][c] = sum g*y, g = dz masked by (zmask > 0) or by
 * (y*scale+shift > 0) when relu_self.  ldm == 0 with a non-NULL zmask: zmask is the BIT mask written by sf_bn_act
 * ([M][C/8] bytes); same convention in sf_bn_bwd_apply. */
int sf_bn_bwd_reduce(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* zmask, int32_t ldm, const void* y,
                     int32_t ldy, const float* scale, const float* shift, int relu_self, float* part,
                     sf_stream_t stream);
/* dgamma/dbeta (fp32, unscaled by inv_loss_scale) and coef[3][C] with dy = k1*g + k2 + k3*y */
int sf_bn_bwd_finalize(float* part, int32_t nblk, int32_t C, int32_t Creal, float count, const float* gamma, const float* mean,
                       const float* rstd, float inv_loss_scale, float* dgamma, float* dbeta, int accumulate,
                       float* coef, sf_stream_t stream);
int sf_bn_bwd_apply(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* zmask, int32_t ldm, const void* y,
                    int32_t ldy, const float* scale, const float* shift, int relu_self, const float* coef, void* dy,
                    int32_t lddy, void* gout, int32_t ldg, sf_stream_t stream);

/* ---- nn.MaxPool3d([1,kH,kW],[1,sH,sW],[0,pH,pW]) fused with the producer's BN(+ReLU) --
 * stem_helper.py:190-201 (bn -> relu -> pool_layer). */
int sf_pool_fwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW, int32_t sH, int32_t sW,
                int32_t pH, int32_t pW, const void* y, int32_t ldy, const float* scale, const float* shift, int relu,
                void* out, int32_t ldo, void* argmax, int32_t cls, sf_stream_t stream);
/* g[N,T,H,W,C] = gradient w.r.t. the BatchNorm output (pool + ReLU backward) from the forward's byte argmax table
 * [N,T,Ho,Wo][C] (window-local index kh*kW+kw of the first maximum, as recorded by torch's max_pool3d; with relu, 0xFF for a
 * window whose maximum is not positive -- the ReLU mask `pooled > 0` -- written by sf_pool_fwd(relu = 1), kH*kW <= 255).
 * `pooled` / `ldp` are kept in the signature and not read (round 6); `relu` must be the forward's. */
int sf_pool_bwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW, int32_t sH, int32_t sW,
                int32_t pH, int32_t pW, const void* pooled, int32_t ldp, const void* argmax, int relu,
                const void* dout, int32_t lddo, void* g, int32_t ldg, int32_t cls, sf_stream_t stream);
/* cls != 0 (both calls): TOKEN tensors -- every sample's T*H*W rows are preceded by one cls-token row that passes
 * through the pool unchanged; replaces attention_pool(x, pool_skip=nn.MaxPool3d) of attention.py:13-45, 485-498. */

/* ---- nn.MaxPool3d(pool_size, stride = pool_size, padding 0) in front of conv_phi / conv_g of the Nonlocal block
 * (nonlocal_helper.py:96-114); argmax: byte table [N,To,Ho,Wo][C] */
int sf_pool3d_fwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kT, int32_t kH, int32_t kW, const void* x,
                  int32_t ldx, void* out, int32_t ldo, void* argmax, sf_stream_t stream);
int sf_pool3d_bwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kT, int32_t kH, int32_t kW,
                  const void* argmax, const void* dout, int32_t lddo, void* dx, int32_t lddx, sf_stream_t stream);

/* ---- layout: clips arrive NCTHW fp32 (tools/train_net.py:79-98) */
int sf_ncthw_to_cl(const float* x, int32_t N, int32_t C, int64_t S, int32_t Cp, void* out, sf_stream_t stream);
int sf_cl_to_ncthw(const void* x, int32_t ld, int32_t N, int32_t C, int64_t S, float* out, sf_stream_t stream);

/* =====================================================================================================
 * Token-space entry points (MViT pooled attention, Mlp; also used by Nonlocal and X3D).  Token tensors are fp16
 * [rows][C] with a row pitch in elements (multiple of 8); parameters, statistics and their gradients are fp32.
 * ===================================================================================================== */

/* Batched GEMM  Y[z][m][n] = sum_k A[z][m][k] * W[z][n][k] (+ bias[n]) (+ resid[z][m][n]),  z = b*bh + j,
 * operand z at base + b*s?_b + j*s?_h (elements).  nbatch = 1, bh = 1 is a plain nn.Linear forward
 * (y = x W^T + b: attention.py:193,195,482; common.py:19-21) or its data gradient (W = weight^T operand).
 * In attention: scores = q k^T (attention.py:355) and attn @ v (:379) per (batch, head). */
int sf_bgemm(int64_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias,
             const void* resid, int32_t ldr, void* Y, int32_t ldy, int32_t nbatch, int32_t bh, int64_t sa_b, int64_t sa_h,
             int64_t sw_b, int64_t sw_h, int64_t sy_b, int64_t sy_h, int64_t sr_b, int64_t sr_h, int32_t resid_row0,
             float alpha, sf_stream_t stream);
/* alpha scales the products before bias / residual (0 = 1): the 1/N normalisation of the Nonlocal "dot_product"
 * affinity (nonlocal_helper.py:130-132). */
/* resid_row0: the residual is added to rows m >= resid_row0 only (residual pooling skips the cls row,
 * attention.py:381-385).  N need not be a multiple of 8 when ldy covers N rounded up to 8 (pad columns <- 0). */
/* fp32 side rows of a token residual stream (round 4).  MultiScaleBlock adds two branch outputs per block to the stream
 * (attention.py:500-510: x = x_res + drop_path(x_block); x = x + drop_path(x_mlp)); the reference keeps that stream in fp32
 * (autocast leaves additions alone).  Rows m with m % period == 0 (period = tokens per sample: the class-token rows; period 1:
 * every row) carry an fp32 copy at side row m / period, pitch ld floats.  `in`: residual operand rows (NULL: the 16-bit
 * residual operand is used for them too), `out`: the sums.  The 16-bit output row is round(out row). */
typedef struct sf_rows32 {
    const float* in;
    float* out;
    int32_t ld, period;
} sf_rows32;
/* nn.Linear forward with the residual addition of the stream in its epilogue: Y = A W^T + bias + resid, side rows summed from
 * the fp32 accumulators (attention.py:393 proj, common.py:31 fc2 followed by attention.py:502 / :510). */
int sf_gemm_rows32(int64_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias,
                   const void* resid, int32_t ldr, void* Y, int32_t ldy, const sf_rows32* side, sf_stream_t stream);
/* Batched "TN" GEMM  Out[z][r][c] = scale * sum_m P[z][m][r] * X[z][m][c]  (fp16 out): the attention gradients
 * dV = P^T dO and dK = dS^T q (autograd of attention.py:355,379). */
int sf_bgemm_tn(int64_t M, int32_t R, int32_t Kc, const void* P, int32_t ldp, const void* X, int32_t ldx, void* Out,
                int32_t ldo, float scale, int32_t nbatch, int32_t bh, int64_t sp_b, int64_t sp_h, int64_t sx_b,
                int64_t sx_h, int64_t so_b, int64_t so_h, sf_stream_t stream);

/* nn.LayerNorm(C, eps) over the last dim (attention.py:240-268, 428, 456; video_model_builder.py:1032) */
int sf_layernorm_fwd(int64_t M, int32_t C, const void* x, int32_t ldx, const float* gamma, const float* beta, float eps,
                     void* y, int32_t ldy, float* mean, float* rstd, sf_stream_t stream);
/* same; rows that have an fp32 side copy (side->in) are normalised from it (norm1 / norm2 / the final norm reading the stream) */
int sf_layernorm_fwd_rows32(int64_t M, int32_t C, const void* x, int32_t ldx, const float* gamma, const float* beta, float eps,
                            void* y, int32_t ldy, float* mean, float* rstd, const sf_rows32* side, sf_stream_t stream);
int sf_layernorm_bwd_blocks(int64_t M, int32_t C);   /* rows of `part` */
/* dx = LN backward (+ resid); part[blk][0][c] = sum dy*xhat, part[blk][1][c] = sum dy -> sf_colsum_finalize */
int sf_layernorm_bwd(int64_t M, int32_t C, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* gamma,
                     const float* mean, const float* rstd, const void* resid, int32_t ldr, void* dx, int32_t lddx,
                     float* part, sf_stream_t stream);
/* same, and part[blk][2][c] = sum of resid, part[blk][3][c] = sum of the stored dx (part is [blocks][4][C]): the bias gradients of
 * the Linear layers on either side of the LayerNorm -- in MultiScaleBlock (attention.py:491-514) mlp.fc2.bias from
 * resid = d(block output) and attn.proj.bias from dx -- without a column-sum pass of their own */
int sf_layernorm_bwd_sums(int64_t M, int32_t C, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* gamma,
                          const float* mean, const float* rstd, const void* resid, int32_t ldr, void* dx, int32_t lddx,
                          float* part, sf_stream_t stream);
/* column sums of an [M][C] tensor (bias gradients): part[blk][0][c] */
int sf_colsum_blocks(int64_t M, int32_t C);
int sf_colsum(int64_t M, int32_t C, const void* x, int32_t ldx, float* part, sf_stream_t stream);
/* out0[c % fold] (+)= scale * sum_blk part[blk][0][c], out1 likewise from part[blk][1][c] (either may be NULL);
 * `part` is scratch (folded in place) */
int sf_colsum_finalize(float* part, int32_t nblk, int32_t C, int32_t fold, float* out0, float* out1, float scale,
                       int accumulate, sf_stream_t stream);
/* n finalizes in one launch (the backward of a MultiScaleBlock ends in ~9: LayerNorm affine gradients, bias gradients;
 * attention.py:428-514).  `items` is a HOST array, passed to the kernel by value; two items must not share an output; tables of
 * more than 2048 partial rows go through sf_colsum_finalize. */
typedef struct sf_colfin_item {
    float* part;
    int32_t nblk, C, fold;
    float* out0;
    float* out1;
    float scale;
    int32_t accumulate;
    int32_t row_stride;   /* table rows of [2][C] floats between two partial rows (0 / 1: dense; 2: the pairs of a [blk][4][C] table) */
} sf_colfin_item;
int sf_colsum_finalize_batch(const sf_colfin_item* items, int32_t n, sf_stream_t stream);
/* out[i] (+)= scale * sum_b part[b*row_len + offset + i] */
int sf_rows_sum(const float* part, int32_t nblk, int64_t row_len, int64_t offset, int32_t n, float* out, float scale,
                int accumulate, sf_stream_t stream);
/* nn.GELU (exact erf; common.py:20) on contiguous arrays of n elements (n % 8 == 0) */
int sf_gelu_fwd(int64_t n, const void* h, void* a, sf_stream_t stream);
int sf_gelu_bwd(int64_t n, const void* h, const void* da, void* dh, sf_stream_t stream);

/* Depthwise Conv3d(C, C, k, stride, padding, groups = C) on channels-last rows; C may be heads*Cw with one weight
 * [Cw][taps] shared by the heads (MViT pool_q/k/v, attention.py:227-266); cls != 0: token tensors whose cls row is
 * routed around the convolution (attention.py:24-36).  Also X3DTransform.b (resnet_helper.py:214-224) and the
 * X3D stem's (5,1,1) temporal conv (stem_helper.py:267-275).  ldx / ldy = pitches of input / output rows.
 * Any kernel extent with kT*kH*kW*Cw <= 12288 (LDS weight image): 3x3x3 / (5,1,1) take the W-blocked stencils, wider planes
 * (MViTv1's stride+1 pooling kernels 1x5x5, 1x9x9) the generic ones, their weight gradient in chunks of 9 taps. */
typedef struct sf_dw_desc {
    int32_t N, C, Cw, cls;
    int32_t Ti, Hi, Wi, To, Ho, Wo;
    int32_t kT, kH, kW, sT, sH, sW, pT, pH, pW;
    int32_t ldx, ldy;
    int32_t Cwreal;   /* rows of the fp32 weight (0 = Cw); channels [Cwreal, Cw) are zero padding (X3D widths 54, 108) */
} sf_dw_desc;
int sf_dwconv_fwd_blocks(const sf_dw_desc* d);       /* rows of stat_part */
/* rows of the statistics table of sf_dwconv_fwd that belong to one sample when the table is sample-major (rows [n * r, (n + 1) * r)
 * = sample n), else 0: per-sample channel sums (SE squeeze, operators.py:38-45) can then be read from the table (ABI 23) */
int sf_dwconv_fwd_sample_rows(const sf_dw_desc* d);
/* stat_part (optional): [blocks][2][C] per-block sum / sum of squares of the outputs (BatchNorm statistics) */
int sf_dwconv_fwd(const sf_dw_desc* d, const void* x, const float* w, void* y, float* stat_part, sf_stream_t stream);
int sf_dwconv_dgrad(const sf_dw_desc* d, const void* dy, const float* w, void* dx, sf_stream_t stream);
/* ... that also leaves the per-workgroup column sums of dx (cls row included) in slot 0 of sum_part[sf_dwconv_dgrad_sum_rows(d)][2][C]
 * (slot 1 unspecified); rows == 0: the kernel this geometry takes cannot.  MViT's qkv bias gradient = the column sums of d(qkv),
 * which the three pooling data gradients write (attention.py:318-330) (ABI 23) */
int sf_dwconv_dgrad_sum_rows(const sf_dw_desc* d);
int sf_dwconv_dgrad_sums(const sf_dw_desc* d, const void* dy, const float* w, void* dx, float* sum_part, sf_stream_t stream);
int64_t sf_dwconv_wgrad_workspace(const sf_dw_desc* d);
int sf_dwconv_wgrad(const sf_dw_desc* d, const void* x, const void* dy, float* dw, float out_scale, int zero_first,
                    void* workspace, int64_t workspace_bytes, sf_stream_t stream);
/* PAIR forms (ABI 23): two depthwise convolutions of the SAME geometry d on two tensors (x, x2) with separate weights in one launch
 * per direction -- MViT pool_k / pool_v of a block (attention.py:227-266).  sf_dwconv_pair_ok(d) == 1 when the geometry is taken
 * (3x3x3 / padding 1 plane sweeps in all three directions, C % 32 == 0); results equal two single calls bit for bit.
 * sf_dwconv_wgrad_pair needs 2 x sf_dwconv_wgrad_workspace(d) bytes. */
int sf_dwconv_pair_ok(const sf_dw_desc* d);
int sf_dwconv_fwd_pair(const sf_dw_desc* d, const void* x, const void* x2, const float* w, const float* w2, void* y, void* y2,
                       sf_stream_t stream);
int sf_dwconv_dgrad_pair(const sf_dw_desc* d, const void* dy, const void* dy2, const float* w, const float* w2, void* dx, void* dx2,
                         sf_stream_t stream);
int sf_dwconv_wgrad_pair(const sf_dw_desc* d, const void* x, const void* x2, const void* dy, const void* dy2, float* dw, float* dw2,
                         float out_scale, int zero_first, int zero_first2, void* workspace, int64_t workspace_bytes,
                         sf_stream_t stream);

/* Pooled attention (attention.py:354-385).  Tokens are [B][N][heads*D]; scores [B][heads][Nq][lds]. */
typedef struct sf_attn_desc {
    int32_t B, heads, D, cls;
    int32_t Nq, qT, qH, qW;        /* Nq = cls + qT*qH*qW */
    int32_t Nk, kT, kH, kW;        /* Nk = cls + kT*kH*kW */
    int32_t rows_h, rows_w, rows_t; /* rows of rel_pos_h / rel_pos_w / rel_pos_t */
} sf_attn_desc;
/* Decomposed relative positions (cal_rel_pos_spatial / cal_rel_pos_temporal, attention.py:64-147).  The contractions
 * with the tables run as GEMMs over the concatenated table Tab = [rel_pos_h; rel_pos_w; rel_pos_t]:
 *   forward   G = q_unscaled Tab^T (sf_bgemm), rq[(b,q,head)][kH+kW+kT] = G[row][column of table row j]   (gather)
 *   backward  E[row][column of table row j] = drq[row][j] (scatter, zero elsewhere); dq += E Tab (sf_bgemm);
 *             dTab = E^T q (sf_conv_wgrad on the 1x1x1 geometry).
 * idx_h [qH][kH], idx_w [qW][kW], idx_t [qT][kT] are the reference's dist_* tables as int32; cls rows get zeros. */
int sf_relpos_gather(const sf_attn_desc* d, const void* G, int32_t ldg, const int32_t* idx_h, const int32_t* idx_w,
                     const int32_t* idx_t, float* rq, sf_stream_t stream);
int sf_relpos_scatter(const sf_attn_desc* d, const float* drq, const int32_t* idx_h, const int32_t* idx_w,
                      const int32_t* idx_t, void* E, int32_t lde, sf_stream_t stream);
/* The concatenated table Tab = [rel_pos_h; rel_pos_w; rel_pos_t] (fp32 parameters, D columns) as 16-bit GEMM operands:
 * t16 [TRp][D] (rows beyond the tables zero, TRp % 8 == 0) and t16t = t16^T [D][TRp]; and back: rows of dtab [TRp][D] fp32
 * into the three parameter gradients (acc_*: accumulate instead of overwrite). */
int sf_relpos_pack(const float* rel_h, const float* rel_w, const float* rel_t, int32_t rows_h, int32_t rows_w, int32_t rows_t,
                   int32_t D, int32_t TRp, void* t16, void* t16t, sf_stream_t stream);
int sf_relpos_unpack(const float* dtab, int32_t rows_h, int32_t rows_w, int32_t rows_t, int32_t D, float* grad_h, float* grad_w,
                     float* grad_t, int32_t acc_h, int32_t acc_w, int32_t acc_t, sf_stream_t stream);
/* in place: s <- softmax_k(scale*s + bias), bias from rq (NULL = none); pad columns [Nk, lds) <- 0 */
int sf_softmax_fwd(const sf_attn_desc* d, void* s, int32_t lds, float scale, const float* rq, sf_stream_t stream);
/* in place: dp <- scale * P*(dp - sum_k P*dp); drq (optional) <- per-(kh|kw|kt) sums of the unscaled dS */
int sf_softmax_bwd(const sf_attn_desc* d, void* dp, const void* prob, int32_t lds, float scale, float* drq,
                   sf_stream_t stream);
/* Fused attention core -- replaces attention.py:355-385 (scores, decomposed rel-pos bias, softmax, attn @ v, residual
 * pooling) without materialising the [B, heads, Nq, Nk] score tensors (online softmax, fp32 statistics).
 *   o[b][q][head*D + :] = softmax_k(scale * q.k + bias(q, k)) v  (+ q on the rows >= cls when `residual`)
 * q, o, dq: [B][Nq][.] rows of pitch ldq / ldo / lddq; k, v, dk, dv: [B][Nk][.] rows of pitch ldk / lddk; head h occupies
 * columns [h*D, (h+1)*D), D in {32, 64, 96, 128}.  rq (optional) is sf_relpos_gather's output
 * [(b*Nq + q)*heads + head][kH + kW + kT] fp32 and `onehot` the constant fp16 matrix [roundup(Nk, 32)][64] with
 * onehot[key][j] = 1 for j in {kh, kH + kw, kH + kW + kt} of a non-cls key (zero rows for the cls key and the padding):
 * bias(q, key) = sum_j rq[q][j] onehot[key][j] runs on the matrix cores.  lse / delta are
 * [(b*heads + head)*Nq + q] fp32 scratch (caller-owned) that the forward / backward fill.  sf_attn_bwd returns dq (incl.
 * the residual's dout on the rows >= cls), dk, dv and drq (the bias gradient, input of sf_relpos_scatter); `workspace`
 * (>= sf_attn_bwd_workspace(d) bytes, may be 0) holds the dK/dV partials of the query splits. */
int sf_attn_fwd(const sf_attn_desc* d, const void* q, int32_t ldq, const void* k, const void* v, int32_t ldk, float scale,
                const float* rq, const void* onehot, int32_t residual, void* o, int32_t ldo, float* lse,
                sf_stream_t stream);
int64_t sf_attn_bwd_workspace(const sf_attn_desc* d);
int sf_attn_bwd(const sf_attn_desc* d, const void* q, int32_t ldq, const void* k, const void* v, int32_t ldk, float scale,
                const float* rq, const void* onehot, int32_t residual, const void* o, const void* dout, int32_t ldo,
                const float* lse, float* delta, void* dq, int32_t lddq, void* dk, void* dv, int32_t lddk, float* drq,
                void* workspace, int64_t workspace_bytes, sf_stream_t stream);
/* RoI head -- replaces head_helper.py:85-97 / 116-133 (ResNetRoIHead): AvgPool3d([T,1,1]) then ROIAlign(res x res,
 * spatial_scale, sampling_ratio 0, aligned) fused with MaxPool2d(res).
 *   sf_tmean_fwd: m[b][hw][c] = mean_t x[b][t][hw][c]   (x channels-last fp16 rows of pitch ldx, m fp32 [B*HW][C])
 *   sf_roi_align_max_fwd: out[r][col0 + c] = max over the res*res bins of ROIAlign(m, rois[r]) (rois [R][5] fp32 =
 *       batch index, x1, y1, x2, y2 in input pixels); argmax [R][C] uint8 records the winning bin
 *   sf_roi_align_max_bwd: dm (pre-zeroed by the caller, fp32) += the gradient routed to the arg-max bin's bilinear samples
 *       (fp32 atomics: overlapping boxes); sf_tmean_bwd: dx[b][t][hw][c] = dm[b][hw][c] / T.
 * ROIAlign is detectron2.layers.ROIAlign in the reference (not vendored): see oracle/video_ref.py:roi_align. */
int sf_tmean_fwd(int32_t B, int32_t T, int64_t HW, int32_t C, const void* x, int32_t ldx, float* m, sf_stream_t stream);
int sf_tmean_bwd(int32_t B, int32_t T, int64_t HW, int32_t C, const float* dm, void* dx, int32_t lddx, sf_stream_t stream);
int sf_roi_align_max_fwd(int32_t R, int32_t B, int32_t H, int32_t W, int32_t C, int32_t res, float scale, int32_t aligned,
                         const float* m, const float* rois, float* out, int32_t ldo, int32_t col0, void* argmax,
                         sf_stream_t stream);
int sf_roi_align_max_bwd(int32_t R, int32_t B, int32_t H, int32_t W, int32_t C, int32_t res, float scale, int32_t aligned,
                         const float* rois, const float* dout, int32_t lddo, int32_t col0, const void* argmax, float* dm,
                         sf_stream_t stream);
/* Linear layers of the Mlp with the GELU fused into the GEMM epilogue (slowfast/models/common.py:25-34):
 *   mode 1 (fc1 forward):   Y = A W^T + bias,  aux = gelu(Y)        (Y is kept for the backward)
 *   mode 2 (fc2 data grad):  Y = (A W^T) * gelu'(aux)                (aux = the saved pre-activation)
 * A [M][K], W [N][K], Y / aux [M][N] fp16, row pitches in elements. */
int sf_gemm_act(int64_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias,
                void* Y, int32_t ldy, int32_t mode, void* aux, int32_t ldaux, sf_stream_t stream);
/* sf_gemm_act + the column sums of the tile it stores (round 3): colsum_part[row][2][N] fp32, slot 0 of every row = the sum of
 * Y over the rows of one M tile (slot 1 unused), *colsum_rows (HOST pointer) = rows written.  With mode 2 (Y = d(loss)/d(fc1
 * output)) the sums are the bias gradient of the Mlp's fc1 (common.py:25-34) once folded by sf_colsum_finalize -- the
 * separate sf_colsum pass over the widest gradient tensor of every block goes away.  colsum_part holds ceil(M / 128) rows. */
int sf_gemm_act_colsum(int64_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* W, int32_t ldw,
                       const float* bias, void* Y, int32_t ldy, int32_t mode, void* aux, int32_t ldaux,
                       float* colsum_part, int32_t colsum_part_rows, int32_t* colsum_rows, sf_stream_t stream);
/* Input side -- replaces tensor_normalize (slowfast/datasets/utils.py:278-297), the THWC -> CTHW
 * permute (datasets/kinetics.py:375-408) and pack_pathway_output (datasets/utils.py:78-111) for one pathway:
 *   out[n][to][h][w][c] = ((frames[n][t_index[to]][h][w][s] / 255) - mean[s]) / std[s]   (s = c, or 2 - c when `reverse`),
 * frames uint8 [N][Tin][H][W][3] on the device, out fp16 [N][Tout][H][W][4] (4th channel zero) = the buffer the stems read
 * as W pairs; t_index [Tout] int32 on the device (NULL = identity; the Slow pathway passes linspace(0, T-1, T/alpha)). */
int sf_pack_clip_u8(const void* frames, int32_t N, int32_t Tin, int32_t H, int32_t W, const int32_t* t_index, int32_t Tout,
                    float mean0, float mean1, float mean2, float std0, float std1, float std2, int32_t reverse, void* out,
                    sf_stream_t stream);
/* Stochastic depth -- replaces drop_path() (slowfast/models/common.py:46-59) at the two residual additions of
 * MultiScaleBlock (attention.py:500-510): y[m] = (resid ? resid[m] : 0) + scale[m / rows_per_sample] * x[m], with
 * scale[b] = floor(keep_prob + u_b) / keep_prob sampled by the caller.  Rows are fp16 [M][C], C % 8 == 0. */
int sf_row_scale_add(const void* x, int32_t ldx, const float* scale, int64_t rows_per_sample, const void* resid,
                     int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t C, sf_stream_t stream);
/* same with fp32 side rows of the stream (sf_rows32 above): side rows read side->in (or resid), write side->out */
int sf_row_scale_add_rows32(const void* x, int32_t ldx, const float* scale, int64_t rows_per_sample, const void* resid,
                            int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t C, const sf_rows32* side, sf_stream_t stream);
/* xt[b][head][c][k] = x[b][k][head*D + c], zero for k in [Nk, ldk): K-contiguous operand for P.V and dS.K */
int sf_transpose_heads(const void* x, int32_t ldx, void* xt, int32_t ldk, int32_t B, int32_t Nk, int32_t heads, int32_t D,
                       sf_stream_t stream);

/* =====================================================================================================
 * X3D: SE block (operators.py:15-59), Swish (pytorchvideo), X3DHead average pool (head_helper.py:413-438).
 * Rows of a tensor are (n, pos), S positions per sample, C channels (padded to 8).
 * ===================================================================================================== */
int sf_sample_chunks(int64_t S, int32_t C);          /* part needs N * chunks * 2 * C floats */
/* out[n][c] = mean_pos relu?(y*scale + shift)  (nn.AdaptiveAvgPool3d(1) of SE on the BatchNorm output; the X3DHead
 * AvgPool3d over the whole (T,H,W) extent after conv_5_bn + ReLU); scale/shift NULL = identity */
int sf_sample_mean(int32_t N, int64_t S, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                   int relu, float* part, float* out, sf_stream_t stream);
/* h = relu(W1 m + b1), gate = sigmoid(W2 h + b2) per sample; m, gate rows have pitch Cp >= C (pad gate = 0) */
int sf_se_gate_fwd(int32_t N, int32_t C, int32_t Cp, int32_t F, const float* m, const float* w1, const float* b1,
                   const float* w2, const float* b2, float* h, float* gate, sf_stream_t stream);
/* z = act(gate[n][c] * (y*scale + shift)), act = swish (x*sigmoid(x)) or relu; gate NULL = 1 */
int sf_gate_act_fwd(int32_t N, int64_t S, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                    const float* gate, int swish, void* z, int32_t ldz, sf_stream_t stream);
/* dgate[n][c] = sum_pos dz * act'(gate*u) * u,  u = y*scale + shift */
int sf_gate_grad(int32_t N, int64_t S, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                 const void* dz, int32_t lddz, const float* gate, int swish, float* part, float* dgate, sf_stream_t stream);
/* per-sample backward of the gate: dpre2 = dgate*g*(1-g), dpre1 = (W2^T dpre2) masked by h > 0, dm = W1^T dpre1 */
int sf_se_gate_bwd(int32_t N, int32_t C, int32_t Cp, int32_t F, const float* gate, const float* h, const float* w1,
                   const float* w2, const float* dgate, float* dpre2, float* dpre1, float* dm, sf_stream_t stream);
/* out[i][j] (+)= scale * sum_n a[n*lda + i] * b[n*ldb + j]  (SE weight/bias gradients; b NULL: J = 1, b = 1) */
int sf_outer_sum(const float* a, int32_t lda, const float* b, int32_t ldb, int32_t N, int32_t I, int32_t J, float* out,
                 float scale, int accumulate, sf_stream_t stream);
/* du = dz * act'(gate*u) * gate + dmean[n][c] / S  (gradient w.r.t. the BatchNorm output u); dmean NULL = 0 */
int sf_gate_act_bwd(int32_t N, int64_t S, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                    const float* gate, int swish, const void* dz, int32_t lddz, const float* dmean, void* du, int32_t lddu,
                    sf_stream_t stream);
/* the same pass, also leaving the per-workgroup column sums of du and du * y in bn_part[sf_gate_act_bwd_bn_rows()][2][C]: the
 * reduction half of the backward of the BatchNorm in front of the gate (X3DTransform.b_bn, resnet_helper.py:226-250), which
 * sf_bn_bwd_finalize takes in place of sf_bn_bwd_reduce's table (ABI 23) */
/* one pass for the backward of SE-gated Swish: du0 = dz * act'(gate * u) * gate stored, sums[N][3][C] = per-sample sums of
 * dz * act'(gate * u) * u (the gate's gradient), du0 and du0 * y; part: [N * sf_sample_chunks(S, C)][4][C] scratch (ABI 23) */
int sf_gate_bwd_sums(int32_t N, int64_t S, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                     const void* dz, int32_t lddz, const float* gate, int swish, void* du0, int32_t lddu, float* part,
                     float* sums, sf_stream_t stream);
/* sf_bn_bwd_apply for a gradient that lacks a per-sample constant: dy = k1 * (dz + sample_add[row / S][c]) + k2 + k3 * y (ABI 23) */
int sf_bn_bwd_apply_sample(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* y, int32_t ldy, const float* coef,
                           const float* sample_add, int64_t S, void* dy, int32_t lddy, sf_stream_t stream);
int sf_gate_act_bwd_bn_rows(int32_t N, int64_t S, int32_t C);
int sf_gate_act_bwd_bn(int32_t N, int64_t S, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift,
                       const float* gate, int swish, const void* dz, int32_t lddz, const float* dmean, void* du, int32_t lddu,
                       float* bn_part, sf_stream_t stream);

/* ---- training-step glue on the flat gradient memory -- replaces tools/train_net.py:150-172 (GradScaler.unscale_ / step /
 * update, clip_grad_norm_ / clip_grad_value_, optimizer.get_grad_norm_, misc.check_nan_losses) and optimizer.step()
 * (slowfast/models/optimizer.py:100-140: SGD with momentum / dampening / nesterov, AdamW), SURVEY.md 8f-1.  No host sync:
 * everything the next launch needs travels through the control block `ctl` (8 fp32 words in device memory):
 *   [0] loss scale  [1] growth tracker  [2] found_inf of this step  [3] global gradient norm (unscaled, mean over ranks)
 *   [4] multiplier applied to the raw gradients = clip_coef / (world * scale)  [5] clean optimizer steps  [6] skipped steps. */
int sf_flat_blocks(int64_t n);                                  /* rows of `part` ([rows][2] fp32) for sf_flat_sumsq */
int sf_flat_sumsq(const float* g, int64_t n, float* part, sf_stream_t stream);     /* g: any 4-byte aligned range of the buffer (a
                                                                 * bucket at a time: the partial rows simply add up) */
/* one workgroup: norm / found_inf / clip coefficient into ctl; GradScaler.update() when `dynamic` */
int sf_step_control(const float* part, int32_t nblk, float* ctl, float world, float clip_norm, int dynamic, float growth,
                    float backoff, int32_t growth_interval, sf_stream_t stream);
/* segs: [nseg] {int64 start, int64 end, int32 group, int32 pad}; blk_seg / blk_off: per launched block its segment and its first
 * element inside it (multiples of 1024); lr / wd: HOST arrays, one entry per parameter group.  Skipped when ctl[2] != 0. */
int sf_flat_sgd(float* param, const float* grad, float* mom, const void* segs, const int32_t* blk_seg, const int32_t* blk_off,
                int32_t nblocks, const float* ctl, const float* lr, const float* wd, int32_t ngroups, float clip_val,
                float momentum, float dampening, int nesterov, sf_stream_t stream);
int sf_flat_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const void* segs, const int32_t* blk_seg,
                  const int32_t* blk_off, int32_t nblocks, const float* ctl, const float* lr, const float* wd, int32_t ngroups,
                  float clip_val, float beta1, float beta2, float eps, sf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SFAMD_H */
