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

#define SF_ABI_VERSION 3
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
} sf_conv_desc;

int sf_abi_version(void);
const char* sf_backend(void);     /* "gfx950" (product) or "hostsim" (CPU test build of the same sources) */
const char* sf_last_error(void); /* host string, thread-local */

/* ---- Conv3d -- replaces nn.Conv3d at resnet_helper.py:331-369 (BottleneckTransform a/b/c),
 * resnet_helper.py:485-493 (ResBlock.branch1), stem_helper.py:182-189 (ResNetBasicStem.conv),
 * video_model_builder.py:147-154 (FuseFastToSlow.conv_f2s), and their autograd backward. */
int sf_conv_weight_ld(const sf_conv_desc* d, int32_t* ldf, int32_t* ldd);
/* fp32 [Co][Cw][kT][kH][kW] -> fp16 forward operand wf[Co][ldf] and (optional) dgrad operand wd[Ci][ldd] */
int sf_prep_weights(const sf_conv_desc* d, const float* w, void* wf, void* wd, sf_stream_t stream);
/* number of 128-row tiles = rows of `stat_part` */
int sf_conv_fwd_mtiles(const sf_conv_desc* d);
/* y = conv(act(x)), act(x) = x or relu?(x*in_scale + in_shift) applied on the fly (zero padding is
 * applied AFTER act, as in the reference where the padded tensor is the post-ReLU activation).
 * stat_part (optional) receives per-tile per-channel sum / sum of squares: [mtiles][2][Co] fp32. */
int sf_conv_fwd(const sf_conv_desc* d, const void* x, const void* wf, const float* in_scale, const float* in_shift,
                int in_relu, const float* bias, void* y, float* stat_part, sf_stream_t stream);
/* dx = conv_transpose(dy, w) (+ resid), dx pitch = d->ldx, dy pitch = d->ldy */
int sf_conv_dgrad(const sf_conv_desc* d, const void* dy, const void* wd, const void* resid, int32_t ldr, void* dx,
                  sf_stream_t stream);
/* dw[Co][Cw][taps] = (zero_first ? 0 : dw) + out_scale * sum_m dy[m] (x) act(x)[m].  The reduction over positions
 * is split; `workspace` (>= sf_conv_wgrad_workspace(d) bytes, caller-owned) holds the per-split partials, which a
 * second kernel sums in a fixed order (no atomics: results are run-to-run reproducible). */
int64_t sf_conv_wgrad_workspace(const sf_conv_desc* d);
int sf_conv_wgrad(const sf_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, int in_relu,
                  const void* dy, float* dw, float out_scale, int zero_first, void* workspace, int64_t workspace_bytes,
                  sf_stream_t stream);

/* ---- BatchNorm3d -- replaces nn.BatchNorm3d built by batchnorm_helper.py:16-37 (get_norm) at every
 * *_bn call site of resnet_helper.py / stem_helper.py / video_model_builder.py:155-159, plus the
 * nn.ReLU and the residual add of resnet_helper.py:512-521. */
/* `part` is scratch: long tables are folded in place before the final reduction (contents are destroyed).
 * nblk > 0: training (partials -> batch statistics, running stats updated when non-null);
 * nblk == 0: eval (running statistics).  Outputs scale = gamma*rstd, shift = beta - mean*scale. */
int sf_bn_finalize(float* part, int32_t nblk, int32_t C, float count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                   float* save_mean, float* save_rstd, sf_stream_t stream);
/* out = relu?( y*scale+shift [+ r*rscale+rshift | + r] ); scale == NULL means identity */
int sf_bn_act(int64_t M, int32_t C, const void* y, int32_t ldy, const float* scale, const float* shift, const void* r,
              int32_t ldr, const float* rscale, const float* rshift, int relu, void* out, int32_t ldo,
              sf_stream_t stream);
int sf_bn_bwd_blocks(int64_t M, int32_t C); /* rows of `part` for the two calls below */
/* part[blk][0][c] = sum g, part[blk][1][c] = sum g*y, g = dz masked by (zmask > 0) or by
 * (y*scale+shift > 0) when relu_self */
int sf_bn_bwd_reduce(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* zmask, int32_t ldm, const void* y,
                     int32_t ldy, const float* scale, const float* shift, int relu_self, float* part,
                     sf_stream_t stream);
/* dgamma/dbeta (fp32, unscaled by inv_loss_scale) and coef[3][C] with dy = k1*g + k2 + k3*y */
int sf_bn_bwd_finalize(float* part, int32_t nblk, int32_t C, float count, const float* gamma, const float* mean,
                       const float* rstd, float inv_loss_scale, float* dgamma, float* dbeta, int accumulate,
                       float* coef, sf_stream_t stream);
int sf_bn_bwd_apply(int64_t M, int32_t C, const void* dz, int32_t lddz, const void* zmask, int32_t ldm, const void* y,
                    int32_t ldy, const float* scale, const float* shift, int relu_self, const float* coef, void* dy,
                    int32_t lddy, void* gout, int32_t ldg, sf_stream_t stream);

/* ---- nn.MaxPool3d([1,kH,kW],[1,sH,sW],[0,pH,pW]) fused with the producer's BN(+ReLU) --
 * stem_helper.py:190-201 (bn -> relu -> pool_layer). */
int sf_pool_fwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW, int32_t sH, int32_t sW,
                int32_t pH, int32_t pW, const void* y, int32_t ldy, const float* scale, const float* shift, int relu,
                void* out, int32_t ldo, void* argmax, sf_stream_t stream);
/* g[N,T,H,W,C] = gradient w.r.t. the BatchNorm output (pool + ReLU backward) from the forward's pooled output
 * (ReLU mask: pooled > 0) and its byte argmax table [N,T,Ho,Wo][C] (window-local index kh*kW+kw of the first
 * maximum, as recorded by torch's max_pool3d) */
int sf_pool_bwd(int32_t N, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kH, int32_t kW, int32_t sH, int32_t sW,
                int32_t pH, int32_t pW, const void* pooled, int32_t ldp, const void* argmax, int relu,
                const void* dout, int32_t lddo, void* g, int32_t ldg, sf_stream_t stream);

/* ---- layout: clips arrive NCTHW fp32 (tools/train_net.py:79-98) */
int sf_ncthw_to_cl(const float* x, int32_t N, int32_t C, int64_t S, int32_t Cp, void* out, sf_stream_t stream);
int sf_cl_to_ncthw(const void* x, int32_t ld, int32_t N, int32_t C, int64_t S, float* out, sf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SFAMD_H */
