/* libpunet_hip.so - C ABI of the MI355X-native (gfx950 / CDNA4) Probabilistic U-Net self-training hot path.
 *
 * Drop-in boundary.  The reference (/root/reference, pure Python) has no FFI of its own: its hot path
 * is reached through nn.Module methods (prob_utils/my_models/probabilistic_unet.py:285-371) and trainer bodies
 * (prob_utils/my_trainer/ *.py).  Each entry point below replaces the ATen op(s) that the cited reference line
 * dispatches; the host-side mirror of the reference API that calls them lives in
 * probabilistic_domain_adaptation_amd/{my_models,my_trainer,my_predictions} (ctypes binding: _lib.py; the stub a
 * reference maintainer would add is shown in INTEGRATION.md).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (a PyTorch-ROCm tensor's
 *    data_ptr(), or hipMalloc); the library owns nothing and allocates nothing.
 *  - activations are NHWC fp32: x[b][y][x][c]; a [B,1,H,W] tensor (logits, labels, masks) has the same bytes in NCHW.
 *  - conv weights are [ky][kx][cin][cout] fp32 (the canonical parameter layout of this build; the Python side exposes
 *    them under the reference's state-dict names/shapes [cout,cin,ky,kx] as permuted views of the same memory).
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is enqueued on it,
 *    no hidden synchronisation, no host reads -> every call is hipGraph-capturable.
 *  - no hidden process state: the library keeps nothing between calls except a per-device cache of the CU count;
 *    every entry point may be called from several threads at once.
 *  - return 0 on success, negative on error (PUNET_ERR_ARG = -1 bad argument, PUNET_ERR_HIP = -2 launch failure);
 *    punet_last_error() returns a thread-local message.  Nothing throws across the ABI.
 *  - `workspace` pointers are caller-provided scratch; the matching *_workspace_floats() gives the required size.
 *  - all reductions are two-pass with fixed summation order (bit-reproducible run to run; no float atomics).
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* punet_last_error(void);
const char* punet_build_info(void);
int punet_device_ok(void);

/* ---- K1: nn.Conv2d(3x3 p1 | 1x1) [+ ReLU] with cin >= 8, fp32 MFMA (unet_blocks.py:19-24, probabilistic_unet.py:56-61,
 *      168-173).  Optional second source = the channel concat of unet_blocks.py:56 ([up, bridge]) without materialising
 *      it.  bias_img_stride = 0: bias[cout]; = cout: bias[b][cout] (Fcomb's folded z term, probabilistic_unet.py:206-212).
 *      c0, c1 multiples of 8; cout multiple of 4. */
int punet_conv_fwd(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias,
                   int bias_img_stride, float* y, int cout, int B, int H, int W, int ksize, int relu, void* stream);

/* ---- K2: autograd input gradient of K1 (convolution_backward, input): dx[.., cin_off : cin_off+cin] of the conv whose
 *      weight is w[k][k][cin_total][cout]; epilogue multiplies by (mask > 0) when mask != NULL (ReLU backward of the
 *      tensor that dx belongs to) and adds into dx when accumulate != 0 (skip connections). */
int punet_conv_dgrad(const float* dz, int cout, const float* w, int cin_total, int cin_off, int cin, float* dx,
                     const float* mask, int accumulate, int B, int H, int W, int ksize, void* stream);

/* ---- K2 fast form: the dgrad's weight slab is first brought into forward-kernel orientation,
 *      wt[k*k-1-t][cout][cin_total] = w[t][cin_total][cout] (once per step for all layers), after which the input gradient
 *      is exactly the K1 kernel.  cin_total, cin_off, cin multiples of 4. */
int punet_flip_transpose_weights(const float* w, float* wt, int cin, int cout, int ksize, void* stream);
int punet_conv_dgrad_t(const float* dz, int cout, const float* wt, int cin_total, int cin_off, int cin, float* dx,
                       const float* mask, int accumulate, int B, int H, int W, int ksize, void* stream);

/* ---- K1x6 / K2x6: the same 3x3 convolution / input gradient at fp32 accuracy on the BF16 matrix cores: every operand is
 *      split into three bf16 planes (3 x 8 = 24 significand bits) and each product evaluated as hh+hm+mh+hl+lh+mm with fp32
 *      accumulation (dropped terms <= 2^-26 relative).  Weights are pre-split once per optimizer step:
 *      planes[3][9][K/8][n_row][8] bf16, plane_stride elements apart; mode 0: K = cin, n = cout (forward);
 *      mode 1: K = cout, n = cin, taps flipped (input gradient).  The conv computes output channels
 *      [n_off, n_off + n_out) of the n_row available; mask/accumulate/relu/bias as in K1/K2. */
/* Arithmetic options of K1x6/K2x6/K3x6 travel WITH EACH CALL in an options word (no process state; thread-safe):
 *   products: bf16 products per fp32 product - 6 (default) = fp32-accurate parity arithmetic; 3 = two planes, hh+hm+mh, 16
 *     significand bits; 1 = plain bf16.  3 and 1 are opt-in fast modes, the counterpart of the reference scripts'
 *     mixed_precision=True (LIVECell/livecell_mt.py:107); they are NOT parity modes.
 *   shape: MFMA shape of the 3x3 forward / input-gradient kernels - 16 (default) = conv3x3_x6b_kernel on
 *     v_mfma_f32_16x16x32_bf16 for layers whose contracted channel count is a multiple of 32 (the energy-cheaper shape of a
 *     power-limited part, weights staged by LDS-DMA); 32 = conv3x3_x6_kernel on v_mfma_f32_32x32x16_bf16 (always used for
 *     other channel counts and for 1x1).  Same results to rounding; every kernel test runs with both.
 * opts = 0 selects both defaults. */
#define PUNET_X6_OPTS(products, shape) ((int)(products) | ((int)(shape) << 8))
/* punet_conv3x3_splitk only, OR-ed into opts: how the last partial round is dealt - 0 (default) = by the library's cost model,
 * 1 = never split, n >= 2 = runs of n - 1 quads (a quad = 32 contracted channels) whatever the cost model says (tests, A/B) */
#define PUNET_X6_SPLITK(n) ((int)(n) << 16)
/* OR-ed into opts: lay the tiles of THIS launch over each image separately even where the library would fold the batch into one
 * virtual image row (round 5: widths that are not whole 32-pixel tiles while B * W is - 88, 176, 48 ... at batch 4; results are
 * bit-identical either way, tests/test_gpu_fold.py).  punet_conv_wgrad_x6 honours the bit too: its 32-pixel column strips then run
 * per image (folded and unfolded weight gradients sum the same pixels in another order: equal to rounding). */
#define PUNET_X6_NOFOLD (1 << 24)
/* compute units of the current device (grids are sized from it: persistent kernels launch one residency round) */
int punet_compute_units(void);
int punet_split_weights_x6(const float* w, void* planes, long plane_stride, int cin, int cout, int ksize, int mode,
                           void* stream);
/* every 3x3 slab of a flat parameter buffer in ONE launch.  table (device, int64) has nconv+1 rows of 4:
 * {w_off, cin, cout, first_unit}, first_unit = running sum of 9*cin*cout/8, last row = {0,0,0,total_units};
 * slab j's planes land at planes + 3*w_off (bf16 elements) with plane stride 9*cin*cout. */
int punet_split_weights_x6_batch(const float* flat, void* planes, const long* table, int nconv, long total_units,
                                 int mode, void* stream);
int punet_conv3x3_x6(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                     int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                     int accumulate, int relu, int B, int H, int W, int opts, void* stream);
/* The same with the ReLU gate as BITS (1/32 of the fp32 gate's traffic and no per-lane load chain in the epilogue).
 * A [B,H,W,C] activation has punet_gate_dwords(B,H,W,C) = B*H*ceil(W/32)*ceil(C/64)*64 dwords: the block {row y,
 * pixels 32t..32t+31, channels 64cb..64cb+63} owns dwords ((((b*H + y)*ceil(W/32) + t)*ceil(C/64) + cb)*64 + l),
 * l = (x % 32) + 32*((c % 8)/4), and bit 16*((c % 64)/32) + 4*((c % 32)/8) + (c % 4) of dword l is
 * (activation[b][y][x][c] > 0) - the lane / accumulator-register mapping of the kernels' epilogue.
 * gate_out (forward, relu != 0): written for the output y; gate_bits (input gradient): the gate of the tensor dx belongs
 * to, replaces `mask`. */
size_t punet_gate_dwords(int B, int H, int W, int C);
int punet_conv3x3_x6_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                           int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out,
                           const float* mask, const void* gate_bits, void* gate_out, int accumulate, int relu, int B,
                           int H, int W, int opts, void* stream);
/* Forward 3x3 convolution that ALSO writes AvgPool2d(2, 2) of its output (DownConvBlock = pool + 3 convs,
 * unet_blocks.py:17-24 / probabilistic_unet.py:54: the last conv of a block feeds the next block's pool): pool_out is
 * [B, H/2, W/2, n_out]; H and W must be even (ceil_mode never engages); bit-identical to punet_pool2x2_fwd of y. */
int punet_conv3x3_x6_pooled(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                            int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out,
                            void* gate_out, float* pool_out, int relu, int B, int H, int W, int opts, void* stream);
/* Stream-K tail (round 5; tile quantization of the persistent grid): the 3x3 forward / input-gradient
 * kernels deal ntiles = B*ceil(H/8)*ceil(W/32)*ceil(n_out/64) equal tiles to 2 workgroups per CU; when that is not a whole
 * number of rounds (704 x 704 patches of MitoEM/mitoem_*.py: 2.06 rounds at the 88 x 88 level, 7.56 at 352 x 352) the tiles of
 * the last, partial round are dealt as one range of K-quads in equal runs over the workgroups, the partial accumulators meet
 * in `scratch`, and the workgroup that completes a tile sums its parts in a fixed order and runs the epilogue: deterministic
 * (bit-identical run to run; differs from the unsplit launch by summation order only), no workgroup ever waits for another.
 * punet_conv3x3_splitk is the superset of punet_conv3x3_x6_gated / _pooled (f16_planes = 0: bf16 planes by `opts`) and
 * punet_conv3x3_f16 (f16_planes = 2 | 1) with that scratch: `scratch_bytes` <= punet_conv_splitk_scratch_bytes() bytes of
 * device memory, 16-byte aligned, whose first 4096 bytes (arrival counters) are ZERO before the first use and are left zero
 * by every launch; one scratch serves every launch of one stream (launches of different streams need their own).
 * scratch = NULL: no split (= the entry points above).  Used where the layer's contracted channel count is a multiple of 32
 * (conv3x3_x6b_kernel); a no-op elsewhere. */
size_t punet_conv_splitk_scratch_bytes(void);
int punet_conv3x3_splitk(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                         int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                         const void* gate_bits, void* gate_out, float* pool_out, int accumulate, int relu, int B, int H, int W,
                         int opts, int f16_planes, const float* amax_x0, const float* amax_x1, const float* amax_w, float* amax_y,
                         void* scratch, size_t scratch_bytes, void* stream);
/* The same kernel for 1x1 convolutions (the Fcomb layers, probabilistic_unet.py:168-173): planes from
 * punet_split_weights_x6(..., ksize = 1, ...), c0/c1 multiples of 8; always the six-product parity arithmetic. */
int punet_conv1x1_x6_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                           int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out,
                           const float* mask, const void* gate_bits, void* gate_out, int accumulate, int relu, int B,
                           int H, int W, void* stream);

/* ---- K3: autograd weight + bias gradient of K1 (convolution_backward, weight/bias), fp32 MFMA, deterministic split-K.
 *      x has cs channels and fills rows [ci_off, ci_off+cs) of dw[k][k][cin_total][cout]; dbias may be NULL.
 *      dw/db (+)= scale * grad when accumulate != 0, else = scale * grad. */
size_t punet_wgrad_workspace_floats(int cs, int cout, int ksize, int nslices);
int punet_wgrad_slices(int cs, int cout, int B, int H, int W);
int punet_conv_wgrad(const float* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                     float* dbias, float* workspace, int nslices, int B, int H, int W, int ksize, int accumulate,
                     float scale, void* stream);

/* ---- K3x6: the 3x3 weight (+ bias) gradient with the same three-plane bf16 arithmetic as K1x6 (fp32-accurate); operands
 *      are fetched with the transposing LDS read ds_read_b64_tr_b16.  Arguments as punet_conv_wgrad (ksize = 3); nslices
 *      from punet_wgrad_x6_slices, workspace from punet_wgrad_workspace_floats(cs, cout, 3, nslices). */
int punet_wgrad_x6_slices(int cs, int cout, int B, int H, int W); /* split-K slices (one resident workgroup per CU) */
int punet_conv_wgrad_x6(const float* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                        float* dbias, float* workspace, int nslices, int B, int H, int W, int accumulate, float scale,
                        int opts, void* stream);

/* ---- K1s: first-layer 3x3 conv with cin in 1..8 (image / [image, mask], probabilistic_unet.py:39-42,118; the reference's
 *      scripts use 1..4: weights in registers; 5..8: weights in LDS / two passes over channel sub-ranges) and its
 *      weight + bias gradient (no input gradient exists). */
int punet_conv3x3_small_fwd(const float* x, int cin, const float* w, const float* bias, float* y, int cout, int B,
                            int H, int W, int relu, void* stream);
/* the same, publishing max |y| into an amax slot (fp16-operand modes, below) */
int punet_conv3x3_small_fwd_amax(const float* x, int cin, const float* w, const float* bias, float* y, int cout, int B, int H,
                                 int W, int relu, float* amax_y, void* stream);
size_t punet_small_wgrad_workspace_floats(int cin, int cout);
int punet_conv3x3_small_wgrad(const float* x, int cin, const float* dz, int cout, float* dw, float* dbias,
                              float* workspace, int B, int H, int W, int accumulate, float scale, void* stream);

/* ---- K4: nn.AvgPool2d(2, 2, ceil_mode=True) (unet_blocks.py:17, probabilistic_unet.py:54) and backward
 *      (dx (+)= (xmask > 0) * dy / window). */
int punet_pool2x2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int punet_pool2x2_bwd(const float* dy, const float* xmask, float* dx, int B, int H, int W, int C, int accumulate,
                      void* stream);

/* ---- K5: F.interpolate(bilinear, scale_factor=2, align_corners=True) (unet_blocks.py:51) and backward (gather form,
 *      dx = (xmask > 0) * ...). H, W are the INPUT sizes. */
int punet_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int punet_upsample2x_bwd(const float* dy, const float* xmask, float* dx, int B, int H, int W, int C, void* stream);
/* K4/K5 backward with the ReLU gate of x given as bits (punet_gate_dwords layout, as published by punet_conv3x3_x6_gated);
 * xbits != NULL takes precedence over xmask. */
int punet_pool2x2_bwd_gated(const float* dy, const float* xmask, const void* xbits, float* dx, int B, int H, int W, int C,
                            int accumulate, void* stream);
int punet_upsample2x_bwd_gated(const float* dy, const float* xmask, const void* xbits, float* dx, int B, int H, int W, int C,
                               void* stream);

/* ---- K5x: the first convolution of an UpConvBlock, y = ReLU(conv3x3(cat[up(x), bridge]) + b) (unet_blocks.py:51-57), with
 *      the `up` source contracted at LOW resolution: bilinear interpolation (per channel) and the convolution's channel
 *      contraction (per pixel) commute, W_t . up(x) = up(W_t . x), so
 *          conv3x3(up(x))[p] = sum_t up(z_t)[p + t - 1],   z = 1x1 convolution of x with the [cup][9 cout] matrix w9
 *      - a quarter of the multiply-adds of that half of the layer, and the up-sampled tensor is never materialised.
 *      punet_upconv_relayout: the slab w[3][3][cup + cb][cout] -> w9[cup][9 cout] (w9[ci][t cout + co] = w[t][ci][co]) and the
 *        bridge rows as a contiguous slab wb[3][3][cb][cout] (both then go through punet_split_weights_x6).
 *      punet_upconv_gather_fwd: y <- act(y + sum_t up(z_t)[. + t - 1]) in place, y [B][2h][2w][cout] holding the bridge half's
 *        convolution incl. bias, z [B][h][w][9][cout]; zero padding at the high resolution as in the reference; gate_out
 *        (nullable, cout % 64 == 0) receives (y > 0) in the punet_gate_dwords layout.
 *      punet_upconv_gather_bwd: the adjoint, dzt[b][j][t][c] = sum_q up_weight(q, j) dz[b][q - t + 1][c] with dz [B][2h][2w][cout],
 *        dzt [B][h][w][9][cout] (gather form, deterministic); the input gradient w.r.t. x is then ONE 1x1 input gradient of dzt
 *        with w9.  Interpolation weights are K5's. */
/* where K5x applies (one rule for every graph walker) and the `up`-channel count below which a forward pass with a backward
 * behind it keeps the layer as written */
int punet_upconv_eligible(int cup, int cb, int cout, int ksize);
int punet_upconv_min_cup_grad(void);
int punet_upconv_relayout(const float* w, int cup, int cb, int cout, float* w9, float* wb, void* stream);
int punet_upconv_gather_fwd(const float* z, float* y, void* gate_out, int relu, int B, int h, int w, int cout, void* stream);
int punet_upconv_gather_bwd(const float* dz, float* dzt, int B, int h, int w, int cout, void* stream);
/*      punet_upconv_wgrad_x6: the weight gradient of the `up` rows in the same form, dw[t][ci][co] (+)= scale * sum_j x[j][ci]
 *        dzt[j][t][co] for ci < cup (rows [0, cup) of every tap's [cin_total][cout] block; six-product arithmetic, deterministic
 *        split-K: nslices from punet_upconv_wgrad_slices, workspace of punet_upconv_wgrad_workspace_floats floats) - a quarter
 *        of the 3x3 weight gradient's multiply-adds, and the up-sampled tensor is needed by nothing any more. */
int punet_upconv_wgrad_slices(int cup, int cout, int B, int h, int w);
size_t punet_upconv_wgrad_workspace_floats(int cup, int cout, int nslices);
int punet_upconv_wgrad_x6(const float* x, int cup, const float* dzt, int cout, float* dw, int cin_total, float* workspace,
                          int nslices, int B, int h, int w, int accumulate, float scale, void* stream);

/* ---- K6: latent head (probabilistic_unet.py:126-137): mean over H, W then 1x1 conv C -> J = 2*latent_dim.
 *      muls[b][:L] = mu, muls[b][L:] = log_sigma.  mean[b][C] is kept for the backward. */
size_t punet_chan_sum_workspace_floats(int B, int P, int C);
int punet_chan_sum(const float* x, float* out, float* workspace, int B, int P, int C, float scale, void* stream);
int punet_latent_head_fwd(const float* enc, const float* w, const float* bias, float* mean, float* muls,
                          float* workspace, int B, int P, int C, int J, void* stream);
int punet_latent_head_bwd(const float* d_muls, const float* mean, const float* w, const float* enc, float* dw, float* db,
                          float* dmean, float* denc, int B, int P, int C, int J, int accumulate, void* stream);
/* z[n][b][:] = mu[b] + exp(log_sigma[b]) * eps[n][b]   (Independent(Normal).rsample, probabilistic_unet.py:141,302,349) */
int punet_reparam(const float* muls, const float* eps, float* z, int N, int B, int L, void* stream);
/* kl[0] = mean_b KL(q_b || p_b)  (probabilistic_unet.py:332,351-353) */
int punet_latent_kl(const float* muls_q, const float* muls_p, float* kl, int B, int L, void* stream);
/* d(coef * B * kl)/d muls + reparameterisation gradient of the upstream dz; coef = gscale * beta / B */
int punet_latent_kl_bwd(const float* muls_q, const float* muls_p, const float* eps, const float* dz, float coef,
                        float* d_muls_q, float* d_muls_p, int B, int L, void* stream);

/* per-image KL kl_rows[B] (nullable) and its batch mean kl_mean[0] (probabilistic_unet.py:324-339,351): z == NULL -> analytic
 * (torch.distributions.kl, :332); z[B][L] != NULL -> the single-sample estimate log q(z) - log p(z) (:334-338). */
int punet_latent_kl_rows(const float* muls_q, const float* muls_p, const float* z, float* kl_rows, float* kl_mean, int B,
                         int L, void* stream);
/* punet_latent_kl_bwd with elbo()'s two options (probabilistic_unet.py:341-358): kl_sampled != 0 <-> analytic_kl=False
 * (estimate at z = mu_q + sigma_q * eps); z_is_mean != 0 <-> reconstruct_posterior_mean=True (dz reaches mu_q only). */
int punet_latent_kl_bwd_ex(const float* muls_q, const float* muls_p, const float* eps, const float* dz, float coef,
                           int kl_sampled, int z_is_mean, float* d_muls_q, float* d_muls_p, int B, int L, void* stream);
/* loc[b][d] = mu, scale[b][d] = exp(log_sigma): the Normal(loc, scale) the reference stores (probabilistic_unet.py:136-141) */
int punet_latent_split(const float* muls, float* loc, float* scale, int B, int L, void* stream);

/* ---- layout / concat helpers at the API edge.  torch.cat([patch, segm], 1) of the posterior's input
 *      (probabilistic_unet.py:115-118) on NHWC tensors with ca, cb channels; NCHW <-> NHWC of the caller's patches. */
int punet_concat_nhwc(const float* a, int ca, const float* b, int cb, float* out, long P, void* stream);
int punet_nchw_to_nhwc(const float* x, float* y, int B, int C, long HW, void* stream);
int punet_nhwc_to_nchw(const float* x, float* y, int B, int C, long HW, void* stream);

/* ---- FixMatch distribution alignment (fixmatch_trainer.py:77-84): target class frequencies of (y >= 0.5) counted exactly
 *      on the device (no torch.unique host sync), ratio = source / target, out = clip(y * ratio[class], 0, 1);
 *      ratio[2] (nullable) receives the two ratios; count_ws = 8 bytes of scratch. */
int punet_align_distribution(const float* y, const float* source_distribution, float* out, float* ratio,
                             unsigned long long* count_ws, long P, void* stream);

/* ---- Fcomb pieces (probabilistic_unet.py:200-214) used by the single-sample training path */
int punet_fcomb_zbias(const float* z, const float* w1, const float* b1, float* zb, int NB, int L, int CH, void* stream);
/* backward of the folded z term: S[b][CH] = per-image column sums of dZ1 (punet_chan_sum) */
int punet_fcomb_z_bwd(const float* S, const float* z, const float* w1, float* dw1, float* dz, int B, int L, int CH,
                      int accumulate, void* stream);
int punet_dot_bias(const float* h, const float* w3, const float* b3, float* logits, long P, int C, void* stream);
size_t punet_dot_bwd_workspace_floats(int C);
int punet_dot_bwd(const float* h, const float* dlogit, const float* w3, float* dz, float* dw3, float* db3,
                  float* workspace, long P, int C, int accumulate, void* stream);

/* ---- K7: persistent latent-sampler + Fcomb + consensus kernel (mean_teacher_trainer.py:72-88, fixmatch_trainer.py:37-54,
 *      punet_predictions.py:29-33,108-124).  zb[N][B][CH] from punet_fcomb_zbias.  y = mean_n sigmoid(logit_n);
 *      z = fraction of samples with sigmoid >= upper or <= lower (masking = 0) or its (== 1) indicator (masking = 1);
 *      logits[N][P] optional.  CH = num_filters[0] in {32, 64}; nmid = no_convs_fcomb - 2 middle layers. */
int punet_fcomb_sample(const float* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                       const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z,
                       float* logits, int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking,
                       void* stream);

/* The same kernel in the split-plane arithmetic of K1x6 (three bf16 planes per operand, six products on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation: fp32-accurate): 2.67x less matrix-pipe time.  Same arguments. */
int punet_fcomb_sample_x6(const float* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                          const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z,
                          float* logits, int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking,
                          void* stream);

/* ---- K7t: the single-sample Fcomb of the TRAINING path in one persistent kernel: forward (probabilistic_unet.py:200-214 with
 *      the z term folded into zb[B][CH]), BCE criterion with the consensus mask (:360-369) and the whole backward - dfeat
 *      (gated by feat > 0, i.e. the gradient w.r.t. the pre-activation of the U-Net's last conv) and, as per-workgroup slabs in
 *      `workspace`, every weight / bias gradient.  punet_fcomb_train_reduce then sums the slabs into the gradient tensors
 *      (w1 has w1_rows = CH + latent_dim rows, rows [0, CH) are written) and produces S[B][CH], the per-image column sums
 *      that punet_fcomb_z_bwd takes.  Supported: CH in {32, 64}, at most one middle layer (no_convs_fcomb <= 3), H*W % 32 == 0;
 *      otherwise (and for the Dice criterion) the caller uses the per-op entry points above. */
int punet_fcomb_train_supported(int CH, int nmid, int HW);
size_t punet_fcomb_train_workspace_floats(int B, int HW, int CH, int nmid);
int punet_fcomb_train_fwd_bwd(const float* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                              const float* w3, const float* b3, const float* target, const float* consm, float gscale,
                              float* logits, float* dfeat, float* loss_sum, float* workspace, int B, int HW, int CH, int nmid,
                              void* stream);
/* the same with the Dice criterion of rl_swap=True (probabilistic_unet.py:347; every LIVECell / MitoEM script), recalled torch_em
 * definition [parity unpinned]: the ratio is global, so the fused kernel runs twice (forward + partial sums, then forward +
 * backward with the reduced sums); loss_out[0] = 1 - 2 sum(p t) / max(sum p^2 + sum t^2, 1e-7).  Same workspace, fp32 features. */
int punet_fcomb_train_fwd_bwd_dice(const float* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                                   const float* w3, const float* b3, const float* target, const float* consm, float gscale,
                                   float* logits, float* dfeat, float* loss_out, float* workspace, int B, int HW, int CH, int nmid,
                                   void* stream);
int punet_fcomb_train_reduce(const float* workspace, float* dw1, int w1_rows, float* db1, float* dwmid, float* dbmid, float* dw3,
                             float* db3, float* S, float* chan_workspace, int B, int HW, int CH, int nmid, int accumulate,
                             void* stream);

/* ---- "bf16s": the opt-in reduced-precision fast mode with bf16 STORAGE of the activations on the no-gradient forward path
 *      (inference, teacher / weak pass).  [B,H,W,C] bf16 tensors; one bf16 MFMA per product; fp32 accumulate,
 *      bias, interpolation weights and latent head.  NOT a parity mode. */
int punet_conv3x3_small_fwd_bf16s(const float* x, int cin, const float* w, const float* bias, void* y, int cout, int B, int H,
                                  int W, int relu, void* stream);                       /* fp32 image in, bf16 out */
int punet_conv3x3_bf16s(const void* x0, int c0, const void* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                        int n_off, const float* bias, void* y, int n_out, void* gate_out, void* pool_out, int relu, int B, int H,
                        int W, void* stream);                        /* gate_out, pool_out nullable (gate bits; fused AvgPool2d(2,2)) */
/* student pass of the same mode: weight gradient with the activation operand stored as bf16 (dz fp32), the fused Fcomb training
 * kernel reading bf16 features, the latent head's backward gated by the encoder output's gate BITS instead of its fp32 values */
int punet_conv_wgrad_bf16s(const void* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off, float* dbias,
                           float* workspace, int nslices, int B, int H, int W, int accumulate, float scale, void* stream);
int punet_fcomb_train_fwd_bwd_bf16s(const void* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                                    const float* w3, const float* b3, const float* target, const float* consm, float gscale,
                                    float* logits, float* dfeat, float* loss_sum, float* workspace, int B, int HW, int CH,
                                    int nmid, void* stream);
/* gate bits (punet_gate_dwords layout) of a bf16 [B,H,W,C] activation: (y > 0) */
int punet_gate_bits_bf16(const void* y, void* bits, int B, int H, int W, int C, void* stream);
int punet_latent_head_bwd_bits(const float* d_muls, const float* mean, const float* w, const void* enc_bits, float* dw, float* db,
                               float* dmean, float* denc, int B, int H, int W, int C, int J, int accumulate, void* stream);
int punet_upsample2x_fwd_bf16s(const void* x, void* y, int B, int H, int W, int C, void* stream);
int punet_latent_head_fwd_bf16s(const void* enc, const float* w, const float* bias, float* mean, float* muls, float* workspace,
                                int B, int P, int C, int J, void* stream);
int punet_fcomb_sample_x6_bf16s(const void* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                                const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z,
                                float* logits, int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking,
                                void* stream);

/* ---- fp16-operand fast modes (the reference scripts' mixed_precision=True, LIVECell/livecell_mt.py:107, is fp16
 *      autocast + GradScaler).  "f16x3": every fp32 operand = two fp16 planes h + l, products hh + hl + lh on
 *      v_mfma_f32_*_f16 with fp32 accumulation (22-23 significand bits at half the matrix-pipe time of the six-product parity
 *      arithmetic); "f16": one plane (11 bits, what autocast computes with).  fp16 has 5 exponent bits, so every tensor is
 *      brought into range by an EXACT power of two from its amax: an amax slot is one device float holding (a bound of) max |v|,
 *      published by the producing kernel with an order-independent atomic max (bit-reproducible); the consumer scales by
 *      2^(14 - floor(log2 amax)) before the split and the epilogue rescales by the exact inverse - the per-tensor counterpart of
 *      the reference's one global loss scale, with no overflow and no skipped steps.  NOT parity modes.
 *      Weight planes: punet_split_weights_f16[_batch] = amax per slab (slot j of wamax for table row j) + planes in the layout
 *      of punet_split_weights_x6 (16-bit elements, planes 0 and 1).  amax_* arguments may be NULL (scale 1 / not published);
 *      a slot that receives a publication must have been zeroed (or hold an earlier bound of the same tensor). */
int punet_split_weights_f16(const float* w, void* planes, long plane_stride, int cin, int cout, int mode, float* amax_w,
                            void* stream);
int punet_split_weights_f16_batch(const float* flat, void* planes, const long* table, int nconv, long total_units, int mode,
                                  float* wamax, int compute_amax, void* stream);
/* forward (bias / relu / gate_out / pool_out) or input gradient (mask | gate_bits, accumulate) as punet_conv3x3_x6_gated /
 * _pooled; planes = 2 ("f16x3") or 1 ("f16") */
int punet_conv3x3_f16(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                      int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                      const void* gate_bits, void* gate_out, float* pool_out, int accumulate, int relu, int B, int H, int W,
                      int planes, const float* amax_x0, const float* amax_x1, const float* amax_w, float* amax_y, void* stream);
/* weight (+ bias) gradient as punet_conv_wgrad_x6; x16 != 0: x is STORED as fp16 (planes = 1, unscaled) */
int punet_conv_wgrad_f16(const void* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off, float* dbias,
                         float* workspace, int nslices, int B, int H, int W, int accumulate, float scale, int planes, int x16,
                         const float* amax_x, const float* amax_dz, void* stream);
/* slot = max(slot, max |x|) over n floats (x 16-byte aligned; zero_first != 0 clears the slot first) - for tensors whose producer
 * publishes no amax; and a BOUND composed from other slots, out = ca * a + cb * b (a or b may be NULL; out may alias b): e.g. the
 * average-pool backward's output is bounded by |dy| / 4 (+ what it accumulates onto), the x2 up-sampling backward's by 4 |dy| */
int punet_amax(const float* x, long n, float* slot, int zero_first, void* stream);
int punet_amax_bound(const float* a, float ca, const float* b, float cb, float* out, void* stream);

/* "f16s" = "f16" + fp16 STORAGE of the activations ([B,H,W,C] fp16, unscaled as under the reference's autocast, saturating at
 * fp16's largest finite value; gradients stay fp32 and are scaled per tensor): the twins of the punet_*_bf16s entry points above
 * (the weight gradient with a stored-fp16 operand is punet_conv_wgrad_f16(..., planes = 1, x16 = 1, NULL, amax_dz)) */
int punet_conv3x3_small_fwd_f16s(const float* x, int cin, const float* w, const float* bias, void* y, int cout, int B, int H, int W,
                                 int relu, void* stream);
int punet_conv3x3_f16s(const void* x0, int c0, const void* x1, int c1, const void* wplanes, long plane_stride, int n_row, int n_off,
                       const float* bias, void* y, int n_out, void* gate_out, void* pool_out, int relu, int B, int H, int W,
                       const float* amax_w, void* stream);
int punet_fcomb_train_fwd_bwd_f16s(const void* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                                   const float* w3, const float* b3, const float* target, const float* consm, float gscale,
                                   float* logits, float* dfeat, float* loss_sum, float* workspace, int B, int HW, int CH, int nmid,
                                   void* stream);
int punet_gate_bits_f16(const void* y, void* bits, int B, int H, int W, int C, void* stream);
int punet_upsample2x_fwd_f16s(const void* x, void* y, int B, int H, int W, int C, void* stream);
int punet_latent_head_fwd_f16s(const void* enc, const float* w, const float* bias, float* mean, float* muls, float* workspace, int B,
                               int P, int C, int J, void* stream);
int punet_fcomb_sample_x6_f16s(const void* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                               const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z,
                               float* logits, int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking,
                               void* stream);

/* ---- K8: the reduction half of K7 on given logits[N][P] (unit parity of the consensus rule) */
int punet_consensus(const float* logits, int N, long P, float upper, float lower, int masking, float* y, float* z,
                    void* stream);

/* ---- K9: reconstruction loss forward + d/dlogit (probabilistic_unet.py:347-369).  consm (nullable, float) multiplies
 *      BOTH logits and targets as :363-364 does.  dlogit = gscale * dloss/dlogit.
 *      BCE: loss_sum[0] = sum of BCEWithLogits(reduction none).  Dice: [recalled torch_em definition, parity unpinned]. */
size_t punet_loss_workspace_floats(void);
int punet_bce_fwd_bwd(const float* logits, const float* target, const float* consm, float gscale, float* dlogit,
                      float* loss_sum, float* workspace, long P, void* stream);
int punet_dice_fwd_bwd(const float* logits, const float* target, const float* consm, float gscale, float* dlogit,
                       float* loss_out, float* workspace, long P, void* stream);

/* ---- E1: dice_score sums (my_utils/util.py:17-44): sums3 = [sum(gt*seg), sum(gt), sum(seg)] */
int punet_dice_score_sums(const float* seg, const float* gt, float thr_seg, int use_thr_seg, float thr_gt, int use_thr_gt,
                          float* sums3, float* workspace, long P, void* stream);

/* ---- K10: l2_regularisation (utils.py:32-40) over a chunk table {offset, length, tensor index} (int64 triples) */
int punet_l2_reg(const float* w, const int64_t* chunks, int nchunks, int ntensors, float* partial, float* norms,
                 float* reg_out, void* stream);
int punet_l2_reg_add_grad(const float* w, float* g, const int64_t* chunks, int nchunks, const float* norms, float coef,
                          void* stream);

/* ---- K11 + K12: torch.optim.Adam step (LIVECell/livecell_mt.py:83 defaults) fused with the EMA teacher update
 *      (mean_teacher_trainer.py:52-55; adamt_trainer.py:40-43) on flat buffers.  teacher may be NULL (no EMA);
 *      do_adam = 0 performs the EMA alone.  The gradient is read as g * gscale (1/world_size after all-reduce). */
int punet_adam_ema(float* p, const float* g, float* m, float* v, float* teacher, long n, float gscale, float lr,
                   float beta1, float beta2, float eps, int step, float ema_momentum, int do_adam, void* stream);
/* The same step with its step-dependent scalars taken from DEVICE memory, so that a step captured in a hipGraph replays
 * correctly: punet_adam_hyper advances the state block hstate (double[8]: Adam steps taken, lr, beta1, beta2, EMA momentum,
 * EMA rule 0 none | 1 constant (mean_teacher_trainer.py:52-55) | 2 warm-up min(1 - 1/(it+1), m) (adamt_trainer.py:41),
 * trainer iteration it, gradient scale) and writes hyper (float[4]: gradient scale, lr / (1 - beta1^t), sqrt(1 - beta2^t),
 * this step's EMA momentum); punet_adam_ema_dev reads hyper.  teacher == NULL: no EMA. */
int punet_adam_hyper(double* hstate, float* hyper, void* stream);
int punet_adam_ema_dev(float* p, const float* g, float* m, float* v, float* teacher, long n, const float* hyper, float beta1,
                       float beta2, float eps, void* stream);
int punet_scale(float* x, long n, float s, void* stream);

/* ---- K13: weak / strong view generation on NCHW fp32 planes (the transform chains of LIVECell/livecell_fm.py:43-67 and
 *      MitoEM/common.py:50-68: my_standardize_torch (prob_utils/my_utils/util.py:9-14), then randomly applied Gaussian
 *      blur, additive Gaussian noise, contrast).  The host draws the random decisions; the device does the arithmetic.
 *      stats[b] = {mean, Bessel-corrected std} over the n = C*H*W values of item b (fp64 accumulation).
 *      params[b] = {kernel size (odd, 1 = no blur), sigma, noise scale (0 = none), contrast alpha (1 = none)}.
 *      out = cmean + alpha * ((blur((x - mean) / (std + eps)) + noise_scale * noise) - cmean); stats == NULL skips the
 *      standardisation, noise == NULL the noise.  Blur: separable, taps exp(-0.5 (i/sigma)^2) / sum, reflect padding.
 *      tmp: B*C*H*W floats, must not alias x or out. */
size_t punet_standardize_workspace_doubles(int B);
int punet_standardize_stats(const float* x, float* stats, double* workspace, int B, long n, void* stream);
int punet_augment_view(const float* x, const float* stats, const float* noise, const float* params, int max_ksize,
                       float contrast_mean, float eps, float* tmp, float* out, int B, int C, int H, int W, void* stream);

/* ==== Step-level runtime (csrc/runtime.hip): the graph walk as native host code =========================================
 * The entry points above replace single ATen ops; these replace the METHODS the reference's trainers call - a host in any
 * language drives a self-training step without knowing the layer graph:
 *     punet_rt_pseudo_labels   = teacher.forward(x1, None, False) + N x sigmoid(teacher.sample()) + consensus
 *                                (mean_teacher_trainer.py:72-88, fixmatch_trainer.py:37-54, punet_predictions.py:29-33)
 *     punet_rt_elbo_fwd_bwd    = model.forward(x2, y, True); model.elbo(y, z); loss.backward()
 *                                (probabilistic_unet.py:285-293, 341-371; mean_teacher_trainer.py:113-119)
 *     punet_rt_l2_reg_grad     = the three l2_regularisation terms and their gradient (utils.py:32-40, mean_teacher_trainer.py:115-117)
 *     punet_adam_hyper + punet_adam_ema_dev (above) = optimizer.step() + _momentum_update()
 * Covered: num_classes = 1, num_filters[0] in {32, 64}, no_convs_fcomb in {2, 3}, filters multiples of 8, input_channels <= 6,
 * BCE or Dice criterion, the six-product parity arithmetic or the fp16-operand fast modes - every configuration the reference's
 * scripts build.
 * Parameters: ONE flat fp32 buffer in the reference's state-dict order, conv weights [ky][kx][cin][cout] (punet_rt_param_entry
 * lists name / offset / logical shape of every tensor).  The runtime owns NOTHING on the device: every call carves its
 * activations, gate bits, gradients, weight planes and split-K slabs out of a caller-provided ARENA (punet_rt_arena_bytes) and
 * reads two small index tables from a caller-provided TABLES buffer (punet_rt_tables_bytes / punet_rt_init_tables, once).  The
 * arena's BASE must be 256-byte aligned (hipMalloc'ed memory is; a host that sub-allocates must round up): the runtime aligns the
 * offsets of what it carves, the kernels read it with 16-byte vector loads, and a misaligned base is refused (PUNET_ERR_ARG).  No
 * allocation, no synchronisation, no host read inside a call: each is hipGraph-capturable.  x is NHWC fp32 ([B][H][W][C]; for
 * C = 1 the same bytes as NCHW); target / consm / y / z / logits are [B][1][H][W].  Not thread-safe per handle. */
typedef struct PunetRt PunetRt;
typedef struct {
    int input_channels, num_classes, levels, latent_dim, no_convs_fcomb;
    int num_filters[8];
} PunetRtConfig;
int punet_rt_create(const PunetRtConfig* cfg, PunetRt** out);
void punet_rt_destroy(PunetRt* rt);
/* arithmetic of the 3x3 convolutions of every later call: 6 = six-product parity arithmetic (default), 3 = "f16x3", 1 = "f16"
 * (the fp16-operand fast modes above: amax slots are carved from the arena and handled inside the runtime) */
int punet_rt_set_arithmetic(PunetRt* rt, int products);
/* criterion of every later punet_rt_elbo_fwd_bwd: 0 = BCEWithLogits (rl_swap=False), 1 = Dice (rl_swap=True; recalled torch_em
 * definition, parity unpinned) */
int punet_rt_set_criterion(PunetRt* rt, int dice);
/* K5x for every later call on this handle: 1 (default) = the up-blocks' entry convolutions contract their `up` source at low
 * resolution (punet_upconv_* above; the six-product parity arithmetic only), 0 = as written, on the materialised up-sampled
 * tensor (an A/B switch; results differ by summation order only). */
int punet_rt_set_upconv(PunetRt* rt, int on);
long punet_rt_param_floats(const PunetRt* rt);
long punet_rt_unet_floats(const PunetRt* rt); /* [0, this) of the flat buffers = the U-Net's tensors (final first in the backward) */
int punet_rt_num_entries(const PunetRt* rt);
int punet_rt_param_entry(const PunetRt* rt, int i, char* name, int cap, long* off, long* numel, int* shape4);
size_t punet_rt_tables_bytes(const PunetRt* rt);
int punet_rt_init_tables(PunetRt* rt, void* device_tables, void* stream);
size_t punet_rt_arena_bytes(PunetRt* rt, int B, int H, int W, int N, int training); /* exact: a dry walk of the call's graph */
size_t punet_rt_last_arena_used(const PunetRt* rt);
int punet_rt_pseudo_labels(PunetRt* rt, const float* flat, const float* x, int B, int H, int W, const float* eps, int N, float upper,
                           float lower, int masking, float* y, float* z, float* logits, void* arena, size_t arena_bytes, void* stream);
int punet_rt_elbo_fwd_bwd(PunetRt* rt, const float* flat, float* gflat, const float* x, const float* target, const float* consm,
                          const float* eps, int B, int H, int W, float beta, float gscale, int accumulate, float* rec_out,
                          float* kl_out, float* logits_out, void (*after_unet)(void*), void* ctx, void* arena, size_t arena_bytes,
                          void* stream);
int punet_rt_l2_reg_grad(PunetRt* rt, const float* flat, float* gflat, float coef, float* reg_out, void* arena, size_t arena_bytes,
                         void* stream);
/* ONE call per self-training iteration (MeanTeacherTrainer._train_epoch_impl, mean_teacher_trainer.py:101-131; FixMatchTrainer with
 * labeller == student and ema == 0): pseudo-labels from `labeller` on view x1, ELBO forward / backward of `student` on view x2,
 * regulariser, Adam with the EMA of the teacher fused in.  hstate / hyper: the device state block of punet_adam_hyper (lr, step
 * count, EMA rule, gradient scale 1 / world_size live there).  scalars[3] (device) = {reconstruction loss, kl, reg};
 * loss = scalars[0] + beta * scalars[1] + reg_weight * scalars[2].  hooks (nullable) let a data-parallel host enqueue its
 * all-reduces: after_unet - the slice [0, punet_rt_unet_floats) of gflat is final and enqueued; before_update - all of it is. */
typedef struct {
    void (*after_unet)(void* ctx);
    void (*before_update)(void* ctx);
    void* ctx;
} PunetRtHooks;
int punet_rt_self_training_step(PunetRt* rt, float* student, float* labeller, float* gflat, float* exp_avg, float* exp_avg_sq,
                                double* hstate, float* hyper, int ema, const float* x1, const float* x2, int B, int H, int W,
                                const float* eps_pl, int N, const float* eps_s, float upper, float lower, int masking, int use_consensus,
                                float beta, float reg_weight, float adam_beta1, float adam_beta2, float adam_eps, float* y, float* z,
                                float* scalars, const PunetRtHooks* hooks, void* arena, size_t arena_bytes, void* stream);
/* the other trainers, one call per iteration each: supervised (PUNetTrainer / PseudoTrainerPUNet, punet_trainer.py:24-36,
 * punet_target_trainer.py:105-118; z nullable; scalars[3] = {rec, kl, reg}) and joint (AdaMTTrainer / AdaMatchTrainer,
 * adamt_trainer.py:89-128, adamatch_trainer.py:62-107: source + target ELBO at scale 1/2 each, one optimizer step;
 * scalars[5] = {rec_src, kl_src, rec_tgt, kl_tgt, reg}) */
int punet_rt_supervised_step(PunetRt* rt, float* model, float* gflat, float* exp_avg, float* exp_avg_sq, double* hstate, float* hyper,
                             const float* x, const float* y, const float* z, int B, int H, int W, const float* eps, float beta,
                             float reg_weight, float adam_beta1, float adam_beta2, float adam_eps, float* scalars,
                             const PunetRtHooks* hooks, void* arena, size_t arena_bytes, void* stream);
int punet_rt_joint_step(PunetRt* rt, float* student, float* labeller, float* gflat, float* exp_avg, float* exp_avg_sq, double* hstate,
                        float* hyper, int ema, const float* xs, const float* ys, int Bs, const float* xt1, const float* xt2, int B, int H,
                        int W, const float* eps_sup, const float* eps_pl, int N, const float* eps_s, float upper, float lower, int masking,
                        int use_consensus, float beta, float reg_weight, float adam_beta1, float adam_beta2, float adam_eps, float* y,
                        float* z, float* scalars, const PunetRtHooks* hooks, void* arena, size_t arena_bytes, void* stream);

#ifdef __cplusplus
}
#endif
