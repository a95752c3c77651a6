/*
 * savfi_hip.h -- C ABI of libsavfi_hip.so, the MI355X (gfx950) kernels behind the
 * MAML inner-loop adaptation path of scene-adaptive video frame interpolation.
 *
 * This header is the drop-in boundary.  Each entry point replaces one piece of the
 * reference's hot path (file:line are relative to the reference checkout):
 *
 *   savfi_sepconv_fwd_f32      kernel_Sepconv_updateOutput            sepconv/sepconv_op/sepconv.py:5-30, launch :280-291
 *   savfi_sepconv_bwd_f32      kernel_Sepconv_updateGradVertical      sepconv/sepconv_op/sepconv.py:138-163, launch :343-356
 *                              kernel_Sepconv_updateGradHorizontal    sepconv/sepconv_op/sepconv.py:165-190, launch :358-371
 *                              kernel_Sepconv_updateGradInput         sepconv/sepconv_op/sepconv.py:32-63,  launch :328-341
 *   savfi_voxelwarp_fwd_f32    flow/mask split + meshgrid + 2x grid_sample + blend
 *   savfi_voxelwarp_bwd_f32                                          voxelflow/core/models/voxel_flow.py:471-509, :9-17
 *   savfi_avgpool2x2_fwd/bwd_f32  2x2 average pooling                 sepconv/model.py:176-187, rrin/unet.py:146, superslomo/model.py:66
 *   savfi_flowwarp_fwd/bwd_f32 pixel-flow backward warp              superslomo/model.py:231-307, rrin/model.py:8-20
 *   savfi_pixel_unshuffle_f32  pixel_shuffle(scale<1)                 model_utils.py:202-217 (else branch)
 *   savfi_pixel_shuffle_f32    pixel_shuffle(scale>=1)                model_utils.py:202-217 (if branch)
 *   savfi_mt_update_f32        LSLR / Meta-SGD update_sgd/adam/adamax inner_loop_optimizers.py:136-244, :324-425
 *   savfi_mt_update_bwd_f32    (autograd of the above w.r.t. the learning rates)
 *   savfi_mt_mean_f32          per-tensor mean of grads (L2F embedding) meta_learning_system.py:249-253
 *   savfi_mt_scale_f32         gamma_i * w_i (L2F attenuation)        meta_learning_system.py:267-268
 *   savfi_l1_mse_f32           nn.L1Loss / nn.MSELoss                 loss.py:287-290
 *   savfi_upsample2x_fwd/bwd_f32  bilinear x2 up-sampling                 sepconv/model.py:191,213-234; voxel_flow.py:400-414
 *   savfi_upsample2x_window_fwd/bwd_f32  the same map on a window (SepConv Subnets on the frame area)  sepconv/model.py:309-349
 *   savfi_bias_act_fwd/bwd_f32 conv bias add + (Leaky)ReLU and their backward + bias gradient
 *                                                                     sepconv/model.py:172-194, model_utils.py:957-990
 *   savfi_conv3x3_f32          F.conv2d 3x3 / stride 1 (+ bias, activation) and its data gradient
 *   savfi_conv3x3_wgrad_f32    its weight gradient                    model_utils.py:308-366 (MetaConv2dLayer.forward -> F.conv2d)
 *   savfi_conv3x3_tasks_f32, savfi_conv3x3_wgrad_tasks_f32   the same for T tasks with their own fast weights in ONE launch
 *   savfi_conv3x3_filters_f32, savfi_conv3x3_tasks_pre_f32   filter transforms of forward + data gradient in one launch / convolution on a transformed filter
 *                              (the sequential task loop meta_learning_system.py:366 run in lockstep)
 *   savfi_convk_filters_f32, savfi_convk_tasks_pre_f32, savfi_convk_wgrad_tasks_f32   F.conv2d K x K (K = 3, 5, 7) / stride 1 and its data gradient as a direct
 *                              implicit GEMM on the bf16 matrix cores from error-free 3-way operand splits (fp32-equivalent)
 *                                                                     voxelflow/core/models/voxel_flow.py:357-470 (5x5 layers),
 *                                                                     superslomo/model.py:547-646 (7x7 / 5x5), model_utils.py:308-366
 *   savfi_ca_pool/mlp_fwd/mlp_bwd/apply_f32   channel attention + residual of CAIN's RCAB   model_utils.py:931-953, :957-990
 *   savfi_frames_u8_to_f32     HWC uint8 frames -> normalised fp32 NCHW  data/vimeo_septuplet.py:68-80, data/video.py:44-51
 *   savfi_*_workspace_floats / savfi_bias_act_scratch_floats: sizes of the caller-owned scratch buffers (return int64_t)
 *
 * Conventions (all functions):
 *   - extern "C", return int: 0 = ok; >0 = hipError_t reported by the launch;
 *     <0 = argument error (SAVFI_E_*).  No exceptions cross the boundary.
 *   - every pointer is a DEVICE pointer to a contiguous fp32 NCHW buffer owned by the
 *     caller, unless the parameter is documented as a host array.  The library never
 *     allocates device memory, never synchronises, and is re-entrant.
 *   - `stream` is a hipStream_t passed as void* (the caller passes
 *     torch.cuda.current_stream().cuda_stream); launches are asynchronous on it.
 */
#ifndef SAVFI_HIP_H_
#define SAVFI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVFI_ABI_VERSION 20

#define SAVFI_OK            0
#define SAVFI_E_NULL       (-1)  /* a required pointer is NULL                          */
#define SAVFI_E_SHAPE      (-2)  /* a dimension is <= 0 or inconsistent                  */
#define SAVFI_E_UNSUPPORTED (-3) /* e.g. scale factor / rule id not implemented          */
#define SAVFI_E_TOOBIG     (-4)  /* index arithmetic would overflow 32-bit element index */

/* ABI version of the loaded library (== SAVFI_ABI_VERSION it was built against). */
int savfi_version(void);

/* ------------------------------------------------------------------------------------
 * Separable local convolution (SepConv), K taps per axis (K = 51 in the model).
 *   in  [B, C, Ho+K-1, Wo+K-1]   v, h [B, K, Ho, Wo]   out [B, C, Ho, Wo]
 *   out[b,c,y,x] = sum_{fy<K} sum_{fx<K} in[b,c,y+fy,x+fx] * v[b,fy,y,x] * h[b,fx,y,x]
 * ---------------------------------------------------------------------------------- */
int savfi_sepconv_fwd_f32(const float* in, const float* v, const float* h, float* out,
                          int B, int C, int Ho, int Wo, int K, void* stream);

/* Gradients of the above.  Any of gI / gV / gH may be NULL (= not needed, like
 * needs_input_grad in sepconv.py:319-321).  gI is the mathematically exact adjoint
 * (the reference's kernel has an off-by-one bounds test, sepconv.py:51,54).
 *   gO [B,C,Ho,Wo]  gI [B,C,Ho+K-1,Wo+K-1]  gV, gH [B,K,Ho,Wo]
 * Outputs are fully overwritten (no pre-zeroing needed). */
int savfi_sepconv_bwd_f32(const float* in, const float* v, const float* h, const float* gO,
                          float* gI, float* gV, float* gH,
                          int B, int C, int Ho, int Wo, int K, void* stream);

/* The same op on tap tensors that are slices of ONE interleaved buffer: sample b of v / h (and of gV / gH) starts tap_bstride planes of
 * Ho * Wo floats after sample b - 1 (tap_bstride = K: the contiguous layout of the two entry points above).  The package's SepConv plugin
 * evaluates the reference's four Subnets (sepconv/model.py:183-194, :239-242, :346-347) as one task-batched launch per layer; their taps are
 * a [B * 4, K, Ho, Wo] tensor (sample 4 b + s = sub-network s) that the two local convolutions read -- and their filter gradients write --
 * in place with tap_bstride = 4 * K.  K = 51, C = 3, Wo % 4 == 0, every tensor below 2^31 bytes; SAVFI_E_UNSUPPORTED otherwise (the
 * caller copies the slices and uses the contiguous entry points).  No gI (frames carry no gradient on this path). */
/* 1 when the strided / frames8 / pair entry points take the problem in this process (shapes, sizes AND the A/B environment switches that
 * make them return SAVFI_E_UNSUPPORTED), 0 otherwise: what a caller asks before it chooses the interleaved tap tensor. */
int savfi_sepconv_taps_strided_supported(int B, int C, int Ho, int Wo, int K, int tap_bstride);
int savfi_sepconv_fwd_taps_strided_f32(const float* in, const float* v, const float* h, float* out, int B, int C, int Ho, int Wo, int K,
                                       int tap_bstride, void* stream);
int savfi_sepconv_bwd_taps_strided_f32(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH, int B,
                                       int C, int Ho, int Wo, int K, int tap_bstride, void* stream);

/* Frames of 8-bit images.  The reference hands the op decoded PNG frames: ToTensor's k / 255 with k = 0..255 (data/vimeo_septuplet.py:24-36;
 * sepconv/model.py:346-347 only replication-pads them).  255 * in is then an integer that one bf16 holds exactly, and an fp32 product
 * in * tap costs three exact bf16 products (k times the tap's three bf16 pieces) instead of six; the sum is scaled by 1/255 once.
 *   savfi_frames8_classify_f32   x [n] floats -> cls [SAVFI_FRAMES8_WORDS] words (device, 16-byte aligned, need not be initialised):
 *                                word i != 0 = classifier workgroup i met an element that is not fl32(k / 255) to within 2 ulp
 *   savfi_sepconv_{fwd,bwd}_frames8_f32   the two strided entry points above with the words of `in` as an extra argument.  BOTH kernel
 *                                variants are launched and the words select one ON THE DEVICE (no host round trip, graph-capture safe):
 *                                all words zero -> the three-product kernel, otherwise the six-product kernel of the entry points above.
 *                                Same results to fp32 rounding for frames that qualify, identical results for frames that do not.
 * tap_bstride = K for contiguous tap tensors.  K = 51, C = 3, Wo % 4 == 0; SAVFI_E_UNSUPPORTED otherwise (use the entry points above).
 * taps_unit16 != 0: v and h are UNIT-MAJOR -- a sample is [Ho][Wo / 16][K][16] instead of [K][Ho][Wo], the layout
 *   savfi_conv3x3_tasks_pre_unit16_f32 writes (Wo % 16 == 0): the 51 taps of 16 neighbouring pixels are one contiguous run of
 *   51 x 64 bytes instead of 64-byte pieces of 51 planes.  The sample stride (tap_bstride planes of Ho * Wo floats) is unchanged, and
 *   gV / gH are written [K][Ho][Wo] (what the producing convolution's gradient kernels read) unless bit 1 is set as well (taps_unit16 = 3,
 *   backward only): then gV / gH are unit-major too -- for a producer whose data gradient reads that layout (savfi_conv3x3_dgrad_in_unit16_f32)
 *   and whose weights need no gradient in this pass. */
#define SAVFI_FRAMES8_WORDS 256
int savfi_frames8_classify_f32(const float* x, int64_t n, unsigned* cls, void* stream);
int savfi_sepconv_fwd_frames8_f32(const float* in, const float* v, const float* h, float* out, const unsigned* cls, int B, int C, int Ho,
                                  int Wo, int K, int tap_bstride, int taps_unit16, void* stream);
int savfi_sepconv_bwd_frames8_f32(const float* in, const float* v, const float* h, const float* gO, float* gV, float* gH,
                                  const unsigned* cls, int B, int C, int Ho, int Wo, int K, int tap_bstride, int taps_unit16, void* stream);

/* The two filter-gradient calls of an interleaved tap tensor as ONE launch (the plugin's tail: sepconv/model.py:346-347 evaluates
 * FunctionSepconv twice, on frame 0 with sub-networks 0 / 1 and on frame 1 with sub-networks 2 / 3, and adds the results -- their backward
 * passes share the cotangent gO).  taps, gtaps [4 B][K][Ho][Wo] (sample 4 b + s = sub-network s: v0, h0, v1, h1); in0, in1 the frames with
 * their own classifier words.  The three-product kernel runs when BOTH frames qualify.  A launch of these kernels has a fixed cost of about
 * 24 us at 256 x 448, so one launch over 2 B samples is faster than two over B; the results are the same bit for bit.  taps_unit16 as above.
 * The forward of the pair the same way: out [B][2][C][Ho][Wo] holds the two local convolutions of every sample, the caller adds them. */
int savfi_sepconv_fwd_pair_frames8_f32(const float* in0, const float* in1, const float* taps, float* out /* [B][2][C][Ho][Wo] */,
                                       const unsigned* cls0, const unsigned* cls1, int B, int C, int Ho, int Wo, int K, int taps_unit16,
                                       void* stream);
int savfi_sepconv_bwd_pair_frames8_f32(const float* in0, const float* in1, const float* taps, const float* gO, float* gtaps,
                                       const unsigned* cls0, const unsigned* cls1, int B, int C, int Ho, int Wo, int K, int taps_unit16,
                                       void* stream);

/* Diagnostic of the wave-specialised filter-gradient kernel (csrc/sepconv_ws.hip): number of bounded in-kernel waits that
 * gave up since the library was loaded on the current device.  0 on a healthy build; > 0 means a launch's numbers are wrong
 * (the kernel never hangs).  Synchronises with the device.  -1: the counter could not be read. */
int savfi_sepconv_ws_errors(void);
/* The same count WITHOUT a device synchronisation.  savfi_sepconv_ws_watch() -- once per device, outside a stream capture -- maps one host
 * word into the device; a wait that gives up also adds to that word (system-scope atomic), which the host sees at the latest when the
 * launch has completed.  savfi_sepconv_ws_errors_peek() reads the word: the caller checks it wherever it has synchronised anyway (the
 * package: after reading an iteration's loss, meta_learning_system.py) and refuses the iteration's numbers when it is non-zero.
 * -1: not armed on this device. */
int savfi_sepconv_ws_watch(void);
int savfi_sepconv_ws_errors_peek(void);
/* Clears the current device's mapped word and returns the count it held (-1: not armed): for a caller that has handled the reported
 * time-out (the product drops the meta-iteration BEFORE its outer optimizer step and raises) and goes on. */
int savfi_sepconv_ws_errors_reset(void);
/* Test hook: the spin limit of the kernels' bounded waits (default 1 << 19 spins of s_sleep 2).  A NEGATIVE limit makes every wait that does
 * not find its flag at once give up -- tests/ provoke the error path with it.  *previous (may be NULL) receives the old limit; launches issued afterwards use the new one. */
int savfi_sepconv_ws_debug_spin_limit(int limit, int* previous);

/* ------------------------------------------------------------------------------------
 * VoxelFlow warp + blend (syn_type 'inter').
 *   frames [B,6,H,W] (I0 = ch 0..2, I1 = ch 3..5), x3 [B,3,H,W] = tanh output
 *   flow = 0.5*x3[:,0:2] (normalised units), mask = 0.5*(1+x3[:,2])
 *   out[b,c] = mask * bilinear(I0, grid - flow) + (1-mask) * bilinear(I1, grid + flow)
 *   with grid = linspace(-1,1), align_corners=True, padding_mode='border'.
 * bwd: g_x3 [B,3,H,W] always written; g_frames [B,6,H,W] may be NULL (frames are data).
 *      When g_frames != NULL it must be zero-filled by the caller (scatter-add).
 * ---------------------------------------------------------------------------------- */
int savfi_voxelwarp_fwd_f32(const float* frames, const float* x3, float* out,
                            int B, int H, int W, void* stream);
int savfi_voxelwarp_bwd_f32(const float* frames, const float* x3, const float* gO,
                            float* g_x3, float* g_frames,
                            int B, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------
 * 2x2 / stride 2 average pooling of `planes` = N*C maps [H,W] -> [H/2,W/2] (floor): torch.nn.AvgPool2d(2, 2)
 * sepconv/model.py:176-187, F.avg_pool2d(x, 2) rrin/unet.py:146, superslomo/model.py:66.
 *   fwd: out[y][x] = (in[2y][2x] + in[2y][2x+1] + in[2y+1][2x] + in[2y+1][2x+1]) / 4
 *   bwd: gin [planes,H,W] fully written: gout[y/2][x/2] / 4, zero in an odd last row / column
 * ---------------------------------------------------------------------------------- */
int savfi_avgpool2x2_fwd_f32(const float* in, float* out, int64_t planes, int H, int W, void* stream);
int savfi_avgpool2x2_bwd_f32(const float* gout, float* gin, int64_t planes, int H, int W, void* stream);
/* the adjoint of "pool AND keep": an activated map y [planes,H,W] feeds the pooling and, unchanged, a skip connection (an encoder block
 * of sepconv/model.py:176-187 + the decoder's `tensorUpsample + tensorConv`); gout [planes,H/2,W/2] and gskip [planes,H,W] are the two
 * consumers' cotangents (either may be NULL), and the (leaky) ReLU derivative of y's producer is applied in the same pass (y NULL: none):
 *   gin = (gout[y/2][x/2] / 4 + gskip) * (y > 0 ? 1 : slope)
 * -- what autograd runs as the pooling's adjoint, an accumulation and ReLU's backward: three element-wise passes. */
int savfi_avgpool2x2_bwd_fused_f32(const float* gout, const float* gskip, const float* y, float slope, float* gin, int64_t planes, int H, int W,
                                   void* stream);

/* ------------------------------------------------------------------------------------
 * Backward warping by a pixel-unit flow (SuperSloMo backWarp superslomo/model.py:231-307, RRIN warp
 * rrin/model.py:8-20: meshgrid + 2*(x/W-0.5) normalisation + F.grid_sample(bilinear, zeros, align_corners=False)).
 *   img [N,C,H,W], flow [N,2,H,W] (u, v in pixels) -> out [N,C,H,W]
 *   out[n,c,y,x] = bilinear(img[n,c]; x + u - 0.5, y + v - 0.5), corners outside the image count as zero
 *   (the half-pixel offset is what the reference's normalisation amounts to under align_corners=False).
 * bwd: gflow [N,2,H,W] = dL/d(u,v), fully written (a gather: deterministic).  No image gradient: the warped
 *      images are network inputs on this path.
 * ---------------------------------------------------------------------------------- */
int savfi_flowwarp_fwd_f32(const float* img, const float* flow, float* out, int N, int C, int H, int W, void* stream);
int savfi_flowwarp_bwd_f32(const float* img, const float* flow, const float* gout, float* gflow,
                           int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------
 * Pixel (un)shuffle with the reference's channel order.
 *   unshuffle: in [B,C,H,W] -> out [B,C*r*r,H/r,W/r], out ch = c*r*r + i*r + j
 *              out[b, c*r*r+i*r+j, y, x] = in[b, c, y*r+i, x*r+j]
 *   shuffle:   in [B,C*r*r,H,W] -> out [B,C,H*r,W*r]   (exact inverse)
 * Dimensions passed are those of `in`.  Each is the other's autograd adjoint.
 * ---------------------------------------------------------------------------------- */
int savfi_pixel_unshuffle_f32(const float* in, float* out, int B, int C, int H, int W, int r, void* stream);
int savfi_pixel_shuffle_f32(const float* in, float* out, int B, int C, int H, int W, int r, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused multi-tensor inner-loop update (one launch per <= SAVFI_MT_MAX_TENSORS tensors;
 * the library chunks internally).  Host arrays of device pointers, length n.
 *
 *   rule: SAVFI_RULE_SGD      out = w - lr * g
 *         SAVFI_RULE_ADAM     m = b1*m + (1-b1)*g ; s = b2*s + (1-b2)*g*g   (m, s updated IN PLACE)
 *                             out = w - (lr/bc1) * m / (sqrt(s)/sqrt(bc2) + eps)
 *         SAVFI_RULE_ADAMAX_LSLR   m = b1*m + (1-b1)*g (in place) ; out = w - (lr/bc1) * m / (|g| + eps)
 *         SAVFI_RULE_ADAMAX_MSGD   out = w - (lr/bc1) * ((1-b1)*g) / (|g| + eps)   (state untouched)
 *       (the two Adamax forms are the reference's as-implemented arithmetic,
 *        inner_loop_optimizers.py:201-244 and :385-425)
 *   lr_mode: SAVFI_LR_SCALAR   lr[i] points at ONE float (LSLR: &table_i[num_step])
 *            SAVFI_LR_ELEMENT  lr[i] points at numel[i] floats (Meta-SGD)
 *   bc1[i], sqrt_bc2[i]: host float arrays with the per-tensor bias corrections
 *                   1-b1^step_i and sqrt(1-b2^step_i) (computed in double by the caller like
 *                   the reference does in Python floats, then rounded); ignored by SGD.
 *   beta1, beta2, eps are doubles so that (1-beta) is rounded to fp32 from the double
 *   difference, as the reference's Python-float arithmetic does.
 *   m, s: host arrays of device pointers (may be NULL for rules that do not use them).
 *   coef[i] (optional device float*, may be NULL array): when non-NULL the kernel also
 *   writes the per-element d out / d lr = -(update direction) so that the autograd
 *   backward w.r.t. lr is a plain product (see savfi_mt_update_bwd_f32).
 * ---------------------------------------------------------------------------------- */
#define SAVFI_RULE_SGD          0
#define SAVFI_RULE_ADAM         1
#define SAVFI_RULE_ADAMAX_LSLR  2
#define SAVFI_RULE_ADAMAX_MSGD  3
#define SAVFI_LR_SCALAR   0
#define SAVFI_LR_ELEMENT  1
#define SAVFI_MT_MAX_TENSORS 48

int savfi_mt_update_f32(int rule, int lr_mode, int n,
                        const float* const* w, const float* const* g, const float* const* lr,
                        float* const* m, float* const* s, float* const* out, float* const* coef,
                        const int64_t* numel, const float* bc1, const float* sqrt_bc2,
                        double beta1, double beta2, double eps, void* stream);

/* Backward of the update w.r.t. the learning rates (first-order MAML: g is a constant).
 *   dir[i] = the `coef` buffer written by the forward (= d out / d lr, per element), scale = 1;
 *            for SGD pass dir = g and scale = -1 (no coef buffer needed).
 *   lr_mode ELEMENT: g_lr[i][e]  = scale * g_out[i][e] * dir[i][e]
 *   lr_mode SCALAR : g_lr[i][0] += scale * sum_e g_out[i][e] * dir[i][e]   (one float per tensor,
 *                    accumulated with one atomic per workgroup: the caller zero-fills it)
 * g_w is the identity (g_w = g_out) and needs no kernel. */
int savfi_mt_update_bwd_f32(int lr_mode, int n,
                            const float* const* g_out, const float* const* dir, float* const* g_lr,
                            const int64_t* numel, float scale, void* stream);

/* L2F: per-tensor mean -> out_vec[i] (device float[n], zero-filled by the caller, accumulated
 * with one atomic per workgroup);  per-tensor scale out[i] = gamma[i]*w[i] (gamma: device
 * float[n]).  scale_bwd: g_w[i] = gamma[i]*g_out[i] (g_w may be NULL or hold NULL entries);
 * g_gamma[i] += sum_e g_out[i][e]*w[i][e] (zero-filled by the caller; may be NULL). */
int savfi_mt_mean_f32(int n, const float* const* x, const int64_t* numel, float* out_vec, void* stream);
int savfi_mt_scale_f32(int n, const float* const* w, const float* gamma, float* const* out,
                       const int64_t* numel, void* stream);
int savfi_mt_scale_bwd_f32(int n, const float* const* g_out, const float* const* w, const float* gamma,
                           float* const* g_w, float* g_gamma, const int64_t* numel, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused mean-reduced L1 / MSE loss and its gradient, `rows` independent reductions per launch:
 *   a, b [rows, n];  kind 0: result[r] = mean |a[r]-b[r]| ; kind 1: mean (a[r]-b[r])^2   (fully overwritten)
 *   `scratch`: savfi_l1_mse_scratch_floats(rows, n) floats of caller-owned device memory (per-workgroup partial sums,
 *   added in a fixed order: the value is bit-reproducible).
 *   bwd: g_a[r] = g_loss[r]*sign(a-b)/n or g_loss[r]*2*(a-b)/n.
 * rows = 1 is nn.L1Loss / nn.MSELoss (loss.py:287-290); rows > 1 serves the per-sample losses of tasks adapted in lockstep.
 * ---------------------------------------------------------------------------------- */
int64_t savfi_l1_mse_scratch_floats(int rows, int64_t n);
int savfi_l1_mse_f32(int kind, const float* a, const float* b, float* result, float* scratch, int rows, int64_t n, void* stream);
int savfi_l1_mse_bwd_f32(int kind, const float* a, const float* b, const float* g_loss, float* g_a,
                         int rows, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution epilogue: bias + activation, in place on the conv output z [N,C,H*W]:
 *   z <- act(z + bias[c]),  act(x) = x > 0 ? x : slope*x   (slope 0 ReLU, 0.2 LeakyReLU, 1 bias only)
 * bwd: gz = gy * act'(y) with y the forward OUTPUT (sign(y) == sign(z+b) for slope >= 0);
 *      gbias[c] = sum over n, hw of gz (may be NULL; fully overwritten; deterministic: per-workgroup partial sums
 *      go to `scratch` (caller-owned, savfi_bias_act_scratch_floats(N, C, HW) floats) and are added in a fixed order).
 *      gz may alias gy; gz may be NULL when only the bias gradient is wanted (slope 1: gz == gy).
 * ---------------------------------------------------------------------------------- */
int savfi_bias_act_fwd_f32(float* z, const float* bias, int N, int C, int HW, float slope, void* stream);
int64_t savfi_bias_act_scratch_floats(int N, int C, int HW);
int savfi_bias_act_bwd_f32(const float* gy, const float* y, float* gz, float* gbias, float* scratch,
                           int N, int C, int HW, float slope, void* stream);

/* ------------------------------------------------------------------------------------
 * Bilinear x2 up-sampling of `planes` = N*C maps [H,W] -> [2H,2W] with ATen's source-index rules
 * (align_corners 1: torch.nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
 *  sepconv/model.py:191,213-234;  0: F.interpolate(scale_factor=2, align_corners=False), voxel_flow.py:400-414).
 * bwd is the exact adjoint, computed as a gather (gin fully overwritten, deterministic).
 * ---------------------------------------------------------------------------------- */
int savfi_upsample2x_fwd_f32(const float* in, float* out, int planes, int H, int W, int align_corners, void* stream);
int savfi_upsample2x_bwd_f32(const float* gout, float* gin, int planes, int H, int W, int align_corners, void* stream);

/* Windowed form of the same map: `in` holds the crop rows [sy0,sy0+Hs) x cols [sx0,sx0+Ws) of the virtual
 * [planes,H,W] source, `out` the window rows [oy0,oy0+Hw) x cols [ox0,ox0+Ww) of the virtual [planes,2H,2W]
 * result (identical values to the full op on that window).  SAVFI_E_SHAPE if the window reads a source pixel
 * outside the crop.  bwd: gout is the window, gin the crop (fully overwritten; crop pixels no window output
 * reads get 0).  Used for SepConv's sub-networks (sepconv/model.py:196-245), whose 51-tap maps are consumed
 * only on the un-padded frame area (sepconv/model.py:346-349 crops the result). */
int savfi_upsample2x_window_fwd_f32(const float* in, float* out, int planes, int H, int W, int sy0, int sx0,
                                    int Hs, int Ws, int oy0, int ox0, int Hw, int Ww, int align_corners, void* stream);
int savfi_upsample2x_window_bwd_f32(const float* gout, float* gin, int planes, int H, int W, int sy0, int sx0,
                                    int Hs, int Ws, int oy0, int ox0, int Hw, int Ww, int align_corners, void* stream);
/* the same adjoint with the (leaky) ReLU derivative of the up-sampled map's PRODUCER folded into its store:
 *   gin[i] *= (y[i] > 0 ? 1 : slope),   y [planes,Hs,Ws] = that producer's activated output = the op's forward input (y == NULL: plain)
 * -- what autograd runs as ReLU's backward between the two (reference: nn.ReLU after the Subnets' third convolution and after every
 * Basic block in front of an Upsample, sepconv/model.py:172-245); here one element-wise pass over the map less. */
int savfi_upsample2x_window_bwd_masked_f32(const float* gout, const float* y, float slope, float* gin, int planes, int H, int W,
                                           int sy0, int sx0, int Hs, int Ws, int oy0, int ox0, int Hw, int Ww, int align_corners,
                                           void* stream);

/* ----------------------------------------------------------------------------------
 * 3x3 / stride 1 / zero-pad 1 convolution of the backbones (sepconv/model.py:172-245 Basic / Subnet /
 * Upsample blocks through model_utils.py:308-366 MetaConv2dLayer -> F.conv2d; cain, voxelflow likewise):
 * Winograd F(2x2,3x3) with its 16 batched GEMMs on the fp32 matrix cores, one fused kernel.
 *   mode 0  forward:        out[N,Co,H+2p-2,W+2p-2] = act(conv2d(x[N,Ci,H,W], w[Co,Ci,3,3], zero pad p) + bias)
 *                           (p = pad in {0,1}; bias may be NULL)
 *   mode 1  data gradient of that convolution:  x = gy[N,Co,H,W] -> out = gx[N,Ci,H+2-2p,W+2-2p]
 *                           (bias ignored, pass slope 1)
 * act(v) = v > 0 ? v : slope * v  (slope 1 = none, 0 = ReLU).  `workspace` is caller-owned device memory of
 * savfi_conv3x3_workspace_floats(same N, Ci, Co, H, W, pad, mode) floats; it receives the transformed filter and, for
 * deep layers whose reduction channels are split over workgroups, the partial outputs; it may be reused by the next
 * call on the same stream.
 * ---------------------------------------------------------------------------------- */
int64_t savfi_conv3x3_workspace_floats(int N, int Ci, int Co, int H, int W, int pad, int mode);
int savfi_conv3x3_f32(const float* x, const float* w, const float* bias, float* out, float* workspace,
                      int N, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream);

/* The same convolution for T tasks adapted in lockstep (the task loop of meta_learning_system.py:366 as ONE launch per
 * layer): T filter sets w [T,Co,Ci,3,3], bias [T,Co]; the N samples are ordered sample-major, sample n belongs to task
 * n % T and is convolved with that task's filters (N % T == 0).  T = 1 is savfi_conv3x3_f32. */
int64_t savfi_conv3x3_tasks_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad, int mode);
int savfi_conv3x3_tasks_f32(const float* x, const float* w, const float* bias, float* out, float* workspace,
                            int N, int T, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream);

/* The two halves of savfi_conv3x3_tasks_f32, for a training step that runs the forward pass AND the data gradient on the
 * same filters (F.conv2d and its autograd backward, model_utils.py:308-366): the Winograd transform of both uses in ONE
 * launch, and the convolution on an already transformed filter.
 *   savfi_conv3x3_filters_f32      u_fwd / u_bwd (either may be NULL): the mode-0 / mode-1 transform of w [T,Co,Ci,3,3],
 *                                  savfi_conv3x3_filter_floats(T, Ci, Co, mode) floats each
 *   savfi_conv3x3_tasks_pre_f32    savfi_conv3x3_tasks_f32 with `u` (of the same mode) in place of w; `workspace`:
 *                                  savfi_conv3x3_tasks_pre_workspace_floats(...) floats (partial outputs of split
 *                                  launches; 0 for most shapes, then it may be NULL) */
int64_t savfi_conv3x3_filter_floats(int T, int Ci, int Co, int mode);
/* Layers of at most 512 -> 512 channels run on Winograd F(4x4, 3x3) (csrc/winograd4.h: 36 points per 4 x 4 outputs, a quarter of the
 * direct multiplies; fp32 rounding 3e-7 rms / <= 1e-5 max of the result's scale), deeper ones on F(2x2, 3x3).  The form follows the
 * channel counts alone, so a transformed filter is valid for every map size.  savfi_conv3x3_f4_workgroups: the workgroups the F(4x4)
 * kernel would launch for this call (32 tiles of 4 x 4 pixels x 32 produced channels each, x the reduction split of a deep layer on a
 * small map), 0 for an F(2x2) layer -- a caller with another kernel for launches that cannot fill the chip routes by this count. */
int64_t savfi_conv3x3_f4_workgroups(int N, int Ci, int Co, int H, int W, int pad, int mode);
/* The caller's choice of the form per layer AND map: bit 1 of every `mode` argument of the savfi_conv3x3_* functions (mode | 2) and `form`
 * = 2 below select the F(2x2) kernel whatever the channel counts -- for launches too small for F(4x4) to pay (it rounds 5x coarser; an
 * Adam-type inner rule turns that into flipped steps of elements whose gradient is rounding noise).  form = 0 / bit clear: by the channel
 * counts.  A transformed filter is valid for the form it was made for only.  The plain entry points are form 0. */
int savfi_conv3x3_filters_form_f32(const float* w, float* u_fwd, float* u_bwd, int T, int Ci, int Co, int form, void* stream);
int savfi_conv3x3_filters_multi_form_f32(const float* const* w, float* const* u_fwd, float* const* u_bwd, const int* T, const int* Ci,
                                         const int* Co, const int* form, int n, void* stream);
int savfi_conv3x3_dgrad_masked_form_f32(const float* gy, const float* u, const float* mask, float mask_slope, float* gx, float* workspace,
                                        int N, int T, int Ci, int Co, int H, int W, int pad, int form, void* stream);
int savfi_conv3x3_filters_f32(const float* w, float* u_fwd, float* u_bwd, int T, int Ci, int Co, void* stream);
/* n layers in one launch per 56 (layer, mode) jobs; entry i is savfi_conv3x3_filters_f32(w[i], u_fwd[i], u_bwd[i], T[i], Ci[i], Co[i]). */
int savfi_conv3x3_filters_multi_f32(const float* const* w, float* const* u_fwd, float* const* u_bwd, const int* T, const int* Ci,
                                    const int* Co, int n, void* stream);
int64_t savfi_conv3x3_tasks_pre_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad, int mode);
int savfi_conv3x3_tasks_pre_f32(const float* x, const float* u, const float* bias, float* out, float* workspace,
                                int N, int T, int Ci, int Co, int H, int W, int pad, int mode, float slope, void* stream);
/* Data gradient (mode 1) on a transformed filter with the (leaky) ReLU derivative of the layer that PRODUCED this convolution's input
 * folded into the output stage: gx = dgrad(gy) * (mask > 0 ? 1 : mask_slope); mask [N,Ci,H+2-2pad,W+2-2pad] is the convolution's
 * forward input (= the producer's activated output).  conv -> ReLU -> conv chains (sepconv/model.py:172-245 Basic / Subnet blocks):
 * the producer's backward then needs no element-wise pass of its own.  workspace: as savfi_conv3x3_tasks_pre_f32, mode 1. */
int savfi_conv3x3_dgrad_masked_f32(const float* gy, const float* u, const float* mask, float mask_slope, float* gx,
                                   float* workspace, int N, int T, int Ci, int Co, int H, int W, int pad, void* stream);

/* savfi_conv3x3_tasks_pre_f32, forward (mode 0) only, with the result written UNIT-MAJOR: out[n] is [Ho][Wo / 16][Co][16] instead of
 * [Co][Ho][Wo] -- the 16-pixel units of the SepConv kernels with a unit's Co x 64 bytes contiguous (the layout `taps_unit16` of
 * savfi_sepconv_{fwd,bwd}_frames8_f32 reads).  The SepConv plugin's last Subnet convolution (reference sepconv/model.py:183-194: the
 * 51 -> 51 layer behind the bilinear x2) writes its taps this way.  Needs Wo % 16 == 0, a sample below 2^31 bytes and a launch without
 * a reduction split (savfi_conv3x3_unit16_supported says so: 1 / 0); SAVFI_E_UNSUPPORTED otherwise.  No workspace. */
int savfi_conv3x3_unit16_supported(int N, int T, int Ci, int Co, int H, int W, int pad);
int savfi_conv3x3_tasks_pre_unit16_f32(const float* x, const float* u, const float* bias, float* out, int N, int T, int Ci, int Co,
                                       int H, int W, int pad, float slope, void* stream);
/* The data gradient of that layer on a cotangent that is unit-major as well: gy[n] is [H][W / 16][Co][16] (savfi_sepconv_bwd_frames8_f32 with
 * taps_unit16 = 3), gx [N][Ci][H+2-2pad][W+2-2pad] as always; u = the layer's data-gradient filter transform (savfi_conv3x3_filters_f32).
 * W % 16 == 0, no reduction split (savfi_conv3x3_in_unit16_supported: 1 / 0); SAVFI_E_UNSUPPORTED otherwise. */
int savfi_conv3x3_in_unit16_supported(int N, int T, int Ci, int Co, int H, int W, int pad);
int savfi_conv3x3_dgrad_in_unit16_f32(const float* gy, const float* u, float* gx, int N, int T, int Ci, int Co, int H, int W, int pad,
                                      void* stream);

/* Weight gradient of the same convolution (zero padding `pad` in {0,1}), NCHW in and out, deterministic:
 *   gw[Co,Ci,3,3] = sum over n,y,x of gz[n,co,y,x] * x[n,ci,y+a-pad,x+b-pad]      x [N,Ci,H,W], gz [N,Co,H+2pad-2,W+2pad-2]
 * `workspace`: savfi_conv3x3_wgrad_workspace_floats(same N, Ci, Co, H, W, pad) floats of caller-owned device memory
 * (per-workgroup partial blocks, added in a fixed order). */
int64_t savfi_conv3x3_wgrad_workspace_floats(int N, int Ci, int Co, int H, int W, int pad);
int savfi_conv3x3_wgrad_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int Ci, int Co,
                            int H, int W, int pad, void* stream);

/* Per-task weight gradients of T tasks in lockstep: gw [T,Co,Ci,3,3], gw[t] sums over the samples n with n % T == t. */
int64_t savfi_conv3x3_wgrad_tasks_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad);
int savfi_conv3x3_wgrad_tasks_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci,
                                  int Co, int H, int W, int pad, void* stream);

/* The same weight gradient in Winograd form F(3x3, 2x2) (2.25x fewer multiplies; same contract, its own workspace size;
 * deterministic; sums in a different order than the direct form: results agree to fp32 rounding of the reduction). */
int64_t savfi_conv3x3_wgrad_wino_tasks_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad);
int savfi_conv3x3_wgrad_wino_tasks_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci,
                                       int Co, int H, int W, int pad, void* stream);
/* ... and the bias gradient with it (round 5): gb [T][Co], gb[t][co] = sum of gz over the samples n % T == t and the map -- the gradient of
 * a bias added to this convolution's output (model_utils.py:308-366 MetaConv2dLayer: F.conv2d(..., bias)).  The weight gradient reads gz
 * anyway; the separate pass (savfi_bias_act_bwd_f32 with gz = NULL) read the whole map once more.  Sums in a fixed order (deterministic).
 * workspace: savfi_conv3x3_wgrad_wino_tasks_bias_workspace_floats(...) floats. */
int64_t savfi_conv3x3_wgrad_wino_tasks_bias_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int pad);
int savfi_conv3x3_wgrad_wino_tasks_bias_f32(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci,
                                            int Co, int H, int W, int pad, void* stream);

/* ----------------------------------------------------------------------------------
 * Direct K x K convolution, K in {3, 5, 7}, stride 1, zero padding `pad` (0 .. K-1), T filter sets (sample n uses set n % T):
 * F.conv2d of the backbones' layers that are not 3x3 (VoxelFlow's 5x5, Super SloMo's 7x7 / 5x5 heads) and of 3x3 layers where
 * Winograd rounding is not acceptable.  Arithmetic: every fp32 operand is split exactly into three bf16 pieces
 * (a = a1 + a2 + a3), the product sum runs as SIX bf16 MFMAs per K-slab (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2) with fp32
 * accumulation: the dropped terms are < 2^-26 of a product, the result is as close to fp64 as an fp32 fmaf chain.
 *   savfi_convk_filters_f32     p_fwd / p_bwd (either may be NULL): w [T,Co,Ci,K,K] packed as bf16 triples in MFMA fragment
 *                               order for the forward pass (mode 0) / the data gradient (mode 1);
 *                               savfi_convk_filter_floats(T, Ci, Co, K, mode) 4-byte units each, caller-owned
 *   savfi_convk_tasks_pre_f32   mode 0: out[N,Co,H+2p-K+1,W+2p-K+1] = act(conv2d(x[N,Ci,H,W], w, pad p) + bias[T,Co])
 *                               mode 1: x = gy[N,Co,H,W] -> out = gx[N,Ci,H+K-1-2p,W+K-1-2p]   (bias ignored, pass slope 1)
 *                               `packed` = the buffer of the same mode from savfi_convk_filters_f32
 *                               `precise` != 0: the five cross terms accumulate apart from the a1b1 sum (3x closer to float64
 *                               than an fp32 fmaf chain; for networks that amplify convolution rounding, ~10-15 % slower)
 * ---------------------------------------------------------------------------------- */
int64_t savfi_convk_filter_floats(int T, int Ci, int Co, int K, int mode);
int savfi_convk_filters_f32(const float* w, float* p_fwd, float* p_bwd, int T, int Ci, int Co, int K, void* stream);
/* The same for n layers in ONE launch per 56 (layer, mode) jobs: arrays of n pointers / shapes (host memory); entry i is
 * savfi_convk_filters_f32(w[i], p_fwd[i], p_bwd[i], T[i], Ci[i], Co[i], K[i]).  A MAML inner step re-packs every layer's fast weights
 * after every update (reference: nothing to pack -- F.conv2d reads the fast weight, model_utils.py:354-366); one launch per step
 * instead of one per layer and pass. */
int savfi_convk_filters_multi_f32(const float* const* w, float* const* p_fwd, float* const* p_bwd, const int* T, const int* Ci,
                                  const int* Co, const int* K, int n, void* stream);
int savfi_convk_tasks_pre_f32(const float* x, const float* packed, const float* bias, float* out, int N, int T, int Ci,
                              int Co, int H, int W, int K, int pad, int mode, float slope, int precise, void* stream);
/* the same fold for the direct kernels: gx = savfi_convk_tasks_pre_f32(mode 1)(gy) * (mask > 0 ? 1 : mask_slope).  K = 3 and
 * precise = 0 only (the layers of the conv -> act -> conv chains); SAVFI_E_UNSUPPORTED otherwise: the caller masks the plain data
 * gradient itself (savfi_bias_act_bwd_f32 without bias), as hip_ops.convk_tasks_pre does. */
int savfi_convk_dgrad_masked_f32(const float* gy, const float* packed, const float* mask, float mask_slope, float* gx, int N, int T,
                                 int Ci, int Co, int H, int W, int K, int pad, int precise, void* stream);
/* `reflect` != 0 (mode 0 only, pad < H, W): the border of width `pad` mirrors the image instead of reading zeros, i.e.
 * conv2d(nn.ReflectionPad2d(pad)(x), w) without the padded copy -- CAIN's MetaConvNorm (reference model_utils.py:821-848).  Its data
 * gradient is savfi_convk_tasks_pre_f32(mode 1, pad 0) on gy (the gradient of the padded extent) folded by
 * savfi_reflect_pad_bwd_f32; its weight gradient savfi_convk_wgrad_tasks_reflect_f32. */
int savfi_convk_tasks_pre_reflect_f32(const float* x, const float* packed, const float* bias, float* out, int N, int T, int Ci,
                                      int Co, int H, int W, int K, int pad, int mode, float slope, int precise, int reflect,
                                      void* stream);
/* gx[planes,H,W] = adjoint of nn.ReflectionPad2d(pad) applied to gp[planes,H+2pad,W+2pad] (gather: deterministic, no zero fill) */
int savfi_reflect_pad_bwd_f32(const float* gp, float* gx, int planes, int H, int W, int pad, void* stream);
/* the same fold plus a second cotangent of the unpadded map: gx = fold(gp) + add (add[planes,H,W]; NULL = savfi_reflect_pad_bwd_f32).
 * CAIN's RCAB (reference model_utils.py:957-990) pads x for its first convolution and adds x to its result: both gradients of x in
 * one pass (ABI 20) */
int savfi_reflect_pad_bwd_add_f32(const float* gp, const float* add, float* gx, int planes, int H, int W, int pad, void* stream);
/* xp[planes,H+2pad,W+2pad] = nn.ReflectionPad2d(pad)(x[planes,H,W]) (reference model_utils.py:829, :838), pad < H, W: the padded copy
 * the Winograd kernels read (ABI 20) */
int savfi_reflect_pad_fwd_f32(const float* x, float* xp, int planes, int H, int W, int pad, void* stream);

/* Weight gradient of the same convolution (same arithmetic; deterministic: per-workgroup partial blocks added in a fixed order):
 *   gw[T,Co,Ci,K,K], gw[t] = sum over samples n with n % T == t, y, x of gz[n,co,y,x] * x[n,ci,y+ky-pad,x+kx-pad]
 *   x [N,Ci,H,W], gz [N,Co,H+2pad-K+1,W+2pad-K+1]; `workspace`: savfi_convk_wgrad_workspace_floats(...) floats, caller-owned. */
int64_t savfi_convk_wgrad_workspace_floats(int N, int T, int Ci, int Co, int H, int W, int K, int pad);
int savfi_convk_wgrad_tasks_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci, int Co,
                                int H, int W, int K, int pad, int precise, void* stream);
/* the same with x's border of width `pad` mirrored (see savfi_convk_tasks_pre_reflect_f32) */
int savfi_convk_wgrad_tasks_reflect_f32(const float* x, const float* gz, float* gw, float* workspace, int N, int T, int Ci, int Co,
                                        int H, int W, int K, int pad, int precise, int reflect, void* stream);
/* the same (precise = 0) that also hands out the bias gradient gb [T,Co] = sums of gz over the samples n % T == t and the map: the sums ride
 * on the kernel's staging of gz (fixed order, bit-reproducible; gw is bit-identical to the call above), which saves the bias pass over
 * the map (reference: the bias gradient autograd returns for MetaConv2dLayer's F.conv2d, model_utils.py:308-366).  Only for the shapes
 * savfi_convk_wgrad_sums_bias() answers 1 for (3 x 3 layers of >= 48 -> 48 channels: the all-taps kernel, csrc/convk_wgrad.hip);
 * SAVFI_E_UNSUPPORTED otherwise.  Same workspace as above. */
int savfi_convk_wgrad_sums_bias(int N, int T, int Ci, int Co, int H, int W, int K, int pad);
int savfi_convk_wgrad_tasks_bias_f32(const float* x, const float* gz, float* gw, float* gb, float* workspace, int N, int T, int Ci, int Co,
                                     int H, int W, int K, int pad, int reflect, void* stream);

/* ----------------------------------------------------------------------------------
 * Channel attention + residual of CAIN's RCAB (model_utils.py:931-953 MetaCALayer, :957-990 MetaRCAB):
 *   s = mean_hw(t);  y = sigmoid(W2 relu(W1 s + b1) + b2);  out = t * y + x        t, x, out [N,C,H,W]; T weight sets (n % T)
 *   savfi_ca_pool_f32       s[plane] = scale * sum_hw a[plane][.] (* b[plane][.] when b != NULL)       planes = N*C
 *   savfi_ca_mlp_fwd_f32    y [N,C], a1 [N,Cr] (hidden activations, kept for the backward) from s [N,C]; w1 [T,Cr,C], b1 [T,Cr],
 *                           w2 [T,C,Cr], b2 [T,C]; C <= 1024, Cr <= 64
 *   savfi_ca_mlp_bwd_f32    r [N,C] = sum_hw g*t  ->  ds [N,C] (gradient w.r.t. s, times inv_hw) and gw1, gb1, gw2, gb2 per task
 *   savfi_ca_apply_f32      out = a * y[plane] + x          (x != NULL: forward, x = the skip connection)
 *                           out = a * y[plane] + ds[plane]  (x == NULL: backward, gradient w.r.t. t)
 * ---------------------------------------------------------------------------------- */
int savfi_ca_pool_f32(const float* a, const float* b, float* s, int64_t planes, int hw, float scale, void* stream);
int savfi_ca_mlp_fwd_f32(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* a1,
                         int N, int T, int C, int Cr, void* stream);
int savfi_ca_mlp_bwd_f32(const float* r, const float* s, const float* y, const float* a1, const float* w1, const float* w2,
                         float* ds, float* gw1, float* gb1, float* gw2, float* gb2, int N, int T, int C, int Cr, float inv_hw,
                         void* stream);
int savfi_ca_apply_f32(const float* a, const float* y, const float* x, const float* ds, float* out, int64_t planes, int hw,
                       void* stream);
/* forward of the same block with the MLP inside the apply launch (Cr <= 16, C <= 256; SAVFI_E_UNSUPPORTED otherwise: the two calls
 * above): out = a * y + x with y = sigmoid(W2 relu(W1 s + b1) + b2) of the sample, s [N,C] from savfi_ca_pool_f32; y [N,C] and
 * a1 [N,Cr] are written for the backward, bit-identical to savfi_ca_mlp_fwd_f32's (ABI 20) */
int savfi_ca_apply_mlp_f32(const float* a, const float* s, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* x, float* out, float* y, float* a1, int N, int T, int C, int Cr, int hw, void* stream);
/* backward of the same with the MLP's backward inside the apply launch (same limits): gt = g * y + ds, ds recomputed per workgroup,
 * r [N,C] = sum_hw g * t from savfi_ca_pool_f32; gw1 [T,Cr,C], gb1 [T,Cr], gw2 [T,C,Cr], gb2 [T,C] from T workgroups at the front of
 * the grid; every result bit-identical to savfi_ca_mlp_bwd_f32 + savfi_ca_apply_f32 (ABI 20) */
int savfi_ca_apply_bwd_mlp_f32(const float* g, const float* r, const float* s, const float* y, const float* a1, const float* w1,
                               const float* w2, float* gt, float* gw1, float* gb1, float* gw2, float* gb2, int N, int T, int C, int Cr,
                               int hw, void* stream);

/* ----------------------------------------------------------------------------------
 * Per-plane mean removal of CAIN's input frames (model_utils.py:11-15 sub_mean; cain/model.py:70-94):
 *   mean[p] = sum_i x[p][i] / hw;  out[p][i] = x[p][i] - mean[p]            x, out [planes][hw] (out may alias x), mean [planes]
 * Two launches, fixed summation order, no cleared memory (safe inside a captured hipGraph, where ATen's multi-workgroup
 * reduction is not: csrc/submean.hip).  `workspace`: savfi_sub_mean_workspace_floats(planes, hw) floats, caller-owned.
 * ---------------------------------------------------------------------------------- */
int64_t savfi_sub_mean_workspace_floats(int64_t planes, int hw);
int savfi_sub_mean_f32(const float* x, float* out, float* mean, float* workspace, int64_t planes, int hw, void* stream);

/* ----------------------------------------------------------------------------------
 * Frame staging (data/vimeo_septuplet.py:68-80, data/video.py:44-51: channel swap, HWC->CHW, .float()/255,
 * Normalize): src = N decoded frames, uint8 [N,H,W,3] on the DEVICE (copied there as bytes);
 * dst[n][c][y][x] = (src[n][y][x][swap_rb ? 2-c : c] / div - mean_c) / std, fp32 [N,3,H,W]; one mean per OUTPUT
 * channel (Super SloMo subtracts 0.429 / 0.431 / 0.397, data/vimeo_septuplet.py:31-35).
 * ---------------------------------------------------------------------------------- */
int savfi_frames_u8_to_f32(const unsigned char* src, float* dst, int64_t N, int H, int W, int swap_rb, float div,
                           float mean_c0, float mean_c1, float mean_c2, float std, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAVFI_HIP_H_ */
