/* libflairhip -- C ABI of the MI355X (gfx950) tiled-segmentation hot path.
 *
 * The reference (kezakool/flair-for-aigle) is pure Python and has no FFI of its own: the seam this
 * library replaces is the set of torch / segmentation_models_pytorch calls listed per entry point
 * below (file:line under the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FFA_ERR_* for argument / support errors, or a
 *     positive hipError_t when a launch fails; nothing throws; ffa_last_error() gives the text of
 *     the last failure on the calling thread
 *   - every launch goes to the caller's hipStream_t; the library never allocates device memory,
 *     never synchronises and keeps no mutable global state (workspace is passed in; query sizes
 *     with the *_workspace_bytes functions)
 *   - tensors are raw device pointers, NHWC, contiguous, dtype FFA_BF16 (storage) or FFA_F32,
 *     channel pitch a multiple of 16 elements (8 where noted); pad channels must hold zeros
 *   - all reductions are fixed-order (bitwise reproducible); no float atomics anywhere
 */
#ifndef FLAIRHIP_H
#define FLAIRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ffa_stream_t; /* == hipStream_t */

#define FFA_BF16 0
#define FFA_F32 1

#define FFA_OK 0
#define FFA_ERR_ARG (-1)
#define FFA_ERR_UNSUPPORTED (-2)
#define FFA_ERR_WORKSPACE (-3)

const char* ffa_last_error(void);
int ffa_version(void);
const char* ffa_target_arch(void);

/* ---- convolution (replaces the conv2d forward/backward reached via
 *      flair_hub/models/monotemp_model.py:68-92 -> smp encoder/decoder, called at
 *      flair_hub/models/flair_model.py:376 and :417-419) ------------------------------------------ */

/* preferred block height (output channels per workgroup) for a layer; weights are packed for it */
int ffa_conv_block_co(int kh, int kw, int stride, int cout);
/* Operand layout + block height of a layer: the `bco` value for ffa_pack_conv_weight / ffa_pack_desc_fill /
 * ffa_conv2d / ffa_conv2d_stats / ffa_conv_stat_rows.  Bits 0..11 = rows per block (pad the row count to a multiple
 * of it); FFA_BCO_RING set = the operand is packed for the LDS-DMA ring kernels (bf16: conv3x3_ring16_kernel, the default
 * for 3x3 stride 1 pad 1 layers with >= 64 output channels and whole 64-byte groups of input channels -- FFA_RING=0 in the
 * environment restores the conv_igemm kernels; f32: conv3x3_ring_kernel, FFA_RING=1); bit 13 (0x2000) = the
 * register-resident operand of the thin layers (ffa_thin_*; FFA_THIN=0 disables).  `allow`: bit 0 admits the ring
 * layout, bit 1 the thin one, bit 2 the stem one (0x8000); pass 0 for operands that feed ffa_conv2d_bnbwd or a dil = 2 call. */
#define FFA_BCO_RING 0x1000
#define FFA_BCO_THIN 0x2000
int ffa_conv_plan(int dtype, int kh, int kw, int stride, int cout, int ci_pitch, int allow);
/* The ring kernel called directly: out = relu?(conv3x3(in', w) + bias + residual) with in' = in, or -- when
 * pro_scale / pro_shift are given -- in' = relu(in * pro_scale[c] + pro_shift[c]) evaluated while the input tile is
 * staged (the training-mode BatchNorm + ReLU of the producing layer folded into this convolution's loader: replaces
 * the F.batch_norm + relu pass between two convolutions of smp's Conv2dReLU / torchvision's BasicBlock; zero padding
 * applies to the normalised tensor).  stat_partials as for ffa_conv2d_stats (rows = ffa_conv_stat_rows) or null. */
int ffa_ring_conv3x3(int dtype, const void* in, const void* w_ring, const float* bias, const void* residual, void* out,
                     float* stat_partials, const float* pro_scale, const float* pro_shift, int B, int H, int W, int Ci,
                     int Co, int co_rows, int relu, ffa_stream_t stream);
long long ffa_ring_stat_rows(int B, int H, int W, int co_rows);
int ffa_ring_pack(int dtype, const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose,
                  int co_rows, int ci_pitch, ffa_stream_t stream);
/* batched packing of ring operands (the ring layout has its own descriptor type) */
int ffa_ring_pack_desc_bytes(void);
int ffa_ring_pack_desc_fill(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I,
                            int transpose, int co_rows, int ci_pitch, int dtype);
int ffa_ring_pack_batched(int dtype, const void* descs_device, int n, ffa_stream_t stream);
/* Thin 3x3 stride-1 pad-1 layers (bf16, <= 32 real output rows, input pitch 16 or 32: the decoder's last two blocks and the
 * segmentation head; csrc/conv3x3_thin.hip): the whole weight operand lives in registers, the input halo arrives by
 * LDS-DMA three tiles deep.  Replaces the same F.conv2d as ffa_conv2d for those shapes; `up` reads nearest_x2(in)
 * (F.interpolate(scale_factor=2) in front of smp's DecoderBlock.conv1), `pool` stores the 2x2 sums of the output (its
 * adjoint, for the transposed operand), `_pro` evaluates relu(in * scale + shift) on the staged halo.  ffa_conv2d /
 * ffa_conv2d_upcat / ffa_conv2d_dgrad_upcat dispatch here when the operand was packed for it (ffa_conv_plan bit 13). */
int ffa_thin_eligible(int dtype, int kh, int kw, int stride, int rows_real, int ci_pitch);
long long ffa_thin_pack_bytes(int co_rows, int ci_pitch);
long long ffa_thin_stat_rows(int B, int H, int W, int ci_pitch);
int ffa_thin_conv3x3(const void* in, const void* w_thin, const float* bias, const void* residual, void* out,
                     float* stat_partials, int B, int H, int W, int Ci, int Co, int co_rows, int relu, int up, int pool,
                     ffa_stream_t stream);
int ffa_thin_conv3x3_pro(const void* in, const void* w_thin, const float* bias, const void* residual, void* out,
                         float* stat_partials, const float* pro_scale, const float* pro_shift, int B, int H, int W, int Ci,
                         int Co, int co_rows, int relu, int up, int pool, ffa_stream_t stream);
int ffa_thin_pack(const float* w_oihw, const float* scale, void* dst, int O, int I, int transpose, int co_rows,
                  int ci_pitch, ffa_stream_t stream);
int ffa_thin_pack_desc_bytes(void);
int ffa_thin_pack_desc_fill(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I,
                            int transpose, int co_rows, int ci_pitch);
int ffa_thin_pack_batched(const void* descs_device, int n, ffa_stream_t stream);
/* The ResNet stem (bf16 7x7 stride-2 pad-3 convolution of a <= 8-channel tile stored at pitch 16 into 64 channels;
 * csrc/conv7x7_stem.hip; torchvision's conv1 as smp's ResNetEncoder keeps it): four adjacent taps x 8 channels per
 * K = 32 MFMA step, only the first 16 bytes of every input pixel are staged.  ffa_conv2d / ffa_conv2d_stats dispatch here
 * when the operand was packed for it (ffa_conv_plan with bit 2 of `allow`; bit 15 of the returned code). */
#define FFA_BCO_STEM 0x8000
int ffa_stem_eligible(int dtype, int kh, int kw, int stride, int cout, int ci_pitch);
long long ffa_stem_pack_bytes(void);
long long ffa_stem_stat_rows(int B, int Ho, int Wo);
int ffa_stem_pack(const float* w_oihw, const float* scale, void* dst, int O, int I, ffa_stream_t stream);
int ffa_stem_conv7x7(const void* in, const void* w_stem, const float* bias, void* out, float* stat_partials, int B, int Hi,
                     int Wi, int Ci, int Ho, int Wo, int Co, int relu, ffa_stream_t stream);
int ffa_conv_row_group(int kh);
long long ffa_pack_conv_weight_bytes(int dtype, int co_rows, int ci_pitch, int kh, int kw);
/* OIHW f32 master weight -> kernel operand.  transpose=1 builds the dgrad operand (rows = input
 * channels, taps mirrored).  scale (optional, per row) folds an eval-mode BatchNorm. */
int ffa_pack_conv_weight(int dtype, const float* w_oihw, const float* scale, void* dst, int O, int I, int kh, int kw,
                         int transpose, int co_rows, int ci_pitch, int bco, int rg, ffa_stream_t stream);
/* every conv operand of a network in one launch: descriptors are filled on the host (ffa_pack_desc_fill,
 * ffa_pack_desc_bytes each), copied to the device once, and replayed every step */
int ffa_pack_desc_bytes(void);
int ffa_pack_desc_fill(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I, int kh,
                       int kw, int transpose, int co_rows, int ci_pitch, int bco, int rg, int dtype);
/* descriptor for the column block W[:, col0 : col0 + ncols] of a weight with I_total input channels (one modality's share
 * of a FusionHandler 1x1 convolution, flair_hub/models/flair_model.py:470-475,533-541) */
int ffa_pack_desc_fill_cols(void* host_desc, const float* w_oihw, const float* scale, void* dst, int O, int I_total,
                            int col0, int ncols, int kh, int kw, int transpose, int co_rows, int ci_pitch, int bco, int rg,
                            int dtype);
int ffa_pack_conv_weights_batched(int dtype, const void* descs_device, int n, ffa_stream_t stream);
/* out = relu?( conv(in, w) + bias + residual ).  dil=2 reads `in` through a virtual zero insertion
 * (dgrad of a stride-2 layer, stride must then be 1). */
int ffa_conv2d(int dtype, const void* in, const void* w_packed, const float* bias, const void* residual, void* out,
               int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int co_rows, int bco, int kh, int kw, int stride,
               int pad, int dil, int relu, ffa_stream_t stream);

/* ffa_conv2d that also writes, per pixel tile, the channel sums and sums of squares of the output it stores:
 * stat_partials[rows][2][Co] f32, rows = ffa_conv_stat_rows(B, Ho, Wo, co_rows, bco).  With ffa_bn_finalize this replaces the
 * separate statistics pass of a training-mode BatchNorm that follows the convolution (torch.nn.functional.batch_norm
 * as called by smp's Conv2dReLU / torchvision's BasicBlock). */
long long ffa_conv_stat_rows(int B, int Ho, int Wo, int co_rows, int bco);
/* 1 when ffa_conv2d / ffa_conv2d_stats run this 3x3 stride-1 convolution on the persistent kernel
 * (conv3x3_persist_kernel, blocks that walk several pixel tiles; FFA_CONV_PERSIST=0 disables), else 0.
 * Profiling aid: the two kernels are separate symbols. */
int ffa_conv_is_persistent(int dtype, int B, int Ho, int Wo, int Ci, int co_rows, int bco, int kh, int kw, int stride,
                           int dil);
int ffa_conv2d_stats(int dtype, const void* in, const void* w_packed, const float* bias, const void* residual,
                     void* out, float* stat_partials, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                     int co_rows, int bco, int kh, int kw, int stride, int pad, int dil, int relu,
                     ffa_stream_t stream);

/* ffa_conv2d whose output is the gradient dy of y = relu(bn_train(x)) -- a dgrad convolution feeding a BatchNorm
 * backward: also leaves per tile sum(g), sum(g*x), g = dy masked by x*bn_scale + bn_shift > 0, in
 * stat_partials[rows][2][Co], so that ffa_bn_bwd_partials can skip the reduction pass of ffa_bn_bwd. */
int ffa_conv2d_bnbwd(int dtype, const void* in, const void* w_packed, const void* residual, void* out,
                     float* stat_partials, const void* bnx, const float* bn_scale, const float* bn_shift, int B, int Hi,
                     int Wi, int Ci, int Ho, int Wo, int Co, int co_rows, int bco, int kh, int kw, int stride, int pad,
                     int dil, ffa_stream_t stream);

/* 3x3 stride-1 pad-1 convolution over the virtual tensor cat(nearest_x2(lo), skip) -- what smp's DecoderBlock builds
 * with F.interpolate(scale_factor=2, mode="nearest") + torch.cat before its first conv (reached from
 * flair_model.py:417-419) -- without writing it: lo [B][Hl][Wl][C1], skip [B][2Hl][2Wl][C2] or null.
 * FFA_ERR_UNSUPPORTED when C1 is not a whole number of the kernel's channel groups.  bias (per output channel, e.g.
 * an eval-mode BatchNorm shift) and stat_partials may be null; relu != 0 clamps in the epilogue. */
int ffa_conv2d_upcat(int dtype, const void* lo, const void* skip, const void* w_packed, const float* bias, void* out,
                     float* stat_partials, int B, int Hl, int Wl, int C1, int C2, int Co, int co_rows, int bco,
                     int relu, ffa_stream_t stream);

/* Input gradient of that convolution as the two tensors the decoder block needs -- dlo [B][Ho/2][Wo/2][C1] (2x2 sums,
 * the adjoint of nearest x2) and dskip [B][Ho][Wo][C2] -- without materialising the gradient of the concatenation
 * (replaces ffa_conv2d(dgrad) + ffa_upsample_nearest2x_concat_bwd).  w_packed_t: the transposed operand. */
int ffa_conv2d_dgrad_upcat(int dtype, const void* dy, const void* w_packed_t, void* dlo, void* dskip, int B, int Ho,
                           int Wo, int Cdy, int C1, int C2, int co_rows, int bco, ffa_stream_t stream);
long long ffa_conv_wgrad_workspace_bytes(int dtype, int kh, int kw, int stride, int Co, int Ci, int B, int Ho, int Wo);
int ffa_conv_wgrad(int dtype, const void* x, const void* dy, float* dw_oihw, int B, int Hi, int Wi, int Ci, int Ho,
                   int Wo, int Co, int Co_real, int Ci_real, int kh, int kw, int stride, int pad, int accumulate,
                   void* workspace, long long workspace_bytes, ffa_stream_t stream);

/* Weight gradient of the convolution ffa_conv2d_upcat computes: the input is the virtual cat(nearest_x2(lo), skip),
 * dw is OIHW [Co_real][C1 + C2][3][3] f32.  Workspace: ffa_conv_wgrad_workspace_bytes(dtype, 3, 3, 1, Co, C1 + C2, B,
 * 2*Hl, 2*Wl).  FFA_ERR_UNSUPPORTED when C1 is not a multiple of the kernel's input-channel block. */
int ffa_conv_wgrad_upcat(int dtype, const void* lo, const void* skip, const void* dy, float* dw_oihw, int B, int Hl,
                         int Wl, int C1, int C2, int Co, int Co_real, int accumulate, void* workspace,
                         long long workspace_bytes, ffa_stream_t stream);

/* "Normalise on load" (round 3): a convolution / weight gradient whose input is relu(x * pro_scale[c] + pro_shift[c]),
 * the training-mode BatchNorm + ReLU of the PRODUCING layer (smp Conv2dReLU -> Conv2dReLU, torchvision BasicBlock conv1 ->
 * bn1 -> relu -> conv2: flair_hub/models/monotemp_model.py:68-92), evaluated while the consumer stages x -- the
 * normalised tensor is never written (replaces ffa_bn_apply + ffa_conv2d[_stats] / ffa_conv_wgrad; bit-identical to that
 * sequence, zero padding applied after the normalisation).  bf16, 3x3 stride 1 pad 1; operands in the ring16 / thin
 * layouts of ffa_conv_plan.  up != 0: x is the low-resolution map of the skip-less nearest-x2 form.
 * FFA_ERR_UNSUPPORTED where no prologue kernel exists (the caller then materialises the tensor). */
int ffa_conv2d_pro(int dtype, const void* in, const void* w_packed, const float* bias, const void* residual, void* out,
                   float* stat_partials, const float* pro_scale, const float* pro_shift, int B, int H, int W, int Ci,
                   int Co, int co_rows, int bco, int relu, int up, ffa_stream_t stream);
int ffa_conv_wgrad_pro(int dtype, const void* x, const void* dy, float* dw_oihw, const float* pro_scale,
                       const float* pro_shift, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int Co_real,
                       int Ci_real, int up, int accumulate, void* workspace, long long workspace_bytes,
                       ffa_stream_t stream);

/* ---- U-TAE Sentinel time-series branch (flair_hub/models/multitemp_model.py) ------------------
 * Small NHWC kernels around ffa_conv2d for the temporally shared encoder, the L-TAE and the attention-weighted skip
 * aggregation; evaluation-mode forward. */
/* out[n][y][x] = in[n][refl(y-1)][refl(x-1)]: the padding of nn.Conv2d(padding=1, padding_mode="reflect")
 * (multitemp_model.py:473-482); the convolution itself is ffa_conv2d with pad 0 on the padded tensor */
int ffa_reflect_pad1(int dtype, const void* in, void* out, int N, int H, int W, int C, ffa_stream_t stream);
/* y = [residual +] relu?(GroupNorm(x)): nn.GroupNorm of ConvLayer (:464-468, 4 groups over an image) and of LTAE2d
 * (:224-231, 16 groups over the dates of one pixel).  Sample s starts at element (s / Q) * stride_hi +
 * (s % Q) * stride_lo and has `inner` positions, inner_stride elements apart, of C contiguous channels. */
int ffa_group_norm(int dtype, const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                   long long samples, int Q, long long stride_hi, long long stride_lo, int inner,
                   long long inner_stride, int C, int groups, float eps, int relu, ffa_stream_t stream);
/* PositionalEncoder (:287-313): out[n][r*d + j] = sin|cos(pos[n] / period^(2*(j/2)/d)), repeated `repeat` times */
int ffa_positional_encoding(const float* pos, float* out, int n, int d, int repeat, float period, ffa_stream_t stream);
/* x[n][p][c] += vec[n][c] (f32 vec): "out + positional_encoder(bp)" (:270) */
int ffa_add_rowvec(int dtype, void* x, const float* vec, int N, int P, int C, ffa_stream_t stream);
/* MultiHeadAttention with one learnt query per head + masked softmax over the dates (:337-403), per pixel:
 * k [B][T][P][n_head*d_k], v [B][T][P][n_head*d_v], Q f32 [n_head][d_k], pad u8 [B][T] ->
 * out [B][P][n_head*d_v], attn f32 [n_head][B][T][P] */
int ffa_ltae_attention(int dtype, const void* k, const void* v, const float* Q, const unsigned char* pad, void* out,
                       float* attn, int B, int T, int P, int n_head, int d_k, int d_v, ffa_stream_t stream);
/* Temporal_Aggregator(mode="att_group") (:609-628,640-654): out[b][p][c] = sum_t attn[c/(C/n_head)][b][t][p] *
 * x[b][t][p][c], padded dates dropped when use_pad; attn already at the resolution of x */
int ffa_temporal_aggregate(int dtype, const void* x, const float* attn, const unsigned char* pad, void* out, int B,
                           int T, int P, int C, int n_head, int use_pad, ffa_stream_t stream);
/* TemporallySharedBlock.smart_forward (:420-447): pad[n] = image n is all `value`; padded images come out as `value` */
int ffa_detect_pad_images(const float* x, unsigned char* pad, int N, long long per_image, float value,
                          ffa_stream_t stream);
int ffa_mask_images(int dtype, void* x, const unsigned char* pad, int N, long long per_image, float value,
                    ffa_stream_t stream);
/* -- training of the branch: backward of the pieces above (autograd of multitemp_model.py's modules) -- */
int ffa_reflect_pad1_bwd(int dtype, const void* dpad, void* dx, int N, int H, int W, int C, ffa_stream_t stream);
/* GroupNorm backward, ffa_group_norm's geometry; y = [res +] relu?(GN(x)): the residual's gradient is dy itself.
 * partial f32 [samples][2][C] receives per-sample (sum dy' xhat, sum dy'): the caller sums the rows (ffa_column_sums) */
int ffa_group_norm_bwd(int dtype, const void* x, const void* dy, void* dx, const float* gamma, const float* beta,
                       float* partial, long long samples, int Q, long long stride_hi, long long stride_lo, int inner,
                       long long inner_stride, int C, int groups, float eps, int relu, ffa_stream_t stream);
/* ffa_ltae_attention with the attention dropout of ScaledDotProductAttention (:399): drop f32 [n_head][B][T][P] = 0 or
 * 1 / (1 - p) (nullable); attn = the dropped-out masks (what the reference returns), prob = the clean softmax (kept) */
int ffa_ltae_attention_train(int dtype, const void* k, const void* v, const float* Q, const unsigned char* pad,
                             const float* drop, void* out, float* attn, float* prob, int B, int T, int P, int n_head,
                             int d_k, int d_v, ffa_stream_t stream);
/* its backward: dout [B][P][n_head*d_v], dattn_ext f32 [n_head][B][T][P] (nullable: gradient the aggregators send to the
 * returned masks) -> dk, dv (layouts of k, v) and dq_partial f32 [ffa_ltae_attention_bwd_blocks()][n_head*d_k] (rows to sum) */
int ffa_ltae_attention_bwd_blocks(int B, int P, int n_head);
int ffa_ltae_attention_bwd(int dtype, const void* k, const void* v, const float* Q, const unsigned char* pad,
                           const float* drop, const float* prob, const void* dout, const float* dattn_ext, void* dk,
                           void* dv, float* dq_partial, int B, int T, int P, int n_head, int d_k, int d_v,
                           ffa_stream_t stream);
int ffa_temporal_aggregate_bwd(int dtype, const void* x, const float* attn, const unsigned char* pad, const void* dout,
                               void* dx, float* dattn, int B, int T, int P, int C, int n_head, int use_pad,
                               ffa_stream_t stream);
/* y = x * m elementwise (nn.Dropout with a pre-scaled keep mask; its own backward) */
int ffa_mul(int dtype, const void* x, const void* m, void* y, long long n, ffa_stream_t stream);
/* y = (xs[0] + ... + xs[n-1]) / divisor elementwise, 1 <= n <= 4 (host array of device pointers): FusionHandler's
 * case 3, torch.mean(torch.stack([maps of several time-series branches]), dim=0), flair_hub/models/flair_model.py:496-501;
 * with n = 1, divisor = branches: its backward */
int ffa_mean_stack(int dtype, const void* const* xs, int n, float divisor, void* y, long long numel, ffa_stream_t stream);

/* ---- measurement only: kernel timing session.  Between ffa_ktime_begin(max_launches) and ffa_ktime_end() the 3x3 MFMA
 *      launches of the process (ffa_conv2d on the ring operand, ffa_conv_wgrad on 64-channel blocks) carry their
 *      own start / stop events (hipExtLaunchKernelGGL): ffa_ktime_end waits for them and returns, in launch order, the
 *      kernels' durations in milliseconds and a tag (1: ring16 8x32 tiles, 2: ring16 16x16, 4: ring16 128-co blocks,
 *      16: wgrad64); its return value is the number of timed launches.  Not for use under stream capture. */
int ffa_ktime_begin(int max_launches);
int ffa_ktime_end(float* ms, int* tags, int cap);

/* ---- optimizer step: Adam / AdamW over all parameter tensors (csrc/optim.hip).  Replaces torch.optim.AdamW / Adam as
 *      built at flair_hub/tasks/tasks_module.py:385-389 (same update, same state tensors: exp_avg, exp_avg_sq, a
 *      per-parameter device step counter that already counts this update).  Host arrays of device pointers, one entry per
 *      f32 parameter tensor; lr is a device scalar; ceil(n / 72) launches, descriptors ride in the kernel arguments. */
int ffa_adamw_multi(int n_tensors, void* const* p, const void* const* g, void* const* m, void* const* v,
                    const void* const* step, const long long* numel, const float* lr, double beta1, double beta2,
                    float eps, float weight_decay, int decoupled, int maximize, ffa_stream_t stream);

/* ---- Swin-Transformer encoder + UPerNet decoder (the reference's default `swin_*-upernet` architecture:
 *      configs/train/config_models.yaml:5, configs/config_model_zonal_segmentation.yaml:26, resolved through
 *      flair_hub/models/monotemp_model.py:64-92 by smp.create_model("upernet", "tu-swin_...")).
 *      Token tensors are NHWC [B][H][W][C]; evaluation-mode forward. ------------------------------------------- */
#define FFA_ACT_NONE 0
#define FFA_ACT_GELU 1
/* nn.Linear (+ nn.GELU) (+ residual add) on tokens: out[m][n] = act(sum_k a[m][k] w[n][k] + bias[n]) + residual[m][n];
 * w is nn.Linear.weight's own [N][K] layout in the dtype of a.  bf16: MFMA token GEMM (K % 32 == 0, N % 8 == 0, pitches
 * % 8 == 0); f32: plain-FMA parity kernel with the same epilogues (any K, N), for gradient checks, not a speed path. */
int ffa_linear(int dtype, const void* a, long long lda, const void* w, const float* bias, const void* residual,
               long long ldr, void* out, long long ldc, int M, int K, int N, int act, ffa_stream_t stream);
/* The same GEMM with the training-time epilogues: act = FFA_ACT_GELU with `aux` != NULL also stores the (bf16) pre-activation
 * to aux [M][ldaux] (Mlp.fc1, kept for the backward pass); act = FFA_ACT_DGELU multiplies the result by gelu'(aux) (the input
 * gradient of Mlp.fc2 carried through the activation); row_scale[m / rows_per_scale] multiplies the result before the
 * residual add (timm's DropPath: 0 or 1 / keep_prob per sample). */
#define FFA_ACT_DGELU 2
#define FFA_ACT_RELU 3 /* relu(a w^T + bias): a 1x1 convolution with folded BatchNorm + ReLU on tokens */
int ffa_linear_ex(int dtype, const void* a, long long lda, const void* w, const float* bias, const void* residual,
                  long long ldr, void* out, long long ldc, int M, int K, int N, int act, void* aux, long long ldaux,
                  const float* row_scale, int rows_per_scale, ffa_stream_t stream);
/* nn.Linear's weight gradient dW[n][k] = sum_m dy[m][n] x[m][k] (f32 [N][K]; accumulate != 0 adds to dw): transposed-
 * operand MFMA GEMM over the tokens, split over token ranges with a fixed-order reduction (deterministic); f32 operands take
 * a plain-FMA kernel with the same slabs and reduction (parity mode) */
long long ffa_linear_wgrad_workspace_bytes(int M, int N, int K);
int ffa_linear_wgrad(int dtype, const void* x, long long ldx, const void* dy, long long ldy, float* dw,
                     float* dbias /* nullable: [N] column sums of dy from the same pass */, int M, int K, int N,
                     int accumulate, void* workspace, long long workspace_bytes, ffa_stream_t stream);
/* PatchEmbed's Conv2d(kernel = stride = ps) as a gather: out[b][y][x][(dy*ps + dx)*C + c] = in[b][y*ps+dy][x*ps+dx][c] */
int ffa_space_to_depth(int dtype, const void* in, void* out, int B, int Ho, int Wo, int C, int ps, ffa_stream_t stream);
/* nn.LayerNorm(C) over rows of C contiguous channels; stats (nullable) receives (mean, rstd) per row for the backward pass */
int ffa_layer_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats, long long rows,
                   int C, float eps, ffa_stream_t stream);
/* PatchMerging's 2x2 gather + nn.LayerNorm(4C): x [B][H][W][C] -> y [B][H/2][W/2][4C] */
int ffa_patch_merge_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats, int B,
                         int H, int W, int C, float eps, ffa_stream_t stream);
/* backward of the two: dx, dgamma, dbeta (f32) from x (the forward's input), dy and the forward's row statistics;
 * deterministic (chunk partials in the workspace, summed in a fixed order) */
long long ffa_layer_norm_bwd_workspace_bytes(long long rows, int C);
int ffa_layer_norm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats,
                       const void* dres /* nullable: added to dx (the residual around the normalised branch) */, void* dx,
                       float* dgamma, float* dbeta, long long rows, int C, void* workspace, long long workspace_bytes,
                       ffa_stream_t stream);
int ffa_patch_merge_norm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats, void* dx,
                             float* dgamma, float* dbeta, int B, int H, int W, int C, void* workspace,
                             long long workspace_bytes, ffa_stream_t stream);
/* SwinTransformerBlock._attn without the two projections: cyclic shift, padding to the window grid, window partition,
 * softmax(scale q k^T + relative position bias + shift mask) v, window reverse, crop, un-shift -- by index arithmetic on
 * qkv [B][H][W][3C] (channel = which*C + head*32 + d); qkv_bias [3C] is what a padding token projects to;
 * table [(2 ws - 1)^2][heads] is relative_position_bias_table */
int ffa_window_attention(int dtype, const void* qkv, void* out, const float* qkv_bias, const float* table, int B, int H,
                         int W, int C, int heads, int ws, int shift, float scale, ffa_stream_t stream);
/* backward of ffa_window_attention (bf16 MFMA kernel; f32 plain-FMA parity kernel): dqkv [B][H][W][3C] from qkv and dout [B][H][W][C]; dtable [(2 ws - 1)^2][heads]
 * and dbias_pad [3C] (gradient reaching the qkv bias through the padding tokens) are written (per-window partial rows in
 * the workspace, summed in a fixed order; only LDS atomics inside a block) */
long long ffa_window_attention_bwd_workspace_bytes(int B, int H, int W, int C, int heads, int ws);
int ffa_window_attention_bwd(int dtype, const void* qkv, const void* dout, void* dqkv, const float* qkv_bias,
                             const float* table, float* dtable, float* dbias_pad, int B, int H, int W, int C, int heads,
                             int ws, int shift, float scale, void* workspace, long long workspace_bytes,
                             ffa_stream_t stream);
/* nn.GELU() (erf form), elementwise; n a multiple of 8 */
int ffa_gelu(int dtype, const void* x, void* y, long long n, ffa_stream_t stream);
/* nn.AdaptiveAvgPool2d(S) of smp's PSPModule: x [B][H][W][C] -> y [B][S][S][C] */
int ffa_adaptive_avg_pool(int dtype, const void* x, void* y, int B, int H, int W, int C, int S, ffa_stream_t stream);
/* F.interpolate(mode="bilinear", align_corners=...) written into channels [y_off, y_off + C) of y [B][Ho][Wo][y_pitch],
 * plus an optional dense addend [B][Ho][Wo][C] (FPNBlock: upsample + lateral) */
int ffa_bilinear_slice(int dtype, const void* x, const void* addend, void* y, int B, int Hi, int Wi, int Ho, int Wo,
                       int C, int y_pitch, int y_off, int align_corners, ffa_stream_t stream);

/* y[m] = x[m] * row_scale[m / rows_per_scale] (DropPath factor on a gradient ahead of the weight / bias reductions) */
int ffa_scale_rows(int dtype, const void* x, void* y, const float* row_scale, long long rows, int C, int rows_per_scale,
                   ffa_stream_t stream);
/* bilinear x2 followed by bilinear x1/2 (both align_corners=False) as one separable 3-tap filter with replicated edges:
 * UPerNet's placeholder FPN stage + the resize back to stride 4; symmetric, so it is its own backward.  x / y are channel
 * slices [x_off, x_off + C) / [y_off, y_off + C) of tensors with pitches x_pitch / y_pitch and the same [B][H][W] */
int ffa_updown2x_slice(int dtype, const void* x, void* y, int B, int H, int W, int C, int x_pitch, int x_off, int y_pitch,
                       int y_off, ffa_stream_t stream);
/* out[c] = sum_m x[m][c] (f32): nn.Linear's bias gradient, any width; deterministic */
long long ffa_column_sums_workspace_bytes(long long rows, int C);
int ffa_column_sums(int dtype, const void* x, float* out, long long rows, int C, void* workspace,
                    long long workspace_bytes, ffa_stream_t stream);
/* backward of ffa_bilinear_slice w.r.t. x (the addend's gradient is the slice itself) and of ffa_adaptive_avg_pool */
int ffa_bilinear_slice_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int y_pitch,
                           int y_off, int align_corners, ffa_stream_t stream);
int ffa_adaptive_avg_pool_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int S, ffa_stream_t stream);

/* ---- BatchNorm2d + ReLU + residual, MaxPool2d(3,2,1) (smp ResNet-34 encoder / UnetDecoder blocks;
 *      SURVEY.md Appendix C) ----------------------------------------------------------------------- */
long long ffa_bn_workspace_bytes(int C);
int ffa_bn_stats(int dtype, const void* x, long long npix, int C, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                 float* mean_out, float* rstd_out, void* workspace, long long workspace_bytes, ffa_stream_t stream);
/* ffa_bn_stats without the pass over the tensor: partials[nparts][2][C] come from ffa_conv2d_stats */
int ffa_bn_finalize(const float* partials, long long nparts, long long npix, int C, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* mean_out, float* rstd_out, void* workspace,
                    long long workspace_bytes, ffa_stream_t stream);
int ffa_bn_eval_params(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, ffa_stream_t stream);
int ffa_bn_apply(int dtype, const void* x, const void* residual, void* y, const float* scale, const float* shift,
                 long long npix, int C, int relu, ffa_stream_t stream);
/* relu: 0 none, 1 mask from the stored output y, 2 mask recomputed from x (y may be null; no residual) */
int ffa_bn_bwd(int dtype, const void* x, const void* dy, const void* y, const float* gamma, const float* beta,
               const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, long long npix,
               int C, int relu, void* workspace, long long workspace_bytes, ffa_stream_t stream);
/* ffa_bn_bwd in separate calls: stages bit 0 = reduction + finalize (dgamma, dbeta and the apply coefficients, left in
 * the workspace), bit 1 = apply (dx, dres) from the coefficients of an earlier call on the SAME workspace, bit 2 = the
 * reduction kernel alone, bit 3 = the finalize kernel alone (1 == 4 | 8); 3 = everything.
 * Lets a harness bracket each kernel with its own events (bench.py's HBM roofline entry). */
int ffa_bn_bwd_stages(int dtype, const void* x, const void* dy, const void* y, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta,
                      long long npix, int C, int relu, void* workspace, long long workspace_bytes, int stages,
                      ffa_stream_t stream);
/* ffa_bn_bwd as ONE co-resident kernel (bf16): x and the masked dy stay in registers across two grid-wide barriers
 * -- three passes over memory instead of five, one launch instead of three.  sync = four uint32 of device memory,
 * zero before the first call and owned by the library afterwards (sync[3] != 0: a barrier timed out, results of
 * that call are invalid).  FFA_ERR_UNSUPPORTED when dy + x do not fit the chip's register files (use ffa_bn_bwd). */
int ffa_bn_bwd_fused(int dtype, const void* x, const void* dy, const void* y, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta,
                     long long npix, int C, int relu, void* workspace, long long workspace_bytes, unsigned* sync,
                     ffa_stream_t stream);
/* ffa_bn_bwd (ReLU mask recomputed from x) when sum(g) / sum(g*x) per tile already exist (ffa_conv2d_bnbwd) */
int ffa_bn_bwd_partials(int dtype, const void* x, const void* dy, const float* partials, long long nparts,
                        const float* gamma, const float* beta, const float* mean, const float* rstd, void* dx,
                        float* dgamma, float* dbeta, long long npix, int C, void* workspace, long long workspace_bytes,
                        ffa_stream_t stream);
int ffa_channel_sums(int dtype, const void* x, long long npix, int C, float* sum_out, float* sumsq_out,
                     void* workspace, long long workspace_bytes, ffa_stream_t stream);
int ffa_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int C,
                         ffa_stream_t stream);
/* add (optional, shape of dx): a second gradient of the same tensor, e.g. the U-Net skip branch of the stem output */
int ffa_maxpool3x3s2_bwd(int dtype, const void* dy, const uint8_t* idx, const void* add, void* dx, int B, int H, int W,
                         int C, ffa_stream_t stream);

/* ---- layout hand-over at the model boundary (batch dict tensors are NCHW f32:
 *      flair_hub/data/dataloader.py:105-257, flair_zonal_detection/dataset.py:174-209) -------------- */
int ffa_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int Cp, ffa_stream_t stream);
/* uint8 NCHW raster tiles -> (x - mean[c]) / std[c] as NHWC compute tensor (the zonal dataset's normalisation,
 * flair_hub/data/utils_data/norm.py:37-44 reached from flair_zonal_detection/dataset.py:174-209, done on the device) */
int ffa_u8_nchw_to_nhwc(int dtype, const uint8_t* src, void* dst, int B, int C, int H, int W, int Cp,
                        const float* mean, const float* stdv, ffa_stream_t stream);
/* the same pass for the other sample types rasters come in (uint16 SPOT / Sentinel reflectances, int16, float32
 * elevation models): dst = (src - mean[c]) / std[c] */
#define FFA_SRC_U8 0
#define FFA_SRC_U16 1
#define FFA_SRC_I16 2
#define FFA_SRC_F32 3
int ffa_raw_nchw_to_nhwc(int dtype, int src_kind, const void* src, void* dst, int B, int C, int H, int W, int Cp,
                         const float* mean, const float* stdv, ffa_stream_t stream);
int ffa_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, int Cp, ffa_stream_t stream);

/* ---- decoder resampling (smp DecoderBlock nearest x2 + cat; flair_model.py:318-327 interpolate_map) */
int ffa_upsample_nearest2x_concat_fwd(int dtype, const void* lo, const void* skip, void* out, int B, int Hl, int Wl,
                                      int C1, int C2, ffa_stream_t stream);
/* dskip may be null: only dlo (2x2 sums of the first C1 channels) is written and the caller uses dcat[..., C1:] */
int ffa_upsample_nearest2x_concat_bwd(int dtype, const void* dcat, void* dlo, void* dskip, int B, int Hl, int Wl,
                                      int C1, int C2, ffa_stream_t stream);
int ffa_bilinear_fwd(int dtype, const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                     ffa_stream_t stream);
/* backward in gather form (each source value sums its destination footprint in a fixed order); the workspace
 * arguments are kept for ABI stability: the query returns 0 and workspace may be null */
long long ffa_bilinear_bwd_workspace_bytes(int B, int Hi, int Wi, int C);
int ffa_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                     void* workspace, long long workspace_bytes, ffa_stream_t stream);

/* ---- loss / prediction (flair_hub/tasks/module_setup.py:150-161 CrossEntropyLoss(weight);
 *      flair_hub/tasks/tasks_module.py:153,155,158; flair_zonal_detection/inference.py:300;
 *      flair_zonal_detection/postprocess.py:9-30) --------------------------------------------------- */
long long ffa_softmax_ce_workspace_bytes(void);
int ffa_softmax_ce(int dtype, const void* logits, const uint8_t* targets, const float* class_weights,
                   const float* grad_scale, float* loss, float* wsum_out, void* dlogits, uint8_t* pred,
                   long long npix, int K, int Cp, void* workspace, long long workspace_bytes, ffa_stream_t stream);
/* ffa_softmax_ce that also returns dlogit_sums[Cp]: per class, the sum over all pixels of the dlogits values it wrote
 * = the bias gradient of the convolution that produced the logits (torch: grad of segmentation_head.0.bias,
 * flair_hub/models/checkpoint.py:226), taken in the same pass instead of a column-sum pass over dlogits.  Needs dlogits
 * and 16-byte aligned tensors; FFA_ERR_UNSUPPORTED otherwise (then: ffa_channel_sums over dlogits). */
int ffa_softmax_ce_sums(int dtype, const void* logits, const uint8_t* targets, const float* class_weights,
                        const float* grad_scale, float* loss, float* wsum_out, void* dlogits, uint8_t* pred,
                        float* dlogit_sums, long long npix, int K, int Cp, void* workspace, long long workspace_bytes,
                        ffa_stream_t stream);

/* x[0..n) *= scale[0] in place, scale a device scalar; a no-op pass when the scalar is exactly 1.  Used on the
 * dlogits ffa_softmax_ce wrote in the forward pass (for an upstream gradient of 1) when autograd hands the loss a
 * different grad_output -- replaces the second softmax pass torch.autograd would make (tasks_module.py:155). */
int ffa_scale_inplace(int dtype, void* x, long long n, const float* scale, ffa_stream_t stream);
int ffa_predict_u8(int dtype, int mode, const void* logits, uint8_t* out, int B, int H, int W, int K, int Cp, int y0,
                   int x0, int h, int w, ffa_stream_t stream);
int ffa_onehot_to_index(const float* onehot, uint8_t* idx, int B, int K, int H, int W, ffa_stream_t stream);
/* counts[target][pred] += 1 (int64, accumulating): the IoU-metric state of tasks_module.py:210-212 */
int ffa_confusion_matrix(const uint8_t* pred, const uint8_t* target, long long n, int K, long long* counts,
                         ffa_stream_t stream);

/* ---- host-side tile bookkeeping, bit-exact with the reference's float64 arithmetic
 *      (flair_zonal_detection/slicing.py:51-112, flair_zonal_detection/inference.py:318-335) -------- */
typedef struct {
  double left, bottom, right, top; /* kept area (margins removed, clipped to the zone) */
  double x0, y0, x1, y1;           /* full tile box */
  long long row, col;              /* id = "1-{row}-{col}" */
} ffa_tile_t;
/* returns the number of tiles (<= capacity written to out; call with capacity 0 to count), < 0 on error.
 * clamp_is_pyfloat: 1 when the zone bounds are Python floats (rasterio array_bounds), which selects
 * Python's correctly-rounded round(v, 6) for clamped coordinates in the duplicate filter. */
long long ffa_slice_grid(double min_x, double min_y, double max_x, double max_y, double ref_left, double ref_bottom,
                         int patch_size, int margin, double resolution, ffa_tile_t* out, long long capacity);
typedef struct {
  int col_off, row_off, width, height; /* rasterio Window */
  int skip;                            /* 1 when the clipped window is empty */
} ffa_window_t;
int ffa_write_window(double left, double top, double img_left, double img_bottom, double img_right, double img_top,
                     double out_res, int pred_h, int pred_w, ffa_window_t* out);

/* ---- GeoTIFF block codecs (host side, no GPU): flair_zonal_detection/geotiff.py -------------------- */
/* The reference reads its input rasters and writes its LZW-compressed uint8 prediction rasters through
 * rasterio / GDAL / libtiff (flair_zonal_detection/dataset.py:89-117, inference.py:157-208 "compress": "lzw",
 * :342-352 dst.write).  These are the per-block byte codecs of that format: libtiff-compatible LZW (MSB-first
 * codes, early width change, table reset at 4094 entries) and the Predictor = 2 horizontal differencing. */
long long ffa_tiff_lzw_bound(long long n);
/* returns bytes written (stops at EndOfInformation, end of input or cap), < 0 on a corrupt stream */
long long ffa_tiff_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long cap);
/* returns the encoded size, FFA_ERR_WORKSPACE when cap < what the stream needs (ffa_tiff_lzw_bound(n) suffices) */
long long ffa_tiff_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long cap);
/* in place over rows x row_samples native-endian samples of sample_bytes (1, 2, 4); stride = samples between
 * horizontal neighbours of one band; undo != 0 accumulates (read), 0 differences (write) */
int ffa_tiff_hpredict(void* buf, long long rows, long long row_samples, int sample_bytes, int stride, int undo);

/* ---- hardware layout probes (tests only) ---------------------------------------------------------- */
int ffa_probe_tr16(const uint16_t* src, uint16_t* dst, ffa_stream_t stream);
int ffa_probe_mfma(const float* A, const float* B, float* D, int use_f32, ffa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLAIRHIP_H */
