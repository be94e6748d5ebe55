/* libfea_unet.so -- C ABI of the MI355X-native UNet hot path.
 *
 * The reference (erichou1/fea) has no FFI layer: its hot path is the
 * nn.Module `FloorPlanSegmentationModel` (data_model/models.py:6-46) whose
 * arithmetic is dispatched by PyTorch to ATen/cuDNN kernels.  Each entry
 * point below replaces one of those dispatches; the citation names the
 * reference line whose nn layer it stands for.  A Python binding (ctypes, see
 * INTEGRATION.md and fea_amd/_lib.py) is what the reference side would add.
 *
 * Conventions
 *  - every function returns 0 (FEA_OK) or a negative FEA_E_* code; the text of
 *    the last error on the calling thread is fea_last_error().
 *  - no function allocates, synchronises or creates streams: it validates its
 *    arguments on the host and enqueues kernels on `stream` (a hipStream_t
 *    passed as void*; NULL = the default stream).  The caller owns every
 *    buffer, including workspaces, and keeps them alive until the stream work
 *    has completed.
 *  - partial-sum tables (`stats`, `part`) are scratch: the functions that reduce them
 *    (fea_bn_finalize, fea_bn_bwd_finalize, fea_reduce_partials[_n]) may fold rows in
 *    place (tables of >= 4096 rows), so a table is CONSUMED by the first call that reduces it:
 *    reduce each table once.
 *  - `dtype` is FEA_F32 or FEA_BF16: the storage type of activations and
 *    packed weights.  Accumulation is always fp32.  Parameters, gradients,
 *    BatchNorm statistics and logits are always fp32.
 *  - activations are NHWC ("channels last"); `ld*` is the distance, in
 *    elements, between two consecutive pixels, so a tensor may be a channel
 *    slice of a wider buffer (that is how torch.cat, models.py:42-45, is
 *    made free).  Channel counts seen by the GEMM kernels are multiples of 64.
 *  - a "128-byte row" holds CK consecutive channels: CK = 64 (bf16) / 32 (fp32).
 */
#ifndef FEA_UNET_H
#define FEA_UNET_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEA_OK 0
#define FEA_E_BADARG (-1)
#define FEA_E_UNSUPPORTED_SHAPE (-2)
#define FEA_E_WORKSPACE (-3)
#define FEA_E_LAUNCH (-4)

#define FEA_F32 0
#define FEA_BF16 1

#define FEA_PIXEL_TILE 256 /* output pixels per workgroup of the conv kernels */

int fea_version(void);
const char* fea_last_error(void);
/* Work-split factor of the persistent kernels (fea_conv3x3p*, fea_conv3x3_wgrad in bf16): factor x 256 workgroups of
 * 1 / factor of the tile range each.  No counterpart in the reference (its multi-GPU mode, nn.DataParallel at
 * train_direct.py:23-25, has no kernels of its own).  Default 1; the data-parallel trainer sets 4 so that workgroups
 * displaced by the all-reduce's channel kernels do not serialise behind their peers.  Process-wide; set it before the
 * first forward (table and workspace sizes, fea_conv3x3p_rows / fea_wgrad_ws_bytes, depend on it). */
int fea_set_persistent_split(int factor);
int fea_get_persistent_split(void);

/* ---- layout / packing ---------------------------------------------------- */

/* x[N,C,H,W] fp32 (NCHW, the reference's input layout, models.py:36) ->
 * 3x3 im2col rows out[N,H,W,Kpad] with k = tap*C + c (tap = 3*dy+dx, zero
 * padding 1, zero fill for k >= 9*C).  Kpad is a multiple of CK.  enc1.0
 * (models.py:10,28: Conv2d(in_channels,64,3,padding=1)) then runs as a 1x1
 * GEMM over these rows. */
int fea_im2col3x3_nchw(const float* x, void* out, int N, int C, int H, int W, int Kpad, int dtype, void* stream);

/* NHWC (ld) -> NCHW fp32, and back; used at the module boundary and by tests. */
int fea_nhwc_to_nchw_f32(const void* x, int ld, float* out, int N, int C, int H, int W, int dtype, void* stream);
int fea_nchw_f32_to_nhwc(const float* x, void* out, int ld, int N, int C, int H, int W, int dtype, void* stream);

/* Pack nn.Conv2d(ci,co,3,padding=1).weight [Co,Ci,3,3] fp32 (models.py:28,31)
 * into GEMM-B rows.  forward:  out[(chunk*9+tap)][co][j] = w[co][chunk*CK+j][tap]
 *                    dgrad:    out[(chunk*9+tap)][ci][j] = w[chunk*CK+j][ci][8-tap]
 * (dgrad of a stride-1 pad-1 3x3 conv is the same conv with flipped taps and
 * swapped channel roles). */
int fea_pack_conv3x3(const float* w, void* out, int Co, int Ci, int dgrad, int dtype, void* stream);
/* enc1.0: [Co,C,3,3] -> out[0][co][k], k = tap*C + c, zero for k >= 9C (Kpad = CK multiple). */
int fea_pack_conv3x3_im2col(const float* w, void* out, int Co, int C, int Kpad, int dtype, void* stream);
/* nn.ConvTranspose2d(ci,co,2,stride=2).weight [Ci,Co,2,2] (models.py:15-21):
 * forward:  out[chunk][ab*Co+co][j] = w[chunk*CK+j][co][a][b]     (GEMM N = 4*Co)
 * bwd-data: out[ab*(Co/CK)+chunk][ci][j] = w[ci][chunk*CK+j][a][b] (GEMM N = Ci)  */
int fea_pack_upconv2x2(const float* w, void* out, int Ci, int Co, int bwd, int dtype, void* stream);

/* All weight packs of a training step in one launch.  No counterpart in the reference (cuDNN takes nn.Conv2d's OIHW
 * weights as they are, models.py:10-24): the packs above exist because the kernels here want GEMM-ordered bf16 copies,
 * and a step re-makes ~43 of them, most far smaller than a launch is long.  jobs_dev: device array of njobs (1..64)
 * FeaPackJob in ascending first_block order; job i owns blocks [first_block_i, first_block_i + fea_pack_job_blocks(...));
 * nblocks = their sum.  kind 0: fea_pack_conv3x3(a=Co, b=Ci, c=dgrad); 1: fea_pack_conv3x3_frag(Co, Ci, dgrad);
 * 2: fea_pack_upconv2x2(a=Ci, b=Co, c=bwd); 3: fea_pack_conv3x3_first(a=Co, b=C).  Results are bit-identical to the
 * single calls. */
typedef struct FeaPackJob { const float* src; void* dst; int kind, a, b, c, first_block, reserved; } FeaPackJob;
int fea_pack_job_blocks(int kind, int a, int b, int c);
int fea_pack_batch(const void* jobs_dev, int njobs, int nblocks, int dtype, void* stream);

/* First layer, bf16 fast path: enc1[0] = Conv2d(in_channels,64,3,padding=1) (models.py:10,28)
 * straight from the fp32 NCHW input (9*C <= 64): the im2col gather happens inside the kernel.
 * Pack: out[ks][co/16][lane][8] bf16, (9C+31)/32 * Co * 32 elements. */
int fea_pack_conv3x3_first(const float* w, void* out, int Co, int C, int dtype, void* stream);
int fea_conv3x3_first(const float* x_nchw, const void* w_packed, const float* bias, void* y, int ldy, float* stats,
                      int N, int C, int H, int W, int Cout, int dtype, void* stream);
/* Rows of the BatchNorm partial table `stats` that fea_conv3x3_first writes for this shape (same layer, models.py:28-29): with
 * Cout = 64, H % 8 = 0 and W % 32 = 0 a persistent kernel serves the layer and writes one row per wave (<= 3,072 x the
 * persistent split); 0 = one row per 256-pixel tile, fea_conv_num_tiles(). */
int fea_conv3x3_first_rows(int N, int C, int H, int W, int Cout, int dtype);
/* its weight gradient: dW[co,c,3,3] = sum_p dy[p,co] x[p+tap,c] from dy NHWC bf16 and x fp32 NCHW.
 * ws: fp32 workspace of fea_first_wgrad_ws_bytes() bytes. */
size_t fea_first_wgrad_ws_bytes(int N, int H, int W, int Cout);
int fea_conv3x3_first_wgrad(const void* dy, int lddy, const float* x_nchw, float* dW, void* ws, size_t ws_bytes,
                            int N, int C, int H, int W, int Cout, int dtype, void* stream);
/* The same with the BatchNorm + ReLU backward of enc1[1..2] (models.py:29-30) fused in: `da` is the gradient of the
 * activated tensor and `y` the layer's pre-BN output; the conv-output gradient fea_bn_relu_bwd_apply would write,
 * scale * (g - coef[2c] - xhat * coef[2c+1]) with g = da * [relu(bn(y)) > 0], is formed on the fly and never stored (the
 * first layer has no data gradient, train_direct.py:104: the weight gradient is its only reader).  part[rows][Cout][2],
 * rows = fea_first_wgrad_bnbwd_rows(): slot 0 = per-split sums of that gradient (conv bias gradient via fea_reduce_partials). */
int fea_first_wgrad_bnbwd_rows(int N, int H, int W, int Cout);
int fea_conv3x3_first_wgrad_bnbwd(const void* da, int ldda, const void* y, int ldy, const float* scale, const float* shift,
                                  const float* save_mean, const float* save_invstd, const float* coef,
                                  const float* x_nchw, float* dW, float* part, void* ws, size_t ws_bytes,
                                  int N, int C, int H, int W, int Cout, int dtype, void* stream);

/* ---- implicit-GEMM convolutions (MFMA) ----------------------------------- */

/* y = conv3x3(x, w_packed) + bias          models.py:28,31 (forward) and the
 * data gradient of the same layers (autograd of train_direct.py:104) when
 * called with dgrad-packed weights and bias = NULL.
 * stats (nullable): per-workgroup partial BatchNorm sums, fp32
 * [fea_conv_num_tiles(N,H,W)][Cout][2] = (sum y, sum y^2) over the tile's
 * valid pixels, taken from the fp32 accumulators (models.py:29,32 train mode). */
int fea_conv3x3(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                float* stats, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* Data gradient of conv3x3 (w_packed = dgrad pack, GEMM K = Cin = channels of dy, GEMM N =
 * Cout = channels of da) whose output da is the activation gradient of a conv->BN->ReLU
 * layer with pre-BN output y (models.py:28-33 backward).  The first pass of that layer's
 * BatchNorm backward is fused into the store loop: part[fea_conv_num_tiles()][Cout][2] =
 * per-tile (sum g, sum g*xhat), g = da where relu(y*scale+shift) > 0, xhat = (y-mean)*invstd;
 * feed it to fea_bn_bwd_finalize with nblk = fea_conv_num_tiles().  bf16 only. */
int fea_conv3x3_dgrad_bnred(const void* dy, int lddy, const void* w_packed, void* da, int ldda,
                            const void* y, int ldy_bn, const float* scale, const float* shift,
                            const float* save_mean, const float* save_invstd, float* part,
                            int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* Persistent variants of fea_conv3x3 / fea_conv3x3_dgrad_bnred for the shallow levels (models.py:10-11,19,21: enc1,
 * enc2, dec2, dec1 -- Cin and Cout in {64,128}, H % 8 == 0, W % 32 == 0, bf16): one workgroup per CU walks a range of
 * pixel tiles with the weight fragments resident in registers.  w_frag: fea_pack_conv3x3_frag (same arguments as
 * fea_pack_conv3x3, always MFMA fragment order).  The partial-sum tables they write have fea_conv3x3p_rows() rows
 * (2 per workgroup, not one per pixel tile); fea_conv3x3p_rows() == 0 means the shape is not served: use the
 * plain entry points.  Same results as the plain entry points up to fp32 summation order.  With bias == NULL
 * (a data gradient) fea_conv3x3p's `stats` carries only sum y (slot 1 is 0).  fea_conv3x3p_bias_relu is the
 * persistent form of fea_conv3x3_bias_relu (declared below). */
int fea_pack_conv3x3_frag(const float* w, void* out, int Co, int Ci, int dgrad, int dtype, void* stream);
int fea_conv3x3p_rows(int N, int H, int W, int Cin, int Cout, int dtype);
int fea_conv3x3p(const void* x, int ldx, const void* w_frag, const float* bias, void* y, int ldy,
                 float* stats, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
int fea_conv3x3p_dgrad_bnred(const void* dy, int lddy, const void* w_frag, void* da, int ldda,
                             const void* y, int ldy_bn, const float* scale, const float* shift,
                             const float* save_mean, const float* save_invstd, float* part,
                             int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
int fea_conv3x3p_bias_relu(const void* x, int ldx, const void* w_frag, const float* bias, void* y, int ldy,
                           void* pool, int ldpool, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* Eval-mode inference with BatchNorm folded into the packed weights and the bias
 * (w' = w * gamma/sqrt(var+eps) per output channel, b' = (b - mean) * that + beta):
 * Conv2d -> BatchNorm2d(eval) -> ReLU of models.py:28-33 in one kernel,
 * y = relu(conv3x3(x, w') + b') written as the activated tensor (channel slice `ldy`);
 * when `pool` is given (H, W even) also MaxPool2d(2) of it (models.py:24,38-41),
 * [N][H/2][W/2] pixels `ldpool` elements apart.  bf16 only. */
int fea_conv3x3_bias_relu(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                          void* pool, int ldpool, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* The same for the first layer (x fp32 NCHW, weights from fea_pack_conv3x3_first). */
int fea_conv3x3_first_bias_relu(const float* x, const void* w_packed, const float* bias, void* y, int ldy,
                                int N, int C, int H, int W, int Cout, int dtype, void* stream);
/* 1x1 GEMM over pixel rows (enc1.0 on im2col rows); same epilogue as above. */
int fea_conv1x1(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                float* stats, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
int fea_conv_num_tiles(int N, int H, int W);

/* y[n,2i+a,2j+b,co] = bias[co] + sum_ci x[n,i,j,ci] w[ci,co,a,b]   models.py:15-21,42-45.
 * (H,W) are the INPUT dims; y is [N,2H,2W] with pixel stride ldy, so it can be
 * the first Cout channels of the concat buffer. */
int fea_upconv2x2_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                      int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* dx[n,i,j,ci] = sum_{a,b,co} dy[n,2i+a,2j+b,co] w[ci,co,a,b]; (H,W) = dx dims. */
int fea_upconv2x2_bwd_data(const void* dy, int lddy, const void* w_packed, void* dx, int lddx,
                           int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* fea_upconv2x2_bwd_data fused with the first pass of the BatchNorm backward of the layer that
 * FEEDS the transposed conv (conv -> BN -> ReLU, models.py:31-33): also writes
 * part[pixel tile][Cin][2] = per-tile (sum g, sum g*xhat), g = dx * (relu(bn(y)) > 0), exactly
 * as fea_conv3x3_dgrad_bnred does (rows: fea_conv_num_tiles(N,H,W)).  bf16 only. */
int fea_upconv2x2_bwd_data_bnred(const void* dy, int lddy, const void* w_packed, void* dx, int lddx,
                                 const void* y, int ldy_bn, const float* scale, const float* shift,
                                 const float* save_mean, const float* save_invstd, float* part,
                                 int N, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* ---- weight gradients (MFMA, split over pixels, deterministic) ----------- */

/* conv3x3: dW[co,ci,3,3] += sum_p dy[p,co] x[p+tap,ci] ; dW is fp32 OIHW and
 * is OVERWRITTEN.  ws: fp32 workspace of fea_wgrad_ws_bytes() bytes.
 * mode 0 = conv3x3, 1 = im2col/1x1 (dW[co, C, 3,3] from k = tap*C+c; Cin=Kpad, Creal=C),
 * mode 2 = ConvTranspose2d 2x2/s2 (rows m = ci of x, cols n = co of dy, dW[ci,co,2,2]). */
size_t fea_wgrad_ws_bytes(int mode, int N, int H, int W, int Cm, int Cn, int dtype);
int fea_conv3x3_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dW, void* ws, size_t ws_bytes,
                      int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
int fea_conv1x1_im2col_wgrad(const void* dy, int lddy, const void* xcol, int ldx, float* dW, void* ws, size_t ws_bytes,
                             int N, int H, int W, int Kpad, int Creal, int Cout, int dtype, void* stream);
/* (H,W) = dims of x (low res); dy is [N,2H,2W]. */
int fea_upconv2x2_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dW, void* ws, size_t ws_bytes,
                        int N, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* ---- BatchNorm + ReLU + MaxPool (bandwidth-bound) ------------------------- */

/* Reduce the conv kernels' partial sums (fp64 accumulate) and produce, per
 * channel: scale = gamma*invstd, shift = beta - mean*scale, save_mean,
 * save_invstd; update running_mean/var (momentum, UNBIASED var) when
 * running_* != NULL.                         nn.BatchNorm2d, models.py:29,32 */
int fea_bn_finalize(const float* stats, int ntiles, long long count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream);
/* eval mode: scale/shift from running stats. */
int fea_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, int C, void* stream);
/* a = relu(y*scale + shift) -> out_full (pixel stride ld_full; may be the skip
 * half of a concat buffer) and, when out_pool != NULL, maxpool2x2(a) -> dense
 * [N,H/2,W/2,C].                       models.py:29-30,32-33 and :24,38-41 */
int fea_bn_relu_pool(const void* y, const float* scale, const float* shift, void* out_full, int ld_full,
                     void* out_pool, int N, int H, int W, int C, int dtype, void* stream);
/* Backward of conv->BN(train)->ReLU(->pool) for one layer.
 *   g = (da_full ? da_full : 0) + route(dpool)   with route = max-pool backward
 *       (first max in row-major window order, recomputed from y) ; g *= (a > 0)
 *   pass 1: partial sums (sum g, sum g*xhat) -> part[nblk][C][2]
 *   finalize: dgamma, dbeta, and the per-channel coefficients c1,c2
 *   pass 2: dy = scale*(g - c1 - xhat*c2) -> dy (dense, C channels) and partial
 *           sums of dy (conv bias gradient) -> part[nblk][C][2] slot 0.
 * fea_bn_bwd_blocks() gives nblk for sizing `part` (fp32 [nblk][C][2]). */
int fea_bn_bwd_blocks(int N, int H, int W, int C);
int fea_bn_relu_bwd_reduce(const void* da_full, int ld_da, const void* dpool, const void* y,
                           const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                           float* part, int N, int H, int W, int C, int dtype, void* stream);
int fea_bn_bwd_finalize(const float* part, int nblk, long long count, const float* gamma, const float* save_invstd,
                        float* dgamma, float* dbeta, float* coef /*[C][2]*/, int C, void* stream);
int fea_bn_relu_bwd_apply(const void* da_full, int ld_da, const void* dpool, const void* y,
                          const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                          const float* coef, void* dy, float* part, int N, int H, int W, int C, int dtype, void* stream);
/* sum over blocks of slot `slot` of part[nblk][C][2] -> out[C] (conv/upconv bias gradients). */
int fea_reduce_partials(const float* part, int nblk, int C, int slot, float* out, void* stream);
/* the same for the first n channels of rows that are ld channels wide.  With the `stats` of a
 * dgrad fea_conv3x3 call that writes the concat-buffer gradient, slot 0 of channels [0,C) is the
 * ConvTranspose2d bias gradient (models.py:15-21,42-45 backward). */
int fea_reduce_partials_n(const float* part, int nblk, int ld, int n, int slot, float* out, void* stream);
/* per-channel sum over all pixels of a [npix, C] (ld) tensor -> part[nblk][C][2] slot 0
 * (ConvTranspose2d bias gradient). */
int fea_channel_sum(const void* x, int ld, long long npix, int C, float* part, int* nblk_out, int dtype, void* stream);

/* ---- 1x1 head (models.py:23,46) -------------------------------------------- */
/* logits[N,Ccls,H,W] fp32 NCHW = out.weight[Ccls,64] . a[N,H,W,64] + bias */
int fea_head_fwd(const void* a, int lda, const float* w, const float* bias, float* logits,
                 int N, int H, int W, int Cin, int Ccls, int dtype, void* stream);
/* da[N,H,W,64] = W^T dlogits ; part[nblk][Ccls*(Cin+1)] partial sums of dW (and db in the
 * last column); nblk = fea_head_bwd_blocks(). */
int fea_head_bwd_blocks(int N, int H, int W);
int fea_head_bwd(const float* dlogits, const void* a, int lda, const float* w, void* da, int ldda,
                 float* part, int N, int H, int W, int Cin, int Ccls, int dtype, void* stream);
int fea_head_bwd_finalize(const float* part, int nblk, float* dW, float* db, int Cin, int Ccls, void* stream);

/* ---- loss / mask / IoU (not in the reference) -------- */
/* sums[Ccls][4] (fp64) = (sum bce, sum p*t, sum p, sum t) per class over N,H,W. */
int fea_bce_dice_sums(const float* logits, const uint8_t* target, double* sums, float* part,
                      int N, int Ccls, long long HW, void* stream);
int fea_loss_blocks(int N, int Ccls, long long HW);
/* fea_bce_dice_sums WITHOUT a caller workspace, for callers that cannot allocate (code under HIP
 * graph capture, tight evaluation / serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The per-block partial sums go to a fixed 1 MiB device array inside
 * the library (one copy per GPU, held until the process ends); the call allocates nothing, frees
 * nothing and gives bit-identical `sums` to fea_bce_dice_sums.  Limit: fea_loss_blocks(N, Ccls, HW)
 * <= 65536, else FEA_E_WORKSPACE (batch 64 x 6 classes x 512^2 needs 12288).  The first call on a
 * device looks the array's address up (make it once before capturing a graph).  All calls on one
 * GPU share the array: calls that overlap in time on one GPU (two streams) race -- serialise them. */
int fea_bce_dice_sums_static(const float* logits, const uint8_t* target, double* sums,
                             int N, int Ccls, long long HW, void* stream);
/* loss[0] = BCE(mean) + 1 - mean_c dice_c ; dlogits = d loss / d logits * gscale */
int fea_bce_dice_grad(const float* logits, const uint8_t* target, const double* sums, float* loss, float* dlogits,
                      float gscale, int N, int Ccls, long long HW, void* stream);
/* mask = logits > 0 (nullable out); inter/union int64[Ccls] are ACCUMULATED (caller zeroes).
   target == NULL: threshold only (inference without ground truth, inference.py:39-40): mask is required, inter/uni
   are not touched and may be NULL. */
int fea_mask_iou(const float* logits, const uint8_t* target, uint8_t* mask, long long* inter, long long* uni,
                 int N, int Ccls, long long HW, void* stream);
/* Head + loss fused for training loops: the logits and their gradient stay in registers.
 *   fea_head_loss_sums: logits = out.weight . a + bias (models.py:23,46) formed per pixel, (sum bce, sum p*t, sum p,
 *     sum t) per class -> sums[Ccls][4] (fp64) as fea_bce_dice_sums gives them; logits (nullable) are stored only on
 *     request.  part: fp32 workspace of fea_head_bwd_blocks(N,H,W) * Ccls * 4 elements.
 *   fea_head_loss_bwd: recomputes the logits, forms dlogits = d loss / d logits * gscale from `sums` in registers, and
 *     does fea_head_bwd's work: da = W^T dlogits and part[nblk][Ccls*65] for fea_head_bwd_finalize; loss[0] (nullable)
 *     = BCE(mean) + 1 - mean_c dice_c.
 * Need H*W % 4 == 0 and Ccls <= 8 (fea_head_loss_supported); otherwise use the separate entry points. */
int fea_head_loss_supported(int H, int W, int Ccls);
int fea_head_loss_sums(const void* a, int lda, const float* w, const float* bias, const uint8_t* target,
                       float* logits, double* sums, float* part, int N, int H, int W, int Cin, int Ccls,
                       int dtype, void* stream);
int fea_head_loss_bwd(const void* a, int lda, const float* w, const float* bias, const uint8_t* target,
                      const double* sums, float* loss, void* da, int ldda, float* part, float gscale,
                      int N, int H, int W, int Cin, int Ccls, int dtype, void* stream);
/* The same two entry points with BatchNorm + ReLU of dec1[4..5] (models.py:32-33) applied in their loaders: `y` is the
 * PRE-BatchNorm output of dec1[3]; relu(y * bn_scale + bn_shift), rounded to the compute dtype as fea_bn_relu_pool stores
 * it, is formed in registers, so the activated tensor out = self.out(d1) reads (models.py:46) is never written.
 * bf16, H*W % 16 == 0, up to 6 classes: fea_head_loss_bn_supported(). */
int fea_head_loss_bn_supported(int N, int H, int W, int Ccls, int dtype);
int fea_head_loss_sums_bn(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                          const uint8_t* target, float* logits, double* sums, float* part, int N, int H, int W, int Cin,
                          int Ccls, int dtype, void* stream);
int fea_head_loss_bwd_bn(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                         const uint8_t* target, const double* sums, float* loss, void* da, int ldda, float* part, float gscale,
                         int N, int H, int W, int Cin, int Ccls, int dtype, void* stream);
/* dec1[4..5]'s BatchNorm + ReLU BACKWARD inside the fused head + loss backward as well: the gradient da of the activated
 * tensor is never stored.  _bnred = fea_head_loss_bwd_bn without the da store, plus the first pass of the BatchNorm backward
 * (what fea_bn_relu_bwd_reduce computes from da and y) -> bn_part[fea_head_bwd_blocks()][64][2]; after fea_bn_bwd_finalize,
 * _bnapply forms da again and stores dy = the conv-output gradient fea_bn_relu_bwd_apply would write (bit-identical), with
 * per-block sums of dy in bn_part slot 0 (conv bias gradient via fea_reduce_partials).  Same support as the _bn forms. */
int fea_head_loss_bwd_bnred(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* save_mean,
                            const float* save_invstd, const float* w, const float* bias, const uint8_t* target,
                            const double* sums, float* loss, float* part, float* bn_part, float gscale,
                            int N, int H, int W, int Cin, int Ccls, int dtype, void* stream);
int fea_head_loss_bwd_bnapply(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* save_mean,
                              const float* save_invstd, const float* coef, const float* w, const float* bias,
                              const uint8_t* target, const double* sums, void* dy, int lddy, float* bn_part, float gscale,
                              int N, int H, int W, int Cin, int Ccls, int dtype, void* stream);
/* target = grey > 127 */
int fea_binarise(const uint8_t* grey, uint8_t* out, long long n, void* stream);


/* ---- optimiser step --------------------------------------
 * torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.AdamW.step()
 * (train_direct.py:28-32,105-106) over the flat gradient buffer, two launches:
 * fea_grad_sumsq: sumsq[0] (fp64, device) = sum of g[i]^2, i < n; part: fea_sumsq_blocks(n) floats.
 * fea_adamw_clip_step: seg = DEVICE table [nseg][3] of (parameter chunk address, element offset
 * of the chunk in g/m/v, element count <= fea_opt_chunk()); the gradient is scaled by
 * min(1, max_norm / (sqrt(sumsq) + 1e-6)) on the fly (max_norm <= 0: no clipping) and never
 * rewritten.  AdamW arithmetic as torch.optim.AdamW: p *= 1 - lr*wd; m = lerp(m, g, 1-b1);
 * v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  Hyper-parameters
 * are doubles: the derived constants are formed in double as Python does, then rounded to fp32. */
int fea_opt_chunk(void);
int fea_sumsq_blocks(long long n);
int fea_grad_sumsq(const float* g, long long n, float* part, double* sumsq, void* stream);
int fea_adamw_clip_step(const long long* seg, int nseg, const float* g_flat, float* m_flat, float* v_flat,
                        const double* sumsq, double max_norm, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int step, void* stream);

/* ---- input pipeline ------------------------------------------------
 * The reference's per-image transform, custom_dataset.py:23-28 (applied at :69-71):
 * Resize((S,S)) on a PIL image -> ToTensor() -> Normalize(mean, std).  The resize is
 * Pillow's antialiased bilinear resample (the library torchvision forwards PIL images to),
 * reproduced bit for bit in fixed point. */
/* HOST helpers (no GPU work): taps per output sample; tables bounds[2*out] = (first input
 * sample, count) and coeffs[out*ksize] (22-bit fixed point) as Pillow computes them.
 * fea_resize_coeffs returns ksize (>0) or a negative error. */
int fea_resize_ksize(int in_size, int out_size);
int fea_resize_coeffs(int in_size, int out_size, int* bounds, int* coeffs);
/* HOST helper: lut[c*256+v] = ((float)v / 255 - mean[c]) / std[c] in float32, the exact
 * value ToTensor + Normalize give for byte v of channel c. */
int fea_normalize_lut(const float* mean, const float* stdv, int C, float* lut);
/* src u8 [src_h][src_w][C] (C = 1 or 3, device) -> tmp u8 [src_h][W][C] -> dst:
 *   epilogue 0: u8  [H][W][C]  resized image
 *   epilogue 1: f32 [C][H][W]  lut[c][v]         (device lut of fea_normalize_lut: the module's input)
 *   epilogue 2: u8  [C][H][W]  v > threshold     (mask target, threshold 127)
 * x/y tables: DEVICE copies of fea_resize_coeffs(src_w, W) / (src_h, H). */
int fea_resize_u8(const uint8_t* src, int src_h, int src_w, int C, uint8_t* tmp, void* dst, int H, int W,
                  const int* xbounds, const int* xcoef, int xks, const int* ybounds, const int* ycoef, int yks,
                  int epilogue, const float* lut, int threshold, void* stream);

#ifdef __cplusplus
}
#endif
#endif
