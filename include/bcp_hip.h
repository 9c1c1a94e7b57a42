/* include/bcp_hip.h -- C ABI of libbcp_hip.so: the MI355X (gfx950) kernels behind the BCP training hot path.
 *
 * The reference (DeepMed-Lab-ECNU/BCP) has NO plugin / FFI interface: its seam is a set of Python
 * callables.  This header is the boundary a maintainer would bind instead; every
 * entry point names the reference code it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every function returns 0 (BCP_OK) or a negative BCP_E* code; bcp_last_error() (thread-local)
 *     explains the last failure.  Nothing throws, nothing calls exit().
 *   - all pointers are DEVICE pointers owned by the caller (outputs and workspaces included), EXCEPT the small host-side
 *     arguments that say so in their comment: `box6` / `affine6` (six ints / doubles read when the call is made), option and
 *     error strings, and the bcp_comm_* handles;
 *     `*_workspace_bytes` tells how much scratch an op needs.  The library allocates nothing and
 *     never synchronises: work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - activations are channels-last fp32: [N][D][H][W][C] (2-D: D = 1).  float* must be 16-B aligned.
 *   - labels / masks are uint8.  A "box" is int[6] = {d0, h0, w0, size_d, size_h, size_w}: the
 *     zero region of the reference's mask (mask = 1 outside the box).
 *   - `accumulate` != 0 means "+=" into the output (gradient accumulation over the two student
 *     forwards, skip-connection gradient joins).
 */
#ifndef BCP_HIP_H
#define BCP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCP_OK 0
#define BCP_EINVAL (-1)
#define BCP_ELAUNCH (-2)
#define BCP_EUNSUP (-3)

enum { BCP_ACT_NONE = 0, BCP_ACT_RELU = 1, BCP_ACT_LRELU = 2 };
enum { BCP_LOSS_LA = 0, BCP_LOSS_ACDC = 1 };
enum { BCP_CAST_I64_U8 = 0, BCP_CAST_F32_U8 = 1, BCP_CAST_U8_F32 = 2, BCP_CAST_U8_I64 = 3 };
enum { BCP_PACK_DOWN_FWD = 0, BCP_PACK_DOWN_DGRAD = 1, BCP_PACK_UP_FWD = 2, BCP_PACK_UP_DGRAD = 3, BCP_PACK_PW_FWD = 4, BCP_PACK_PW_DGRAD = 5 };
enum { BCP_WG_DOWN = 0, BCP_WG_UP = 1, BCP_WG_PW = 2 };

/* ---- library ------------------------------------------------------------------------------- */
/* ABI revision = 100 * round + change counter.  Bumped whenever an exported signature changes; a binding must refuse a library whose
 * bcp_version() differs from the header it was written against (bcp_amd/_lib.py does: a stale in-tree .so then fails at load, not
 * with shifted arguments inside a launch). */
#define BCP_ABI_VERSION 511
int bcp_version(void);
const char* bcp_last_error(void);
/* process-wide tuning / test switches (the library never reads the environment): name = a field of bcp::Options
 * (csrc/common.h: conv3_p, splitk, res_pcu, res_nt, res_tile2d_vox, conv3_cfg, wgrad_nt, wgrad_tile, tn_groups, cc_tile,
 * conv3_b6*, wgrad_b6*: which shapes run on the bf16 matrix pipe and with which tiles; norm_slabs, conv3_xcd; round 6: norm_fuse_fin, norm_own,
 * wgrad_reduce_flat, gemm_pipe (the statistics GEMMs' row-block walk prefetches the next block), cc_fuse_select / cc_border_dedupe (the largest-CC chain's
 * fused selection and wave-level pair exchange), the measurement switches gemm_walk, gemm_stat_r, norm_apply_cap, norm_apply_vec, cc_count_tile, and the MEASUREMENT-ONLY whatif bits that leave launches out -- wrong results, for pricing a change); value = decimal integer(s), comma separated for the array-valued ones; "" restores the default.
 * HOST strings.  NOT thread-safe and not per-stream: ONE Options struct per process, read by every launch on every stream and
 * device.  Set options before work is enqueued, never concurrently with launches from another thread (the launch entry points
 * themselves are re-entrant across streams / devices as long as the options stay put). */
int bcp_set_option(const char* name, const char* value);
/* writes the gcnArchName of the current device ("gfx950...") */
int bcp_device_arch(char* buf, int n);
/* hipEvent-based timing on an arbitrary stream (bench.py uses it for per-kernel roofline numbers) */
int bcp_event_create(void** ev);
int bcp_event_record(void* ev, void* stream);
int bcp_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int bcp_event_destroy(void* ev);

/* ---- bidirectional copy-paste mix: a*mask + b*(1-mask)  (LA_BCP_train.py:248-251, ACDC_BCP_train.py:372-373,
 *      train_pancreas.py:155-156; mask from utils/BCP_utils.py:18-28 context_mask / ACDC generate_mask).
 *      16-byte vector path only: W * C must be a multiple of 4 (80, 96 and 256 in the reference's configurations); other extents are
 *      rejected with BCP_EINVAL. */
int bcp_mix_box(const float* a, const float* b, float* out, int N, int D, int H, int W, int C, const int* box6 /* HOST */, void* stream);

/* ---- pseudo-labels (LA_BCP_train.py:57-60 get_cut_mask; ACDC_BCP_train.py:112-114 get_ACDC_masks) ---------- */
int bcp_plabel_bin(const float* logits /*[n_vox][2]*/, uint8_t* out, long long n_vox, float thres, void* stream);
int bcp_plabel_argmax4(const float* logits /*[n_pix][4]*/, uint8_t* out, long long n_pix, void* stream);

/* ---- largest connected component (LA_BCP_train.py:65-77, pancreas_utils.py:284-296, ACDC_BCP_train.py:89-109)
 *      connectivity = number of axes allowed to differ: 3-D 3 -> 26-conn, 2 -> 18, 1 -> 6; 2-D (D=1) 2 -> 8, 1 -> 4.
 *      seg values 0..nclass; per sample and per class the largest component is kept (ties: first in raster order). */
size_t bcp_cc_workspace_bytes(int N, int D, int H, int W, int nclass);
int bcp_cc_largest(const uint8_t* seg, uint8_t* out_u8_or_null, float* out_f32_or_null, int N, int D, int H, int W, int nclass,
                   int connectivity, void* workspace, void* stream);
/*      (ABI 511) pseudo-label + largest-CC as ONE chain, straight from the channel-last logits: get_cut_mask(out, nms=1)
 *      (LA_BCP_train.py:57-63, pancreas_utils.py:275-281; C = 2, nclass = 1) and get_ACDC_masks(output, nms=1) (ACDC_BCP_train.py:112-117;
 *      C = 4, nclass = 3, thres ignored).  seg_out receives what bcp_plabel_bin / bcp_plabel_argmax4 would write, out_* what
 *      bcp_cc_largest(seg_out, ..) would -- the same bits as the two calls, one launch fewer in front of the chain; workspace as above. */
int bcp_plabel_cc_largest(const float* logits /*[N][D][H][W][C]*/, int C, float thres, uint8_t* seg_out, uint8_t* out_u8_or_null,
                          float* out_f32_or_null, int N, int D, int H, int W, int nclass, int connectivity, void* workspace, void* stream);

/* ---- masked Dice + CE "mix_loss" (utils/BCP_utils.py:58-69 + utils/losses.py:47-77 [flavour LA, C=2];
 *      ACDC_BCP_train.py:167-179 + utils/losses.py:102-134 [flavour ACDC, C=4]).  mask_or_null: explicit uint8 mask
 *      (1 = image term) or NULL to use the box.  out3: LA {loss, ce, dice}; ACDC {dice, ce, (dice+ce)/2}.
 *      bwd writes d(g_dice*dice + g_ce*ce)/dlogits (LA: g_dice = g_ce = 0.5); g_dev_or_null = device float[g_dev_n] of upstream
 *      gradients multiplied in on the device, so autograd never has to read a scalar back (g_dev_n = 2: {dice, ce}; 1: one gradient
 *      for both terms).
 *      prev_out3_or_null + total_or_null (round 4, together or not at all): the SECOND mix_loss call of a training step passes the
 *      first call's out3 and receives the step's total loss, summed in the reference's fp32 order -- LA / pancreas loss_l + loss_u
 *      (LA_BCP_train.py:255, train_pancreas.py:166), ACDC ((unl_dice + l_dice) + (unl_ce + l_ce)) / 2 (ACDC_BCP_train.py:381-384) --
 *      so no elementwise launches sit between the forward and the backward pass. */
size_t bcp_mixloss_workspace_bytes(int N, int C);
int bcp_mixloss_fwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* mask_or_null, const int* box6 /* HOST */,
                    int N, int D, int H, int W, int C, int flavour, float w_img, float w_patch, void* workspace, float* out3,
                    const float* prev_out3_or_null, float* total_or_null, void* stream);
int bcp_mixloss_bwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* mask_or_null, const int* box6 /* HOST */,
                    int N, int D, int H, int W, int C, int flavour, const void* workspace, float g_dice, float g_ce, const float* g_dev_or_null,
                    int g_dev_n, float* dlogits, void* stream);
/* Round 5: BOTH mix_loss calls of a self-training step (LA_BCP_train.py:252-254, ACDC_BCP_train.py:370-377, train_pancreas.py:160-165) in one
 * launch pair.  logits = [2N] samples, the grouped student forward's output: call 1 = samples 0 .. N-1 with (img_l, patch_l, w_img, w_patch),
 * call 2 = samples N .. 2N-1 with (img_l2, patch_l2, w_img2, w_patch2); mask / box shared ([N] samples).  out6 = the two calls' out3 back to
 * back, total = the step's loss as prev_out3 / total of bcp_mixloss_fwd form it.  Results are bit-identical to the two calls. */
size_t bcp_mixloss_pair_workspace_bytes(int N, int C);
int bcp_mixloss_pair_fwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* img_l2, const uint8_t* patch_l2,
                         const uint8_t* mask_or_null, const int* box6 /* HOST */, int N, int D, int H, int W, int C, int flavour,
                         float w_img, float w_patch, float w_img2, float w_patch2, void* workspace, float* out6, float* total, void* stream);
int bcp_mixloss_pair_bwd(const float* logits, const uint8_t* img_l, const uint8_t* patch_l, const uint8_t* img_l2, const uint8_t* patch_l2,
                         const uint8_t* mask_or_null, const int* box6 /* HOST */, int N, int D, int H, int W, int C, int flavour,
                         const void* workspace, float g_dice, float g_ce, const float* g_dev_or_null, int g_dev_n, float* dlogits, void* stream);

/* ---- utils/losses.py:79-134 `DiceLoss.forward(inputs, target, mask=None, weight=None, softmax=False)` as the CLASS the ACDC
 *      script instantiates (ACDC_BCP_train.py:66) and calls on F.softmax(output) (:170-176): `probs` are PROBABILITIES in any
 *      dense layout (strides in elements: channel, voxel, sample -- NCHW-contiguous: V, 1, C*V; NHWC: 1, C, V*C), per-class
 *      sums over the WHOLE batch, squared denominators, smooth 1e-10 with a mask / 1e-5 without (:95-112), per-class weights
 *      (HOST float[C] or NULL = ones), result / C.  mask_mode: 0 none, 1 dense uint8 [N][V] (non-zero = counted), 2 = ones
 *      with a zero box (box6 as for bcp_mix_box), 3 = its complement.  out = float[1 + C] {loss, class_wise_dice...}.
 *      bwd writes d(g * g_dev[0] * loss)/dprobs in the layout of `probs`.  C in 2..4. */
size_t bcp_dice_prob_workspace_bytes(int C);
int bcp_dice_prob_fwd(const float* probs, long long cstride, long long vstride, long long nstride, const uint8_t* target,
                      const uint8_t* mask_or_null, int mask_mode, const int* box6 /* HOST, modes 2/3 */, int N, int D, int H, int W, int C,
                      const float* weight_host_or_null, void* workspace, float* out, void* stream);
int bcp_dice_prob_bwd(const float* probs, long long cstride, long long vstride, long long nstride, const uint8_t* target,
                      const uint8_t* mask_or_null, int mask_mode, const int* box6 /* HOST */, int N, int D, int H, int W, int C,
                      const void* workspace, const float* g_dev_or_null, float g, float* dprobs, void* stream);

/* ---- norm + activation (+Dropout3d channel scale, +elementwise dropout mask, +residual)
 *      (nn.BatchNorm3d/2d train mode networks/VNet.py:18-26, networks/unet.py:21-28; nn.InstanceNorm3d pancreas/Vnet.py:93;
 *      ReLU / LeakyReLU(0.01); Dropout3d VNet.py:165,211; Dropout unet.py:23; skip add VNet.py:220-233).
 *      G = 1: BatchNorm over all rows; G = N without gamma/beta: InstanceNorm; G > 1 WITH gamma/beta/running stats: "grouped
 *      BatchNorm" = G consecutive BatchNorm calls in one launch (statistics per group, running stats updated group after
 *      group in order) -- how the two student / teacher batches of a BCP step are normalised separately yet launched together.
 *      stats = float[5][G][C] {mean, rstd, scale = gamma*rstd, beta, unbiased var}; z = (y - mean)*scale + beta.
 *      Elementwise Dropout (nn.Dropout, unet.py:23): either elem_mask (uint8 keep bits, [rows][C]) or -- round 4 -- mask_seed_or_null +
 *      mask_p_keep: the keep bit of element i is EVALUATED in the kernels from the 64-bit seed in device memory, exactly the bit
 *      bcp_bernoulli_dev(out, n, p_keep, ., as_u8 = 1, seed_dev) would write at out[i]; forward and backward of a layer get the same seed
 *      and no mask tensor exists.  Both NULL: no elementwise dropout.  elem_scale = 1 / (1 - p) either way.
 *      Launches (round 6): statistics pass -> finalize -> apply pass; where the statistics pass leaves <= 128 partial rows per group and
 *      C <= 256 (tensors of a few MB) the apply pass reduces them itself and there is no finalize launch (option norm_fuse_fin; the
 *      fp64 sums are then added in a different fixed order: the statistics can differ in the last bit, deterministically). */
size_t bcp_norm_workspace_bytes(int G, long long rows_per_group, int C);
int bcp_norm_fwd(const float* y, int G, long long rows_per_group, int C, const float* gamma, const float* beta, float* running_mean,
                 float* running_var, float momentum, float eps, int act, const float* chan_scale, long long rows_per_sample,
                 const uint8_t* elem_mask, float elem_scale,
                 const unsigned long long* mask_seed_or_null /* DEVICE u64 */, float mask_p_keep, const float* residual, float* stats,
                 void* workspace, const double* partial_in_or_null /* [G][nb_in][C][2] from bcp_conv3_fwd_stats */, int nb_in, float* out,
                 long long out_ld /* row stride of out in floats, 0 = C.  Wider: out is the first C channels of a concat buffer
                                     (networks/unet.py:56 torch.cat([skip, up]): the skip is WRITTEN there, never copied) */,
                 float* amax_out_or_null /* device float <- max |out| (round 4: the x_amax of the conv that reads out, see bcp_conv3_fwd) */,
                 void* stream);
int bcp_norm_bwd(const float* y, const float* da, int G, long long rows_per_group, int C, const float* stats, int act,
                 const float* chan_scale, long long rows_per_sample, const uint8_t* elem_mask, float elem_scale,
                 const unsigned long long* mask_seed_or_null, float mask_p_keep, float* dgamma,
                 float* dbeta, int accumulate, void* workspace, const double* partial_in_or_null, int nb_in, float* dy,
                 float* amax_out_or_null /* device float <- max |dy|: the x_amax of the dgrad conv and of the weight gradient that read dy */,
                 void* stream);
/* partial_in: (sum dz, sum dz*xhat) partials [G][nb_in][C][2] computed by the caller -- the statistics pass over (y, da) is
 * skipped (not available together with chan_scale / elem_mask).  Producer: bcp_conv3_dgrad_bwdstats (round 3: the epilogue of the
 * bf16-pipe dgrad kernels, where the extra vector work overlaps the matrix pipe). */

/* Deep levels (rows_per_group <= 4096: the 128- / 256-channel levels of the V-Nets, the U-Net's deepest level --
 * networks/VNet.py:74-86,101-113 block_four .. block_six, networks/unet.py down4 / up1): the producing conv leaves its raw split-K
 * slabs (bcp_conv3_fwd_raw) and the ROW-MAJOR statistics pass of the norm sums them on its way in (bias first, then the slabs front
 * to back -- bit-identical to the slab-sum launch it replaces) and writes the sum once for the apply pass: one launch and one
 * round trip of the tensor fewer per layer on the step's critical path.  Same kernels and arithmetic as bcp_norm_fwd / bcp_norm_bwd
 * otherwise.  bcp_norm_slabs_ok: 1 when the shape is served (and option norm_slabs is on, the default).
 *   fwd: slabs = float[nslab][slab_stride] (nslab = 1: y itself); ysum = bias + slab 0 + slab 1 + ... (always written);
 *        out_or_null = NULL: statistics only.
 *   bwd: da_sum = sum of da_slabs (always written; it is also the gradient a skip connection forwards).
 * (Round 3's one-launch form of these -- a workgroup owning four channels of every row, TA-bound -- lives in tools/attic/norm_small.hip.) */
int bcp_norm_slabs_ok(int G, long long rows_per_group, int C);
int bcp_norm_fwd_slabs(const float* slabs, int nslab, long long slab_stride, const float* bias_or_null, float* ysum, int G,
                       long long rows_per_group, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, int act, const float* chan_scale, long long rows_per_sample, const uint8_t* elem_mask,
                       float elem_scale, const unsigned long long* mask_seed_or_null, float mask_p_keep, const float* residual, float* stats,
                       void* workspace, float* out_or_null, float* amax_out_or_null, void* stream);
int bcp_norm_bwd_slabs(const float* y, const float* da_slabs, int nslab, long long slab_stride, float* da_sum, int G,
                       long long rows_per_group, int C, const float* stats, int act, const float* chan_scale, long long rows_per_sample,
                       const uint8_t* elem_mask, float elem_scale, const unsigned long long* mask_seed_or_null, float mask_p_keep,
                       float* dgamma, float* dbeta, int accumulate, void* workspace, float* dy, float* amax_out_or_null, void* stream);

/* ---- 3x3x3 / 3x3 convolution, pad 1 (nn.Conv3d networks/VNet.py:17, nn.Conv2d networks/unet.py:19-25) on fp32 MFMA.
 *      KD = 3 (3-D) or 1 (2-D, D = 1).  Weights are packed once per optimizer step from the torch layout
 *      [Cout][Cin][KD*9]: wp_fwd feeds bcp_conv3_fwd(x -> y); wp_dgrad feeds the SAME entry point as
 *      bcp_conv3_fwd(dy -> dx, Cin/Cout swapped).  Channel counts must be multiples of 4 (padded to 16 inside). */
size_t bcp_conv3_packed_weight_floats(int Cin, int Cout, int KD);
int bcp_conv3_pack_weight(const float* w, float* wp_fwd_or_null, float* wp_dgrad_or_null, int Cin, int Cout, int KD, void* stream);
/* every conv layer of a network in one launch: descs = device array of n packed structs (40 B, 8-B aligned):
 * { const float* w; float* wp; int Cout, Cin, T, K16, N16, dgrad; }  with T = KD*9, K16/N16 = GEMM K/N extents padded to 16
 * (fwd: K16 = Cin16, N16 = Cout16; dgrad: K16 = Cout16, N16 = Cin16). */
int bcp_conv3_pack_many(const void* descs_dev, int n, void* stream);
size_t bcp_conv3_fwd_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD); /* split-K slabs of the deep levels; may be 0 */
/* x_amax_or_null (round 4, here and in _fwd_stats / _dgrad_bwdstats / _fwd_raw): device float holding max |x| over the WHOLE input
 * tensor (an upper bound is enough), as the norm pass that wrote x leaves it (bcp_norm_fwd ... amax_out).  With it, shapes that have a
 * two-plane fp16 instance take their operands as x * 2^ex = h0 + h1 (fp16) -- three v_mfma_f32_16x16x32_f16 per K block instead of six
 * bf16 ones; power-of-two pre-scales from the tensor's and the layer's weights' own maxima, undone exactly on the accumulator; error
 * against fp64 1.0-1.3x the fp32 kernel's.  NULL (or option conv3_f16 = 0): three bf16 planes, no scale needed. */
int bcp_conv3_fwd(const float* x, const float* wp, const float* bias_or_null, float* y, int N, int D, int H, int W, int Cin, int Cout,
                  int KD, int accumulate, void* workspace_or_null, const float* x_amax_or_null, void* stream);
/* fused variant: the conv epilogue also emits the (sum, sum^2) partials of y that bcp_norm_fwd needs, for `groups`
 * consecutive sample ranges; rows = bcp_conv3_stat_rows(...) (0: unavailable for this shape -> use bcp_conv3_fwd);
 * stat_partial = double[groups][rows][Cout][2], handed to bcp_norm_fwd as partial_in with nb_in = rows. */
int bcp_conv3_stat_rows(int N, int D, int H, int W, int Cin, int Cout, int KD, int groups, int has_workspace);
int bcp_conv3_fwd_stats(const float* x, const float* wp, const float* bias_or_null, float* y, int N, int D, int H, int W, int Cin,
                        int Cout, int KD, void* workspace_or_null, double* stat_partial, int groups, const float* x_amax_or_null, void* stream);
/* dgrad with the CONSUMER's norm-backward statistics in its epilogue (bf16-pipe kernels; autograd of Conv3d/Conv2d followed by
 * BatchNorm/InstanceNorm backward, networks/VNet.py:17-26): the output da feeds the norm layer whose pre-norm tensor is y_prev and
 * whose statistics are stats_prev ([5][groups][Cout] as bcp_norm_fwd leaves them); stat_partial = double[groups][rows][Cout][2]
 * receives (sum dz, sum dz * xhat), dz = da * act'(z), rows = bcp_conv3_bwdstat_rows(...) (0: unavailable -> bcp_conv3_fwd), and
 * goes to bcp_norm_bwd as partial_in / nb_in.  Cin / Cout are those of THIS launch (Cin = channels of dy, Cout = channels of da).
 * Only for norm layers without chan_scale / elem_mask. */
int bcp_conv3_bwdstat_rows(int N, int D, int H, int W, int Cin, int Cout, int KD, int groups);
int bcp_conv3_dgrad_bwdstats(const float* dy, const float* wp_dgrad, float* da, int N, int D, int H, int W, int Cin, int Cout, int KD,
                             const float* y_prev, const float* stats_prev, int act, void* workspace_or_null, double* stat_partial,
                             int groups, const float* dy_amax_or_null, void* stream);
/* raw variant for the deep levels: the kernel's split-K partial slabs are the result -- slabs = float[nslabs][N*D*H*W*Cout], no bias,
 * no slab-sum launch; bcp_norm_fwd_slabs / bcp_norm_bwd_slabs sum them on their way in.  nslabs = bcp_conv3_fwd_nslabs(...) under the
 * current options (1..8; 0: shape not served in raw mode -> bcp_conv3_fwd).  Forward and dgrad alike.  bcp_conv3_fwd_raw is told how
 * many slabs the caller allocated and refuses (nothing launched) when the launch would write a different number. */
int bcp_conv3_fwd_nslabs(int N, int D, int H, int W, int Cin, int Cout, int KD);
/* round 6: which section of the packed weight the LAST forward / dgrad launch issued by this thread read -- 1 = the fp32 pack, 2 = the three
   bf16 planes, 4 = the two fp16 planes (0: none yet).  bcp_conv3_pack_many writes, per descriptor, only the sections named in bits 8-10 of the
   descriptor's last word (0 = all three; bit 0 stays the dgrad flag; bit 11: "the previous descriptor packs the same weight tensor" -- its
   |max| partials are taken from there instead of reading the weights again): a host that repacks every weight every step (the BCP loop: the
   optimiser and the EMA change all of them) learns from this query which sections each layer's launches read and stops writing the others --
   300 of the 500 MB per LA step.  The host must hold every section a launch reads: bcp_amd/networks/_hipnet.py packs partially only in front
   of REPLAYS of recorded passes whose launches it has observed, and fully before anything else. */
int bcp_conv3_last_section(void);
int bcp_conv3_fwd_raw(const float* x, const float* wp, float* slabs, int nslab, int N, int D, int H, int W, int Cin, int Cout, int KD,
                      const float* x_amax_or_null, void* stream);
/* which matrix pipe serves bcp_conv3_fwd / bcp_conv3_fwd_stats for this shape under the current options (no launch): 0 = fp32 MFMA
 * (v_mfma_f32_16x16x4_f32), 1 = bf16 MFMA with three-piece operands (fp32-equivalent results; csrc/conv3b.hip).  Measurement record only. */
size_t bcp_conv3_fwd_path(int N, int D, int H, int W, int Cin, int Cout, int KD);
/* operand planes of the kernel serving this shape when the launch carries the tensors' maxima: 3 = three bf16 planes (six MFMAs per K block),
 * 2 = two fp16 planes (three MFMAs), 0 = not on the 16-bit matrix pipe.  wgrad != 0: the layer's weight gradient.  Measurement record only. */
int bcp_conv3_planes(int N, int D, int H, int W, int Cin, int Cout, int KD, int wgrad);
size_t bcp_conv3_wgrad_path(int N, int D, int H, int W, int Cin, int Cout, int KD);   /* the same for bcp_conv3_wgrad (csrc/conv3bw.hip) */
size_t bcp_conv3_wgrad_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD);
int bcp_conv3_wgrad(const float* x, const float* dy, float* dw /*[Cout][Cin][KD*9]*/, int N, int D, int H, int W, int Cin, int Cout,
                    int KD, int accumulate, void* workspace, const float* x_amax_or_null, const float* dy_amax_or_null /* both given: two
                    fp16 planes per operand, see bcp_conv3_fwd */, void* stream);
/* first layer, Cin = 1 -> Cout = 16 (torch weight layout used directly) */
int bcp_conv3_c1_fwd(const float* x, const float* w, const float* bias_or_null, float* y, int N, int D, int H, int W, int KD, void* stream);
/* fused variant, as bcp_conv3_fwd_stats: stat_partial = double[groups][rows][16][2], rows = bcp_conv3_c1_stat_rows(...) (0: unavailable) */
int bcp_conv3_c1_stat_rows(int N, int D, int H, int W, int KD, int groups);
int bcp_conv3_c1_fwd_stats(const float* x, const float* w, const float* bias_or_null, float* y, int N, int D, int H, int W, int KD,
                           double* stat_partial, int groups, void* stream);
/* first layer + its norm with RECOMPUTE (networks/VNet.py:17-26 block_one; networks/unet.py:19-28 in_conv): y = conv + bias never
 * reaches HBM -- pass 1 takes the statistics from the accumulators, pass 2 repeats the 27-tap MFMAs and stores
 * out = act((y - mean) * scale + beta) [* elem_mask * elem_scale]; the backward recomputes y next to da and writes dy (for
 * bcp_conv3_c1_wgrad).  stats as bcp_norm_fwd; results bit-identical to bcp_conv3_c1_fwd_stats + bcp_norm_fwd / bcp_norm_bwd. */
size_t bcp_conv3_c1_norm_workspace_bytes(int N, int D, int H, int W, int KD, int groups);   /* 0: groups do not divide N */
int bcp_conv3_c1_norm_fwd(const float* x, const float* w, const float* bias_or_null, int N, int D, int H, int W, int KD, int groups,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, int act,
                          const uint8_t* elem_mask, float elem_scale, const unsigned long long* mask_seed_or_null, float mask_p_keep,
                          float* stats, void* workspace, float* out, float* amax_out_or_null /* |max| slots of out, see bcp_norm_fwd */, void* stream);
int bcp_conv3_c1_norm_bwd(const float* x, const float* w, const float* bias_or_null, const float* da, int N, int D, int H, int W, int KD,
                          int groups, const float* stats, int act, const uint8_t* elem_mask, float elem_scale,
                          const unsigned long long* mask_seed_or_null, float mask_p_keep, float* dgamma, float* dbeta,
                          int accumulate, void* workspace, float* dy, void* stream);
int bcp_conv3_c1_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int KD, int accumulate, void* workspace,
                       void* stream);
/* Round 5: bcp_conv3_c1_norm_bwd with THIS layer's weight gradient (networks/VNet.py:17-26 block_one / networks/unet.py:19-28 in_conv,
 * backward) folded into its second pass: dy is never written (the first layer has no dgrad -- its weight gradient was dy's only reader) and
 * bcp_conv3_c1_wgrad is not needed.  dw = float[16][1][T], += when dw_accumulate; equal to the two calls it replaces up to fp32 summation
 * order.  dgamma / dbeta / accumulate as in bcp_conv3_c1_norm_bwd. */
size_t bcp_conv3_c1_norm_bwd_wgrad_workspace_bytes(int N, int D, int H, int W, int KD, int groups);
int bcp_conv3_c1_norm_bwd_wgrad(const float* x, const float* w, const float* bias_or_null, const float* da, int N, int D, int H, int W, int KD,
                                int groups, const float* stats, int act, const uint8_t* elem_mask, float elem_scale,
                                const unsigned long long* mask_seed_or_null, float mask_p_keep, float* dgamma, float* dbeta,
                                int accumulate, void* workspace, float* dw, int dw_accumulate, void* stream);

/* ---- k=2,s=2 conv (networks/VNet.py:74), k=2,s=2 transposed conv (networks/VNet.py:101), 1x1 conv (networks/unet.py:48).
 *      (D,H,W) are always the FINE grid dims.  Packed B matrices via bcp_k2_pack_weight(kind). */
int bcp_k2_pack_weight(const float* w, float* bp /*K*N floats*/, int Cin, int Cout, int kind, void* stream);
/* Every k2 / 1x1 layer of a network in ONE launch: bcp_k2_pack_desc fills a 64-byte descriptor (host memory) for
 * (w, bp, kind); the caller concatenates them, keeps the array in device memory and calls bcp_k2_pack_many(descs, n)
 * whenever the weights changed (i.e. once per training step). */
int bcp_k2_pack_desc(const float* w, float* bp, int Cin, int Cout, int kind, void* desc_out /*64 bytes, host*/);
int bcp_k2_pack_many(const void* descs_dev, int n, void* stream);
int bcp_down_fwd(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout, void* stream);
int bcp_down_dgrad(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout, int accumulate, void* stream);
int bcp_up_fwd(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* round 6: the same two forwards leaving the norm statistics of their output -- stat_partial[groups][rows][Cout][2] doubles (sum y, sum y^2)
   for bcp_norm_fwd(partial_in, nb = rows): the norm layer behind every Conv3d(k=2,s=2) / ConvTranspose3d(k=2,s=2) of the reference
   (networks/VNet.py:74-86, 101-113) then skips its statistics pass over y.  rows = bcp_k2_stat_rows(kind: 0 down / 1 transposed, ...);
   0: not available for this shape (use the plain entry points).  (D, H, W): the FINE extents, as above. */
int bcp_k2_stat_rows(int kind, int N, int D, int H, int W, int Cin, int Cout, int groups);
int bcp_down_fwd_stats(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout, double* stat_partial, int groups, void* stream);
int bcp_up_fwd_stats(const float* x, const float* bp, const float* bias, float* y, int N, int D, int H, int W, int Cin, int Cout, double* stat_partial, int groups, void* stream);
/* ... and the two dgrads leaving the BACKWARD statistics of the norm layer in front of the k2s2 / transposed conv (the autograd of
   networks/VNet.py:74-86, 101-113 followed by that of the BatchNorm / InstanceNorm + ReLU before it): the output -- after the optional += of
   a skip gradient -- is da of that layer; y_prev / stats_prev: its pre-norm tensor (laid out like dx) and statistics table (bcp_norm_fwd);
   stat_partial[groups][rows][Cin][2] doubles = (sum dz, sum dz * xhat) partial rows for bcp_norm_bwd(partial_in, nb_in = rows).  Same
   contract as bcp_conv3_dgrad_bwdstats (no chan_scale / elem_mask on that layer).  rows = bcp_k2_bwdstat_rows(kind: 0 = down-conv dgrad,
   1 = transposed-conv dgrad; (D, H, W) the FINE extents); 0: not available for this shape. */
int bcp_k2_bwdstat_rows(int kind, int N, int D, int H, int W, int Cin, int Cout, int groups);
int bcp_down_dgrad_bwdstats(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout, int accumulate, const float* y_prev, const float* stats_prev, int act, double* stat_partial, int groups, void* stream);
int bcp_up_dgrad_bwdstats(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout, int accumulate, const float* y_prev, const float* stats_prev, int act, double* stat_partial, int groups, void* stream);
/* Round 6: ConvTranspose3d(k=2,s=2) -> BatchNorm3d / InstanceNorm3d -> ReLU (+ the decoder's skip add) with the conv output RECOMPUTED instead
   of stored (networks/VNet.py:101-113 UpsamplingDeconvBlock and :268-283): the layer is HBM-bound, y = up(x) is an eighth-size tensor times a
   32 x 128 matrix, so both passes of the norm recompute it -- forward: statistics pass (nothing stored) -> finalize -> apply pass writing
   out = act((y - mean) * scale + shift) + residual; backward: statistics pass over (recomputed y, da) -> finalize -> apply pass writing dy.
   Per element the arithmetic of bcp_up_fwd + bcp_norm_fwd / bcp_norm_bwd; statistics summed as bcp_up_fwd_stats does.  stats: float[5][G][C]
   as bcp_norm_fwd leaves it (mean, rstd, scale, shift, unbiased variance); running statistics updated group after group as there.
   rows = bcp_up_norm_rows(...) > 0: shape served; workspace = bcp_up_norm_workspace_bytes(...).  (D, H, W): the FINE extents. */
int bcp_up_norm_rows(int N, int D, int H, int W, int Cin, int Cout, int groups);
size_t bcp_up_norm_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int groups);
int bcp_up_fwd_norm(const float* x, const float* bp, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int groups, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, int act, const float* residual_or_null, float* stats, void* workspace, float* out, float* amax_out_or_null, void* stream);
int bcp_up_norm_bwd(const float* x, const float* bp, const float* bias, const float* da, int N, int D, int H, int W, int Cin, int Cout, int groups, const float* stats, int act, float* dgamma, float* dbeta, int accumulate, void* workspace, float* dy, void* stream);
int bcp_up_dgrad(const float* dy, const float* bp, float* dx, int N, int D, int H, int W, int Cin, int Cout, int accumulate, void* stream);
int bcp_pw_fwd(const float* x, const float* bp, const float* bias_or_null, float* y, long long rows, int Cin, int Cout, void* stream);
size_t bcp_tn_workspace_bytes(long long M, int K, int N);
int bcp_k2_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Cin, int Cout, int kind, int accumulate,
                 void* workspace, void* stream);
/* 16 -> {2,4} pointwise output conv (networks/VNet.py:210 out_conv) and its backward (dx, dw +=, db +=);
 * bwd workspace = (Cout*17) doubles */
int bcp_pw16_fwd(const float* x, const float* w, const float* bias_or_null, float* y, long long nvox, int Cout, void* stream);
int bcp_pw16_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long nvox, int Cout,
                 int accumulate, void* workspace, void* stream);
/* The same head with the LAST 3x3x3 conv's normalisation + activation + Dropout3d channel scale applied on the way in: x_raw is the raw
 * conv output, stats = [5][G][16] as bcp_norm_fwd(out = NULL: statistics only) leaves them, chan_scale nullable [N][16] -- the
 * 16-channel activation at full resolution is never materialised (networks/VNet.py:213-216 out_conv after block_nine + dropout).
 * bcp_pw16_bwd_norm recomputes it for the weight gradient and returns dx = gradient w.r.t. that activation (for bcp_norm_bwd). */
int bcp_pw16_fwd_norm(const float* x_raw, const float* stats, const float* chan_scale, int N, int G, int act, const float* w, const float* bias,
                      float* y, long long nvox, int Cout, void* stream);
int bcp_pw16_bwd_norm(const float* x_raw, const float* stats, const float* chan_scale, int N, int G, int act, const float* dy, const float* w,
                      float* dx, float* dw, float* db, long long nvox, int Cout, int accumulate, void* workspace, void* stream);
/* Round 5: the head's backward THROUGH that norm layer in one call -- bcp_pw16_bwd_norm + bcp_norm_bwd (networks/VNet.py:213-216 backward of
 * out_conv(dropout(block_nine(...)))) without the 16-channel activation gradient ever written: pass 1 takes dw / db and the norm's backward
 * statistics from (x_raw, dy), pass 2 recomputes da = dy * w per voxel and writes dy_raw = the gradient w.r.t. x_raw.  dgamma / dbeta: the
 * norm's affine gradients (nullable pair; norm_accumulate: +=), accumulate: dw / db +=.  amax_out_or_null: |max| slots of dy_raw. */
size_t bcp_pw16_bwd_norm_bwd_workspace_bytes(int N, int G, long long nvox);
int bcp_pw16_bwd_norm_bwd(const float* x_raw, const float* stats, const float* chan_scale, int N, int G, int act, const float* dy, const float* w,
                          float* dy_raw, float* dw, float* db, float* dgamma_or_null, float* dbeta_or_null, int norm_accumulate, long long nvox,
                          int Cout, int accumulate, void* workspace, float* amax_out_or_null, void* stream);
/* column sums of [rows][C] (bias gradients of convs not followed by a norm); workspace = C doubles */
int bcp_colsum(const float* x, long long rows, int C, float* out, int accumulate, void* workspace, void* stream);

/* ---- 2-D U-Net plumbing (networks/unet.py:36-57): MaxPool2d(2), bilinear x2 align_corners=True, channel concat ---- */
/* ldx: row stride of x in floats (0 = C): x may be the first C channels of a concat buffer (bcp_norm_fwd out_ld) */
/* amax_src / amax_dst (both or neither): also copies x's |max| slots (see "per-tensor |max|" below) into the slots of the concat buffer
 * x lives in, which the upsample (bcp_bilinear2x_fwd amax_io) then max-reduces its half into */
int bcp_maxpool2d_fwd(const float* x, int ldx, float* y, int N, int H, int W, int C, const float* amax_src_or_null, float* amax_dst_or_null,
                      void* stream);
/* nn.MaxPool3d(3, stride=2), forward only: pool(x5), the V-Net's second return value (networks/VNet.py:246,286-290); x [N][D][H][W][C]
 * -> y [N][(D-3)/2+1][(H-3)/2+1][(W-3)/2+1][C] */
int bcp_maxpool3d_k3s2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, void* stream);
/* dx = scatter(dy) [+ dx when accumulate] [+ add: a second gradient of x with row stride ld_add, e.g. the skip half of the concat
 * buffer's gradient -- the join that used to be a bcp_copy_channels(accumulate) launch] */
int bcp_maxpool2d_bwd(const float* x, int ldx, const float* dy, float* dx, int N, int H, int W, int C, int accumulate,
                      const float* add_or_null, int ld_add, void* stream);
/* amax (round 4, the |max| a conv needs for its fp16 planes, see bcp_conv3_fwd), through the U-Net's skip concatenation: bcp_copy_channels
 * initialises the concat buffer's slot with the skip tensor's |max| (amax_src_or_null; NULL: 0), bcp_bilinear2x_fwd max-reduces what it
 * writes INTO the slot (amax_io_or_null) -- in that order on one stream. */
int bcp_bilinear2x_fwd(const float* x, float* y, int N, int H, int W, int C, int ldy, int y_off, float* amax_io_or_null, void* stream);
int bcp_bilinear2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, int lddy, int dy_off, void* stream);
int bcp_copy_channels(const float* src, float* dst, long long rows, int C, int ld_src, int src_off, int ld_dst, int dst_off,
                      int accumulate, const float* amax_src_or_null, float* amax_dst_or_null, void* stream);

/* ---- optimiser / mean teacher (utils/BCP_utils.py:78-81 update_ema_variables, ACDC_BCP_train.py:123-129 update_model_ema,
 *      torch.optim.SGD LA_BCP_train.py:218, torch.optim.Adam pancreas/dataloaders.py:182) over FLAT fp32 buffers ------- */
/* alpha is a double so that (1 - alpha) is formed exactly as python forms it before the fp32 multiply */
int bcp_ema(float* dst, const float* src, long long n, double alpha, void* stream);
int bcp_sgd(float* p, const float* g, float* buf, float* ema_or_null, long long n, float lr, float momentum, float weight_decay,
            float grad_scale, int first_step, double ema_alpha, void* stream);
int bcp_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, int step,
             float grad_scale, void* stream);

/* ---- validation path on the device
 *      bcp_norm_eval: BatchNorm in eval() mode -- running statistics, no update (networks/VNet.py:18-26 under model.eval());
 *        a = act((y - running_mean) * gamma / sqrt(running_var + eps) + beta) [+ residual]; gamma / beta nullable.
 *      bcp_sw_accumulate / bcp_sw_finish: the sliding-window bookkeeping of utils/test_3d_patch.py:test_single_case
 *        (:117-136): score[x0+i][y0+j][z0+k] += softmax(logits_patch[i][j][k])[cls], cnt += 1; then score /= cnt and
 *        label = score > thres.  logits_patch is [px][py][pz][C] (C = 2 or 4), score / cnt are [X][Y][Z] float32.
 *      bcp_overlap_counts: counts = {|A & B|, |A|, |B|} (device uint64[3], zeroed by the call) for the Dice / Jaccard metrics
 *        (medpy.metric.binary.dc / jc as used by utils/test_3d_patch.py:29-33,180-186). */
int bcp_norm_eval(const float* y, long long rows, int C, const float* gamma_or_null, const float* beta_or_null, const float* running_mean,
                  const float* running_var, float eps, int act, const float* residual_or_null, float* out, void* stream);
int bcp_sw_accumulate(const float* logits_patch, float* score, float* cnt, int X, int Y, int Z, int px, int py, int pz, int x0, int y0,
                      int z0, int C, int cls, void* stream);
int bcp_sw_finish(float* score, const float* cnt, uint8_t* label, long long n, float thres, void* stream);
int bcp_overlap_counts(const uint8_t* pred, const uint8_t* gt, long long n, int cls /* 0: != 0; > 0: == cls */, unsigned long long* counts,
                       void* stream);

/* ---- device-side input pipeline, LA: RandomRotFlip + RandomCrop of dataloaders/dataset.py:52-59,173-214 as one
 *      gather: dst[P0][P1][P2] = pad(flip(rot90(src[n0][n1][n2], k, axes=(0,1)), flip_axis), (pw,ph,pd))[w1:, h1:, d1:];
 *      elem_bytes = 4 (float32 image) or 1 (uint8 label).  The caller draws k, flip_axis, w1, h1, d1 (np.random, reference order).
 *      flip_axis = -1 with k = 0: no rotation / flip -- the pancreas RandomCrop / CenterCrop (pancreas/dataloaders.py:22-91), which
 *      only pad and crop. */
int bcp_crop_rotflip(const void* src, void* dst, int elem_bytes, int n0, int n1, int n2, int k, int flip_axis, int pw, int ph, int pd,
                     int w1, int h1, int d1, int P0, int P1, int P2, void* stream);

/* ---- device-side input pipeline, ACDC: RandomGenerator of dataloaders/dataset.py:69-88 as one gather:
 *      dst[OH][OW] = zoom_nearest(stage1(src[H][W])), stage1 = identity (mode 0) | flip(rot90(src, k), flip_axis) (mode 1,
 *      dataset.py:52-59) | scipy.ndimage.rotate(src, angle, order=0, reshape=False) (mode 2, dataset.py:62-66; affine6 = HOST
 *      pointer to {m00, m01, m10, m11, off0, off1} of scipy's rotation, computed by the caller with scipy.special.cosdg / sindg).
 *      elem_bytes = 4 (float32 image) or 1 (uint8 label).  The caller draws the random numbers (random.random / np.random,
 *      reference order). */
int bcp_acdc_augment(const void* src, void* dst, int elem_bytes, int H, int W, int mode, int k, int flip_axis, const double* affine6,
                     int OH, int OW, void* stream);

/* ---- small utilities ---------------------------------------------------------------------------- */
int bcp_cast(const void* in, void* out, long long n, int kind, void* stream);
int bcp_axpy(float* y, const float* x, long long n, float a, void* stream);
int bcp_bernoulli(void* out, long long n, float p_keep, float keep_value, int as_u8, unsigned long long seed, void* stream);
/* The same draw with the seed read from DEVICE memory at run time (a launch captured in a HIP graph keeps its arguments), and the
 * store that refreshes such seeds: host_values is a HOST pointer to n <= 16 values, which travel as kernel arguments. */
int bcp_bernoulli_dev(void* out, long long n, float p_keep, float keep_value, int as_u8, const unsigned long long* seed_dev, void* stream);
int bcp_store_u64(unsigned long long* dst, int n, const unsigned long long* host_values, void* stream);

/* ---- HIP graphs (no reference counterpart: its host path is PyTorch's eager dispatch).  Everything launched on `stream` (and on
 * streams that fork from / join it through events) between begin and end becomes one executable graph; bcp_graph_launch replays it
 * with one call.  Used by bcp_amd/plan.py for whole network passes. */
int bcp_graph_begin_capture(void* stream);
int bcp_graph_end_capture(void* stream, void** graph_exec);
int bcp_graph_launch(void* graph_exec, void* stream);
int bcp_graph_destroy(void* graph_exec);

/* ---- launch-plan replay in C (no reference counterpart).  A recorded network pass is a constant list of calls to the entry points
 * above: bcp_replay_add appends one -- fn = the entry point's address, shape = its argument classes, one character each (p pointer,
 * i int, l long long, f float, d double, u unsigned long long, z size_t), slots = nargs 8-byte argument images (HOST pointer, copied:
 * ints sign-extended, floats in the low four bytes) -- and bcp_replay_run makes the calls in order, returning the first non-zero
 * status.  Shapes outside csrc/replay_shapes.inc (generated from the binding table) are refused with BCP_EUNSUP-style errors. */
int bcp_replay_create(void** handle);
int bcp_replay_add(void* handle, void* fn, const char* shape, const void* slots, int nargs);
int bcp_replay_count(void* handle);
int bcp_replay_run(void* handle);
/* measurement twin (bench.py's per-op table): the same walk with HIP events recorded around chosen entries on the entry's own stream --
 * ev_before[i] in front of entry i, ev_after[i] behind it (hipEvent_t as void*, NULL: none), streams[i] = the stream entry i launches on;
 * three HOST arrays of bcp_replay_count(handle) elements */
int bcp_replay_run_timed(void* handle, void* const* ev_before, void* const* ev_after, void* const* streams);
int bcp_replay_destroy(void* handle);
/* stream ordering inside such a list: `waiter` waits for everything enqueued on `signaller` so far (event record + wait; capturable) */
int bcp_stream_wait_stream(void* waiter, void* signaller);

/* ---- data-parallel gradient exchange: all-reduce (sum, in place) of the flat fp32 gradient buffer over RCCL /
 *      xGMI.  No reference counterpart for LA / ACDC (its only multi-GPU code is nn.DataParallel, pancreas/dataloaders.py:14);
 *      this is the exchange a one-process-per-GPU run needs between loss.backward() and optimizer.step()
 *      (LA_BCP_train.py:265-267).  librccl.so is dlopen()ed on first use.  `comm`, `id128`, `world` are HOST pointers;
 *      `buf` is a DEVICE pointer; the collective is enqueued on `stream` and nothing synchronises the host.
 *      Bootstrap: rank 0 calls bcp_comm_unique_id and hands the 128 bytes to every rank (file, TCP store, ...); every rank
 *      then calls bcp_comm_init_rank (collective). */
int bcp_comm_available(void);                                  /* 1 when librccl.so could be loaded */
int bcp_comm_unique_id(void* id128);
int bcp_comm_init_rank(void** comm, int world, int rank, const void* id128);
int bcp_comm_count(void* comm, int* world);                    /* ranks the communicator really spans */
int bcp_allreduce_f32(void* comm, float* buf, long long n, void* stream);
int bcp_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* BCP_HIP_H */
