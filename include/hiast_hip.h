/*
 * hiast_hip.h — C ABI of libhiast_hip.so, the MI355X (gfx950) kernels behind HIAST's
 * self-training hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into caller-owned memory unless marked "host";
 *     the library never allocates, frees or keeps a pointer after the call returns;
 *   - head-side tensors (logits, probabilities, ASPP I/O of K1, fp32 BN of K10, K2-K8, K15, K16) are contiguous
 *     NCHW fp32; the trunk kernels (K1b, K9, K9c, K9d, K10b) take CHANNELS-LAST rows [B,H,W,C] in fp32, bf16 or
 *     bf16 split planes (hi|lo slabs) as stated per entry point; label maps are uint8 or int64 (is_i64 flag);
 *   - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*); they never
 *     synchronise the device, so a caller's DDP/compute overlap is preserved;
 *   - return value: 0 = ok, <0 = argument error (HIAST_E_*), >0 = a hipError_t;
 *   - thread-safe for distinct streams; the only state is the tuning environment (read once), the thread-local
 *     co-scheduling hint (hiast_igemm_set_cosched) and the process-wide CU reserve (hiast_set_reserve_cus).
 *
 * Each entry point cites the reference call it stands in for (paths under
 * bupt-ai-cz/HIAST `code/`).  The reference has no native code: these replace
 * torch/cuDNN/numpy calls made from its Python.
 */
#ifndef HIAST_HIP_H
#define HIAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIAST_ABI_VERSION 6

#define HIAST_E_ARG   (-1) /* null pointer / non-positive extent */
#define HIAST_E_RANGE (-2) /* extent outside what the kernels are built for */
#define HIAST_E_WS    (-3) /* workspace too small */

/* fp16 bit patterns 0x0000..0x3C00 (0.0 .. 1.0): one histogram bin per representable
 * fp16 confidence value; reproduces `probs.astype(np.float16)` exactly
 * (workflows/pseudo_label_generator.py:201). */
#define HIAST_NBINS 15361
/* Σ max-prob is accumulated exactly as an integer: prob * 2^30 (prob >= 2^-7). */
#define HIAST_PROB_FX_SHIFT 30
#define HIAST_MAX_CLASSES 32
/* Operand formats of the 16-bit channels-last kernels (K9c-K9e, K10b, K1b, K18; the `fmt` arguments below):
 *   HIAST_FMT_BF16        rows of bf16 values (slab = 64 channels)
 *   HIAST_FMT_SPLIT_BF16  fp32-class values as bf16 hi|lo planes (slab = 32 channels; inference only)
 *   HIAST_FMT_FP16        rows of IEEE fp16 values — the type the reference trains in under apex O1
 *                         (code/utils/default_config.py:109, code/utils/utils.py:126-132)
 * bf16 and fp16 rows share every layout and kernel structure; the matrix cores run both at the same rate. */
#define HIAST_FMT_BF16 1
#define HIAST_FMT_SPLIT_BF16 2
#define HIAST_FMT_FP16 3
#define HIAST_IGNORE 255

typedef void* hiast_stream_t;

int hiast_version(void);
/* host: static string for a negative HIAST_E_* code, "hip error" otherwise */
const char* hiast_error_string(int code);

/* ---- device geometry and the CU reserve (ABI 6; host) -------------------------------------------------------------------
 * Stands in for nothing in the reference (apex / NCCL on NVSwitch hardware has no such knob): under DDP + SyncBN
 * (code/workflows/trainer/base_trainer.py:43-56, code/utils/utils.py:103-105) 208 chained [C,2] all-reduces per step each need a
 * free compute unit for RCCL's kernel while the side streams keep every CU busy with 60-160 us tile-kernel blocks.
 * hiast_device_cus(): compute units of the current device (hipDeviceGetAttribute; 0 when no device is visible).
 * hiast_set_reserve_cus(n): from now on every one-block-per-CU / persistent launch of this library (xconv*, stem*, pass 1,
 *   weight-gradient work splits) sizes its grid to CUs - n (n is rounded up to a multiple of 8, one per XCD; n <= CUs / 2);
 *   returns the previous value.  Process-wide; call it once before the first launch (work splits, i.e. partial-sum row counts
 *   and summation orders, depend on it).
 * hiast_stream_create_reserved(&s, n): a stream on which NO kernel (ours or torch's) can be placed on n of the CUs (queue CU
 *   mask, hipExtStreamCreateWithCUMask; n / 8 CUs of every XCD) — what keeps the non-persistent tile kernels off them.
 *   Destroy with hiast_stream_destroy. */
int hiast_device_cus(void);
int hiast_set_reserve_cus(int n);
int hiast_get_reserve_cus(void);
int hiast_stream_create_reserved(hiast_stream_t* out, int reserve);
int hiast_stream_destroy(hiast_stream_t s);

/* ---- K2: bilinear upsample, align_corners=True ------------------------------------
 * F.interpolate(logits, size, mode='bilinear', align_corners=True):
 *   sseg/models/segmentors/self_training_segmentor.py:27, source_only_segmentor.py:19,
 *   workflows/validator.py:45,52, workflows/trainer/base_trainer.py:169,171.
 * in [B,C,h,w] -> out [B,C,H,W].  bwd is the exact adjoint (gin overwritten). */
int hiast_upsample_bilinear_ac_fwd(const float* in, float* out, int B, int C, int h, int w,
                                   int H, int W, hiast_stream_t stream);
int hiast_upsample_bilinear_ac_bwd(const float* gout, float* gin, int B, int C, int h, int w,
                                   int H, int W, hiast_stream_t stream);

/* ---- K3: pseudo-label pass 1 -------------------------------------------------------
 * F.softmax(logits,1).max(1) on the upsampled logits + the per-class fp16 confidence
 * lists of workflows/pseudo_label_generator.py:192-201, as one fused kernel:
 * logits_lr [B,C,h,w] (low-res head output; h==H,w==W accepted) ->
 *   maxprob f32 [B,H,W], argmax u8 [B,H,W] (first max on ties),
 *   hist u32 [C,HIAST_NBINS] += count of pixels of class c whose fp16(maxprob) has
 *   bit pattern `bin` (caller zeroes hist; bins accumulate across calls).
 * workspace (optional, hiast_plabel_pass1_workspace_bytes(C) bytes, 4-byte aligned, private to
 * the call until the stream has passed it): the kernel counts into a scattered copy of the
 * histogram there (hot counters on different memory lines) and a second launch adds it into
 * hist; NULL = count straight into hist (same sums, slower on peaked class distributions). */
size_t hiast_plabel_pass1_workspace_bytes(int C);
int hiast_plabel_pass1(const float* logits_lr, int B, int C, int h, int w, int H, int W,
                       float* maxprob, uint8_t* argmax, uint32_t* hist, void* workspace,
                       size_t workspace_bytes, hiast_stream_t stream);

/* ---- K4: pseudo-label pass 2 -------------------------------------------------------
 * BasePseudoGenerator.select_and_save_confident_label, pseudo_label_generator.py:67-105:
 *   plbl = argmax; plbl[maxprob < thr[argmax]] = 255; per-image per-class pixel counts;
 *   per-class Σ maxprob over kept pixels.
 * thr_up f32 [C] = smallest float32 >= the float64 threshold (so the fp32 compare equals
 * the reference's float32<float64 compare); NULL = no threshold ('NT' policy).
 * count i64 [B,C] and sumprob_fx u64 [C] (Σ prob*2^30, exact) are ACCUMULATED (caller zeroes). */
int hiast_plabel_pass2(const float* maxprob, const uint8_t* argmax, const float* thr_up,
                       int B, int C, int64_t HW, uint8_t* plbl, int64_t* count,
                       uint64_t* sumprob_fx, hiast_stream_t stream);

/* ---- K3b: CBST confidence sample -------------------------------------------------------
 * CBSTPseudoGenerator.get_constant_threshold, workflows/pseudo_label_generator.py:142-158:
 *   for c: tmp = probs_pred[lbls_pred == c].astype(float16); list_c.extend(tmp[0:len(tmp):interval])
 * over ONE batch: maxprob f32 [N], argmax u8 [N] (N = B*H*W, raster order) -> hist u32 [C,HIAST_NBINS] += the
 * kept pixels' fp16 bins (pixel kept iff (rank_offset[c] + its rank among the batch's class-c pixels) % interval == 0).
 * rank_offset i64 [C] or NULL (= 0): class-c pixels of this global batch held by lower ranks (sharded generation);
 * class_total i64 [C] or NULL: out, the batch's per-class pixel counts.  workspace: device scratch of
 * hiast_plabel_strided_hist_workspace_bytes(N, C). */
size_t hiast_plabel_strided_hist_workspace_bytes(int64_t N, int C);
int hiast_plabel_strided_hist(const float* maxprob, const uint8_t* argmax, int64_t N, int C, int interval,
                              const int64_t* rank_offset, int64_t* class_total, uint32_t* hist,
                              void* workspace, size_t workspace_bytes, hiast_stream_t stream);

/* ---- K16: multi-scale + flip test-time augmentation, fused tail -------------------------
 * Validator.get_multi_scale_and_flip_logits + argmax, workflows/validator.py:34-55,92, from the LOW-RES head outputs:
 *   out = sum_s interp( softmax(interp(z[s] -> Hs[s] x Ws[s])) + flip_w(softmax(interp(zf[s] -> ...))) -> H x W )
 * z, zf: HOST arrays of n_scales device pointers to [B,C,hs[s],ws[s]] fp32 (zf NULL or zf[s] NULL = no flip);
 * hs, ws, Hs, Ws: host int arrays.  probsum f32 [B,C,H,W] and/or label u8 [B,H,W] (either may be NULL):
 * neither full-resolution logits nor per-scale probability maps are stored. */
#define HIAST_TTA_MAX_SCALES 8
int hiast_tta_fused(const float* const* z, const float* const* zf, const int* hs, const int* ws, const int* Hs,
                    const int* Ws, int n_scales, int B, int C, int H, int W, float* probsum, uint8_t* label,
                    hiast_stream_t stream);

/* ---- K5-K8: fused self-training loss ------------------------------------------------
 * SelfTrainingSegmentor.compute_loss, self_training_segmentor.py:30-53 with
 * losses.py:32-36 (CE), :39-65,75-89 (SoftCE on a region), _kld :153-163, _entropy :140-150,
 * consuming LOW-RES student logits (and low-res teacher logits; the bilinear upsample of
 * forward() and the teacher softmax of consistency_self_training_trainer.py:113-119 are
 * recomputed in-kernel, never materialised).
 *
 * sums f64 [8] (overwritten): 0 Σ_conf -logp[y]   1 Σ_conf Σ_c -logp_c/C   2 Σ_ign Σ_c -p_c logp_c
 *   3 Σ_region Σ_c -q_c logp_c   4 N_conf   5 N_ign   6 #(q_c * -logp_c != 0 in region)   7 unused
 * The reference's losses are then  w_t*s0/s4, w_k*s1/(C*s4), w_e*s2/(C*s5), w_c*s3/s6
 * (0/0 = NaN, as in the reference).
 * region: 0 'ignored' (plbl==255), 1 'confident', 2 'all' (losses.py:77-82).
 * teacher_lr may be NULL (no consistency term; s3 = s6 = 0).
 * plbl: uint8 or int64 [B,H,W] (is_i64).
 * workspace: hiast_st_loss_workspace_bytes(...) bytes of scratch. */
size_t hiast_st_loss_workspace_bytes(int B, int C, int h, int w, int H, int W);
int hiast_st_loss_fwd(const float* logits_lr, const float* teacher_lr, const void* plbl,
                      int plbl_is_i64, int B, int C, int h, int w, int H, int W, int region,
                      double* sums, void* workspace, size_t workspace_bytes,
                      hiast_stream_t stream);
/* d(Σ_i coef_i * loss_i)/d logits_lr, with the loss_i normalised as above.
 * coef f32 [4] (device) = grad_output_i * weight_i for (CE, KLD, ENT, CST).
 * dlogits_lr [B,C,h,w] is overwritten. */
int hiast_st_loss_bwd(const float* logits_lr, const float* teacher_lr, const void* plbl,
                      int plbl_is_i64, int B, int C, int h, int w, int H, int W, int region,
                      const double* sums, const float* coef, float* dlogits_lr,
                      void* workspace, size_t workspace_bytes, hiast_stream_t stream);

/* ---- K1: ASPP head, 4 dilated 3x3 convs summed --------------------------------------
 * ASPP_V2.forward, sseg/models/modules/seg_models/deeplab_v2.py:20-24 (+ autograd):
 *   y = Σ_{i<4} conv3x3(x; W_i, b_i, dilation=dil[i], padding=dil[i]).
 * x [B,Cin,h,w]; W_i [Cout,Cin,3,3]; b_i [Cout]; y [B,Cout,h,w]; Cin % 64 == 0, Cout <= 32.
 * The 36 taps collapse to 33 (the 4 centre taps share one input pixel and are pre-summed).
 * wpack: hiast_aspp_wpack_bytes() scratch written by hiast_aspp_pack_weights (repacked
 * [33][Cin][32] weights + summed bias) and read by fwd / bwd_data; dil: host int[4].
 * workspace: hiast_aspp_workspace_bytes() scratch (split-K partial sums, fixed-order reduce:
 * results are bitwise reproducible; no float atomics). */
size_t hiast_aspp_wpack_bytes(int Cin, int Cout);
size_t hiast_aspp_workspace_bytes(int B, int Cin, int h, int w, int Cout);
int hiast_aspp_pack_weights(const float* w0, const float* w1, const float* w2, const float* w3,
                            const float* b0, const float* b1, const float* b2, const float* b3,
                            int Cin, int Cout, float* wpack, hiast_stream_t stream);
int hiast_aspp_fwd(const float* x, const float* wpack, float* y, int B, int Cin, int h, int w,
                   int Cout, const int* dil, void* workspace, size_t workspace_bytes,
                   hiast_stream_t stream);
/* dx [B,Cin,h,w] overwritten */
int hiast_aspp_bwd_data(const float* dy, const float* wpack, float* dx, int B, int Cin, int h,
                        int w, int Cout, const int* dil, hiast_stream_t stream);
/* dW_i [Cout,Cin,3,3] and db [Cout] (the same bias gradient for every branch) overwritten */
int hiast_aspp_bwd_weight(const float* x, const float* dy, float* dw0, float* dw1, float* dw2,
                          float* dw3, float* db, int B, int Cin, int h, int w, int Cout,
                          const int* dil, void* workspace, size_t workspace_bytes,
                          hiast_stream_t stream);

/* ---- K1b: the same ASPP head on CHANNELS-LAST activations, as one GEMM + a 33-tap shift-add ------------
 * (deeplab_v2.py:20-24 + autograd, as K1).  T[p][tap*Cout+co] = Σ_ci x[p][ci]*W[tap][co][ci] is a plain
 * [B*h*w x Cin] x [Cin x NP] GEMM on the bf16 matrix cores (NP = hiast_aspp2_np(Cout) = 33*Cout rounded up to
 * 128), y[co][q] = bias[co] + Σ_tap T[q+off(tap)][tap*Cout+co]: the feature map is read once instead of 33x.
 * x_nhwc [B,h,w,Cin]; dtype 0 = fp32 rows (split-bf16 arithmetic on the fly), 2 = split planes (K9c format; both
 * fp32-class: the pseudo-label forward), 1 = bf16 (training step under mixed precision; the reference trains under
 * apex O1).  y [B,Cout,h,w] fp32 (NCHW, what the loss / pseudo-label kernels read).  Cin % 128 == 0, Cout <= 32.
 * pack: wt [NP][Cin] fp32, wd [Cin][NP] fp32 (its transpose; may be NULL), bias [Cout].  fwd takes wt itself for
 * dtype 0 and hiast_pack_conv_weight(wt, NP, Cin, 1, planes = dtype) for dtype 1 | 2; bwd takes
 * hiast_pack_conv_weight(wd, Cin, NP, 1, 1).
 * bwd (bf16 x only): dy [B,Cout,h,w] fp32 -> dx_nhwc [B,h,w,Cin] bf16 (NULL = skip), dW_i [Cout,Cin,3,3] fp32 and
 * db [Cout] (all NULL = skip); fp32 accumulation, pixel-range split with a fixed-order reduce (bitwise
 * reproducible, no float atomics).  workspace: hiast_aspp2_workspace_bytes(..., backward) bytes. */
int hiast_aspp2_np(int Cout);
size_t hiast_aspp2_workspace_bytes(int B, int Cin, int h, int w, int Cout, int backward);
int hiast_aspp2_pack_weights(const float* w0, const float* w1, const float* w2, const float* w3,
                             const float* b0, const float* b1, const float* b2, const float* b3, int Cin,
                             int Cout, float* wt, float* wd, float* bias, hiast_stream_t stream);
int hiast_aspp2_fwd(const void* x_nhwc, int dtype, const void* wt, const float* bias, float* y, int B, int Cin,
                    int h, int w, int Cout, const int* dil, void* workspace, size_t workspace_bytes,
                    hiast_stream_t stream);
int hiast_aspp2_bwd(const void* x_nhwc, const float* dy, const void* wd, void* dx_nhwc, float* dw0, float* dw1,
                    float* dw2, float* dw3, float* db, int B, int Cin, int h, int w, int Cout, const int* dil,
                    int fmt /* HIAST_FMT_BF16 | HIAST_FMT_FP16: type of x, dx (and of the gathered dY) */,
                    void* workspace, size_t workspace_bytes, hiast_stream_t stream);

/* ---- K10: BatchNorm2d (+ residual) (+ ReLU), fused ----------------------------------------
 * The conv -> BN -> ReLU / conv -> BN -> (+identity) -> ReLU chains of Bottleneck.forward,
 * sseg/models/modules/resnet.py:78-98 (separate BN / add / ReLU passes in the reference).  "Frozen" BN still
 * uses BATCH statistics in train() mode (utils/utils.py:60-65).
 * x, res, y, dy, dx, dres: [B,C,HW] contiguous, dtype 0 = fp32, 1 = bf16; gamma/beta/statistics fp32 [C]
 * (gamma/beta may be NULL = 1/0).  `part` = per-plane partial sums, double [C][npart][2]; between the
 * stats call and the apply call the caller may sum `part` across ranks (SyncBN) — then pass the summed
 * buffer with npart planes and the GLOBAL element count.
 *   hiast_bn_stats        : part[c][n] = (Σx, Σx²) of plane (n,c)            (npart = B)
 *   hiast_bn_act_apply    : y = relu?(x*scale_c + shift_c (+res));  part == NULL -> inference (running
 *                           statistics), else training: mean/var from part/count, writes save_mean /
 *                           save_invstd [C] and updates running_mean/var (momentum, unbiased variance;
 *                           running_* may be NULL).
 *   hiast_bn_act_bwd_stats: g = dy*(y>0 if relu); part[c][n] = (Σg, Σ g*xhat)
 *   hiast_bn_act_bwd_apply: dx = gamma*invstd*(g - Σg/count - xhat*Σ(g*xhat)/count); dres = g (nullable);
 *                           dgamma = Σ g*xhat, dbeta = Σg (nullable). */
size_t hiast_bn_workspace_bytes(int B, int C);
int hiast_bn_stats(const void* x, int B, int C, int64_t HW, int dtype, double* part, hiast_stream_t stream);
int hiast_bn_act_apply(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, const double* part, int npart, double count,
                       float momentum, float eps, int relu, float* save_mean, float* save_invstd, int B, int C,
                       int64_t HW, int dtype, hiast_stream_t stream);
int hiast_bn_act_bwd_stats(const void* dy, const void* y, const void* x, const float* save_mean,
                           const float* save_invstd, int relu, int B, int C, int64_t HW, int dtype,
                           double* part, hiast_stream_t stream);
int hiast_bn_act_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma,
                           const float* save_mean, const float* save_invstd, const double* part, int npart,
                           double count, int relu, void* dx, void* dres, float* dgamma, float* dbeta, int B,
                           int C, int64_t HW, int dtype, hiast_stream_t stream);

/* ---- K9 (round 1): BN(eval) (+ ReLU) on channels-last rows ------------------------------------------------------------
 * hiast_bn_act_nhwc_infer: y[m][c] = act(x[m][c]*scale_c + shift_c) behind the library 7x7 stem of the inference forwards
 * (sseg/models/modules/resnet.py:177-184) when the fused stem kernels do not apply.  dtype 0: fp32 rows, 1: bf16 rows.
 * (Round 5: the round-1 convolution entries this block used to declare, hiast_conv1x1_bn_act_nhwc / hiast_conv3x3_bn_act_nhwc,
 * had no product caller since hiast_igemm_bn_act (K9c) took every trunk convolution and were REMOVED from the ABI; their
 * translation unit keeps the fp32-row GEMM behind hiast_aspp2_fwd.) */
int hiast_bn_act_nhwc_infer(const void* x, void* y, const float* gamma, const float* beta, const float* mean,
                            const float* var, float eps, int relu, int64_t M, int C, int dtype,
                            hiast_stream_t stream);

/* ---- K9c: the same fused trunk convolutions with operands pre-split into bf16 planes and staged by LDS-DMA ------
 * (resnet.py:78-98 as K9).  An fp32-class activation row of C values is stored as [hi_0..hi_{C-1} | lo_0..lo_{C-1}]
 * bf16 (hi = bf16(v), lo = bf16(v - hi); 4 bytes per value like fp32): x [B,H,W,planes,Cin], y [B,Ho,Wo,planes,Cout],
 * res like y.  planes = 2: hi*hi + lo*hi + hi*lo on the bf16 matrix cores (fp32-class, the pseudo-label forward);
 * planes = 1: plain bf16.  The split is done once by the producer's epilogue instead of by every consumer block.
 * wp: packed bf16 [Cout][taps][planes*Cin] from hiast_pack_conv_weight (w: torch layout [Cout][Cin][taps], taps = 1 | 9;
 * transpose = 1 packs the ADJOINT convolution's weight [Cin][taps flipped][planes*Cout] into wp: running the same kernel
 * on dY with it is the data gradient of a stride-1 convolution — autograd of nn.Conv2d in resnet.py:78-98;
 * transpose = 2 writes the forward weight to wp AND the adjoint to wpt in one pass; wpt is unused otherwise).
 * taps = 9: padding = dilation, stride 1 | 2.  out_f32 = 1: y is fp32 [B,Ho,Wo,Cout] (no residual).  mean == NULL: no
 * BatchNorm (plain GEMM).  Cin % 32 == 0, Cout % 64 == 0, every tensor < 2 GiB, 16-byte aligned.
 * hiast_split_planes: fp32 [M][C] <-> planes [M][2][C] (inverse = 1: x is written; hi + lo is exact in fp32).
 * stats (planes = 1, 16-bit output, no res, relu = 0; may be NULL): [ceil(B*Ho*Wo/256)][Cout][2] fp32, per 256-row block the sums
 * Σy and Σy² of the stored (bf16) outputs — the BatchNorm batch statistics of the training forward come out of the
 * convolution's epilogue (hiast_bn_nhwc_stats_from_partial reduces them) instead of another pass over y.
 * res_gate (planes = 1, 16-bit output, with res and relu = 0; may be NULL): like res; the residual is then added only where res_gate > 0 — the data
 * gradient of a bottleneck's first convolution takes the ReLU-masked gradient of the identity branch (dy of the block
 * output, gated by the block output) in its epilogue instead of a masked copy + a separate add.  gate_mask = 1:
 * res_gate is the [M][Cout/8] bit mask written by hiast_bn_nhwc_apply instead of a tensor of values. */
int hiast_igemm_bn_act(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int B, int H, int W, int Cin,
                       int Cout, int taps, int stride, int dil, int fmt /* HIAST_FMT_*: "planes" above = 2 for SPLIT_BF16, 1 otherwise;
                       FP16: everything said of planes = 1 with fp16 rows */, int out_f32, float* stats,
                       int stats_rows /* rows of `stats` the caller allocated = hiast_igemm_stats_rows(...) (ignored when stats is
                       NULL): the launch returns HIAST_E_ARG when the tile form it takes writes another number of rows (ABI 6;
                       never more than ceil(M / 128) rows) */,
                       const void* res_gate, int gate_mask, hiast_stream_t stream);
/* host: the co-scheduling hint of the CALLING THREAD (thread-local; ABI 6, replaces the HIAST_IGEMM_COSCHED environment
 * variable that round 5 set and unset around every forward): on != 0 = "this thread runs two launch sequences side by side
 * on two streams" — half-chip launches then keep the 256-row tile form (two of them fill the chip) instead of the 128 x 128 /
 * two-blocks-per-CU form.  on < 0 = back to the process default (HIAST_IGEMM_COSCHED, read once).  Returns the previous value
 * (-1 = default).  The hint enters hiast_igemm_stats_rows / hiast_igemm_dgrad_bn_stats_rows and the launches alike; a launch
 * checks the caller's row count against the form it takes. */
int hiast_igemm_set_cosched(int on);
/* host, A/B and tests: force the tile form of the calling thread's launches — 1 = the 128 x 128 form wherever it is instantiated,
 * 0 = never, < 0 = automatic (the HIAST_IGEMM_HALF environment variable, read once, else the measured rule).  Returns the
 * previous value. */
int hiast_igemm_set_half(int v);
/* host: number of partial-sum rows hiast_igemm_bn_act writes into `stats` [rows][Cout][2] for M = B*Ho*Wo output pixels
 * (one row per block of the kernel chosen for the shape) */
int hiast_igemm_stats_rows(int64_t M, int Cin, int Cout, int taps, int fmt);
/* Data gradient of a stride-1 trunk convolution, dA = conv(dy, adjoint weight) (wpt: hiast_pack_conv_weight with
 * transpose; Cin = channels of dy, Cout = channels of dA; bf16 channels-last rows), for the case that A = relu(bn(x)):
 * the epilogue also emits the per-block sums of that BatchNorm's backward pass, partial[rows][Cout][2] = (Σg, Σ g*xhat)
 * with g = dA where gamma*(x-mean)*invstd + beta > 0 (the stored bf16 dA), xhat = (x-mean)*invstd — what
 * hiast_bn_nhwc_bwd_stats(relu = 2) computes with one more read of dA and x.  rows = hiast_igemm_dgrad_bn_stats_rows(M, Cin, Cout, taps) (one row per block row of the tile form the launch takes);
 * hiast_bn_nhwc_stats_from_partial reduces them to the sums hiast_bn_nhwc_bwd_apply takes.  gamma / beta may be NULL
 * (1 / 0).  Replaces the autograd of conv -> bn -> relu (Bottleneck.forward, sseg/models/modules/resnet.py:78-98: cuDNN
 * data gradient + ATen batch_norm_backward reduce, each a pass of its own). */
int hiast_igemm_dgrad_bn_stats(const void* dy, const void* wpt, void* da, int B, int H, int W, int Cin, int Cout, int taps,
                               int dil, const void* bn_x, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, float* partial, int partial_rows /* = hiast_igemm_dgrad_bn_stats_rows(...);
                               HIAST_E_ARG on a mismatch with the tile form the launch takes (ABI 6) */,
                               int fmt /* HIAST_FMT_BF16 | HIAST_FMT_FP16 */, hiast_stream_t stream);
int hiast_igemm_dgrad_bn_stats_rows(int64_t M, int Cin, int Cout, int taps);
/* Data gradient of the trunk's 3x3 / stride-2 / padding-1 convolution (layer2.0.conv2; autograd of nn.Conv2d in
 * Bottleneck.forward, resnet.py:78-98 — round 4, was the library's): dx [B,H,W,Cin] from dy [B,(H-1)/2+1,(W-1)/2+1,Cout]
 * and the adjoint-packed weight (hiast_pack_conv_weight, transpose = 1), 16-bit rows of format fmt. */
int hiast_igemm_dgrad_s2(const void* dy, const void* wpt, void* dx, int B, int H, int W, int Cin, int Cout, int fmt,
                         hiast_stream_t stream);
/* Data gradient of conv1 of an IDENTITY bottleneck (K = 256 -> N = 1024 in layer3) that also delivers the backward sums of the
 * PREVIOUS block's bn3 (round 4; autograd of `out = relu(bn3(conv3(out)) + identity)` followed by the next block's conv1,
 * resnet.py:78-98): dx [M][N] = dy [M][K] x wpt (adjoint packed) + res where the bit of res_gate is set (the identity
 * branch's gradient of THIS block: res = gradient of its output, res_gate = gate bits of its ReLU, [M][N/8] bytes); and
 * partial fp32 [hiast_xconv_dgrad_gated_bn_stats_rows][N][2] = per-block (Σg, Σ g*xhat) with g = stored dx where the bit of
 * bn_mask is set and xhat = (bn_x - save_mean) * save_invstd: what hiast_bn_nhwc_bwd_stats computes in a pass of its own
 * over (dx, bn_x, bn_mask).  K == 256, N % 256 == 0, M >= 4096 (else rows() returns 0: use hiast_igemm_bn_act + that pass). */
int hiast_xconv_dgrad_gated_bn_stats_rows(int64_t M, int K, int N);
int hiast_xconv_dgrad_gated_bn_stats(const void* dy, const void* wpt, const void* res, const void* res_gate, const void* bn_x,
                                     const void* bn_mask, const float* save_mean, const float* save_invstd, void* dx,
                                     float* partial, int64_t M, int K, int N, int fmt, hiast_stream_t stream);
int hiast_pack_conv_weight(const float* w, int N, int K, int taps, int fmt, int transpose, void* wp, void* wpt,
                           hiast_stream_t stream);
int hiast_split_planes(float* x, void* planes, int64_t M, int C, int inverse, hiast_stream_t stream);
/* K9f hiast_stem_tail: bn1 (eval) -> ReLU -> MaxPool2d(3, stride 2, padding 1) of the stem convolution's output
 * (ResNet.forward, sseg/models/modules/resnet.py:180-184: x = conv1(x); bn1; relu; maxpool) in ONE pass, written in the
 * operand format of the trunk kernels.  x: channels-last [B,H,W,C] fp32 (dtype 0) or bf16 (dtype 1; the ReLU output is
 * rounded to bf16 before the maximum, as a bf16 module path does); out: planes = 2: split planes [B,Ho,Wo,2*C] (C % 32 == 0),
 * planes = 1: bf16 [B,Ho,Wo,C]; Ho = (H-1)/2+1, Wo = (W-1)/2+1; C % 8 == 0; gamma / beta may be NULL (1 / 0). */
int hiast_stem_tail(const void* x, int dtype /* 0 fp32 | 1 bf16 | 2 fp16 */, const float* gamma, const float* beta,
                    const float* mean, const float* var, float eps, void* out, int fmt /* HIAST_FMT_* of out */, int B, int H,
                    int W, int C, hiast_stream_t stream);
/* K9j hiast_stem_eval: the whole stem of an inference forward in ONE kernel — conv 7x7 / stride 2 / padding 3 (3 -> 64, no bias)
 * -> bn1 (eval) -> ReLU -> MaxPool2d(3, stride 2, padding 1) (ResNet.forward, sseg/models/modules/resnet.py:180-184), written in
 * the operand format of the trunk kernels; the full-resolution convolution output is never stored.
 * x: fp32 [B,3,H,W] contiguous (NCHW), w: fp32 [64,3,7,7] (nn.Conv2d layout), gamma / beta may be NULL (1 / 0);
 * out: [B,Hp,Wp,planes*64] in format fmt (HIAST_FMT_BF16 | _SPLIT_BF16 | _FP16), Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W).
 * Arithmetic: fmt's 16-bit operands (split planes: hi*hi + lo*hi + hi*lo) with fp32 accumulation. */
int hiast_stem_eval(const float* x, const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                    float eps, void* out, int fmt, int B, int H, int W, hiast_stream_t stream);

/* ---- K9k (round 4): the stem convolution of the TRAINING forward and its weight gradient -----------------------------------
 * `x = self.conv1(x)` of ResNet.forward in model.train() under apex O1 (sseg/models/modules/resnet.py:177-180: a half-
 * precision library convolution) and the autograd weight gradient of that nn.Conv2d (the image needs no data gradient).
 * x fp32 [B,3,H,W] NCHW contiguous, w fp32 [64,3,7,7] -> y [B,Hc,Wc,64] channels-last rows in format fmt (HIAST_FMT_BF16 |
 * HIAST_FMT_FP16; operands rounded to that type, fp32 accumulation), Hc = (H-1)/2+1, Wc = (W-1)/2+1, and partial fp32
 * [hiast_stem_train_blocks(B,H,W)][64][2]: per-block sums Σy, Σy² of the STORED values for the batch-statistics BatchNorm that
 * follows (hiast_bn_nhwc_apply_partial).  hiast_stem_wgrad: dw fp32 [64,3,7,7] from dy (rows like y) and the same x;
 * per-block partials in `workspace` (hiast_stem_wgrad_workspace_bytes), added in a fixed order: bitwise reproducible. */
int hiast_stem_train_blocks(int B, int H, int W);
int hiast_stem_train_fwd(const float* x, const float* w, void* y, float* partial, int fmt, int B, int H, int W,
                         hiast_stream_t stream);
size_t hiast_stem_wgrad_workspace_bytes(int B, int H, int W);
int hiast_stem_wgrad(const float* x, const void* dy, float* dw, int fmt, int B, int H, int W, void* workspace,
                     size_t workspace_bytes, hiast_stream_t stream);
/* hiast_pack_conv_weight for a list of weights in ONE launch (a trunk's 104 convolutions after every optimiser / EMA
 * update).  table: device array of records (mode = the `transpose` argument above; N, K multiples of 64, taps <= 9);
 * one block per 64 x 64 (n, k) tile of one weight: chunk_tensor[b] = record index, chunk_start[b] = index of the tile
 * within that weight, row-major over (N/64, K/64).  max_taps >= the largest `taps` of the records a block of this launch
 * may meet (sizes the LDS tile: list the 1x1 and the 3x3 weights in separate launches to keep the 1x1 blocks small). */
typedef struct { const float* w; void* wp; void* wpt; int32_t N, K, taps, planes /* = HIAST_FMT_* */, mode, pad; } hiast_pack_rec;
int hiast_pack_conv_weight_multi(const hiast_pack_rec* table, const int32_t* chunk_tensor, const int64_t* chunk_start,
                                 int n_chunks, int max_taps, hiast_stream_t stream);

/* ---- K9d: weight gradient of the trunk convolutions on channels-last bf16 activations -------------------------
 * autograd of nn.Conv2d in Bottleneck.forward (resnet.py:78-98) under mixed precision:
 *   dW[n][k][tap] = Σ_m dy[m][n] * x[m + off(tap)][k], bf16 operands, fp32 accumulation, fp32 result in torch's
 *   [Cout][Cin][kh][kw] layout.  dy [B,Ho,Wo,Cout], x [B,H,W,Cin] bf16; taps = 1 | 9 (padding = dilation, stride 1|2).
 * Cin % 256 == 0 and Cout % 256 == 0 (layer3 / layer4 of the trunk: 91 % of its FLOPs).  Pixel ranges are reduced in
 * a fixed order (bitwise reproducible).  workspace: hiast_conv_wgrad_workspace_bytes(...) (0 = unsupported shape). */
size_t hiast_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int taps);
int hiast_conv_wgrad_nhwc(const void* dy, const void* x, float* dw, int B, int H, int W, int Cin, int Cout, int taps,
                          int stride, int dil, int fmt /* HIAST_FMT_BF16 | HIAST_FMT_FP16: type of dy and x */,
                          void* workspace, size_t workspace_bytes, hiast_stream_t stream);

/* Grouped form (round 4): the weight gradients of up to 4 convolutions — the three of one bottleneck, whose backward
 * (resnet.py:78-98 under autograd) produces their operands one after the other — in ONE launch + ONE reduction.  Each job
 * is a hiast_conv_wgrad_nhwc problem (same shape limits); the jobs share the pixel-range split, so the chip is filled
 * once (one 256 x 256 fp32 partial tile per CU) for all of them instead of once per convolution: a layer3 bottleneck
 * writes 64 MB of partials instead of 192 MB and needs 2 launches instead of 6.  `jobs` is a HOST array (read during the
 * call); results are bitwise reproducible (fixed-order reduction) but differ from the one-by-one form in summation order. */
typedef struct { const void* dy; const void* x; float* dw; int32_t B, H, W, Cin, Cout, taps, stride, dil; } hiast_wgrad_job;
size_t hiast_conv_wgrad_group_workspace_bytes(const hiast_wgrad_job* jobs, int njobs);
int hiast_conv_wgrad_group_nhwc(const hiast_wgrad_job* jobs, int njobs, int fmt, void* workspace, size_t workspace_bytes,
                                hiast_stream_t stream);

/* ---- K9h: the same weight gradient for the layers BELOW 256 channels (layer1 / layer2: Cin, Cout in {64, 128, 256, ...,
 * multiples of 128}; 1x1 (stride 1) and 3x3 (any stride, 'same' padding = dil)) — autograd of nn.Conv2d in
 * sseg/models/modules/resnet.py:78-98.  x [B,H,W,Cin], dy [B,Ho,Wo,Cout] (Ho = (H-1)/stride + 1) 16-bit channels-last rows;
 * hiast_conv_wgrad_small_workspace_bytes takes the OUTPUT map size (fmt = HIAST_FMT_BF16 /
 * _FP16), dw fp32 [Cout][Cin][kh][kw], fp32 accumulation, pixel ranges reduced in a fixed order (bitwise reproducible).
 * A strided 1x1 is this call on the subsampled input. */
size_t hiast_conv_wgrad_small_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps);
int hiast_conv_wgrad_small_nhwc(const void* dy, const void* x, float* dw, int B, int H, int W, int Cin, int Cout, int taps,
                                int stride, int dil, int fmt, void* workspace, size_t workspace_bytes, hiast_stream_t stream);

/* ---- K10b: BatchNorm2d (+ residual) (+ ReLU), TRAINING mode, on channels-last bf16 activations [M = B*H*W][C] ------
 * Same arithmetic and passes as K10 (resnet.py:78-98 in train(): batch statistics even with frozen affine
 * parameters, utils/utils.py:60-65) in the layout the convolution kernels of K9c produce / consume.
 * C a power of two in [8, 2048].  sums [C][2] double: (Σx, Σx²) forward, (Σg, Σ g*xhat) backward — all-reduce them
 * across ranks between the stats and the apply call for SyncBN.  workspace: hiast_bn_nhwc_workspace_bytes(C). */
size_t hiast_bn_nhwc_workspace_bytes(int C);
/* fmt (every call below): HIAST_FMT_BF16 | HIAST_FMT_FP16, the type of the activation tensors x, res, y, dy, dx, dres */
int hiast_bn_nhwc_stats(const void* x, int64_t M, int C, double* sums, void* workspace, size_t workspace_bytes, int fmt,
                        hiast_stream_t stream);
int hiast_bn_nhwc_stats_from_partial(const float* partial, int nblk, int C, double* sums, hiast_stream_t stream);
/* single-rank forward straight from the per-block partial sums (no all-reduce point): statistics + apply */
int hiast_bn_nhwc_apply_partial(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, const float* partial, int nblk, double count,
                                float momentum, float eps, int relu, float* save_mean, float* save_invstd, int64_t M,
                                int C, void* mask, int fmt, hiast_stream_t stream);
int hiast_bn_nhwc_apply(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, const double* sums, double count, float momentum,
                        float eps, int relu, float* save_mean, float* save_invstd, int64_t M, int C, void* mask, int fmt,
                        hiast_stream_t stream);
/* apply: mask (may be NULL) receives the bits y > 0, [M][C/8] bytes (bit k of byte (m, g) = channel 8g + k).
 * relu of the backward calls: 0 = no ReLU in the forward, 1 = gate y > 0 read from y, 2 = gate recomputed as
 * x*scale + shift > 0 (only when the forward had NO residual input; y may be NULL and is not read), 3 = y points to the
 * bit mask written by the forward (1/16 of y's bytes) */
int hiast_bn_nhwc_bwd_stats(const void* dy, const void* y, const void* x, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, int relu, int64_t M, int C, double* sums,
                            void* workspace, size_t workspace_bytes, int fmt, hiast_stream_t stream);
int hiast_bn_nhwc_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, const double* sums, double count, int relu,
                            void* dx, void* dres, float* dgamma, float* dbeta, int64_t M, int C, int fmt,
                            hiast_stream_t stream);

/* ---- K11: EMA teacher update ---------------------------------------------------------
 * utils/utils.py:115-123 update_ema_model: ema = ema*gamma + p*(1-gamma) over a list of
 * tensors in ONE launch (gamma, one_minus_gamma: the float32 roundings of the Python
 * doubles, as torch's tensor*scalar does; mul, mul, add - no fma).  table: device array of n_tensors records
 * {float* ema; const float* p; int64 n}; chunk_tensor/chunk_start: device int32/int64
 * arrays of n_chunks entries mapping a 64Ki-element chunk to (tensor, first element). */
typedef struct { float* ema; const float* p; int64_t n; } hiast_ema_rec;
int hiast_ema_update(const hiast_ema_rec* table, const int32_t* chunk_tensor,
                     const int64_t* chunk_start, int n_chunks, float gamma,
                     float one_minus_gamma, hiast_stream_t stream);

/* ---- K18: MaxPool2d(3, stride 2, padding 1) on channels-last bf16 activations ------------------------------------
 * ResNet.forward, sseg/models/modules/resnet.py:184 (x = self.maxpool(x)) in the mixed-precision training forward, and
 * its autograd.  x, y, dy, dx: bf16 [B,H,W,C] / [B,Ho,Wo,C] (Ho = (H-1)/2+1, Wo = (W-1)/2+1), C % 8 == 0;
 * idx: one byte per output element, the position 3*dy + dx of the maximum inside its window (ATen keeps an int64 flat
 * index: 8x the bytes); first maximum in row-major window order, a NaN takes over, as ATen.  The backward gathers, for
 * every input pixel, the gradients of the <= 4 windows whose maximum it is (fp32 sum, one rounding). */
int hiast_maxpool3x3s2_nhwc_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C,
                                int fmt /* HIAST_FMT_BF16 | HIAST_FMT_FP16 */, hiast_stream_t stream);
int hiast_maxpool3x3s2_nhwc_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, int fmt,
                                hiast_stream_t stream);

/* ---- K14: ToTensor + Normalize on the device ------------------------------------------------------
 * transform (sseg/datasets/utils.py:37-55: torchvision ToTensor + Normalize in the DataLoader workers): img uint8
 * [B][H*W][3] (HWC as decoded) -> out float32 [B][3][H*W]; v = float(u8)/255, out = (v - mean[c])/std[c], torch's
 * operations in torch's order (bit-identical).  mean / std: HOST float[3]. */
int hiast_normalize_u8(const uint8_t* img, float* out, int B, int64_t HW, const float* mean, const float* std,
                       hiast_stream_t stream);

/* K11b: copy n_tensors small tensors in one launch (the BatchNorm buffers update_ema_model copies from the student,
 * utils/utils.py:120-123).  table: device array of {dst, src, nbytes}; one block per tensor. */
typedef struct { void* dst; const void* src; int64_t nbytes; } hiast_copy_rec;
int hiast_multi_copy(const hiast_copy_rec* table, int n_tensors, hiast_stream_t stream);

/* ---- K13: Adam step ---------------------------------------------------------------------
 * torch.optim.Adam(betas, weight_decay) as built by utils/utils.py:135-154 and stepped by BaseTrainer.update_model
 * (workflows/trainer/base_trainer.py:127-141): every parameter tensor in ONE launch, torch's single-tensor
 * formulas in its operation order (L2 weight decay folded into the gradient, bias-corrected step).
 * table: device array of records {p, g, m (exp_avg), v (exp_avg_sq), n, lr, bc1 = 1-b1^t, bc2_sqrt = sqrt(1-b2^t)};
 * chunk tables as for K11. */
typedef struct { float* p; const float* g; float* m; float* v; int64_t n; float lr; float bc1; float bc2_sqrt; float step; } hiast_adam_rec;
/* ctl (may be NULL): device control block for mixed precision with dynamic loss scaling (apex amp.scale_loss /
 * torch GradScaler, base_trainer.py:129-131): a one-thread kernel ahead of the update reads the scaler's device scalars
 * grad_scale (may be NULL = 1) and found_inf (may be NULL = 0) and writes skip / 1/scale / the running count of SKIPPED
 * steps (hiast_adam_prepare: ONCE per optimiser step, before the hiast_adam_step launches of that step); the update multiplies every gradient by 1/scale and is skipped entirely (no moment update) when found_inf != 0 —
 * without the host ever reading found_inf.  With ctl the records' bc1 / bc2_sqrt are ignored: a record's `step` field is
 * the number of steps ATTEMPTED on that tensor (this one included, host-side count) and the bias corrections are formed
 * on the device from step - ctl->skipped.  Zero ctl once when the optimiser is created. */
typedef struct { float skipped; float skip; float inv_scale; float pad[5]; } hiast_adam_ctl;
int hiast_adam_prepare(hiast_adam_ctl* ctl, const float* grad_scale, const float* found_inf, hiast_stream_t stream);
int hiast_adam_step(const hiast_adam_rec* table, const int32_t* chunk_tensor, const int64_t* chunk_start,
                    int n_chunks, double beta1, double beta2, float eps, float weight_decay, const hiast_adam_ctl* ctl,
                    hiast_stream_t stream);

/* ---- K15: discriminator input map (adversarial warm-up stage) -----------------------------
 * sseg/models/segmentors/adversarial_warmup_segmentor.py: F.interpolate(logits, size, bilinear,
 * align_corners=True) :36,:41 followed by D_preprocess_fun :26-29 — mode 0 = softmax(dim=1)
 * (AdaptSegNet), mode 1 = prob_2_entropy(softmax) :71-76 (AdvEnt):  -p log2(p + 1e-30) / log2(C).
 * logits_lr [B,C,h,w] -> out [B,C,H,W]; the full-resolution logits are never stored.
 * bwd: gout [B,C,H,W] -> dlogits_lr [B,C,h,w] (overwritten); scratch: B*C*H*W floats. */
int hiast_dinput_fwd(const float* logits_lr, int mode, float* out, int B, int C, int h, int w, int H, int W,
                     hiast_stream_t stream);
int hiast_dinput_bwd(const float* logits_lr, int mode, const float* gout, float* scratch, float* dlogits_lr,
                     int B, int C, int h, int w, int H, int W, hiast_stream_t stream);

/* ---- K12: IoU histograms --------------------------------------------------------------
 * utils/metrics.py:6-19 intersectionAndUnionGPU: pred/target int64 [N]; target==255 is
 * ignored; inter/area_pred/area_tgt i64 [K] ACCUMULATED (caller zeroes). */
int hiast_confusion_hist(const int64_t* pred, const int64_t* target, int64_t N, int K,
                         int64_t* inter, int64_t* area_pred, int64_t* area_tgt,
                         hiast_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HIAST_HIP_H */
