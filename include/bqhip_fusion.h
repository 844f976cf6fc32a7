/* bqhip_fusion.h -- C ABI of the MI355X (gfx950) kernels behind the BLIP 2D-3D fusion half of the hot path and the
 * bf16 / point-major forms of the grouping operator.  Same conventions as bqhip.h (raw device pointers, extents,
 * strides in ELEMENTS, a hipStream_t passed as void*, int status, bq_last_error()).
 *
 * The reference has no native code here: these entry points replace COMPOSITIONS of torch operators in the
 * reference's Python, cited per function.  bridgeqa_amd/fusion_ops.py is the only caller in this repo; the
 * reference-side binding a maintainer would add is in INTEGRATION.md §3.
 *
 * Test infrastructure never links this file's library into the oracle; see oracle/README in DESIGN.md §2. */
#ifndef BQHIP_FUSION_H
#define BQHIP_FUSION_H
#include "bqhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- multi-head attention, head dim 64, bf16 operands, fp32 accumulation (csrc/attn.hip) -------------------
 * Replaces  attn = (q @ k^T) * scale [+ mask]; attn = softmax(attn); attn = dropout(attn); out = attn @ v
 *   models/vit.py:75-83 (Attention.forward), models/med.py:179-217 (BertSelfAttention.forward).
 * Q (B,Lq,H,64), K / V (B,Lk,H,64) by strides (q_bs,q_rs,q_hs = batch, token, head; 64 contiguous elements; V strided
 * like K); O (B,Lq,H,64) by strides; LSE f32 [B*H][Lq] (log2 domain).  mask: NULL or f32 [B][Lkp] additive key mask
 * ALREADY multiplied by log2(e), Lkp a multiple of 64 >= Lk.  p_drop / seed / seed_ptr: dropout on the probabilities
 * as a stateless hash of (seed_ptr[0]*2654435761 + seed, b*H+h, query, key) -- nothing stored, the backward
 * regenerates it.  causal != 0 (Lq == Lk): key j visible to query i only if j <= i (med.py:771-830 decoder mask).
 * No transposed operand copies: V^T / K^T / Q^T / dO^T are read out of the row-major LDS tiles (ds_read_b64_tr_b16). */
BQ_API int bq_attn_fwd(const void *Q, const void *K, const void *V, void *O, float *LSE, const float *mask, int B,
                       int H, int Lq, int Lk, int Lkp, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs,
                       long k_hs, long o_bs, long o_rs, long o_hs, float scale, float p_drop, unsigned seed,
                       const unsigned *seed_ptr, int causal, void *stream);

/* Backward of bq_attn_fwd (what autograd derives from the composition above).  dQ strided like Q, dK/dV like K;
 * dO by its own strides (g_*); O contiguous (B,Lq,H,64) and LSE from the forward; DELTA f32 [B*H][Lq] scratch. */
BQ_API int bq_attn_bwd(const void *Q, const void *K, const void *V, const void *dO, const float *LSE, const void *O,
                       float *DELTA, const float *mask, void *dQ, void *dK, void *dV, int B, int H, int Lq, int Lk,
                       int Lkp, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, long g_bs, long g_rs,
                       long g_hs, float scale, float p_drop, unsigned seed, const unsigned *seed_ptr, int causal,
                       void *stream);

/* Two narrow attentions (1 <= Lq <= 32 queries, more than 128 keys, one key segment, no causal order) in ONE launch per
 * kernel: the two cross-attentions of a twin level (reference med.py:549-614: text queries over cat(image tokens, 3D
 * states) and over cat(object tokens, 2D states)) are latency-bound launches of B * H workgroups each; side by side the
 * short one runs under the long one.  Per side: the arguments of bq_attn_fwd / bq_attn_bwd (out = O in the forward, dQ in
 * the backward; o_* strides describe O in the forward, dO in the backward; O contiguous (B, Lq, H, 64)). */
typedef struct bq_attn_side {
  const void *Q, *K, *V, *dO, *O;
  void *out, *dK, *dV;
  float *LSE, *DELTA;
  const float *mask;
  int Lq, Lk, Lkp;
  long q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, o_bs, o_rs, o_hs;
  unsigned seed;
} bq_attn_side;
BQ_API int bq_attn_fwd_pair(const bq_attn_side *sides, int B, int H, float scale, float p_drop, const unsigned *seed_ptr,
                            void *stream);
BQ_API int bq_attn_bwd_pair(const bq_attn_side *sides, int B, int H, float scale, float p_drop, const unsigned *seed_ptr,
                            void *stream);

/* The attention probabilities of a bq_attn_fwd call, rebuilt from Q, K and its LSE -- what the reference returns under
 * output_attentions (models/med.py:202,223: the softmax BEFORE dropout; BLIP_VQA3D keeps the last level's cross-attention
 * maps, blip_vqa_3d.py:262-281): P f32 (B, H, Lq, Lk) = exp2(scale*log2e * q.k + mask - LSE), causal flag as in the
 * forward.  p_drop = 0 gives that map; the forward's p_drop / seed / seed_ptr give the dropped map it multiplied V with. */
BQ_API int bq_attn_probs(const void *Q, const void *K, const float *LSE, const float *mask, float *P, int B, int H, int Lq,
                         int Lk, int Lkp, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, float scale,
                         float p_drop, unsigned seed, const unsigned *seed_ptr, int causal, void *stream);

/* ABI 6.  Which passes of the UNMASKED attention (no mask, not causal, no dropout: the ViT's, models/vit.py:72-84) run on the
 * resident-grid kernels of csrc/attn.hip -- 3 / 3 / 2 workgroups per CU walking (head, row block) items as one stream of
 * LDS tiles -- instead of one workgroup per block: bit 0 the forward, bit 1 the dQ pass, bit 2 the dK/dV pass.  Default 7;
 * returns the previous mask.  Results are the same to rounding (tests/test_attn_gpu.py compares both against fp32 torch);
 * the switch exists for A/B timing (tools/bench_attn.py). */
BQ_API int bq_attn_set_persistent(int mask);

/* Attention of Lq <= 32 queries over cat(segment 1, segment 2) along the key axis without the concatenated tensor:
 * replaces  encoder_hidden_states = torch.cat([image_embeds | object_embeds, other stream's states], dim=1)  followed
 * by the cross-attention of models/med.py:549-562, 179-217 (BertEncoderTwin / BertSelfAttention).  K / V = segment 1
 * (B,Lk,H,64) by strides k_*, K2 / V2 = segment 2 (B,Lk2,H,64) by strides k2_*.  mask: NULL or f32 [B][Lkp] with
 * Lkp = 64*(ceil(Lk/64) + ceil(Lk2/64)): segment 1's keys at [0,Lk), segment 2's at [64*ceil(Lk/64), ..+Lk2), times
 * log2(e), 0 in the padding.  No causal form.  Backward: dK/dV strided like K, dK2/dV2 like K2. */
BQ_API int bq_attn_fwd2(const void *Q, const void *K, const void *V, const void *K2, const void *V2, void *O, float *LSE,
                        const float *mask, int B, int H, int Lq, int Lk, int Lk2, int Lkp, long q_bs, long q_rs,
                        long q_hs, long k_bs, long k_rs, long k_hs, long k2_bs, long k2_rs, long k2_hs, long o_bs,
                        long o_rs, long o_hs, float scale, float p_drop, unsigned seed, const unsigned *seed_ptr,
                        void *stream);
BQ_API int bq_attn_bwd2(const void *Q, const void *K, const void *V, const void *K2, const void *V2, const void *dO,
                        const float *LSE, const void *O, float *DELTA, const float *mask, void *dQ, void *dK, void *dV,
                        void *dK2, void *dV2, int B, int H, int Lq, int Lk, int Lk2, int Lkp, long q_bs, long q_rs,
                        long q_hs, long k_bs, long k_rs, long k_hs, long k2_bs, long k2_rs, long k2_hs, long g_bs,
                        long g_rs, long g_hs, float scale, float p_drop, unsigned seed, const unsigned *seed_ptr,
                        void *stream);

/* ---- y = LayerNorm(path(dropout(x)) + residual) (csrc/ln.hip) -----------------------------------------------
 * Replaces  hidden = dense(x); hidden = dropout(hidden); hidden = LayerNorm(hidden + input)
 *   models/med.py:236-239 (BertSelfOutput), :313-317 (BertOutput)   [residual != NULL, p_drop]
 * and       x = x + drop_path(f(norm(x)))  followed by the next norm(x)
 *   models/vit.py:106-109 (Block.forward)                            [sum_out != NULL, p_path, rows_per_sample]
 * and plain LayerNorm (residual NULL).  x, residual, y, sum_out bf16 (M,H) row-major, H in {256,512,768,1024};
 * gamma/beta f32 (H); mean/rstd f32 (M) saved for the backward; zero_out: f32 (2,H) cleared by this launch (the
 * backward's dgamma/dbeta accumulator) or NULL. */
BQ_API int bq_drop_add_ln_fwd(const void *x, const void *residual, const float *gamma, const float *beta, void *y,
                              void *sum_out, float *mean, float *rstd, float *zero_out, int M, int H, float eps,
                              float p_drop, float p_path, int rows_per_sample, unsigned seed,
                              const unsigned *seed_ptr, void *stream);
/* dgb f32 (2,H) = dgamma, dbeta ACCUMULATED (must be zero on entry: the forward's zero_out); dsum = gradient that
 * reached sum_out (or NULL); dresidual NULL iff residual NULL. */
BQ_API int bq_drop_add_ln_bwd(const void *x, const void *residual, const float *gamma, const void *dy,
                              const void *dsum, const float *mean, const float *rstd, void *dx, void *dresidual,
                              float *dgb, int M, int H, float eps, float p_drop, float p_path, int rows_per_sample,
                              unsigned seed, const unsigned *seed_ptr, void *stream);

/* ABI 6.  The same backward for a site without dropout whose forward wrote sum_out (the pre-LN residual update of
 * models/vit.py:106-109), from that STORED SUM instead of x and residual: one row read instead of two.  dx = dz * path scale,
 * dresidual = dz; dresidual may be NULL when p_path == 0 (both gradients are the same tensor then).  The normalised value is
 * re-formed from the bf16 sum (bq_drop_add_ln_bwd re-forms it from x + residual in fp32: one rounding of the row apart). */
BQ_API int bq_drop_add_ln_bwd_sum(const void *sum, const float *gamma, const void *dy, const void *dsum, const float *mean,
                                  const float *rstd, void *dx, void *dresidual, float *dgb, int M, int H, float eps,
                                  float p_path, int rows_per_sample, unsigned seed, const unsigned *seed_ptr, void *stream);

/* ---- bias gradient: out[n] = sum_m g[m][n], g bf16 (M,N), out f32 (N) (csrc/ln.hip) -------------------------
 * Replaces grad_output.sum(0) of torch's LinearBackward (every nn.Linear of vit.py / med.py).
 * C = bq_colsum_chunks(M) row chunks; C > 1 needs partial (C*N floats) and, for M <= 2048, counter ((N+255)/256
 * zeroed unsigned ints, left zero; not shared with a concurrent launch).  Fixed summation order. */
BQ_API int bq_colsum_chunks(int M);
BQ_API int bq_colsum_bf16(const void *g, float *out, int M, int N, float *partial, unsigned *counter, void *stream);

/* ---- grouping operator, bf16 / point-major forms (csrc/pn2_ops.hip) -----------------------------------------
 * bq_group_concat with a bf16 (B,3+C,M,S) result / bf16 incoming gradient */
BQ_API int bq_group_concat_bf16(const float *xyz, const float *new_xyz, const float *features, const int32_t *idx,
                                void *out, int B, int C, int N, int M, int S, float radius, int normalize,
                                void *stream);
BQ_API int bq_group_concat_grad_bf16(const void *grad_out, const int32_t *idx, float *grad_features, float *grad_xyz,
                                     float *grad_new_xyz, int B, int C, int N, int M, int S, float radius,
                                     int normalize, void *stream);
/* point-major: feats rows (b, n) of C floats at feats + b*f_bs + n*f_rs; out rows (b, m, s) of ld >= 3+C elements, f32
 * or bf16, elements 3+C .. ld-1 zeroed (ld = 3+C rounded up to 8: 16-byte aligned rows for the SharedMLP GEMM of
 * bq_pwconv_bn_fwd; the next level's grouping reads the same rows without a transpose);
 * same values as bq_group_concat (pointnet2_utils.py:348-359). */
BQ_API int bq_group_concat_pm(const float *xyz, const float *new_xyz, const float *feats, long f_bs, long f_rs,
                              const int32_t *idx, void *out, int out_bf16, int B, int C, int N, int M, int S,
                              float radius, int normalize, int ld, void *stream);
/* grad_out rows of ld elements; grad_feats (B,N,C) point-major, zero_init (or NULL); grad_xyz / grad_new_xyz as
 * bq_group_concat_grad */
BQ_API int bq_group_concat_pm_grad(const void *grad_out, int in_bf16, const int32_t *idx, float *grad_feats,
                                   float *grad_xyz, float *grad_new_xyz, int B, int C, int N, int M, int S,
                                   float radius, int normalize, int ld, void *stream);

/* ---- training-mode BatchNorm2d + ReLU (+ max over nsample) on point-major rows (csrc/bn.hip) ------------------
 * Replaces conv -> BatchNorm2d -> ReLU of one SharedMLP layer (lib/pointnet2/pytorch_utils.py:104-157) and, for the
 * last layer of a set-abstraction module, F.max_pool2d(kernel=[1, nsample]) (pointnet2_modules.py:259-262).
 * x: bf16 (R, C), rows = (b, m, s) = the NHWC convolution output, C a power of two in [8, 2048].
 * bq_bn_stats: batch statistics -> scale = gamma*rstd, shift = beta - mean*scale, mean, rstd (f32 C each);
 *   running_mean / running_var (momentum, unbiased variance) and num_batches_tracked (int64) updated when non-NULL;
 *   partial = bq_bn_chunks(R, 0, 0) * 2C floats of scratch.
 * bq_bn_apply: y = relu?(x*scale + shift), bf16 (R, C); pool != 0: y bf16 (R/S, C) = max over each run of S rows.
 * bq_bn_backward: dy as y; dgb f32 (2, C) = dbeta | dgamma; dx bf16 (R, C); pooled gradients go to the first row
 *   attaining the maximum; partial = bq_bn_chunks(R, S, pool) * 2C floats. */
BQ_API int bq_bn_chunks(long R, int S, int pool);
BQ_API int bq_bn_stats(const void *x, long R, int C, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, long long *num_batches_tracked, float eps, float momentum, float *partial,
                       float *scale, float *shift, float *mean, float *rstd, void *stream);
BQ_API int bq_bn_apply(const void *x, const float *scale, const float *shift, void *y, long R, int C, int S, int relu,
                       int pool, void *stream);
BQ_API int bq_bn_backward(const void *dy, const void *x, const float *scale, const float *shift, const float *mean,
                          const float *rstd, float *partial, float *dgb, void *dx, long R, int C, int S, int relu,
                          int pool, void *stream);
/* ABI 5.  bq_bn_apply_arg: bq_bn_apply that also records, for a pooled layer, which row 0 .. S-1 of every group holds the
 * group's FIRST maximum per channel: arg u8 (R / S, C), may be NULL, S <= 256 -- the index F.max_pool2d's backward routes the
 * gradient to (pointnet2_modules.py:259-262), so that the backward does not search for it again.
 * bq_bn_backward_reduce: the reduction half of bq_bn_backward alone (dgb = dbeta | dgamma; no dx).
 * bq_bn_backward_reduce_arg: the same for a pooled layer from its arg table: one stored pre-activation value read per
 * (group, channel) instead of all S; C <= 256, 256 % C == 0; partial = bq_bn_chunks(R, S, 1) * 2C floats. */
BQ_API int bq_bn_apply_arg(const void *x, const float *scale, const float *shift, void *y, void *arg, long R, int C, int S,
                           int relu, int pool, void *stream);
BQ_API int bq_bn_backward_reduce(const void *dy, const void *x, const float *scale, const float *shift, const float *mean,
                                 const float *rstd, float *partial, float *dgb, long R, int C, int S, int relu, int pool,
                                 void *stream);
BQ_API int bq_bn_backward_reduce_arg(const void *dy, const void *x, const void *arg, const float *scale, const float *shift,
                                     const float *mean, const float *rstd, float *partial, float *dgb, long R, int C, int S,
                                     int relu, void *stream);
/* ---- the backward of a SharedMLP layer in one pass over its activations (csrc/detbwd.hip, ABI 5) ------------------------
 * Replaces, after the BatchNorm reduction (dgb from bq_bn_backward_reduce[_arg]), bq_bn_backward's dx pass, the dX GEMM and
 * bq_wgrad_rows_bf16 of one conv -> BatchNorm2d -> ReLU (-> max_pool2d) layer (lib/pointnet2/pytorch_utils.py:104-157 and
 * its autograd): x bf16 (R, ldx) the layer's input rows (whole padded rows, ldx % 8 == 0, ldx <= 192), p bf16 (R, Nj) the
 * stored pre-activation (Nj = 64 or 128; 256 for a pooled layer with ldx <= 128), dout bf16 (R, Nj), or (R / S, Nj) with arg u8 (R / S, Nj) when pool != 0 (S = 16,
 * 32 or 64, R % S == 0), w bf16 (Nj, ldw) zero beyond the input channels (ldw >= ldx rounded up to 64), scale / shift / mean /
 * rstd f32 (Nj) of the stored pre-activation ->
 *   dx bf16 (R, ldx) = dP w                  (NULL: not needed; the padding columns come out 0 because w's are)
 *   dw f32 (Nj, ldo) = dP^T x                (columns [ldx, ldo) set to 0; ldo % 4 == 0)
 * with dP = scale (g - dbeta / R - xhat dgamma / R) formed tile by tile in LDS and never stored.  part: scratch of
 * bq_sa_bwd_workgroups(R, ldx, Nj, pool, dx != NULL) * Nj * ldo floats (per-workgroup slices, summed in a fixed order: no
 * atomics).  bq_sa_bwd_supported: 1 when a kernel exists for the shape (it fits the 160 KB LDS). */
BQ_API int bq_sa_bwd_supported(int ldx, int Nj, int S, int pool, int need_dx);
BQ_API int bq_sa_bwd_workgroups(long R, int ldx, int Nj, int pool, int need_dx);
BQ_API int bq_sa_bwd_fused(const void *x, const void *p, const void *dout, const void *arg, const void *w, const float *scale,
                           const float *shift, const float *mean, const float *rstd, const float *dgb, void *dx, float *dw,
                           float *part, long R, int ldx, int Nj, int ldw, int ldo, int S, int relu, int pool, void *stream);
/* Deferred activations between the layers of one SharedMLP (ABI 5): the BatchNorm + ReLU between two convolutions is never
 * materialised.  bq_pwconv_bn_fwd_x = bq_pwconv_bn_fwd whose x holds the PREVIOUS layer's stored pre-activation (K = ldx = 64
 * or 128): relu(x xscale + xshift) -- the previous layer's scale / shift as bq_pwconv_bn_fwd wrote them -- is applied to every
 * x tile as it arrives in LDS.  bq_sa_bwd_fused_x = bq_sa_bwd_fused on such an input (the weight gradient contracts with the
 * re-formed activation; dx, required, is the gradient w.r.t. the activation, i.e. the previous layer's dout).
 * xscale == NULL: the plain entries. */
BQ_API int bq_pwconv_bn_fwd_x(const void *x, const float *xscale, const float *xshift, long R, int K, int ldx, const void *w,
                              int ldw, int Kc, int N, void *y, float *partial, const float *gamma, const float *beta,
                              float *running_mean, float *running_var, long long *num_batches_tracked, float eps,
                              float momentum, float *scale, float *shift, float *mean, float *rstd, const float *center,
                              float *shift_acc, void *stream);
/* ... and with the PREVIOUS layer's BatchNorm reduction riding on the pass: that layer's dOut is this call's dx, its stored
 * pre-activation this call's x.  xmean / xrstd: its mean / rstd; red_part: bq_sa_bwd_workgroups(...) * 4 * 2 * ldx floats of
 * scratch; red_dgb f32 (2, ldx) = its dbeta | dgamma (what bq_bn_backward_reduce returns for it).  red_dgb == NULL:
 * bq_sa_bwd_fused_x.  bq_sa_bwd_reduce_supported: 1 when the shape has room for the reduction's table. */
BQ_API int bq_sa_bwd_reduce_supported(int ldx, int Nj, int S, int pool);
BQ_API int bq_sa_bwd_fused_xr(const void *x, const float *xscale, const float *xshift, const float *xmean, const float *xrstd,
                              float *red_part, float *red_dgb, const void *p, const void *dout, const void *arg, const void *w,
                              const float *scale, const float *shift, const float *mean, const float *rstd, const float *dgb,
                              void *dx, float *dw, float *part, long R, int ldx, int Nj, int ldw, int ldo, int S, int relu,
                              int pool, void *stream);
BQ_API int bq_sa_bwd_fused_x(const void *x, const float *xscale, const float *xshift, const void *p, const void *dout,
                             const void *arg, const void *w, const float *scale, const float *shift, const float *mean,
                             const float *rstd, const float *dgb, void *dx, float *dw, float *part, long R, int ldx, int Nj,
                             int ldw, int ldo, int S, int relu, int pool, void *stream);

/* exact (erf) GELU, bf16 (vit.py:23-41 Mlp act_layer=nn.GELU); n % 8 == 0, 16-B aligned */
BQ_API int bq_gelu_fwd_bf16(const void *x, void *y, long n, void *stream);
/* The same over TWO row groups with their own LayerNorm parameters -- the 2D and the 3D text stream of the twin encoder
 * (models/med.py:549-614) stacked in one (M, H) tensor: rows [0, M/2) use gamma / beta, rows [M/2, M) gamma2 / beta2; one
 * launch for both streams' BertSelfOutput / BertOutput tails.  zero_out / dgb: f32 (2, 2, H) = per group dgamma, dbeta. */
BQ_API int bq_twin_drop_add_ln_fwd(const void *x, const void *residual, const float *gamma, const float *beta,
                                   const float *gamma2, const float *beta2, void *y, float *mean, float *rstd,
                                   float *zero_out, int M, int H, float eps, float p_drop, unsigned seed,
                                   const unsigned *seed_ptr, void *stream);
BQ_API int bq_twin_drop_add_ln_bwd(const void *x, const void *residual, const float *gamma, const float *gamma2,
                                   const void *dy, const float *mean, const float *rstd, void *dx, void *dresidual,
                                   float *dgb, int M, int H, float eps, float p_drop, unsigned seed,
                                   const unsigned *seed_ptr, void *stream);

/* ---- multi-tensor AdamW that also writes the bf16 operand copies (csrc/adamw.hip) ----------------------------
 * Replaces torch.optim.AdamW(...).step() of the reference's training step (scripts/train.py:410-417) and the
 * per-weight fp32 -> bf16 casts of the next forward.  table: n records {p, g, m, v, shadow|NULL (device pointers),
 * n (int64 elements), lr, weight_decay (f32)} of bq_adamw_tensor_bytes() bytes each, in device memory; chunks:
 * n_chunks x {tensor index, chunk index} int32 pairs covering every tensor in pieces of bq_adamw_chunk_elems()
 * elements; step: device f32 = the 1-based count of THIS update (graph-replay safe).  amsgrad / maximize: off.
 * grad_clip_value > 0: every gradient element is clamped to [-c, c] as it is read -- the reference's
 * torch.nn.utils.clip_grad_value_(parameters, 1.0) before optimizer.step() (lib/solver.py:407-409) without a pass of
 * its own; <= 0: off. */
BQ_API int bq_adamw_chunk_elems(void);
BQ_API int bq_adamw_tensor_bytes(void);
BQ_API int bq_adamw_multi(const void *table, const void *chunks, int n_chunks, const float *step, float beta1,
                          float beta2, float eps, float grad_clip_value, void *stream);

/* ---- post-processing of the proposals (csrc/nms.hip) -------------------------------------------------------------
 * Replaces the host loops of lib/ap_helper.py:40-178 parse_predictions (the evaluation path).
 * bq_box_point_count: count[b][k] = number of points of scene b inside box k (remove_empty_box, ap_helper.py:88-100,
 *   whose in-hull test over the 8 corners of utils/box_util.py:282-300 get_3d_box is the closed oriented box):
 *   points f32 (B, N, ld) with xyz first, center / size f32 (B, K, 3) (size = l, w, h), heading f32 (B, K) radians
 *   (rotation `roty`); cap > 0 clamps the count (the caller only compares with 5).
 * bq_nms: greedy NMS of every scene (utils/nms.py:40-152 nms_2d_faster / nms_3d_faster / nms_3d_faster_samecls): box
 *   f32 (B, K, 6) = (x1, y1, z1, x2, y2, z2) (2-D: z1 = 0, z2 = 1), score f32 (B, K), cls i32 (B, K) or NULL, valid u8
 *   (B, K) or NULL (boxes left out), keep u8 (B, K) out; old_type: overlap = intersection / the other box's volume
 *   instead of IoU; same_cls: only boxes of the picked box's class are suppressed.  K <= 1024.  Double arithmetic. */
BQ_API int bq_box_point_count(const float *points, const float *center, const float *size, const float *heading, int *count,
                              int B, int N, int ld, int K, int cap, void *stream);
BQ_API int bq_nms(const float *box, const float *score, const int *cls, const unsigned char *valid, unsigned char *keep,
                  int B, int K, float thresh, int old_type, int same_cls, void *stream);

/* ---- multiview projection (csrc/projection.hip) ---------------------------------------------------------------------
 * Replaces the per-frame host loop of lib/projection.py:194-252 ProjectionHelper.compute_projection / :254-276 project
 * as driven by scripts/project_multiview_features.py:103-202 (offline preprocessing).
 * bq_project_points: pix[f][n] (i32, F x N) = y * W + x of the pixel of frame f that sees point n, or -1.  points f32
 *   (N, 3); depth f32 (F, H * W) metres; frames f32 (F, 40) = per frame: world-to-camera 4x4 row-major (16), the six
 *   inward frustum-plane normals (18), frustum corner 2 and corner 4 (6) -- computed by the host as the reference does.
 *   Tests in the reference's order: inside the six planes (round(100 d) / 100 < 0), pinhole projection rounded half to
 *   even, inside the W x H image, depth_min <= depth <= depth_max, |depth - z| <= accuracy.  fp32.
 * bq_fuse_point_features: out f32 (N, C) from pix and the frames' features feat f32 (F, H * W, C) (pixel-major), frames
 *   in order: maxpool = 0 -- an all-zero point takes the vector of the frame that sees it; maxpool = 1 -- a vector that is
 *   not all zero fills an all-zero point and is max-ed into a filled one.  C = 64, 128 or 256. */
BQ_API int bq_project_points(const float *points, const float *depth, const float *frames, int *pix, int F, int N, int W,
                             int H, float fx, float fy, float cx, float cy, float depth_min, float depth_max,
                             float accuracy, void *stream);
BQ_API int bq_fuse_point_features(const int *pix, const float *feat, float *out, int F, int N, int HW, int C, int maxpool,
                                  void *stream);

/* ---- MFMA bf16 GEMM family (csrc/gemm.hip) -----------------------------------------------------------------
 * Replaces every nn.Linear of the fusion half and its autograd: models/vit.py:30-32 (Mlp fc1 / fc2), :51-53 (qkv /
 * proj), timm PatchEmbed as a GEMM over 16x16x3 patches (vit.py:144-145), models/med.py:112-118 (query / key / value),
 * :232 (BertSelfOutput.dense), :295 (BertIntermediate.dense), :310 (BertOutput.dense).
 * One formulation:  out[j][i] = epilogue( sum_kc P(i, kc) * Q(j, kc) ),  bf16 operands, fp32 accumulation
 * (v_mfma_f32_16x16x32_bf16), out row-major over j with i contiguous (leading dimension ldo).
 *   P: K-contiguous  P[i*ldp + kc]  or, with BQ_GEMM_P_XC, contraction-major  P[kc*ldp + i]  (same for Q / ldq / j):
 *     forward y = x W^T + b : P = W, Q = x            (flags 0,                  i = out feature, j = row; with
 *                                                      BQ_GEMM_OUT_F32 (tiles 64 / 32, epilogue NONE / BIAS) y is fp32)
 *     dX = dY W            : P = W (P_XC), Q = dY      (BQ_GEMM_P_XC,             i = in feature,  j = row)
 *     dW = dY^T X          : P = X (P_XC), Q = dY (Q_XC), fp32 out (P_XC|Q_XC|OUT_F32, i = in feature, j = out feature)
 *   epilogue: BQ_GEMM_EPI_NONE; _BIAS: + bias[i] (fp32); _BIAS_GELU: out = bf16(acc + bias), out2 = gelu(out) (exact
 *     erf GELU, vit.py act_layer=nn.GELU / med hidden_act "gelu"); _DGELU: out = acc * gelu'(aux[j][i]) (the backward
 *     of the GELU fused into the dX GEMM of the layer after it).  colsum != NULL: colsum[i] += sum_j out[j][i] (fp32
 *     atomics; the bias gradient of the layer that produced the operand).  In the weight-gradient form
 *     (P_XC | Q_XC | OUT_F32; with ksplit > 1 through atomics onto a zeroed vector) colsum has Nj entries instead and RECEIVES colsum[j] = sum_kc Q(j, kc): the column
 *     sums of dY over the rows = the bias gradient of the same layer, from the same launch (plain stores).
 *   tile: 256 = 256x256 tiles, 8 waves, LDS-DMA pipeline (large M, long contractions: the weight gradients); 128 = 256 (i) x
 *     128 (j) tiles, 4 waves, two workgroups co-resident per CU (large M with SHORT contractions: the forward and
 *     input-gradient forms of the image encoder -- one workgroup's prologue / epilogue runs under the other's MFMAs; bf16
 *     out, K-contiguous Q, no colsum; also the weight-gradient form, colsum allowed); 64 / 32 = 64 x {64,32} tiles (small M; 32 needs a K-contiguous Q).  All problems
 *     of one call run in ONE launch (grouped GEMM) and must share flags / epilogue.
 * Requirements: 16-byte aligned operands, ldp / ldq / Ni / ldo multiples of 8 (ldo of 4 for fp32), Kc a multiple of
 * 64 for K-contiguous operands (any Kc for contraction-major ones), operands below 2 GB. */
#define BQ_GEMM_P_XC 1
#define BQ_GEMM_Q_XC 2
#define BQ_GEMM_OUT_F32 4
#define BQ_GEMM_BACKGROUND 8 /* tile 128 only: ONE persistent workgroup per CU instead of two (half the LDS, a quarter of the
                               wave slots) -- for a large GEMM issued on a side stream beside a chain of short latency-bound
                               kernels: a full persistent grid holds every CU's LDS until it ends and the chain's kernels
                               queue behind it (measured: a 7 us projection took 60 us beside the K/V projection of the image
                               tokens) */
#define BQ_GEMM_EPI_NONE 0
#define BQ_GEMM_EPI_BIAS 1
#define BQ_GEMM_EPI_BIAS_GELU 2
#define BQ_GEMM_EPI_DGELU 3
#define BQ_GEMM_EPI_ADD 5 /* out = acc + aux (aux bf16, laid out like out): in the input-gradient form the gradient that reached the
                            same tensor through another branch (the residual input of the following LayerNorm) is added here
                            instead of by a separate accumulation kernel */
#define BQ_GEMM_EPI_BIAS_CE 4 /* LM head: out = bf16(acc + bias) AND cross-entropy partials from the fp32 values; tile 256
                                 only; field reuse: out2 = f32 partials [2 * ceil(Ni/256)][Nj][3] (max, sum exp(z - max),
                                 sum z over the valid entries of each 128-wide half tile), aux = int32 targets [Nj] (< 0:
                                 ignored row), colsum = f32 [Nj] receives the target's logit, ksplit = number of valid
                                 entries along i (the vocabulary size; Ni may be padded beyond it) */
typedef struct bq_gemm_desc {
  const void *P, *Q;
  void *out;
  const float *bias;
  void *out2;
  const void *aux;
  float *colsum;
  int ldp, ldq, ldo;
  int Ni, Nj, Kc;
  int bias_bf16; /* 0: bias is fp32 (a master parameter), 1: bias is bf16 (a concatenated operand copy) */
  long p_bytes, q_bytes; /* 0: derived from the extents.  Otherwise the true size of the operand buffer: a K-contiguous
                            operand may have rows SHORTER than Kc (e.g. 136 elements of a 3+C = 135 channel point row
                            against Kc = 192) when the other operand is zero-padded to Kc -- the tail of a row then reads
                            the head of the next one (finite values times zeros), bounded by this size */
  int ksplit;            /* > 1 (fp32 out, tile 64): the contraction runs in ksplit pieces accumulated with fp32 atomics
                            into `out`, which the caller zero-fills (weight gradients over millions of rows) */
  /* Batched-row maps (ABI 2).  q_rpb > 0: Q's logical row r (a j row, or a CONTRACTION row when Q is contraction-major)
   * is read at element (r / q_rpb) * q_bstride + (r % q_rpb) * ldq -- a (batch, rows, cols) view whose batches are
   * q_bstride elements apart; o_rpb / o_bstride: the same for the rows of out (and of out2 / aux, laid out like out).
   * Replaces the torch.cat((image tokens, other stream's states), dim=1) in front of every cross-attention K/V
   * projection of the twin text encoder (models/med.py:549-562) and the strided slicing of its gradient: the
   * projection writes each source's rows into its row range of ONE (B, L1 + L2, 2 * D) key/value tensor, the input
   * gradient and the weight gradient read their row range of its gradient.  Maps need: tile 128, 64 or 32 (tile 256:
   * only on a contraction-major Q with q_rpb >= 64); rpb-mapped buffers below 2 GB; q_bstride % 8 == 0, o_bstride % 8. */
  int q_rpb, q_bstride, o_rpb, o_bstride;
  int accum;             /* 1 (fp32 out, tile 64 / 32 only): out += result and colsum += sums with fp32 atomics -- a second
                            row source of a weight gradient an earlier launch on the same stream has stored */
} bq_gemm_desc;
BQ_API int bq_gemm_max_problems(void); /* problems per launch; longer lists are split into several launches */
BQ_API int bq_gemm_bf16(const bq_gemm_desc *problems, int n, int flags, int epilogue, int tile, void *stream);
/* Stream-K form of the 256 x 128 tile kernel (ABI 4; csrc/gemm_mid.hip header): single-problem forward / K-contiguous dX
 * launches with a long contraction whose whole tiles would fill the grid unevenly (the ViT MLP's N = 768 / K = 3072
 * launches: 387 tiles of 48 K tiles on 512 workgroups) are cut into equal runs of K tiles; a cut tile is finished by the
 * workgroup that arrives last, through fp32 slabs in a caller-provided workspace.  bq_gemm_set_workspace registers the
 * workspace the launches of `stream` may use (device memory of >= bq_gemm_workspace_bytes(), its first 64 KB ZEROED once --
 * the kernel leaves the tickets zero); without one, or with ws = NULL, those launches run on whole tiles as before.  At most
 * one launch at a time may use a workspace: one workspace per stream, and a captured launch replays on its capture stream. */
BQ_API long bq_gemm_workspace_bytes(void);
BQ_API int bq_gemm_set_workspace(void *ws, long bytes, void *stream);
/* ABI 5.  Which stream-K forms may run on a stream that has a workspace: bit 0 the 256 x 128 kernel's (round 5's first attempt:
 * slower than whole tiles on every ViT shape), bit 1 the 256 x 256 kernel's (one problem, K-contiguous operands, bf16 out, a
 * contraction of >= 24 K tiles whose tiles would leave >= 15 % of the chip idle: fc2 forward, the input gradient through fc1).
 * Both measured slower than whole tiles on this repo's kernels: default 0.  mode < 0: query.  Returns the
 * previous mode. */
BQ_API int bq_gemm_streamk_mode(int mode);

/* Column sums of a list of bf16 matrices in ONE launch: out[n] += sum_m g[m*ld + n] (fp32 atomics: out must be zeroed
 * by the caller).  Replaces grad.sum(0), the bias gradient of nn.Linear, for every parked linear of a backward pass.
 * N and ld multiples of 8, g 16-byte aligned. */
typedef struct bq_colsum_desc {
  const void *g;
  float *out;
  int M, N, ld;
} bq_colsum_desc;
BQ_API int bq_colsum_grouped_bf16(const bq_colsum_desc *problems, int n, void *stream);

/* ---- multi-tensor bf16 transpose (csrc/transpose.hip) ----
 * dst_t (K, N) = src_t (N, K)^T for a list of weight operands in ONE launch: the K-contiguous second copy of the text
 * side's weights, which lets the input-gradient GEMMs of models/med.py's linears (autograd: grad_output.mm(weight)) run
 * on the forward's operand form.  table: DEVICE array of {const void *src; void *dst; int N, K, ld, tiles_k;} records
 * (bq_transpose_tensor_bytes() each; N, K multiples of 64, ld of 8, tiles_k = K / 64); chunks: DEVICE int32 pairs
 * {tensor, tile} covering every 64 x 64 tile, tile = (n / 64) * tiles_k + k / 64.  max_wgs > 0: a grid of at most that many
 * workgroups walking the tiles (a launch beside a latency-bound chain); <= 0: one workgroup per tile. */
BQ_API int bq_transpose_tensor_bytes(void);
BQ_API int bq_transpose_multi_bf16(const void *table, const void *chunks, int n_chunks, int max_wgs, void *stream);

/* ---- weight gradient of a SharedMLP layer over whole rows (csrc/gemm.hip wgrad_rows_kernel + reduce) ----
 * Replaces the conv weight gradient of autograd for the 1x1 convolutions of lib/pointnet2/pytorch_utils.py:104-157 on
 * point-major rows:  out[j][i] = sum_r Q[r][j] P[r][i]  (P = the layer's input rows (R, ldp) bf16, Q = the gradient
 * w.r.t. the convolution output (R, ldq) bf16, out (Nj, ldo) fp32, columns [Ni, ldo) set to 0).  Every operand row is
 * staged once for all its output tiles and there are no atomics (bq_gemm_bf16's cut weight-gradient form ends every
 * workgroup in 4096 scattered fp32 atomics per 64 x 64 tile, which is what it spends its time on): each of
 * the W = bq_wgrad_rows_workgroups(R, Ni, Nj, workgroups) workgroups stores its share into part (W x Nj x ldo floats of
 * scratch) and a second kernel sums the slices in a fixed order.  Supported: ceil(Ni / 64) in 1..5, ceil(Nj / 64) in
 * {1, 2, 4}, their sum <= 7, not 5 + 1 (bq_wgrad_rows_supported); Ni % 4 == 0, ldp % 8 == 0, ldq % 8 == 0, ldo % 4 == 0,
 * operands below 2 GB.  workgroups <= 0: 256, or 512 when two workgroups fit a CU's LDS (at most 3 units). */
BQ_API int bq_wgrad_rows_supported(int Ni, int Nj);
BQ_API int bq_wgrad_rows_workgroups(long R, int Ni, int Nj, int workgroups);
BQ_API int bq_wgrad_rows_bf16(const void *P, const void *Q, float *out, float *part, long R, int Ni, int Nj, int ldp,
                              int ldq, int ldo, int workgroups, void *stream);

/* ---- SharedMLP layer: 1x1 convolution on point-major rows + BatchNorm statistics in its epilogue (csrc/gemm.hip) ----
 * Replaces conv (1x1, bias=False) -> the statistics pass of BatchNorm2d(train) of one SharedMLP layer
 * (lib/pointnet2/pytorch_utils.py:104-157, :11-36):  y[r][n] = sum_k x[r][k] w[n][k]  for bf16 x (R rows of ldx >= K
 * elements: (b, npoint, nsample) x (3+C) point-major), bf16 w (N rows of ldw >= Kc elements, ZERO beyond K, Kc % 64
 * == 0), y bf16 (R, N), N % 64 == 0; and from the fp32 accumulators (before the rounding of y) the training-mode
 * statistics: scale = gamma * rstd, shift = beta - mean * scale, mean, rstd (f32 N each), running_mean / running_var
 * (momentum, unbiased) and num_batches_tracked updated when non-NULL -- what bq_bn_stats computes from a second pass
 * over y.  bq_bn_apply / bq_bn_backward consume scale / shift / mean / rstd unchanged.
 * center (f32 N, may be NULL; may alias running_mean): the stored y is  conv - center[n]  and shift / mean are those of the
 * STORED values (the normalised output is the same function: training-mode BatchNorm does not see a per-channel offset
 * of its input; running_mean still receives the mean of the convolution itself).  With center = running_mean the bf16
 * rounding of y applies to the deviation from the channel mean instead of to the value: where |mean| >> std the
 * normalised activations keep 8 bits of their OWN scale (the pre-BatchNorm rounding finding).
 * partial: bq_pwconv_records(R, N) * 3 * N floats of scratch. */
BQ_API int bq_pwconv_records(long R, int N);
BQ_API int bq_pwconv_bn_fwd(const void *x, long R, int K, int ldx, const void *w, int ldw, int Kc, int N, void *y,
                            float *partial, const float *gamma, const float *beta, float *running_mean,
                            float *running_var, long long *num_batches_tracked, float eps, float momentum,
                            float *scale, float *shift, float *mean, float *rstd, const float *center, float *shift_acc,
                            void *stream);
/* The second pass of such a layer WITHOUT a bf16 pre-activation in between (ABI 4): the product x W^T once more, BatchNorm's
 * affine map (scale and shift_acc = beta - mean scale as bq_pwconv_bn_fwd wrote them -- shift_acc: optional f32 [N] output of
 * that call, ABI 4, the shift for the UNcentred product), ReLU and -- pool != 0 -- the maximum over every run
 * of S (16, 32 or 64) rows applied to the fp32 accumulators; out bf16 (R, N) or (R / S, N).  Replaces bq_bn_apply on the
 * stored y: same traffic class (reads x instead of y), and the layer's output carries one bf16 rounding instead of two
 * (reference: fp32 throughout, lib/pointnet2/pytorch_utils.py:104-157). */
BQ_API int bq_pwconv_bn_apply(const void *x, long R, int K, int ldx, const void *w, int ldw, int Kc, int N, const float *scale,
                              const float *shift_acc, void *out, int S, int relu, int pool, void *stream);

/* ---- LM head + label-smoothed cross entropy (csrc/lmhead.hip, with BQ_GEMM_EPI_BIAS_CE of bq_gemm_bf16) -------------
 * Replaces prediction_scores = cls(sequence_output) -> .float() -> CrossEntropyLoss(reduction='none',
 * label_smoothing=0.1) of models/med.py:1417-1432 (no fp32 logits tensor; the bf16 logits are stored once).
 * bq_lmhead_ce_combine: partial f32 [nrec][R][3] from the GEMM epilogue, target_logit f32 [R], target int32 [R] (< 0 =
 *   ignore_index) -> loss f32 [R] = (1-eps)(lse - z_t) + eps (lse - mean_v z_v) (0 for ignored rows), lse f32 [R].
 * bq_lmhead_ce_dlogits: logits bf16 (R rows of ld >= V elements) -> IN PLACE grad_loss[r] * (softmax - (1-eps) onehot -
 *   eps/V), zero in ignored rows and in the padding columns v >= V: the operand of the dW / dH / db launches. */
BQ_API int bq_lmhead_ce_combine(const float *partial, const float *target_logit, const int *target, float *loss,
                                float *lse, int R, int nrec, int V, float label_smoothing, void *stream);
BQ_API int bq_lmhead_ce_dlogits(void *logits, const float *lse, const int *target, const float *grad_loss, int R, int V,
                                int ld, float label_smoothing, void *stream);

/* ---- deterministic scatter gradients through an inverted index (csrc/invert.hip, ABI 4) ---------------------------------------
 * Replaces the fp32-atomic scatters of group_points_grad / three_interpolate_grad (lib/pointnet2/_ext_src/src/
 * group_points_gpu.cu:43-75, interpolate_gpu.cu:116-154) by gathers: one sum per destination, terms in ascending position
 * order -- same terms, a fixed order, bitwise reproducible.
 * bq_invert_index: idx int32 (B, L), values in [0, N) -> start int32 [B * N + 1] (CSR over scene * N + value), slots uint32
 *   [B * L]: the positions b * L + l naming each value, ascending (stable radix sort); workspace of
 *   bq_invert_index_workspace_bytes(B * L) bytes.
 * bq_group_concat_pm_grad_gather: the FEATURE gradient of bq_group_concat_pm from the inverted group index (idx (B, M, S) ->
 *   L = M * S): grad_out rows of ld elements (bf16 / fp32), grad_feats f32 (B, N, C) written whole (no zero-fill); optional
 *   grad_xyz f32 (B, N, 3) (the offset channels, / radius under normalize) and grad_new_xyz f32 (B, M, 3) (minus each centre's
 *   own S rows, in order) -- any of the three may be NULL.
 * bq_three_interpolate_grad_gather: idx (B, n, 3) inverted over m known points (L = 3 n); grad_out f32 (B, C, n), weight f32
 *   (B, n, 3) -> grad_points f32 (B, C, m). */
BQ_API size_t bq_invert_index_workspace_bytes(long total);
BQ_API int bq_invert_index(const int32_t *idx, int B, long L, int N, int32_t *start, unsigned *slots, void *workspace,
                           size_t workspace_bytes, void *stream);
BQ_API int bq_group_concat_pm_grad_gather(const void *grad_out, int in_bf16, const int32_t *start, const unsigned *slots,
                                          float *grad_feats, float *grad_xyz, float *grad_new_xyz, int B, int C, int N, int M,
                                          int S, int ld, float radius, int normalize, void *stream);
BQ_API int bq_three_interpolate_grad_gather(const float *grad_out, const int32_t *start, const unsigned *slots,
                                            const float *weight, float *grad_points, int B, int C, int n, int m, void *stream);

/* ---- detection loss and its gradient (csrc/detloss.hip) -------------------------------------------------------------------
 * Replaces compute_vote_loss + compute_objectness_loss + compute_box_and_sem_cls_loss of lib/loss_helper.py:25-193 (over
 * utils/nn_distance.py:6-52) and what autograd derives from them: ~460 torch launches on B x 256 proposals -> two.
 * All float tensors fp32 and contiguous, class labels / masks int64 as the reference's dataset produces them
 * (lib/dataset.py:553-577); seed_inds int32 or int64 (seed_inds_i64).  terms f32 [16]: vote_loss, objectness_loss,
 * center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss, sem_cls_loss, pos_ratio, neg_ratio (the
 * data_dict entries of loss_helper.py:400-430, before the caller's weights and the x10).  g_*: d term / d input, the term named
 * by bq_det_loss_bwd's order, shaped like the input; scratch: int32 [B * G + 1024]. */
typedef struct bq_det_loss_desc {
  const float *seed_xyz, *vote_xyz, *aggregated_vote_xyz, *objectness_scores, *center, *heading_scores,
      *heading_residuals_normalized, *size_scores, *size_residuals_normalized, *sem_cls_scores;
  const void *seed_inds;
  const float *vote_label;
  const void *vote_label_mask; /* int64 (B, N) */
  const float *center_label, *box_label_mask;
  const void *heading_class_label, *size_class_label, *sem_cls_label; /* int64 (B, G) */
  const float *heading_residual_label, *size_residual_label, *mean_size_arr;
  float *terms;
  void *objectness_label; /* int64 (B, K) out */
  float *objectness_mask;
  void *object_assignment; /* int64 (B, K) out */
  float *g_vote_xyz, *g_objectness_scores, *g_center, *g_heading_scores, *g_heading_residuals_normalized, *g_size_scores,
      *g_size_residuals_normalized, *g_sem_cls_scores;
  int *scratch;
  int B, S, VF, N, K, G, NH, NS, NC; /* seeds, votes per seed, points, proposals, GT slots, heading bins, size clusters, classes */
  int cl_ld;                         /* floats per center_label row (>= 3) */
  int seed_inds_i64;
  /* floats per proposal row of the six score tensors and of their g_* buffers: the natural widths (2, NH, NH, NS, 3 NS, NC)
   * for separate tensors, the channel count when they are slices of one (B, K, channels) head output
   * (models/proposal_module.py:19-48 decode_scores) -- the g_* pointers then address one buffer of that shape */
  int ld_objectness_scores, ld_heading_scores, ld_heading_residuals_normalized, ld_size_scores, ld_size_residuals_normalized,
      ld_sem_cls_scores;
  float near_threshold, far_threshold, objectness_weight_neg, objectness_weight_pos; /* 0.3, 0.6, 0.2, 0.8 (loss_helper.py:18-22) */
} bq_det_loss_desc;
BQ_API int bq_det_loss_fwd(const bq_det_loss_desc *d, void *stream);
/* backward: segment i: out[r * ld + c] = g[r * ld + c] * upstream[term] for r < rows, c < width (term < 0: out = 0 -- channels of
 * a packed buffer that no term reads); terms 0..7 = vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg,
 * sem_cls; at most 16 segments, one launch */
typedef struct bq_det_loss_seg {
  const float *g;
  float *out;
  int rows, width, ld, term;
} bq_det_loss_seg;
BQ_API int bq_det_loss_bwd(const bq_det_loss_seg *segments, int n, const float *upstream, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BQHIP_FUSION_H */
