/* tspo_hip.h - C ABI of libtspo_hip.so: the MI355X (gfx950) temporal-sampling
 * hot path of TSPO (CLIP-L frame encode -> temporal scoring head -> top-k /
 * bin-max / Gumbel-top-k frame sampler -> group-relative advantage +
 * policy-gradient reduction -> selector backward -> AdamW).
 *
 * The reference (Hui-design/TSPO) is 100 % Python and has no FFI; the path
 * sits behind a Python module API (model/temporal_agent.py, model/utils.py,
 * llava/model/language_model/llava_qwen.py:131-176,
 * src/open_tspo/trainer/tspo_trainer.py:387-404,587-609).  Each entry point
 * below names the reference lines whose arithmetic it replaces; the Python
 * shim in tspo_amd/ keeps the reference's names and signatures and calls
 * these entries through ctypes.
 *
 * Conventions (all entry points)
 *  - return 0 on success, a negative TSPO_E* code on error; the message is
 *    available from tspo_last_error() (thread-local).  No C++ exception
 *    crosses the ABI.
 *  - every pointer is a DEVICE pointer unless the parameter is documented as
 *    "host"; tensors are contiguous row-major; pointers 16-byte aligned.
 *  - the library never allocates or frees device memory (the caller supplies
 *    the workspace whose size tspo_*_workspace_bytes reports), never
 *    synchronises the device or the stream, and launches only on `stream`
 *    (a hipStream_t passed as void*; NULL = the legacy default stream).
 *  - no global mutable state: re-entrant across threads / ranks.
 *  - index outputs are int64, ascending, as the reference returns them.
 */
#ifndef TSPO_HIP_H
#define TSPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSPO_ABI_VERSION 4

enum tspo_error {
  TSPO_OK = 0,
  TSPO_EINVAL = -1,      /* bad dimension / null pointer / unsupported shape */
  TSPO_EWORKSPACE = -2,  /* workspace too small or misaligned */
  TSPO_ELAUNCH = -3      /* HIP launch error (hipGetLastError) */
};

enum tspo_dtype { TSPO_F32 = 0, TSPO_BF16 = 1, TSPO_F16 = 2, TSPO_U8 = 3 };

typedef void* tspo_stream_t; /* hipStream_t */

int tspo_version(void);
const char* tspo_last_error(void);

/* ------------------------------------------------------------------------
 * Frame samplers                                  (bit-exact integer outputs)
 * ------------------------------------------------------------------------ */

/* Greedy top-k, indices ascending.  Replaces
 *   sort(topk(confidence, min(T,k)).indices)       model/temporal_agent.py:190-192
 *                                                   llava_qwen.py:154-157
 * scores f32 [B,T] -> idx i64 [B, min(T,k)].  Ties -> lowest index; NaN sorts
 * as the largest value (torch.topk convention).  One workgroup per row: 4-pass
 * 8-bit radix select on order-preserving keys, then an index-ordered
 * compaction (so no sort is needed).                                       */
int tspo_topk_sorted(const float* scores, int B, int T, int k, int64_t* idx, tspo_stream_t stream);

/* 'bin-max' selection.  Replaces model/temporal_agent.py:194-210 with
 * model/utils.py:10-16 (anchors round-half-even(i*(T-1)/(k-1)) in double,
 * nearest-anchor bins with first-min ties, first arg-max per bin).
 * scores f32 [B,T] -> idx i64 [B, min(T,k)].                               */
int tspo_binmax(const float* scores, int B, int T, int k, int64_t* idx, tspo_stream_t stream);

/* Gumbel-top-k (Plackett-Luce without replacement) for G rollouts of each of
 * B prompts in one launch.  Replaces gumbel_softmax, model/utils.py:69-80,
 * called G times per step from llava_qwen.py:137 (tspo_trainer.py:514).
 *   z = (logits + g)/tau ; idx = sort(topk(z, k)) ; logp = log(softmax(logits))
 *   probs = (one_hot(idx) - softmax(z)) + softmax(z)      (straight-through)
 * logits f32 [B,T]; noise f32 [B,G,T] of Gumbel(0,1) draws, or NULL to draw
 * in-kernel from Philox4x32-10 keyed by (seed, offset) with counter
 * (t, g, b) - see oracle/tspo_oracle.py:gumbel_noise_philox.
 * Outputs: idx i64 [B,G,k]; logp f32 [B,T] (nullable); probs f32 [B,G,T]
 * (nullable); noise_out f32 [B,G,T] (nullable: the noise actually used).
 * k > T -> TSPO_EINVAL (torch.topk raises, model/utils.py:73).             */
int tspo_gumbel_topk(const float* logits, const float* noise, uint64_t seed, uint64_t offset,
                     int B, int G, int T, int k, float tau,
                     int64_t* idx, float* logp, float* probs, float* noise_out,
                     tspo_stream_t stream);

/* ABI v4.  The same with the prompts of SEVERAL micro-steps in one call (the reference trains with
 * per_device_train_batch_size 1 x gradient_accumulation_steps 2, train_deepspeed.sh:30-31; the two micro-steps see the same
 * weights - no update between them, tspo_trainer.py:500-552 - so their policy math is one batch): prompts come in groups
 * of `prompts_per_offset` (> 0, divides B); prompt b draws the in-kernel noise that prompt b % p of a separate call with
 * offset + b / p would draw, so the coalesced rollouts are bit for bit the sequential ones.  0 = tspo_gumbel_topk.     */
int tspo_gumbel_topk_ex(const float* logits, const float* noise, uint64_t seed, uint64_t offset,
                        int B, int G, int T, int k, float tau,
                        int64_t* idx, float* logp, float* probs, float* noise_out,
                        tspo_stream_t stream, int prompts_per_offset);

/* ------------------------------------------------------------------------
 * Group-relative advantage + policy-gradient reduction
 * ------------------------------------------------------------------------ */

/* A = (r - mean_G r) / (std_G r + eps), unbiased std over the G rollouts of a
 * prompt.  Replaces src/open_tspo/trainer/tspo_trainer.py:587-592.
 * rewards f32 [B,G] -> adv f32 [B,G].                                       */
int tspo_grpo_advantage(const float* rewards, int B, int G, float eps, float* adv, tspo_stream_t stream);

/* Closed-form gradient of the TSPO policy loss w.r.t. the logits
 * (tspo_trainer.py:544,594-609):
 *   L_b = -(1/G) sum_g A_bg * mean_{j in S_bg} exp(lp_j - sg(lp_j))  (= -mean_g A_bg)
 *   dL_b/ds_t = -(1/G) sum_g A_bg (1[t in S_bg]/k - exp(logp_bt))
 * logp f32 [B,T], idx i64 [B,G,k] (ascending, from tspo_gumbel_topk),
 * adv f32 [B,G] -> dlogits f32 [B,T] (multiplied by `scale`, e.g. 1/B),
 * loss f32 [B] (nullable, unscaled).                                        */
int tspo_pg_grad_logits(const float* logp, const int64_t* idx, const float* adv,
                        int B, int G, int T, int k, float scale,
                        float* dlogits, float* loss, tspo_stream_t stream);

/* The two calls above in one launch (rewards -> advantages -> dlogits); `adv`
 * [B,G] is still written (the trainer logs it).  Same results bit for bit.  */
int tspo_grpo_pg_grad(const float* rewards, const float* logp, const int64_t* idx,
                      int B, int G, int T, int k, float eps, float scale,
                      float* adv, float* dlogits, float* loss, tspo_stream_t stream);

/* ------------------------------------------------------------------------
 * Temporal scoring head ("selector", MultiModal_Align)
 * ------------------------------------------------------------------------ */

typedef struct tspo_selector_weights {
  const float* wqkv; /* [3D, D]  rows: Self_q.weight | Self_k.weight | Self_v.weight */
  const float* bqkv; /* [3D] */
  const float* w1;   /* mlp.0.weight [D,D] */
  const float* b1;   /* mlp.0.bias   [D]   */
  const float* w2;   /* mlp.2.weight [D,D] */
  const float* b2;   /* mlp.2.bias   [D]   */
} tspo_selector_weights;

typedef struct tspo_selector_grads { /* same shapes as the weights; overwritten */
  float* wqkv; float* bqkv; float* w1; float* b1; float* w2; float* b2;
} tspo_selector_grads;

/* Bytes of workspace tspo_selector_forward/backward need for these dims (the
 * forward leaves its saved activations there; backward must get the same
 * buffer, untouched).                                                       */
size_t tspo_selector_workspace_bytes(int B, int T, int D, int H, int M, int window);

/* MultiModal_Align.forward, batched over B videos.  Replaces
 * model/temporal_agent.py:116-143 (+ positional_encoding :10-19,
 * create_window_mask :97-104, Simple_SelfAttn :38-79, pair_cosine :106-114):
 *   x_t = x + pe(t/T); q,k,v = Linear(x_t); banded softmax over the key set
 *   {clamp(t - w/2 + i, 0, T-1), i<w} per head; h = mlp(attn) + x;
 *   s = (mean_m cos_eps(h, e_m) + clip) / tau.
 * img f32 [B,T,D], txt f32 [B,M,D], clip f32 [B,T] (NULL = zeros) ->
 * scores f32 [B,T], temporal_attn f32 [B,T,D] (nullable).  D % 64 == 0,
 * D/H <= 128, window >= 1.  The T x T mask / score tensors of the reference
 * are never formed.                                                         */
int tspo_selector_forward(const tspo_selector_weights* w, const float* img, const float* txt, const float* clip,
                          int B, int T, int D, int H, int M, int window, float tau,
                          float* scores, float* temporal_attn,
                          void* workspace, size_t workspace_bytes, tspo_stream_t stream);

/* Backward of the above for dL/dscores f32 [B,T] (autograd in the reference:
 * loss.backward() through tspo_trainer.py:542-609).  Writes the 6 gradient
 * buffers (summed over B and T).  Inputs are frozen features, so no input
 * gradient is produced.                                                     */
int tspo_selector_backward(const tspo_selector_weights* w, const float* img, const float* txt,
                           const float* dscores, int B, int T, int D, int H, int M, int window, float tau,
                           const tspo_selector_grads* grads,
                           void* workspace, size_t workspace_bytes, tspo_stream_t stream);

/* Same two calls with an option word.  TSPO_SEL_BF16X3: the projections / MLP /
 * weight-gradient GEMMs run in split precision on the bf16 MFMA (every fp32
 * operand x = hi + lo, products hi*hi + hi*lo + lo*hi, fp32 accumulate:
 * ~1e-5 relative error instead of fp32's ~1e-6) - an opt-in for the TRAINING
 * step, where the reference itself runs in bf16 (`--bf16`,
 * train_deepspeed.sh:33, scripts/zero3.json:10); flags = 0 is identical to the calls
 * above (exact fp32, the mode the greedy-index parity tests use).
 * TSPO_SEL_ACCUMULATE (backward calls only): the gradients are ADDED to what the
 * gradient buffers hold instead of overwriting them - the second.. micro-step of
 * a gradient-accumulation window (gradient_accumulation_steps 2,
 * train_deepspeed.sh:31) without a scratch bucket and an add pass.          */
#define TSPO_SEL_BF16X3 1
#define TSPO_SEL_ACCUMULATE 2
/* TSPO_SEL_BF16 (tspo_selector_forward_ex only; round 6): the reference's own INFERENCE precision as a first-class path - it loads the
 * scoring head in bf16 (mp_tools/vlmeval/vlm/gen_id_tspo.py:55, tspo_trainer.py:201): the three projections and the two MLP
 * layers take their operands rounded to bf16 (round-to-nearest-even, in registers) with fp32 accumulation on the bf16 MFMA,
 * ONE matrix instruction where the exact mode issues eight; everything between the GEMMs (position add, banded softmax,
 * ReLU, residual, cosine) stays fp32 - fewer roundings than the reference's bf16 module, whose every intermediate is bf16.
 * Not for the greedy-index parity claims (those use flags = 0) and not accepted by the backward calls. */
#define TSPO_SEL_BF16 4
int tspo_selector_forward_ex(const tspo_selector_weights* w, const float* img, const float* txt, const float* clip,
                             int B, int T, int D, int H, int M, int window, float tau,
                             float* scores, float* temporal_attn,
                             void* workspace, size_t workspace_bytes, tspo_stream_t stream, int flags);
int tspo_selector_backward_ex(const tspo_selector_weights* w, const float* img, const float* txt,
                              const float* dscores, int B, int T, int D, int H, int M, int window, float tau,
                              const tspo_selector_grads* grads,
                              void* workspace, size_t workspace_bytes, tspo_stream_t stream, int flags);

/* tspo_grpo_pg_grad + tspo_selector_backward_ex in one call with one launch
 * less: the kernel that turns dscores into dL/dh derives dscores itself from
 * (rewards [B,G], logp [B,T], idx [B,G,k] ascending) - same arithmetic and
 * order as tspo_grpo_pg_grad: adv / loss are bit-identical to the two-call
 * form, gradients agree to rounding (< 1e-6 of their maximum).  G <= 64.  `scale` multiplies dL/dscores (e.g. 1/(B*accum)).
 * flags: TSPO_SEL_BF16X3 and / or TSPO_SEL_ACCUMULATE (add to the gradient buffers: micro-steps 2.. of an accumulation window).
 * Replaces loss.backward() of tspo_trainer.py:587-609 end to end.           */
int tspo_policy_backward(const tspo_selector_weights* w, const float* img, const float* txt,
                         const float* rewards, const float* logp, const int64_t* idx,
                         int B, int T, int D, int H, int M, int window, float tau,
                         int G, int k, float adv_eps, float scale,
                         const tspo_selector_grads* grads, float* adv, float* loss,
                         void* workspace, size_t workspace_bytes, tspo_stream_t stream, int flags);

/* tspo_policy_backward that also leaves the sum of squares of the gradient it
 * wrote (all six trainable tensors) as *n_partials <= 2048 block partials in
 * norm_partials (f32, >= 2048 floats), for tspo_adamw_clip_step_ex: the
 * clip_grad_norm_ of the HF Trainer (tspo_trainer.py via Trainer.training_step)
 * then costs no pass of its own.  Only valid while the gradient bucket is not
 * modified between the two calls (single rank, no gradient accumulation);
 * otherwise use tspo_adamw_clip_step.  norm_partials == NULL: plain
 * tspo_policy_backward.                                                     */
int tspo_policy_backward_ex(const tspo_selector_weights* w, const float* img, const float* txt,
                            const float* rewards, const float* logp, const int64_t* idx,
                            int B, int T, int D, int H, int M, int window, float tau,
                            int G, int k, float adv_eps, float scale,
                            const tspo_selector_grads* grads, float* adv, float* loss,
                            void* workspace, size_t workspace_bytes, tspo_stream_t stream, int flags,
                            float* norm_partials, int* n_partials);

/* ------------------------------------------------------------------------
 * Optimiser on the flat parameter bucket
 * ------------------------------------------------------------------------ */

/* sum of squares of a flat f32 buffer -> out[0] = ||g||_2, out[1] =
 * min(1, max_norm/(norm+1e-6)) * pre_scale  (torch clip_grad_norm_ coefficient;
 * HF Trainer max_grad_norm=1.0).  workspace >= 2048 bytes (512 floats).     */
int tspo_grad_norm_scale(const float* grad, size_t n, float pre_scale, float max_norm, float* out2,
                         void* workspace, size_t workspace_bytes, tspo_stream_t stream);

/* torch.optim.AdamW step (HF Trainer default optimiser for the selector
 * params, train_deepspeed.sh: lr 5e-4, wd 0) on flat f32 buffers.
 * g_eff = grad * grad_scale * (d_grad_scale ? d_grad_scale[1] : 1).         */
int tspo_adamw_step(float* param, const float* grad, float* m, float* v, size_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float grad_scale, const float* d_grad_scale, tspo_stream_t stream);

/* tspo_grad_norm_scale + tspo_adamw_step in two launches instead of three
 * (HF Trainer: clip_grad_norm_(max_grad_norm) then optimizer.step()): the
 * AdamW kernel finishes the norm reduction itself and applies
 * coefficient * pre_scale to the gradient on the fly (grad is not modified).
 * out2 f32 [2] = (||g||_2, applied scale); workspace >= 2048 bytes (512
 * floats: one partial sum of squares per norm block); buffers 16-byte aligned. */
int tspo_adamw_clip_step(float* param, const float* grad, float* m, float* v, size_t n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float pre_scale, float max_norm, float* out2,
                         void* workspace, size_t workspace_bytes, tspo_stream_t stream);

/* tspo_adamw_clip_step with the partial sums of squares of `grad` supplied by
 * the kernel that produced it (tspo_policy_backward_ex): ONE launch.        */
int tspo_adamw_clip_step_ex(float* param, const float* grad, float* m, float* v, size_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                            float pre_scale, float max_norm, float* out2,
                            const float* norm_partials, int n_partials, tspo_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP ViT vision tower + projection  (frame encoder)
 * ------------------------------------------------------------------------ */

typedef struct tspo_clip_config {
  int hidden;   /* C: 1024 for ViT-L/14; % 64 == 0                     */
  int layers;   /* 24                                                   */
  int heads;    /* 16; hidden/heads must be 64                          */
  int mlp;      /* 4096; % 64 == 0                                      */
  int patch;    /* 14                                                   */
  int image;    /* 224 -> S = (image/patch)^2 + 1 = 257 tokens (<= 288) */
  int proj;     /* 768                                                  */
  float ln_eps; /* 1e-5                                                 */
} tspo_clip_config;

typedef struct tspo_clip_layer { /* bf16 matrices [out,in] row-major; f32 vectors */
  const float* ln1_g; const float* ln1_b;
  const void* wqkv;  const float* bqkv;   /* [3C, C]: q_proj | k_proj | v_proj */
  const void* wo;    const float* bo;     /* [C, C]   */
  const float* ln2_g; const float* ln2_b;
  const void* w1;    const float* b1;     /* [mlp, C] */
  const void* w2;    const float* b2;     /* [C, mlp] */
} tspo_clip_layer;

typedef struct tspo_clip_weights {
  tspo_clip_config cfg;
  const void* patch_w;   /* bf16 [C, Kp]: conv weight flattened (c,ky,kx), zero-padded to Kp = roundup(3*p*p, 64) */
  const float* pos_emb;  /* f32 [S, C]: position_embedding, with class_embedding added into row 0 */
  const float* pre_g; const float* pre_b;
  const float* post_g; const float* post_b;
  const void* proj_w;    /* bf16 [proj, C] */
  const tspo_clip_layer* layers; /* HOST array of cfg.layers entries */
} tspo_clip_weights;

size_t tspo_clip_workspace_bytes(const tspo_clip_config* cfg, int n_frames);

/* CLIPModel.get_image_features over n_frames already-resized frames
 * (reference call sites model/temporal_agent.py:166, tspo_trainer.py:401,
 * model/utils.py:32; arithmetic in transformers/models/clip/modeling_clip.py).
 * pixels [N,3,image,image] of pixel_dtype: TSPO_F32/TSPO_BF16/TSPO_F16 =
 * already CLIP-normalised; TSPO_U8 = raw 0..255, (x/255-mean)/std fused into
 * the patch gather.  bf16 MFMA GEMMs with fp32 accumulation, bf16 activations.
 * feat f32 [N, proj].  For batches of >= 64 frames the per-layer LayerNorms
 * are folded into the GEMMs around them (same maths, statistics taken from
 * the bf16 residual stream as stored).                                       */
int tspo_clip_vit_forward(const tspo_clip_weights* w, const void* pixels, int pixel_dtype, int n_frames,
                          float* feat, void* workspace, size_t workspace_bytes, tspo_stream_t stream);

/* tspo_clip_vit_forward with option flags (0 = exactly tspo_clip_vit_forward):
 *   TSPO_CLIP_NO_LN_FOLD  keep the stand-alone LayerNorm passes on large batches too (A/B comparison);
 *   TSPO_CLIP_PRUNE_LAST  evaluate the LAST transformer block for the class-token row only - get_image_features
 *                         pools token 0, the other 256 token rows of that block have no consumer; same features up
 *                         to the different GEMM kernel of the small [n_frames, C] matrices (opt-in, off by default:
 *                         the default executes the full model like the reference);
 *   TSPO_CLIP_FOLD_CACHED the LayerNorm-folded weights of all layers (kept at the front of the workspace, at offsets that do
 *                         not depend on n_frames) were written by an EARLIER call on this workspace with these weights, on this
 *                         stream or one ordered before it: the 2 x layers fold launches are skipped.  The library keeps no state -
 *                         the caller vouches; a call without the flag (re)writes them.  Ignored where nothing is folded.  */
#define TSPO_CLIP_NO_LN_FOLD 1
#define TSPO_CLIP_PRUNE_LAST 2
#define TSPO_CLIP_FOLD_CACHED 4
int tspo_clip_vit_forward_ex(const tspo_clip_weights* w, const void* pixels, int pixel_dtype, int n_frames,
                             float* feat, void* workspace, size_t workspace_bytes, tspo_stream_t stream, int flags);

/* Same computation, with a hipEvent recorded on `stream` after every kernel launch; synchronises on the
 * last event (profiling entry point - the only one that waits) and returns in HOST array host_ms6:
 * [0] GEMM ms, [1] attention ms, [2] LayerNorm ms, [3] patch gather ms, [4] total ms, [5] number of GEMM
 * launches.  bench.py derives roofline.achieved for the GEMM kernel from [0].  flags as for _ex.              */
int tspo_clip_vit_profile(const tspo_clip_weights* w, const void* pixels, int pixel_dtype, int n_frames,
                          float* feat, void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                          float* host_ms6, int flags);

/* On-device CLIPImageProcessor front half: uint8 frames [T,H,W,3] (layout 0) or [T,3,H,W] (layout 1) ->
 * Pillow-exact antialiased bicubic resize + centre crop -> uint8 [T,3,out_h,out_w].  Replaces the per-frame
 * np.array -> PIL.Image -> clip_processor loop of model/temporal_agent.py:156-164 (and tspo_trainer.py:393-399).
 * hcoef/vcoef: int32 [out][k] fixed-point (22 fractional bits) filter taps; hbound/vbound: int32 [out][2] =
 * (first input index, tap count), for the output columns / rows of the crop window only (built on the host by
 * tspo_amd/preprocess.py exactly as Pillow's precompute_coeffs + normalize_coeffs_8bpc).  [ylo, ylo+nrows) is the
 * range of input rows the vertical taps touch.  Bit-exact with PIL.                                        */
size_t tspo_preprocess_workspace_bytes(int T, int nrows, int out_w);
int tspo_preprocess_frames(const uint8_t* frames, int layout, int T, int H, int W,
                           const int32_t* hcoef, const int32_t* hbound, int out_w, int hk,
                           const int32_t* vcoef, const int32_t* vbound, int out_h, int vk,
                           int ylo, int nrows, uint8_t* out,
                           void* workspace, size_t workspace_bytes, tspo_stream_t stream);

/* Same, with the horizontal pass on the matrix pipe: the taps of every 16-column
 * block as signed-byte digit matrices (k = d0 + 2^8 d1 + 2^16 d2) laid out
 * [block][K block of 64 input columns][digit][lane 0..63][16 bytes] (int8),
 * mfma_bias i32 [out_w] = 128 * sum(taps) + 2^21, mfma_xs i32 [out_w/16] = first
 * input column of each block, mfma_nkb (1..4) K blocks per block, mfma_span =
 * widest input span (pixels) of four consecutive blocks.  Bit-identical output;
 * mfma_taps == NULL or an unsupported geometry (out_w % 16, span too wide for
 * LDS) takes the scalar kernels.  tspo_amd/preprocess.py builds the tables.  */
int tspo_preprocess_frames_ex(const uint8_t* frames, int layout, int T, int H, int W,
                              const int32_t* hcoef, const int32_t* hbound, int out_w, int hk,
                              const int32_t* vcoef, const int32_t* vbound, int out_h, int vk,
                              int ylo, int nrows, uint8_t* out,
                              void* workspace, size_t workspace_bytes, tspo_stream_t stream,
                              const int8_t* mfma_taps, const int32_t* mfma_bias, const int32_t* mfma_xs,
                              int mfma_nkb, int mfma_span);

/* torch.nn.CosineSimilarity(dim=-1)(text[b,0,:], feat[b,t,:])  (temporal_agent.py:167).
 * txt f32 [B,M,D] (row 0 of each prompt is used), feat f32 [B,T,D] -> clip f32 [B,T]. */
int tspo_clip_scores(const float* txt, const float* feat, int B, int T, int D, int M, float* clip,
                     tspo_stream_t stream);

/* Generic bf16 MFMA GEMM exposed for tests / micro-benchmarks:
 * C[M,N] = A[M,K] * W[N,K]^T (+bias[N]) ; bf16 in, f32 accumulate, out bf16 or f32.
 * K % 64 == 0.  act (bits 0-7): 0 none, 1 quick_gelu.  residual (bf16 [M,N], nullable) is added; it may alias C (the encoder's
 * out-proj / fc2 update the residual stream in place: every read of an element precedes its store).
 * Bits 8+ of `act` choose the kernel: 0 = automatic (what the encoder uses: 77 for big shapes, else 1), 1 = small-problem
 * 128x128 kernel, 77 = persistent 256x256 kernel with LDS-DMA operands (whole rounds of whole tiles + the 64x64 remainder
 * phase), 83 = the same kernel with the remainder phase off (a partial last round of whole tiles; bit-identical with 77).
 * Every choice computes the same result.  Other values (67-76 schedule A/Bs and probes, 82 = the register-staged kernel of
 * rounds 2-3) exist only in a `python -m tspo_amd.build --dev` library and are rejected with TSPO_EINVAL by the shipped one. */
int tspo_gemm_bf16(const void* A, const void* W, const float* bias, const void* residual,
                   void* C, int out_dtype, int M, int N, int K, int act, tspo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TSPO_HIP_H */
