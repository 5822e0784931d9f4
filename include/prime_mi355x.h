/*
 * prime_mi355x.h — C ABI of libprime_mi355x.so, the MI355X (gfx950 / CDNA4) kernels behind the
 * prime-rl GRPO policy-update hot path.
 *
 * The reference (PrimeIntellect-ai/prime-rl) has NO native boundary on this path:
 * its seams are Python (lm_head.py, loss.py, attn.py, norms.py, ...).  This header is the boundary a
 * maintainer would bind from those seams (ctypes stubs: INTEGRATION.md).  Every entry point cites the
 * reference interface (file:line under src/prime_rl/) whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void*; kernels are only ENQUEUED on it: no allocation,
 *     no host synchronisation, no hidden device-to-host copies.
 *   - return 0 on success, a negative PM_ERR_* otherwise; pm_last_error() gives the message
 *     (thread-local).  No exception crosses the ABI.
 *   - index tensors: labels / token ids int64 (torch.long), segment offsets int32.
 *   - dtype arguments take PM_DTYPE_*.
 *   - re-entrant; no mutable globals beyond a once-initialised device-property cache.
 */
#ifndef PRIME_MI355X_H
#define PRIME_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 2

enum { PM_DTYPE_F32 = 0, PM_DTYPE_BF16 = 1, PM_DTYPE_I64 = 2, PM_DTYPE_U8 = 3 /* the last two: pm_comm_* payloads only */ };

enum {
  PM_OK = 0,
  PM_ERR_INVALID_ARG = -1,   /* null pointer, negative size, unsupported dtype/shape */
  PM_ERR_UNSUPPORTED = -2,   /* shape outside what the kernel was built for (e.g. head_dim != 128) */
  PM_ERR_LAUNCH = -3,        /* hipGetLastError() after enqueue was not hipSuccess */
  PM_ERR_WORKSPACE = -4      /* workspace too small */
};

/* loss types of the rl component: trainer/rl/loss.py:262-282 (setup_rl_loss_fn) */
enum { PM_LOSS_DEFAULT = 0 /* default_loss_fn, loss.py:109-161 */, PM_LOSS_IPO = 1 /* ipo_loss_fn, loss.py:164-196 */ };

/* advantage rules sharing the score_group hook: orchestrator/algo/grpo.py:25-45, algo/max_rl.py:22-28 */
enum { PM_ADV_GRPO = 0, PM_ADV_MAXRL = 1 };

int pm_abi_version(void);
const char* pm_last_error(void);
/* Name of the device the library sees (rocminfo gcnArchName, e.g. "gfx950:sramecc+:xnack-"); writes into buf. */
int pm_device_arch(char* buf_host, int buf_len);

/* --------------------------------------------------------------------------------------------
 * K1. Per-token log-prob gather over the vocab + entropy, on materialised logits.
 *     Replaces selective_log_softmax + compute_entropy + the temperature divide
 *     (trainer/rl/loss.py:48-63, trainer/rl/train.py:436-443) and the inner epilogue of
 *     _SequenceChunkedLogProbEntropyFn.forward (trainer/models/layers/lm_head.py:89-156).
 *
 *       l[n,v]     = float(logits[n,v]) * inv_temp[n]
 *       logz[n]    = logsumexp_v l[n,v]
 *       logprob[n] = l[n,label[n]] - logz[n]            (label gather is an exact indexed load)
 *       entropy[n] = logz[n] - sum_v softmax(l)[v]*l[n,v]
 *
 *     logits: [n_rows, row_stride] row-major, dtype f32 or bf16, vocab = first `vocab` columns.
 *     workspace: pm_logprob_workspace_bytes(n_rows, vocab) bytes (split-vocab partials for small n_rows).
 * ------------------------------------------------------------------------------------------ */
int64_t pm_logprob_workspace_bytes(int64_t n_rows, int64_t vocab);
int pm_logprob_entropy_fwd(const void* logits, int dtype, int64_t n_rows, int64_t vocab, int64_t row_stride,
                           const int64_t* labels, const float* inv_temp,
                           float* logprobs, float* entropy, float* logz,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* Backward of K1 w.r.t. logits (entropy has no backward: lm_head.py:160-161 asserts grad_entropy is None):
 *   dlogits[n,v] = grad_logprobs[n] * inv_temp[n] * ( [v==label[n]] - exp(l[n,v]-logz[n]) )
 * lm_head.py:184-191.  dlogits may alias logits when dtypes match (in-place). */
int pm_logprob_bwd(const void* logits, int dtype, int64_t n_rows, int64_t vocab, int64_t row_stride,
                   const int64_t* labels, const float* inv_temp, const float* logz,
                   const float* grad_logprobs, void* dlogits, int dlogits_dtype, int64_t dlogits_stride,
                   void* stream);

/* Vocab-chunked form (logits chunk = hidden @ W[vocab_start:vocab_start+chunk_vocab].T, never the full [N,V]):
 * online (m, s, t, target) state update per chunk then finalize — lm_head.py:76-86,118-148.
 * state_* are [n_rows] f32; initialise with pm_logprob_chunk_init. */
int pm_logprob_chunk_init(float* state_m, float* state_s, float* state_t, float* state_target, int64_t n_rows, void* stream);
int pm_logprob_chunk_update(const void* logits_chunk, int dtype, int64_t n_rows, int64_t chunk_vocab, int64_t row_stride,
                            int64_t vocab_start, const int64_t* labels, const float* inv_temp,
                            float* state_m, float* state_s, float* state_t, float* state_target, void* stream);
int pm_logprob_chunk_finalize(const float* state_m, const float* state_s, const float* state_t, const float* state_target,
                              int64_t n_rows, float* logprobs, float* entropy, float* logz, void* stream);
/* chunk backward: same formula as pm_logprob_bwd with labels offset by vocab_start (lm_head.py:184-191). */
int pm_logprob_chunk_bwd(const void* logits_chunk, int dtype, int64_t n_rows, int64_t chunk_vocab, int64_t row_stride,
                         int64_t vocab_start, const int64_t* labels, const float* inv_temp, const float* logz,
                         const float* grad_logprobs, void* dlogits, int dlogits_dtype, int64_t dlogits_stride, void* stream);

/* --------------------------------------------------------------------------------------------
 * J1. Fused LM head: logits = hidden . W^T on the MFMA units with the K1 epilogue applied to the accumulator tile — the [N, V]
 *     logits never reach HBM.  Replaces FusedOutputLinear.forward / _SequenceChunkedLogProbEntropyFn
 *     (trainer/models/layers/lm_head.py:42-62, 89-156: 19 chunk GEMMs + ~10 elementwise passes over every chunk) and the
 *     recompute + dlogits part of its backward (lm_head.py:158-191); the two gradient GEMMs (:193-199) stay library calls on the
 *     dlogits this writes.
 *
 *     hidden [n_tokens, hidden] bf16, weight [vocab, hidden] bf16 (both row-major, K-contiguous: no transposed copy of the head);
 *     hidden % 64 == 0.  round_logits_bf16 != 0 rounds the GEMM result to bf16 before the f32 epilogue (the reference's bf16
 *     `hidden_chunk @ w_chunk.t()`, lm_head.py:133); 0 keeps the f32 accumulator.
 *     fwd: logprobs / entropy / logz [n_tokens] f32 as K1.  logits_out (nullable): also store the raw (unscaled) bf16 logits,
 *          row stride logits_ld — for callers that prefer keeping logits to recomputing them in backward.
 *     bwd: dlogits[n, v] = grad_logprobs[n] * inv_temp[n] * ([v == label[n]] - exp(l[n, v] * inv_temp[n] - logz[n])) as bf16 with
 *          row stride dlogits_ld (a multiple of 4); logits are recomputed on the fly.
 *     workspace (fwd): pm_lmhead_workspace_bytes(n_tokens, vocab): per-token partial online-softmax states (32 B per token and
 *          256-column vocab tile from 1024 tokens up: 311 MB at 16 384 tokens x 151 936 — against 5 GB of logits).
 *     Two builds behind these entry points: from 1024 tokens, with round_logits_bf16 != 0, hidden % 128 == 0 and no logits_out (bwd: vocab and
 *          dlogits_ld multiples of 8, 16-byte aligned dlogits), the passes run as MODE 8 / 9 of the GEMM core's tile stream (csrc/gemm.hip: the product of
 *          pm_gemm_bf16 with the log-prob epilogue in place of the result store); otherwise lmhead_kernel (csrc/lmhead.hip).  Same results to f32
 *          summation order over the vocabulary; each build is bitwise reproducible.
 * ------------------------------------------------------------------------------------------ */
int64_t pm_lmhead_workspace_bytes(int64_t n_tokens, int64_t vocab);
int pm_lmhead_logprob_fwd(const void* hidden_bf16, const void* weight_bf16, int64_t n_tokens, int64_t hidden, int64_t vocab,
                          const int64_t* labels, const float* inv_temp, int32_t round_logits_bf16,
                          float* logprobs, float* entropy, float* logz,
                          void* logits_out_bf16 /* nullable */, int64_t logits_ld,
                          void* workspace, int64_t workspace_bytes, void* stream);
int pm_lmhead_logprob_bwd(const void* hidden_bf16, const void* weight_bf16, int64_t n_tokens, int64_t hidden, int64_t vocab,
                          const int64_t* labels, const float* inv_temp, const float* logz, const float* grad_logprobs,
                          int32_t round_logits_bf16, void* dlogits_bf16, int64_t dlogits_ld, void* stream);

/* --------------------------------------------------------------------------------------------
 * K3. Group-relative advantage — GRPOAlgorithm.score_group (orchestrator/algo/grpo.py:25-45),
 *     MaxRLAlgorithm.score_group (algo/max_rl.py:22-28).
 *     rewards: flat [n_rollouts]; group g owns rollouts [group_offsets[g], group_offsets[g+1]).
 *       GRPO : adv = r - mean(r)       (no std division)
 *       length penalty (optional, all three pointers non-null): grpo.py:32-43
 *         frac = w_out*out/max(max(out),1) + w_in*(tot-out)/max(max(tot-out),1) + w_turn*turns/max(max(turns),1)
 *         shaped = r - mean(r)*frac ; adv = shaped - mean(shaped)
 *       MaxRL: adv = mean<=0 ? 0 : (r-mean)/mean
 * ------------------------------------------------------------------------------------------ */
int pm_grpo_advantage(const float* rewards, const int32_t* group_offsets, int32_t n_groups, int mode,
                      const float* num_output_tokens, const float* num_total_tokens, const float* num_turns,
                      float w_output, float w_input, float w_turns,
                      float* advantages, void* stream);

/* Broadcast one scalar per rollout over its mask-True tokens — assign_advantages (algo/routing.py:11-34)
 * + the per-token stream of trace_to_samples (orchestrator/trajectories.py:111-160).
 * token t of rollout r (tokens [rollout_offsets[r], rollout_offsets[r+1])) gets adv[r] if mask[t] else 0. */
int pm_advantage_broadcast(const float* rollout_adv, const int32_t* rollout_offsets, int32_t n_rollouts,
                           const uint8_t* token_mask, float* token_adv, int64_t n_tokens, void* stream);

/* --------------------------------------------------------------------------------------------
 * K2. GRPO loss over a packed micro-batch (or several concatenated ones), forward AND backward in one pass.
 *     Replaces compute_loss (trainer/rl/loss.py:285-403) with default_loss_fn (:109-161) / ipo_loss_fn (:164-196)
 *     for the rl component, ce_loss_fn (:246-259), ref_kl_loss_fn (:199-243), and the two shift_tensor_right
 *     calls of train.py:450-457.
 *
 *     Tokens are flat [n_tokens]; sample s owns [cu_samples[s], cu_samples[s+1]).
 *     trainer_logprobs are the UNSHIFTED head outputs (logprob of token i+1 at position i).  With
 *     shift=1 the kernel reads lp_shifted[i] = (i is the first token of a micro-batch) ? pad_logprob : lp[i-1]
 *     (sample_starts_mb[s] != 0 marks samples that open a micro-batch; null = only sample 0) and writes
 *     the gradient back in unshifted indexing (grad_logprobs[i-1] += dL/dlp_shifted[i]; the last position of a
 *     micro-batch gets 0).
 *
 *     loss_out[0]  = rl/rl_scale + ce/ce_scale + ref_kl/ref_kl_scale        (the scaled loss, loss.py:394)
 *     grad_logprobs[n_tokens] = d loss_out / d trainer_logprobs              (f32)
 *     sample_stats[n_samples][PM_LOSS_NSTATS] raw per-sample sums (see enum) from which the wrapper forms the
 *     reference's metrics dict (loss.py:151-159, 186-194, 234-241, 255-258) without device-host syncs.
 *     scales are max(global_count,1) computed by the caller (train.py:307-323).
 * ------------------------------------------------------------------------------------------ */
enum {
  PM_STAT_RL_LOSS = 0,        /* un-normalised rl loss sum of the sample */
  PM_STAT_RL_N_MASK,          /* count(mask_rl) */
  PM_STAT_RL_N_DROP,          /* count(mask_rl & is_masked) */
  PM_STAT_RL_N_KEEP,          /* count(mask_rl & ~is_masked) */
  PM_STAT_RL_MKL_DROP,        /* sum mismatch_kl over drop */
  PM_STAT_RL_MKL_KEEP,        /* sum mismatch_kl over keep */
  PM_STAT_RL_N_LOW,           /* count(mask_rl & is_masked_low)  (default loss only) */
  PM_STAT_RL_N_HIGH,          /* count(mask_rl & is_masked_high) (default loss only) */
  PM_STAT_RL_N_DROP_POS,      /* count(drop & adv>0) */
  PM_STAT_RL_N_DROP_NEG,      /* count(drop & adv<0) */
  PM_STAT_CE_LOSS,            /* un-normalised ce loss sum */
  PM_STAT_CE_N,               /* count(ce_w != 0) */
  PM_STAT_CE_NLL,             /* sum(-lp) over ce mask (unweighted) */
  PM_STAT_REFKL_LOSS,         /* un-normalised ref_kl loss sum */
  PM_STAT_REFKL_N,            /* count(ref_kl_w != 0) */
  PM_STAT_REFKL_N_DROP,
  PM_STAT_REFKL_N_KEEP,
  PM_STAT_REFKL_MKL_DROP,
  PM_STAT_REFKL_MKL_KEEP,
  PM_STAT_REFKL_SUM,          /* sum(ref_lp - lp) over ref_kl mask */
  PM_LOSS_NSTATS
};

typedef struct {
  int32_t loss_type;        /* PM_LOSS_DEFAULT | PM_LOSS_IPO */
  float dppo_mask_low;      /* DefaultLossConfig, configs/trainer.py:494-507 (0.2) */
  float dppo_mask_high;     /* 0.2 */
  float adv_tau;            /* 1.0 */
  float kl_tau;             /* 1e-3 */
  float ipo_threshold;      /* IPOLossConfig (0.1) */
  float rl_scale;           /* max(global rl token count,1) as float (train.py:307-323); the kernel divides by it */
  float ce_scale;
  float ref_kl_scale;
  int32_t shift;            /* 1: apply shift_tensor_right inside the kernel */
  float pad_logprob;        /* log(1/vocab) (train.py:452) */
} pm_loss_params;

/* per-sample metrics written by the kernel (NaN-free: _safe_mean semantics, loss.py:94-97) */
enum {
  PM_MET_MASKED_MISMATCH_KL = 0, PM_MET_UNMASKED_MISMATCH_KL, PM_MET_IS_MASKED, PM_MET_IS_MASKED_LOW, PM_MET_IS_MASKED_HIGH,
  PM_MET_MASKED_ADV_POS, PM_MET_MASKED_ADV_NEG,                                   /* default/ipo rl metrics, loss.py:151-159 */
  PM_MET_NLL,                                                                     /* ce, loss.py:255-258 */
  PM_MET_REFKL_MASKED_MISMATCH_KL, PM_MET_REFKL_UNMASKED_MISMATCH_KL, PM_MET_REFKL_IS_MASKED, PM_MET_REFKL,  /* loss.py:234-241 */
  PM_LOSS_NMETRICS
};

int pm_grpo_loss_fwd_bwd(const float* trainer_logprobs, const float* inference_logprobs,
                         const float* ref_logprobs /* nullable */, const float* advantages,
                         const uint8_t* loss_mask,
                         const float* rl_weights /* nullable */, const float* ce_weights /* nullable */,
                         const float* ref_kl_weights /* nullable */,
                         const int32_t* cu_samples, const uint8_t* sample_starts_mb /* nullable */,
                         int32_t n_samples, int64_t n_tokens, const pm_loss_params* params_host,
                         float* loss_out, float* grad_logprobs, float* sample_stats /* [n_samples][PM_LOSS_NSTATS] */,
                         float* sample_metrics /* [n_samples][PM_LOSS_NMETRICS] */, void* stream);

/* --------------------------------------------------------------------------------------------
 * Packed-sequence scoring forward/backward (Qwen3 dense), bf16 activations, f32 accumulation.
 * ------------------------------------------------------------------------------------------ */

/* Embedding gather / deterministic scatter-add — nn.Embedding (models/qwen3/modeling_qwen3.py:139,174). */
int pm_embedding_fwd(const void* table_bf16, const int64_t* ids, int64_t n_tokens, int64_t hidden, int64_t vocab,
                     void* out_bf16, void* stream);
/* grad_table_f32[id] += sum of grad_out rows with that id.  `order` = argsort(ids) (stable), ids_sorted = ids[order]. */
int pm_embedding_bwd(const void* grad_out_bf16, const int64_t* ids_sorted, const int64_t* order, int64_t n_tokens,
                     int64_t hidden, int64_t vocab, float* grad_table_f32, void* stream);  /* ids outside [0, vocab) are skipped */
/* The same scatter-add into a WINDOW of the flattened [vocab, hidden] gradient: grad_window_f32 points at element `elem_lo` and holds
 * `elem_count` elements (both multiples of 8; the window may start and end inside a row).  Rows outside the window and ids < 0 (padding)
 * are skipped.  Under ZeRO sharding the last pass's embedding gradient — n_tokens rows, not vocab rows — is all-gathered and each rank adds
 * the rows of its own shard, so the dense reduce-scatter of the embedding bucket (FSDP's reduce-scatter of the embedding gradient,
 * trainer/model.py:823-960) is issued one pass early and overlaps the last pass instead of trailing it (trainer/engine.py). */
int pm_embedding_bwd_window(const void* grad_out_bf16, const int64_t* ids_sorted, const int64_t* order, int64_t n_tokens,
                            int64_t hidden, int64_t vocab, int64_t elem_lo, int64_t elem_count, float* grad_window_f32, void* stream);

/* 2-D transpose of 16-bit elements: dst[c*dst_ld + r] = src[r*src_ld + c].  Replaces the `.transpose(1, 2)` / `.t().contiguous()`
 * copies around the attention and linear layers (models/layers/attn.py:147-158) and feeds the library GEMMs the operand
 * orientation they run fastest in (K-contiguous weights for dgrad, K-contiguous activations for wgrad).  cols and both leading
 * dimensions must be multiples of 8 (16-byte vectors); rows is arbitrary (dst_ld >= rows, e.g. rows rounded up to 8). */
int pm_transpose_2d(const void* src_16bit, int64_t rows, int64_t cols, int64_t src_ld, void* dst_16bit, int64_t dst_ld, void* stream);

/* RMSNorm — RMSNorm.forward (models/layers/norms.py:36-44): y = w * bf16(x*rsqrt(mean(x^2)+eps)); fp32 math.
 * Optional fused residual: if residual != null, x_in = x + residual is written to sum_out (bf16) and normalised. */
int pm_rmsnorm_fwd(const void* x_bf16, const void* residual_bf16 /* nullable */, const void* weight_bf16,
                   int64_t n_rows, int64_t hidden, float eps,
                   void* y_bf16, void* sum_out_bf16 /* nullable unless residual */, float* rstd, void* stream);
/* dx = rstd*(g*w - xhat*mean(g*w*xhat)) (+ dresidual if non-null, fused residual-gradient add);
 * dweight_f32[h] += sum_rows g*xhat  (two-stage deterministic reduction through workspace). */
int64_t pm_rmsnorm_bwd_workspace_bytes(int64_t n_rows, int64_t hidden);
int pm_rmsnorm_bwd(const void* grad_y_bf16, const void* x_bf16, const void* weight_bf16, const float* rstd,
                   const void* dresidual_bf16 /* nullable */, int64_t n_rows, int64_t hidden,
                   void* grad_x_bf16, float* grad_weight_f32 /* accumulated into */,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* cos/sin table — RotaryEmbedding.forward (models/layers/rotary_emb.py:60-73): angle = pos*inv_freq[j] in f32
 * (inv_freq is the module's registered buffer, rotary_emb.py:24,55), cos/sin rounded to bf16;
 * table[n][0:d/2] (the reference's cat(f,f) duplicates it). */
int pm_rope_table(const int64_t* position_ids, int64_t n_tokens, const float* inv_freq, int32_t half_dim,
                  void* cos_bf16, void* sin_bf16, void* stream);

/* Per-head q/k RMSNorm + RoPE (half-split) fused, no transposes — FlashAttention.attn_projections
 * (models/layers/attn.py:123-160) + apply_rotary_pos_emb (rotary_emb.py:83-118).
 * qkv: [n_tokens, (hq+2*hkv)*128] bf16 (fused q|k|v projection output).  Writes q_out [n,hq,128], k_out [n,hkv,128]
 * (normed+roped), and keeps per-(token,head) rstd for backward.  bf16 rounding points follow the reference
 * (norm -> bf16, *w -> bf16, *cos -> bf16, *sin -> bf16, add -> bf16).
 * q_norm_w = k_norm_w = NULL: `use_qk_norm = False` (the Llama block, models/llama/modeling_llama.py:59-60, and HF Qwen2): RoPE
 * only; rstd is not written (may be NULL) and the backward needs neither qkv, rstd, the weight gradients nor a workspace. */
int pm_qknorm_rope_fwd(const void* qkv_bf16, const void* q_norm_w_bf16, const void* k_norm_w_bf16,
                       const void* cos_bf16, const void* sin_bf16,
                       int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float eps,
                       void* q_out_bf16, void* k_out_bf16, float* rstd /* [n, hq+hkv] */, void* stream);
int64_t pm_qknorm_rope_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads);
/* grad wrt the raw q/k slices of qkv, written into grad_qkv[:, 0:(hq+hkv)*128]; dweights accumulated (f32). */
int pm_qknorm_rope_bwd(const void* grad_q_bf16, const void* grad_k_bf16, const void* qkv_bf16,
                       const void* q_norm_w_bf16, const void* k_norm_w_bf16,
                       const void* cos_bf16, const void* sin_bf16, const float* rstd,
                       int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                       void* grad_qkv_bf16, float* grad_q_norm_w_f32, float* grad_k_norm_w_f32,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* out_accumulate_f32[c] += sum over rows of x[r, c] (x bf16 [n_rows, n_cols], row stride ld elements; n_cols, ld multiples of 8):
 * the bias gradient of the q/k/v projections when `attention_bias = True` (nn.Linear(bias=True), models/layers/attn.py:64-72).
 * Deterministic two-stage reduction. */
int64_t pm_column_sum_workspace_bytes(int64_t n_cols);
int pm_column_sum_bf16(const void* x_bf16, int64_t n_rows, int64_t n_cols, int64_t ld, float* out_accumulate_f32,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* token -> start index of its document; cu_seqlens from get_cu_seqlens_from_seq_lens (utils/sequence.py:38-52)
 * without its three host syncs. */
int pm_seq_start_from_cu(const int32_t* cu_seqlens, int32_t n_seqs, int64_t n_tokens, int32_t* seq_start, void* stream);

/* Varlen causal attention with GQA over packed documents — FlashAttention._compute_attention
 * (models/layers/attn.py:94-110; flash_attn_varlen_func(q,k,v,cu,cu,max,max,causal=True)).
 * q [n,hq,128], k,v [n,hkv,128] bf16 (v may be a strided view of qkv: v_row_stride in elements);
 * out [n,hq,128] bf16; lse [hq,n] f32 (natural-log logsumexp of scaled scores).
 * Row i attends to j in [seq_start[i], i].  head_dim must be 128. */
int64_t pm_attn_fwd_workspace_bytes(int64_t n_tokens, int32_t n_kv_heads, int32_t head_dim);
int pm_attn_varlen_fwd(const void* q_bf16, const void* k_bf16, const void* v_bf16, int64_t v_row_stride,
                       const int32_t* seq_start, int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads,
                       int32_t head_dim, float softmax_scale,
                       void* out_bf16, float* lse, void* workspace, int64_t workspace_bytes, void* stream);
int64_t pm_attn_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim);
/* grad_v is written with row stride grad_v_row_stride (so it can land in the v slice of grad_qkv). Deterministic. */
int pm_attn_varlen_bwd(const void* grad_out_bf16, const void* q_bf16, const void* k_bf16, const void* v_bf16,
                       int64_t v_row_stride, const void* out_bf16, const float* lse, const int32_t* seq_start,
                       int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale,
                       void* grad_q_bf16, void* grad_k_bf16, void* grad_v_bf16, int64_t grad_v_row_stride,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Row-window forms of the two calls above — one rank's share of a pass under context parallelism with gathered K / V (the reference's
 * `ring_fa3_varlen_func`, models/layers/ring_attn.py:99-280: all-gather K / V over the cp group, attend with the local q chunk
 * against `local_k_slice`, reduce-scatter dK / dV).  k, v, seq_start (and grad_k, grad_v) hold rows [0, q_row0 + n_q) of the pass,
 * the q-side arrays (q, out, lse [hq][n_q], grad_out, grad_q) only the n_q rows from q_row0 on; q_row0 must be a multiple of 128.
 * grad_k / grad_v receive THIS q window's contribution for every kv row (zeros where no q row of the window attends): the caller
 * sums them over the cp ranks.  q_row0 = 0 is exactly pm_attn_varlen_fwd / _bwd; the workspace is pm_attn_bwd_workspace_bytes(n_q). */
int pm_attn_varlen_fwd_rows(const void* q_bf16, const void* k_bf16, const void* v_bf16, int64_t v_row_stride, const int32_t* seq_start,
                            int64_t q_row0, int64_t n_q, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale,
                            void* out_bf16, float* lse, void* stream);
int pm_attn_varlen_bwd_rows(const void* grad_out_bf16, const void* q_bf16, const void* k_bf16, const void* v_bf16, int64_t v_row_stride,
                            const void* out_bf16, const float* lse, const int32_t* seq_start, int64_t q_row0, int64_t n_q,
                            int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale,
                            void* grad_q_bf16, void* grad_k_bf16, void* grad_v_bf16, int64_t grad_v_row_stride,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* SwiGLU — MLP.forward (models/layers/mlp.py:27-29): act = bf16(silu(gate)) * up, on the fused [n, 2*inter] gate|up. */
int pm_swiglu_fwd(const void* gate_up_bf16, int64_t n_rows, int64_t inter, void* act_bf16, void* stream);
int pm_swiglu_bwd(const void* grad_act_bf16, const void* gate_up_bf16, int64_t n_rows, int64_t inter,
                  void* grad_gate_up_bf16, void* stream);

/* --------------------------------------------------------------------------------------------
 * Gradient post-processing + optimizer on flat f32 shards (train.py:557-570,
 * trainer/utils.py:78-98, optim/__init__.py:101-108 AdamW(lr, wd, betas, fused)).
 * ------------------------------------------------------------------------------------------ */
/* main_grad_f32 += float(grad_bf16) */
int pm_grad_accumulate(float* main_grad_f32, const void* grad_bf16, int64_t n, void* stream);
/* partial sums of squares of a flat f32 buffer; out_sumsq[0] += sum(x^2) (deterministic two-stage). */
int64_t pm_sumsq_workspace_bytes(int64_t n);
int pm_sumsq(const float* x, int64_t n, float* out_sumsq /* device scalar, accumulated into */,
             void* workspace, int64_t workspace_bytes, void* stream);
/* The same for `n_segments` (offset, length) pairs of a flat buffer, one partial per segment (segments <= 1 Mi elements each; int64 pairs on the device),
 * and the deterministic last stage: out[0] += sum(partials). */
int pm_sumsq_segments(const float* base, const int64_t* segments /* [n_segments][2] */, int32_t n_segments, float* partials, void* stream);
int pm_sum_partials(const float* partials, int64_t n, float* out /* device scalar, accumulated into */, void* stream);
/* clip_coef[0] = min(1, max_norm/(sqrt(total_sumsq*grad_scale^2)+1e-6)); grad_norm_out[0] = sqrt(total_sumsq)*grad_scale
 * (torch clip_grad_norm_ semantics, trainer/utils.py:88-98). */
int pm_clip_coef(const float* total_sumsq, float grad_scale, float max_norm, float* clip_coef, float* grad_norm_out, void* stream);
/* Fused AdamW step on a flat shard: g = grad*grad_scale*clip_coef[0]; decoupled weight decay; bias correction from
 * `step` (1-based); writes the bf16 compute copy and (zero_grad) clears the gradient. */
int pm_adamw_step(float* master_f32, float* exp_avg_f32, float* exp_avg_sq_f32, float* grad_f32,
                  void* param_bf16 /* nullable */, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, float grad_scale, const float* clip_coef /* device, nullable=1 */,
                  int32_t zero_grad, void* stream);
/* The reference's other elementwise optimizers on the same flat shards (trainer/optim/__init__.py:74-95), same gradient scaling /
 * clipping / bf16-shadow / zero_grad contract as pm_adamw_step.
 * pm_sgd_step: torch.optim.SGD(lr, momentum, nesterov, weight_decay) — g += wd * p; buf = g at step 1, else momentum * buf + g;
 * g = nesterov ? g + momentum * buf : buf; p -= lr * g.  momentum_buf may be NULL when momentum == 0.
 * pm_sign_sgd_step: SignSGD (trainer/sign_sgd.py:34-51) — p -= lr * wd * p; p -= lr * sign(g).  No optimizer state. */
int pm_sgd_step(float* master_f32, float* momentum_buf_f32 /* nullable */, float* grad_f32, void* param_bf16 /* nullable */, int64_t n,
                float lr, float momentum, float weight_decay, int32_t nesterov, int64_t step, float grad_scale,
                const float* clip_coef /* device, nullable=1 */, int32_t zero_grad, void* stream);
int pm_sign_sgd_step(float* master_f32, float* grad_f32, void* param_bf16 /* nullable */, int64_t n, float lr, float weight_decay,
                     float grad_scale, const float* clip_coef /* device, nullable=1 */, int32_t zero_grad, void* stream);

/* --------------------------------------------------------------------------------------------
 * pm_comm_*: RCCL collectives over xGMI behind the C ABI (SURVEY.md section 8(b) minimum set).
 *
 * What they replace in the reference: the raw NCCL communicator of the trainer -> inference weight push
 * (vllm PyNcclCommunicator on a StatelessProcessGroup: transports/weights/nccl.py:27-68,128-136 sender,
 * inference/vllm/worker/nccl.py:28-89 receiver) — a communicator that lives NEXT to whatever process group the host framework
 * already has — and, optionally, the c10d collectives FSDP2 issues for sharding (trainer/model.py:823-960: all-gather,
 * reduce-scatter) and the i64[3] scale all-reduce (trainer/rl/train.py:318-323).
 *
 * A communicator is created from a 128-byte unique id: one rank calls pm_comm_unique_id() and ships the bytes to the others out
 * of band (the Python side uses a torch TCPStore at weight_broadcast.host:port, configs/trainer.py:557-578); every rank then calls
 * pm_comm_init(&comm, id, n_ranks, rank) with its HIP device current.  All data pointers are DEVICE pointers; every call only
 * ENQUEUES on `stream` (a hipStream_t) and returns — order against compute with events, wait with the stream.  Sums only
 * (ncclSum): that is all the path uses.  pm_comm_group_start/end bracket point-to-point batches (ncclGroupStart/End): the
 * sharded weight push sends every trainer's slice to every inference GPU over its own xGMI link in one group.
 * librccl is resolved with dlopen at first use (the copy already mapped into the process wins); pm_comm_available() == 0 and
 * PM_ERR_UNSUPPORTED everywhere else if there is none.
 * ------------------------------------------------------------------------------------------ */
#define PM_COMM_UNIQUE_ID_BYTES 128
int pm_comm_available(void);
int pm_comm_rccl_version(void);                       /* ncclGetVersion code (e.g. 22107), negative on error */
int pm_comm_unique_id(void* id_host_128);
int pm_comm_init(void** comm_out, const void* id_host_128, int32_t n_ranks, int32_t rank);
int pm_comm_destroy(void* comm);
int pm_comm_rank_size(void* comm, int32_t* rank_out_host, int32_t* size_out_host);
/* recv[i] = sum over ranks of send[i]; in place allowed (send == recv). */
int pm_comm_all_reduce(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream);
/* recv <- root's send (send is ignored on non-root ranks; send == recv allowed). */
int pm_comm_broadcast(void* comm, const void* send, void* recv, int64_t count, int dtype, int32_t root, void* stream);
/* send holds n_ranks * recv_count elements; rank r receives the sum of everybody's r-th block (FSDP gradient reduce-scatter). */
int pm_comm_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype, void* stream);
/* recv holds n_ranks * send_count elements, block r = rank r's send (in place allowed when send == recv + rank*send_count). */
int pm_comm_all_gather(void* comm, const void* send, void* recv, int64_t send_count, int dtype, void* stream);
/* block p of send goes to rank p, block r of recv comes from rank r (Ulysses sequence <-> head exchange). */
int pm_comm_all_to_all(void* comm, const void* send, void* recv, int64_t count_per_peer, int dtype, void* stream);
int pm_comm_send(void* comm, const void* send, int64_t count, int dtype, int32_t peer, void* stream);
int pm_comm_recv(void* comm, void* recv, int64_t count, int dtype, int32_t peer, void* stream);
int pm_comm_group_start(void);
int pm_comm_group_end(void);

/* --------------------------------------------------------------------------------------------
 * MoE routing / permutation / combine (the expert-parallel half) — src/prime_rl/trainer/models/layers/moe.py.
 * Token-major tensors; a "slot" is one (token, j < top_k) selection, flattened slot = token * top_k + j; "sorted row" = position of a
 * slot after the stable sort by expert (TokenReorderer, moe.py:769-809).  The expert GEMMs stay library calls (grouped GEMM).
 *
 * pm_moe_router_fwd — TokenChoiceTopKRouter.forward after the gate GEMM (moe.py:707-746):
 *     scores[N,E] f32 = softmax (score_func 0) | sigmoid (1) of gate_logits (bf16 or f32); selection = forced_idx[N,k] (routed_experts,
 *     moe.py:718-720) | top-k of scores + expert_bias[E] (moe.py:726-728) | top-k of scores (ties -> lowest expert index);
 *     top_scores[N,k] = unbiased selected scores, / (their sum + 1e-20) if route_norm, * route_scale; top_idx[N,k];
 *     counts[E] (nullable; zeroed by the call) = selections per expert (histc, moe.py:739-744) — the same numbers are the differences of
 *     pm_moe_sort's offsets, which is where the hot path takes them from (contended atomics here); confidence[N] = selected probability mass per token
 *     (_selected_probability_mass_sum, moe.py:634-641: the caller sums it).
 * pm_moe_router_bwd — d top_scores -> d gate_logits [N,E] f32 through normalisation, scale, gather and softmax / sigmoid.
 * pm_moe_sort — stable counting sort of top_idx (flattened, n_slots = N*k) by expert: dest_of_slot[n_slots], slot_of_dest[n_slots]
 *     (= torch.argsort(stable=True), moe.py:801), offsets[E+1] (exclusive scan of counts).  Four launches, no host sync.
 * pm_moe_gather_rows — out[r] = x[row_map[r] / divisor] (* scale[row_map[r]] in f32, rounded to bf16): the expert-sorted gather
 *     (moe.py:891-895, divisor = top_k, row_map = slot_of_dest, scale = top_scores when score_before_experts) and, with divisor 1, any
 *     row permutation (expert-parallel regrouping).  row_map[r] < 0 -> zero row.
 * pm_moe_combine — out[t] = bf16(base[t] + sum over the token's k slots of bf16(y[dest_of_slot[t*k+j]] * scores[t*k+j])), added in ascending
 *     sorted-row order in f32 and rounded once: the reference's bf16 scatter_add (moe.py:1043-1051) as a deterministic gather.
 *     scores / base nullable (plain sum = backward of the gather).
 * pm_moe_slot_grads — per sorted row r (slot s = slot_of_dest[r], token t = s / k): rows_out[r] = bf16(a[t] * scores[s]) and
 *     dscore[s] = <a[t], b[r]> in f32: the two gradients of the per-row score multiply (moe.py:894-900).  Either output nullable.
 * pm_moe_ep_row_map — expert-parallel regrouping after the row all-to-all (torchtitan ExpertParallel._token_dispatch, used by
 *     trainer/distributed/expert_parallel.py:47-63): recv_counts[ep][n_local] -> row_map (expert-major row -> received row), its
 *     inverse, and local_offsets[n_local+1]. */
int pm_moe_router_fwd(const void* gate_logits, int32_t logits_dtype, const float* expert_bias /* nullable */, const int32_t* forced_idx /* nullable */,
                      int64_t n_tokens, int32_t n_experts, int32_t top_k, int32_t score_func, int32_t route_norm, float route_scale,
                      float* scores, float* top_scores, int32_t* top_idx, int32_t* counts /* nullable */, float* confidence, void* stream);
int pm_moe_router_bwd(const float* scores, const int32_t* top_idx, const float* grad_top_scores, int64_t n_tokens, int32_t n_experts, int32_t top_k,
                      int32_t score_func, int32_t route_norm, float route_scale, float* grad_logits, void* stream);
int64_t pm_moe_sort_workspace_bytes(int64_t n_slots, int32_t n_experts);
int pm_moe_sort(const int32_t* top_idx, int64_t n_slots, int32_t n_experts, int32_t* dest_of_slot, int32_t* slot_of_dest, int32_t* offsets,
                void* workspace, int64_t workspace_bytes, void* stream);
int pm_moe_gather_rows(const void* x_bf16, const int32_t* row_map, int32_t divisor, const float* scale /* nullable */, int64_t n_rows,
                       int64_t row_elems, void* out_bf16, void* stream);
int pm_moe_combine(const void* y_bf16, const int32_t* dest_of_slot, const float* scores /* nullable */, const void* base_bf16 /* nullable */,
                   int64_t n_tokens, int32_t top_k, int64_t row_elems, void* out_bf16, void* stream);
int pm_moe_slot_grads(const void* a_bf16, const void* b_bf16 /* nullable iff dscore is */, const int32_t* slot_of_dest, int32_t top_k,
                      const float* scores /* nullable */, int64_t n_rows, int64_t row_elems, void* rows_out_bf16 /* nullable */,
                      float* dscore /* nullable */, void* stream);
int pm_moe_ep_row_map(const int32_t* recv_counts, int32_t ep_size, int32_t n_local_experts, int32_t* row_map, int32_t* inv_map,
                      int32_t* local_offsets, void* stream);

/* Grouped expert GEMMs over the expert-sorted rows — _run_experts_grouped_mm_impl (layers/moe.py:161-188: torch._grouped_mm(x,
 * w.transpose(-2, -1), offs=offsets)) and the two gradients autograd derives from it.  offsets[E+1] i32 on the device (rows of expert e =
 * [offsets[e], offsets[e+1])), bf16 operands, f32 accumulation, bf16 results; n_out / k_dim multiples of 128 (pm_grouped_gemm_supported):
 *   nt: y[m][n]     = sum_k x[m][k]  * w[e][n][k]      x [M, K], w [E, N, K] -> y [M, N]
 *   nn: dx[m][k]    = sum_n dy[m][n] * w[e][n][k]      dy [M, N], w [E, N, K] -> dx [M, K]
 *   tn: dw[e][n][k] = sum_{m in group e} dy[m][n] * x[m][k]   -> dw [E, N, K] (zeros for an empty group)
 * Each takes an optional SECOND weight set for the columns [n_first, N) (w2 [E, N - n_first, K]; n_first = N and w2 = NULL for one set), so
 * the experts' gate and up projections (w1, w3 of GroupedExperts, moe.py:436-440) run as one product whose output is the gate|up layout
 * pm_swiglu_fwd consumes, their row gradients as one contraction over both, and their weight gradients as one launch with two outputs. */
int pm_grouped_gemm_supported(int64_t n_rows, int64_t n_out, int64_t k_dim);
int pm_grouped_gemm_debug_set_variant(int32_t value); /* `make VARIANTS=1` libraries only: process-wide override of PM_GROUPED_SCHED_* (see `sched` below) */
int pm_grouped_gemm_nt(const void* x_bf16, const void* w_bf16, const void* w2_bf16 /* nullable */, const int32_t* offsets, int32_t n_experts, int64_t n_rows,
                       int64_t n_first, int64_t n_out, int64_t k_dim, void* y_bf16, int32_t sched, void* stream);
int pm_grouped_gemm_nn(const void* dy_bf16, const void* w_bf16, const void* w2_bf16 /* nullable */, const int32_t* offsets, int32_t n_experts, int64_t n_rows,
                       int64_t n_first, int64_t n_dim, int64_t k_out, void* dx_bf16, int32_t sched, void* stream);
/* The experts' gate and up projections with the SwiGLU activation in the epilogue (layers/moe.py:181-182: F.silu(x @ w1^T) * (x @ w3^T); the reference runs two
 * grouped GEMMs, silu and mul): gate_up[n_rows, 2 inter] exactly as pm_grouped_gemm_nt(w_gate, w_up) writes it and act[n_rows, inter] = pm_swiglu_fwd(gate_up),
 * bit for bit, in one launch.  Both weight tensors [E, inter, k_dim] are read through one buffer descriptor: w_up must lie above w_gate within 2 GiB
 * and a whole number of weight rows (k_dim elements) away (parameters of one flat buffer are) — PM_ERR_UNSUPPORTED otherwise, and the caller keeps the two-launch
 * form.  inter % 128 == 0, k_dim % 128 == 0. */
int pm_grouped_gemm_nt_swiglu_supported(int64_t n_rows, int64_t inter, int64_t k_dim);
int pm_grouped_gemm_nt_swiglu(const void* x_bf16, const void* w_gate_bf16, const void* w_up_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t inter,
                              int64_t k_dim, void* gate_up_bf16, void* act_bf16, int32_t sched, void* stream);
/* The experts' down-projection row gradient with the SwiGLU backward in its epilogue (layers/moe.py:181-183 backward): grad_gate_up[m] = [d_gate | d_up] from
 * (dy[m] . w_down[e(m)]) and the saved gate_up[m]; dy [n_rows, n_dim], w_down [E, n_dim, inter], gate_up / grad_gate_up [n_rows, 2 inter] (may alias).
 * inter % 256 == 0, n_dim % 128 == 0. */
int pm_grouped_gemm_nn_swiglu_bwd_supported(int64_t n_rows, int64_t n_dim, int64_t inter);
int pm_grouped_gemm_nn_swiglu_bwd(const void* dy_bf16, const void* w_down_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_dim, int64_t inter,
                                  const void* gate_up_bf16, void* grad_gate_up_bf16, int32_t sched, void* stream);
int pm_grouped_gemm_tn(const void* dy_bf16, const void* x_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_first, int64_t n_dim,
                       int64_t k_dim, void* dw_bf16, void* dw2_bf16 /* nullable */, int32_t sched, void* stream);
/* The same with the gradient written (accumulate = 0) or accumulated (1) in f32 straight into a main-gradient buffer [E, n_first, K] / [E, N - n_first, K]
 * (gradient accumulation in f32: configs/trainer.py:233-237; the reference's autograd adds bf16 `.grad`s that FSDP then reduces in f32).  n_first, N, K
 * multiples of 256. */
int pm_grouped_gemm_tn_f32(const void* dy_bf16, const void* x_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_first, int64_t n_dim,
                           int64_t k_dim, float* dw_f32, float* dw2_f32 /* nullable */, int32_t accumulate, int32_t sched, void* stream);
/* ... with the gradient-norm pass fused into its epilogue (the grouped counterpart of pm_gemm_bf16_sumsq; the reference computes the norm with
 * torch.nn.utils.clip_grad_norm_ over the reduced gradients, trainer/rl/train.py:566-570): every wave also writes the sum of squares of the FINAL f32
 * values it stored for output tile (expert e, tile t) to tile_sumsq[(e * tiles + t) * 8 + wave]; every tile of every expert is visited by every launch,
 * so after a step's last accumulating launch sum(tile_sumsq) is the squared norm of dw and dw2.  tile_sumsq_len >= pm_grouped_gemm_tn_sumsq_slots(). */
int64_t pm_grouped_gemm_tn_sumsq_slots(int32_t n_experts, int64_t n_dim, int64_t k_dim);
int pm_grouped_gemm_tn_f32_sumsq(const void* dy_bf16, const void* x_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_first, int64_t n_dim,
                                 int64_t k_dim, float* dw_f32, float* dw2_f32 /* nullable */, int32_t accumulate, float* tile_sumsq, int64_t tile_sumsq_len,
                                 int32_t sched, void* stream);

/* --------------------------------------------------------------------------------------------
 * Dense bf16 GEMM of the scoring model (csrc/gemm.hip): OUT[M, N] = sum_K X[M, K] * W[N, K] (+ bias[N]), f32 accumulation on the MFMA units,
 * bf16 or f32 result.  Replaces every `F.linear` / autograd matmul the reference hands to cuBLAS:
 *   forward  (trainer/models/layers/mlp.py:27-29, attn.py:60-84, lm_head.py:118-148)   x [M, K], w [N, K]          x_kmajor = 0, w_kmajor = 0
 *   dgrad    dx = dy @ w            dy [M, K = out], w stored [K = out, N = in]                                   x_kmajor = 0, w_kmajor = 1
 *   wgrad    dw[out][in] = sum_tok dy[tok][out] x[tok][in]: X = dy stored [K = tok, M = out], W = x stored [K = tok, N = in],
 *            f32 output with `accumulate` (gradient accumulation in f32, configs/trainer.py:233-237)              x_kmajor = 1, w_kmajor = 1
 * "kmajor" = the operand is stored [K, rows] (its M / N coordinate is the contiguous one).  ld* are row strides in elements; rows 16-byte
 * aligned.  K % 128 == 0 unless both operands are K-major (pm_gemm_supported).  `bias` (bf16 [N], nullable) is added before the rounding.  A K-major
 * operand whose K x ld reaches past 2 GiB (the LM-head weight gradient: 16 384 x 151 936 dlogits) is contracted in chained parts inside one launch. */
int pm_gemm_supported(int64_t m, int64_t n, int64_t k, int32_t x_kmajor, int32_t w_kmajor);
int pm_gemm_bf16(const void* x_bf16, int64_t ldx, int32_t x_kmajor, const void* w_bf16, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo,
                 int32_t out_dtype /* PM_DTYPE_BF16 | PM_DTYPE_F32 */, const void* bias_bf16 /* nullable */, int32_t accumulate, int64_t m, int64_t n,
                 int64_t k, int32_t sched, void* stream);
/* The weight-gradient form with the gradient-norm pass fused into its epilogue (the reference clips the GLOBAL gradient norm before the optimizer step,
 * trainer/rl/train.py:557-570 `clip_grad_norm_`; here every wave adds up the squares of the FINAL f32 values it stores and writes the sum to
 * tile_sumsq[tile * 8 + wave] — pm_gemm_sumsq_slots(m, n) floats, deterministic, slots of tiles that do not exist are left untouched).  Summing the
 * slots (pm_sum_partials) gives sum(out^2) without reading the 4-byte-per-parameter gradient again. */
int64_t pm_gemm_sumsq_slots(int64_t m, int64_t n);
int pm_gemm_bf16_sumsq(const void* x_bf16, int64_t ldx, int32_t x_kmajor, const void* w_bf16, int64_t ldw, int32_t w_kmajor, float* out_f32, int64_t ldo,
                       int32_t accumulate, int64_t m, int64_t n, int64_t k, float* tile_sumsq, int64_t tile_sumsq_len, int32_t sched, void* stream);
/* The general form: optional per-tile sums of squares (as pm_gemm_bf16_sumsq) and an optional workspace of pm_gemm_workspace_bytes(m, n, k) bytes.  With the
 * workspace, an f32 product without bias whose tile count leaves at most half of the LAST round of 256 tiles occupied (e.g. the q|k|v weight gradient of
 * Qwen3-8B: 384 tiles = 1.5 rounds) computes that round split along K — 2 or 4 workgroups per tile on different XCDs, partial tiles in the workspace, a small
 * reduction kernel adds them in a fixed order (deterministic) — instead of leaving half of the chip idle for a whole tile time. */
int64_t pm_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);
int pm_gemm_bf16_ws(const void* x_bf16, int64_t ldx, int32_t x_kmajor, const void* w_bf16, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo,
                    int32_t out_dtype, const void* bias_bf16 /* nullable */, int32_t accumulate, int64_t m, int64_t n, int64_t k,
                    float* tile_sumsq /* nullable */, int64_t tile_sumsq_len, void* workspace /* nullable */, int64_t workspace_bytes, int32_t sched, void* stream);
/* The fused gate|up projection of the MLP with the SwiGLU activation in its epilogue (mlp.py:27-29: down_proj(act_fn(gate_proj(x)) * up_proj(x)); the
 * reference runs two F.linear, F.silu and a multiply as separate kernels): w = [gate rows ; up rows] ([2 inter, k], row stride ldw, the HF order of a
 * concatenated gate_proj / up_proj), gate_up[m, 2 inter] = x . w^T exactly as pm_gemm_bf16 writes it (kept for the backward pass), and
 * act[m, inter] = bf16(bf16(silu(gate)) * up) computed from the ROUNDED bf16 gate / up values — bit-identical to pm_swiglu_fwd on gate_up — without
 * reading gate_up back from HBM: a tile holds 128 features' gate and up columns in the same lanes.  inter % 128 == 0, k % 128 == 0. */
int pm_gemm_swiglu_supported(int64_t m, int64_t inter, int64_t k);
int pm_gemm_swiglu_bf16(const void* x_bf16, int64_t ldx, const void* w_bf16, int64_t ldw, void* gate_up_bf16, int64_t ldo, void* act_bf16, int64_t ldo2, int64_t m,
                        int64_t inter, int64_t k, int32_t sched, void* stream);
/* The down projection's row gradient with the SwiGLU backward in its epilogue (the backward of mlp.py:27-29; autograd runs mm, silu_backward, two muls):
 * d_act[m, inter] = dy[m, k] . w_down[k, inter] (w_down as stored, [hidden, inter], row stride ldw) is rounded to bf16 as pm_gemm_bf16 would store it and
 * then, still in registers, turned into grad_gate_up[m, 2 inter] = [d_gate | d_up] with the saved gate_up[m, 2 inter] — bit-identical to pm_gemm_bf16
 * followed by pm_swiglu_bwd, without d_act ever reaching HBM.  grad_gate_up may alias gate_up (every element is read before it is overwritten, by the
 * same lane).  inter % 8 == 0, k % 128 == 0. */
int pm_gemm_swiglu_bwd_supported(int64_t m, int64_t inter, int64_t k);
int pm_gemm_swiglu_bwd_bf16(const void* dy_bf16, int64_t lddy, const void* w_down_bf16, int64_t ldw, const void* gate_up_bf16, int64_t ldgu, void* grad_gate_up_bf16,
                            int64_t ldo, int64_t m, int64_t inter, int64_t k, int32_t sched, void* stream);
/* `sched` — the schedule word of a launch, passed WITH EVERY CALL of the eleven entry points above (the library keeps no schedule state):
 * PM_GEMM_SCHED_DEFAULT (-1) = the library default (PM_GEMM_SCHED_PERSISTENT | 3), or a combination of PM_GEMM_SCHED_PERSISTENT (bit 2: one workgroup per CU walking
 * its tiles instead of one workgroup per tile — what an engine that shares the GPU with RCCL kernels clears), PM_GEMM_SCHED_NO_SPLITK (bit 5),
 * PM_GEMM_SCHED_NT_OUT (bit 9: non-temporal result stores of pm_gemm_bf16*), PM_GEMM_SCHED_SWEEP_R3 (bit 10: the round-3 patch sweep order); bits 0-1 are accepted
 * and ignored (rounds 3-4 wrote 3 / 7).  The grouped entry points additionally read PM_GROUPED_SCHED_SMALL / _BIG (bits 16-17: the 128 x 128 / 256 x 256
 * kernels of csrc/grouped_gemm.hip instead of the gemm.hip tile stream — the A/B references of the GPU tests).  Any other bit selects a measured-and-not-adopted
 * build that exists only in a `make VARIANTS=1` library (profiles/r04_gemm.md, r05_gemm.md): PM_ERR_UNSUPPORTED in the product library.
 * pm_gemm_debug_set_variant / pm_grouped_gemm_debug_set_variant: a process-wide OVERRIDE of every launch's word, for the A/B tools of `make VARIANTS=1`
 * libraries only (value >= 0 sets, -1 clears); the product library has no such state and returns PM_ERR_UNSUPPORTED for any value >= 0. */
#define PM_GEMM_SCHED_DEFAULT (-1)
#define PM_GEMM_SCHED_PERSISTENT 4
#define PM_GEMM_SCHED_NO_SPLITK 32
#define PM_GEMM_SCHED_NT_OUT 512
#define PM_GEMM_SCHED_SWEEP_R3 1024
#define PM_GROUPED_SCHED_SMALL (1 << 16)
#define PM_GROUPED_SCHED_BIG (2 << 16)
int pm_gemm_debug_set_variant(int32_t value);

/* --------------------------------------------------------------------------------------------
 * Diagnostics used by tests/ to pin the MFMA fragment layouts the attention kernels assume.
 * C[16,16] = A[16,32] @ B[32,16]^T-free form: c = sum_k a[i][k]*b[j][k]  (both operands K-contiguous, bf16). */
int pm_mfma_probe_16x16x32(const void* a_bf16, const void* b_bf16, float* c_f32, void* stream);
/* Kernel-variant switch for in-process A/B timing (tools/microbench.py; selectable builds exist only in a `make VARIANTS=1` library): kernel 0 = attention
 * forward (default = the three-stage pipelined kernel with the hand-placed tile loop, csrc/attention_fwd_p3.hip.h; 0-15 = option bits of the round-1..5
 * kernel: 1 = asm row max, 2 = skip the O rescale when no row max moved, 4 = row-max reduction as one asm block, 8 = 16-byte epilogue stores, 15 = its
 * shipped form; 64 + OPT = the 8-wave ping-pong experiment of csrc/attention_fwd_pp.hip.h; 128 = the pipelined kernel with compiler-scheduled stages;
 * 129 + V = p3_loop_asm<V>), 1 = dK/dV (1: fragment batch 2, 2: fragment batch 8, 3: 32-row q tiles; + 16: 8-byte epilogue stores), 2 = dQ (1: 32 kv
 * rows per barrier; + 16: 8-byte epilogue stores); value <= 0 restores the default.  Not part of the training path. */
int pm_attn_debug_set_variant(int32_t kernel, int32_t value);
/* gfx950 LDS transpose-read (`ds_read_b64_tr_b16`): one wave, LDS image lds[i] = i (u16, 4096 entries); lane l reads at element
 * index elem_index[l] (device int32[64], multiples of 4) and writes its four 16-bit results to out[4*l .. 4*l+3] (device u16[256]). */
int pm_ds_read_tr16_probe(const int32_t* elem_index, void* out_u16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRIME_MI355X_H */
