// loss.hip — K2: GRPO loss forward+backward over packed samples, and K3: group-relative advantage.
//
// Reference arithmetic (src/prime_rl/): compute_loss (trainer/rl/loss.py:285-403), default_loss_fn (:109-161),
// ipo_loss_fn (:164-196), ref_kl_loss_fn (:199-243), ce_loss_fn (:246-259), _safe_mean (:94-97),
// shift_tensor_right (:76-91, called at trainer/rl/train.py:450-457); GRPOAlgorithm.score_group
// (orchestrator/algo/grpo.py:25-45), MaxRLAlgorithm.score_group (orchestrator/algo/max_rl.py:22-28),
// assign_advantages (orchestrator/algo/routing.py:11-34).
//
// The reference runs ~30 eager launches + 7 boolean-index host syncs PER SAMPLE; here one launch walks every sample of
// the micro-batch (one workgroup per sample = one segment of cu_samples), emits the scaled loss, dL/dlogprob and the raw
// per-sample sums, and a second single-workgroup launch folds the per-sample sums in the reference's order.
// HBM: 17 B/token read (+4 per weight stream, +4 ref) and 4 B/token written; latency-bound at 2k tokens.
#include "common.hip.h"

namespace pm {

constexpr int kLossBlock = 256;

struct TokGrad {
  float g;
};

template <bool HAS_RL_W, bool HAS_CE, bool HAS_REFKL>
__global__ __launch_bounds__(kLossBlock) void grpo_loss_sample_kernel(
    const float* __restrict__ lp_raw, const float* __restrict__ lp_inf, const float* __restrict__ lp_ref, const float* __restrict__ adv,
    const uint8_t* __restrict__ mask, const float* __restrict__ rl_w, const float* __restrict__ ce_w, const float* __restrict__ rk_w,
    const int32_t* __restrict__ cu_samples, const uint8_t* __restrict__ starts_mb, int32_t n_samples, int64_t n_tokens, pm_loss_params P,
    float* __restrict__ grad, float* __restrict__ stats) {
  __shared__ float red[PM_LOSS_NSTATS][kLossBlock / 64];
  const int s = blockIdx.x;
  const int64_t begin = cu_samples[s], end = cu_samples[s + 1];
  const bool opens_mb = (s == 0) || (starts_mb && starts_mb[s] != 0);
  const bool closes_mb = (s == n_samples - 1) || (starts_mb ? (starts_mb[s + 1] != 0) : false);
  const float inv_rl = 1.0f / P.rl_scale, inv_ce = 1.0f / P.ce_scale, inv_rk = 1.0f / P.ref_kl_scale;

  float acc[PM_LOSS_NSTATS];
#pragma unroll
  for (int k = 0; k < PM_LOSS_NSTATS; ++k) acc[k] = 0.f;

  for (int64_t i = begin + threadIdx.x; i < end; i += kLossBlock) {
    // shift_tensor_right fused: position i sees the head output of position i-1 (loss.py:76-91)
    float lp;
    const bool is_pad = P.shift && (i == begin) && opens_mb;
    if (P.shift)
      lp = is_pad ? P.pad_logprob : lp_raw[i - 1];
    else
      lp = lp_raw[i];
    const float li = lp_inf[i];
    const float a = adv[i];
    const bool m = mask[i] != 0;

    // compute_importance_ratio_and_mismatch_kl (loss.py:100-106)
    const float lr = lp - li;
    const float ratio = expf(lr);
    const float mkl = ratio - lr - 1.0f;
    const float pd = expf(lp) - expf(li);

    float g_tok = 0.f;

    // ---- rl component -------------------------------------------------------------------------
    {
      float w = 1.0f;
      bool mrl = m;
      if (HAS_RL_W) {
        w = rl_w[i];
        mrl = m && (w != 0.f);
      }
      bool is_masked, is_low = false, is_high = false;
      const bool pos = a > 0.f, neg = a < 0.f;
      if (P.loss_type == PM_LOSS_IPO) {
        is_masked = fabsf(pd) > P.ipo_threshold;
      } else {
        const bool inv_high = pd > P.dppo_mask_high;
        const bool inv_low = pd < -P.dppo_mask_low;
        is_masked = pos ? inv_high : inv_low;  // torch.where(positive_advantages, high, low): adv==0 uses the low test
        is_high = pos && inv_high;
        is_low = neg && inv_low;
      }
      const bool drop = mrl && is_masked;
      const bool keep = mrl && !is_masked;
      const float a_s = P.adv_tau * a;
      const float pg = keep ? (a_s * ratio) : 0.f;
      const float kl = mrl ? (lr * lr) : 0.f;
      float ptl = -pg + P.kl_tau * kl;
      float dptl = -pg + P.kl_tau * (mrl ? 2.0f * lr : 0.f);  // d ratio/d lp = ratio
      if (HAS_RL_W) {
        ptl *= w;
        dptl *= w;
      }
      acc[PM_STAT_RL_LOSS] += ptl;
      g_tok += dptl * inv_rl;
      acc[PM_STAT_RL_N_MASK] += mrl ? 1.f : 0.f;
      acc[PM_STAT_RL_N_DROP] += drop ? 1.f : 0.f;
      acc[PM_STAT_RL_N_KEEP] += keep ? 1.f : 0.f;
      acc[PM_STAT_RL_MKL_DROP] += drop ? mkl : 0.f;
      acc[PM_STAT_RL_MKL_KEEP] += keep ? mkl : 0.f;
      acc[PM_STAT_RL_N_LOW] += (mrl && is_low) ? 1.f : 0.f;
      acc[PM_STAT_RL_N_HIGH] += (mrl && is_high) ? 1.f : 0.f;
      acc[PM_STAT_RL_N_DROP_POS] += (drop && pos) ? 1.f : 0.f;
      acc[PM_STAT_RL_N_DROP_NEG] += (drop && neg) ? 1.f : 0.f;
    }
    // ---- ce component (loss.py:246-259) -------------------------------------------------------
    if (HAS_CE) {
      const float w = ce_w[i];
      const bool mc = w != 0.f;
      if (mc) {
        acc[PM_STAT_CE_LOSS] += (-lp) * w;
        acc[PM_STAT_CE_N] += 1.f;
        acc[PM_STAT_CE_NLL] += -lp;
        g_tok += (-w) * inv_ce;
      }
    }
    // ---- ref_kl component (loss.py:199-243) ---------------------------------------------------
    if (HAS_REFKL) {
      const float w = rk_w[i];
      const bool mk = w != 0.f;
      if (mk) {
        const float lref = lp_ref[i];
        const bool is_masked = pd < -0.2f;
        const bool keep = !is_masked;
        const float rkl = lref - lp;  // detached in the pg term
        const float pg = keep ? (rkl * ratio) : 0.f;
        const float ptl = (-pg + 1e-3f * (lr * lr)) * w;
        const float dptl = (-pg + 1e-3f * 2.0f * lr) * w;
        acc[PM_STAT_REFKL_LOSS] += ptl;
        g_tok += dptl * inv_rk;
        acc[PM_STAT_REFKL_N] += 1.f;
        acc[PM_STAT_REFKL_N_DROP] += is_masked ? 1.f : 0.f;
        acc[PM_STAT_REFKL_N_KEEP] += keep ? 1.f : 0.f;
        acc[PM_STAT_REFKL_MKL_DROP] += is_masked ? mkl : 0.f;
        acc[PM_STAT_REFKL_MKL_KEEP] += keep ? mkl : 0.f;
        acc[PM_STAT_REFKL_SUM] += rkl;
      }
    }
    // ---- gradient, written in the head's (unshifted) indexing ------------------------------------
    if (P.shift) {
      if (!is_pad) grad[i - 1] = g_tok;  // the pad position has no upstream logprob
    } else {
      grad[i] = g_tok;
    }
  }
  if (P.shift && closes_mb && threadIdx.x == 0 && end > begin) grad[end - 1] = 0.f;  // shifted off the end (loss.py:91)

  // block reduction of the 20 sums
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < PM_LOSS_NSTATS; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) red[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < PM_LOSS_NSTATS) {
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kLossBlock / 64; ++w2) v += red[threadIdx.x][w2];
    stats[(int64_t)s * PM_LOSS_NSTATS + threadIdx.x] = v;
  }
}

__device__ __forceinline__ float safe_mean(float sum, float cnt) { return sum / fmaxf(cnt, 1.0f); }

// Folds per-sample sums in the reference's order: rl_loss = ((0 + l_0) + l_1) + ...; scaled = rl/rl_scale + ce/ce_scale + ref/ref_scale.
__global__ void grpo_loss_finalize_kernel(const float* __restrict__ stats, int32_t n_samples, pm_loss_params P, float* __restrict__ loss_out,
                                          float* __restrict__ metrics) {
  for (int s = threadIdx.x; s < n_samples; s += blockDim.x) {
    const float* st = stats + (int64_t)s * PM_LOSS_NSTATS;
    float* me = metrics + (int64_t)s * PM_LOSS_NMETRICS;
    me[PM_MET_MASKED_MISMATCH_KL] = safe_mean(st[PM_STAT_RL_MKL_DROP], st[PM_STAT_RL_N_DROP]);
    me[PM_MET_UNMASKED_MISMATCH_KL] = safe_mean(st[PM_STAT_RL_MKL_KEEP], st[PM_STAT_RL_N_KEEP]);
    me[PM_MET_IS_MASKED] = safe_mean(st[PM_STAT_RL_N_DROP], st[PM_STAT_RL_N_MASK]);
    me[PM_MET_IS_MASKED_LOW] = safe_mean(st[PM_STAT_RL_N_LOW], st[PM_STAT_RL_N_MASK]);
    me[PM_MET_IS_MASKED_HIGH] = safe_mean(st[PM_STAT_RL_N_HIGH], st[PM_STAT_RL_N_MASK]);
    me[PM_MET_MASKED_ADV_POS] = safe_mean(st[PM_STAT_RL_N_DROP_POS], st[PM_STAT_RL_N_DROP]);
    me[PM_MET_MASKED_ADV_NEG] = safe_mean(st[PM_STAT_RL_N_DROP_NEG], st[PM_STAT_RL_N_DROP]);
    me[PM_MET_NLL] = safe_mean(st[PM_STAT_CE_NLL], st[PM_STAT_CE_N]);
    me[PM_MET_REFKL_MASKED_MISMATCH_KL] = safe_mean(st[PM_STAT_REFKL_MKL_DROP], st[PM_STAT_REFKL_N_DROP]);
    me[PM_MET_REFKL_UNMASKED_MISMATCH_KL] = safe_mean(st[PM_STAT_REFKL_MKL_KEEP], st[PM_STAT_REFKL_N_KEEP]);
    me[PM_MET_REFKL_IS_MASKED] = safe_mean(st[PM_STAT_REFKL_N_DROP], st[PM_STAT_REFKL_N]);
    me[PM_MET_REFKL] = safe_mean(st[PM_STAT_REFKL_SUM], st[PM_STAT_REFKL_N]);
  }
  if (threadIdx.x == 0) {
    float rl = 0.f, ce = 0.f, rk = 0.f;
    for (int s = 0; s < n_samples; ++s) {
      const float* st = stats + (int64_t)s * PM_LOSS_NSTATS;
      rl = rl + st[PM_STAT_RL_LOSS];
      ce = ce + st[PM_STAT_CE_LOSS];
      rk = rk + st[PM_STAT_REFKL_LOSS];
    }
    loss_out[0] = rl / P.rl_scale + ce / P.ce_scale + rk / P.ref_kl_scale;
  }
}

// ---- K3: group-relative advantage --------------------------------------------------------------
// One wave64 per group; group_size is tens of rollouts, so one pass of shuffles (LDS not needed below 64 entries,
// strided loop above).
__global__ __launch_bounds__(64) void grpo_advantage_kernel(const float* __restrict__ rewards, const int32_t* __restrict__ offs, int mode,
                                                           const float* __restrict__ n_out, const float* __restrict__ n_tot,
                                                           const float* __restrict__ n_turns, float w_out, float w_in, float w_turn,
                                                           float* __restrict__ advs) {
  const int g = blockIdx.x;
  const int b = offs[g], e = offs[g + 1];
  const int n = e - b;
  if (n <= 0) return;
  const int lane = threadIdx.x;
  float sum = 0.f;
  for (int i = b + lane; i < e; i += 64) sum += rewards[i];
  sum = wave_sum(sum);
  const float mean = sum / (float)n;
  const bool shaped = (mode == PM_ADV_GRPO) && n_out && n_tot && n_turns;
  if (!shaped) {
    for (int i = b + lane; i < e; i += 64) {
      const float r = rewards[i];
      float a;
      if (mode == PM_ADV_MAXRL)
        a = (mean <= 0.f) ? 0.f : (r - mean) / mean;
      else
        a = r - mean;
      advs[i] = a;
    }
    return;
  }
  // linear length penalty (grpo.py:32-43)
  float mo = 0.f, mi = 0.f, mt = 0.f;
  for (int i = b + lane; i < e; i += 64) {
    mo = fmaxf(mo, n_out[i]);
    mi = fmaxf(mi, n_tot[i] - n_out[i]);
    mt = fmaxf(mt, n_turns[i]);
  }
  // .max().clamp(min=1): the group max may be negative in principle; start from -inf-free 0 only when data is >= 0.
  // Token counts and turn counts are non-negative, so max(0, .) then clamp(min=1) equals clamp(min=1) of the true max.
  mo = fmaxf(wave_max(mo), 1.f);
  mi = fmaxf(wave_max(mi), 1.f);
  mt = fmaxf(wave_max(mt), 1.f);
  float ssum = 0.f;
  for (int i = b + lane; i < e; i += 64) {
    const float frac = w_out * (n_out[i] / mo) + w_in * ((n_tot[i] - n_out[i]) / mi) + w_turn * (n_turns[i] / mt);
    ssum += rewards[i] - mean * frac;
  }
  ssum = wave_sum(ssum);
  const float smean = ssum / (float)n;
  for (int i = b + lane; i < e; i += 64) {
    const float frac = w_out * (n_out[i] / mo) + w_in * ((n_tot[i] - n_out[i]) / mi) + w_turn * (n_turns[i] / mt);
    advs[i] = (rewards[i] - mean * frac) - smean;
  }
}

__global__ __launch_bounds__(256) void advantage_broadcast_kernel(const float* __restrict__ radv, const int32_t* __restrict__ roffs,
                                                                 const uint8_t* __restrict__ tmask, float* __restrict__ tadv) {
  const int r = blockIdx.x;
  const int64_t b = roffs[r], e = roffs[r + 1];
  const float a = radv[r];
  for (int64_t i = b + threadIdx.x; i < e; i += 256) tadv[i] = tmask[i] ? a : 0.f;
}

}  // namespace pm

using namespace pm;

extern "C" int pm_grpo_loss_fwd_bwd(const float* trainer_logprobs, const float* inference_logprobs, const float* ref_logprobs,
                                    const float* advantages, const uint8_t* loss_mask, const float* rl_weights, const float* ce_weights,
                                    const float* ref_kl_weights, const int32_t* cu_samples, const uint8_t* sample_starts_mb,
                                    int32_t n_samples, int64_t n_tokens, const pm_loss_params* params_host, float* loss_out,
                                    float* grad_logprobs, float* sample_stats, float* sample_metrics, void* stream) {
  PM_REQUIRE(trainer_logprobs && inference_logprobs && advantages && loss_mask && cu_samples && params_host && loss_out && grad_logprobs &&
                 sample_stats && sample_metrics,
             PM_ERR_INVALID_ARG, "pm_grpo_loss_fwd_bwd: null pointer");
  PM_REQUIRE(n_samples > 0 && n_tokens > 0, PM_ERR_INVALID_ARG, "pm_grpo_loss_fwd_bwd: n_samples=%d n_tokens=%lld", n_samples, (long long)n_tokens);
  PM_REQUIRE(!(ref_kl_weights && !ref_logprobs), PM_ERR_INVALID_ARG,
             "pm_grpo_loss_fwd_bwd: ref_kl loss type requires ref_logprobs (loss.py:213-214)");
  PM_REQUIRE(params_host->rl_scale > 0 && params_host->ce_scale > 0 && params_host->ref_kl_scale > 0, PM_ERR_INVALID_ARG,
             "pm_grpo_loss_fwd_bwd: scales must be >= 1");
  hipStream_t st = as_stream(stream);
  const pm_loss_params P = *params_host;
  const dim3 grid((unsigned)n_samples), block(kLossBlock);
#define PM_LAUNCH_LOSS(A, B, C)                                                                                                           \
  hipLaunchKernelGGL((grpo_loss_sample_kernel<A, B, C>), grid, block, 0, st, trainer_logprobs, inference_logprobs, ref_logprobs, advantages, \
                     loss_mask, rl_weights, ce_weights, ref_kl_weights, cu_samples, sample_starts_mb, n_samples, n_tokens, P, grad_logprobs, \
                     sample_stats)
  const int key = (rl_weights ? 4 : 0) | (ce_weights ? 2 : 0) | (ref_kl_weights ? 1 : 0);
  switch (key) {
    case 0: PM_LAUNCH_LOSS(false, false, false); break;
    case 1: PM_LAUNCH_LOSS(false, false, true); break;
    case 2: PM_LAUNCH_LOSS(false, true, false); break;
    case 3: PM_LAUNCH_LOSS(false, true, true); break;
    case 4: PM_LAUNCH_LOSS(true, false, false); break;
    case 5: PM_LAUNCH_LOSS(true, false, true); break;
    case 6: PM_LAUNCH_LOSS(true, true, false); break;
    default: PM_LAUNCH_LOSS(true, true, true); break;
  }
#undef PM_LAUNCH_LOSS
  PM_CHECK_LAUNCH("pm_grpo_loss_fwd_bwd");
  hipLaunchKernelGGL(grpo_loss_finalize_kernel, dim3(1), dim3(256), 0, st, sample_stats, n_samples, P, loss_out, sample_metrics);
  PM_CHECK_LAUNCH("pm_grpo_loss_fwd_bwd(finalize)");
  return PM_OK;
}

extern "C" int pm_grpo_advantage(const float* rewards, const int32_t* group_offsets, int32_t n_groups, int mode, const float* num_output_tokens,
                                 const float* num_total_tokens, const float* num_turns, float w_output, float w_input, float w_turns,
                                 float* advantages, void* stream) {
  PM_REQUIRE(rewards && group_offsets && advantages, PM_ERR_INVALID_ARG, "pm_grpo_advantage: null pointer");
  PM_REQUIRE(n_groups >= 0 && (mode == PM_ADV_GRPO || mode == PM_ADV_MAXRL), PM_ERR_INVALID_ARG, "pm_grpo_advantage: bad n_groups/mode");
  if (n_groups == 0) return PM_OK;
  hipLaunchKernelGGL(grpo_advantage_kernel, dim3((unsigned)n_groups), dim3(64), 0, as_stream(stream), rewards, group_offsets, mode,
                     num_output_tokens, num_total_tokens, num_turns, w_output, w_input, w_turns, advantages);
  PM_CHECK_LAUNCH("pm_grpo_advantage");
  return PM_OK;
}

extern "C" int pm_advantage_broadcast(const float* rollout_adv, const int32_t* rollout_offsets, int32_t n_rollouts, const uint8_t* token_mask,
                                      float* token_adv, int64_t n_tokens, void* stream) {
  PM_REQUIRE(rollout_adv && rollout_offsets && token_mask && token_adv, PM_ERR_INVALID_ARG, "pm_advantage_broadcast: null pointer");
  PM_REQUIRE(n_rollouts >= 0 && n_tokens >= 0, PM_ERR_INVALID_ARG, "pm_advantage_broadcast: bad sizes");
  if (n_rollouts == 0) return PM_OK;
  hipLaunchKernelGGL(advantage_broadcast_kernel, dim3((unsigned)n_rollouts), dim3(256), 0, as_stream(stream), rollout_adv, rollout_offsets,
                     token_mask, token_adv);
  PM_CHECK_LAUNCH("pm_advantage_broadcast");
  return PM_OK;
}
