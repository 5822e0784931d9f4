// moe.hip — token-choice MoE routing, expert-sorted permutation and combine for gfx950 (the EP half).
//
// Reference: TokenChoiceTopKRouter / TokenReorderer / MoE.forward in src/prime_rl/trainer/models/layers/moe.py:644-1054 (eager torch:
// F.softmax / sigmoid + torch.topk + histc; argsort(stable) + gather; per-row score multiply; scatter_add), driven per decoder layer by
// models/qwen3_moe/modeling_qwen3_moe.py:61-84 (softmax, norm_topk_prob, scores after the experts).  The expert GEMMs themselves stay
// library calls (grouped GEMM), like the dense projections.
//
// MI355X design — everything here is HBM / latency bound integer and row-copy work, so: one wave64 per token for the router (the E
// scores of a token live in registers: E/64 per lane, top-k by k rounds of a wave arg-max), a four-launch stable counting sort instead
// of a radix sort (E <= 1024 bins, block histograms -> per-(block, expert) bases -> in-block stable ranks by ballot matching; no atomics
// on floating point, bitwise reproducible), 16-byte row copies for the permutations, and the combine written as a GATHER over each
// token's k slots (deterministic; the reference's scatter_add is an atomic accumulation on the GPU) that reproduces the reference's CPU
// result bit for bit: contributions are added in ascending sorted-row order in f32 and rounded to bf16 once.
#include "common.hip.h"

namespace pm {

constexpr int kMaxPerLane = 16;  // experts per lane: E <= 1024
constexpr int kMaxTopK = 16;

// ---------------------------------------------------------------------------------------------------------------------
// Router forward: one wave per token.
//   scores = softmax / sigmoid of the gate logits in f32 (moe.py:707-712), saved for the backward;
//   selection = forced indices (moe.py:718-720) | top-k of scores + bias (moe.py:726-728) | top-k of scores (moe.py:730); ties -> lowest index;
//   gating value = unbiased score, optionally normalised over the selected k and scaled (moe.py:734-737);
//   confidence[token] = selected probability mass before normalisation (moe.py:634-641); counts[e] += 1 per selection (moe.py:739-744).
template <typename T, int kPer>  // kPer = experts per lane (E <= 64 * kPer): the score registers of a token
__global__ __launch_bounds__(256) void moe_router_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ bias, const int32_t* __restrict__ forced,
                                                            int64_t n, int E, int k, int sigmoid, int route_norm, float route_scale,
                                                            float* __restrict__ scores, float* __restrict__ top_scores, int32_t* __restrict__ top_idx,
                                                            int32_t* __restrict__ counts, float* __restrict__ confidence) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n) return;
  constexpr int per = kPer;
  float s[kPer], key[kPer];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int e = i * 64 + lane;
    s[i] = -INFINITY;
    if (i < per && e < E) {
      if constexpr (sizeof(T) == 2) s[i] = bf16_to_f32(logits[tok * E + e]);
      else s[i] = logits[tok * E + e];
      mx = fmaxf(mx, s[i]);
    }
  }
  float total = 0.f;
  if (sigmoid) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = i * 64 + lane;
      if (i < per && e < E) {
        s[i] = 1.0f / (1.0f + expf(-s[i]));
        total += s[i];
      }
    }
    total = wave_sum(total);
  } else {
    mx = wave_max(mx);
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = i * 64 + lane;
      if (i < per && e < E) {
        s[i] = expf(s[i] - mx);
        z += s[i];
      }
    }
    z = wave_sum(z);
    const float inv = 1.0f / z;
#pragma unroll
    for (int i = 0; i < kPer; ++i)
      if (i < per && i * 64 + lane < E) s[i] *= inv;
    total = 1.0f;
  }
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int e = i * 64 + lane;
    key[i] = -INFINITY;
    if (i < per && e < E) {
      scores[tok * E + e] = s[i];
      key[i] = bias ? s[i] + bias[e] : s[i];
    }
  }
  float sel = 0.f;  // lane j < k keeps the j-th selected (unbiased) score and its index
  int sel_e = 0;
  float picked_sum = 0.f;
  for (int j = 0; j < k; ++j) {
    int be;
    float bs;
    if (forced) {
      be = forced[tok * k + j];
      float mine = 0.f;
#pragma unroll
      for (int i = 0; i < kPer; ++i)
        if (i < per && i * 64 + lane == be) mine = s[i];
      bs = wave_sum(mine);  // exactly one lane holds it
    } else {
      float bk = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int e = i * 64 + lane;
        if (i < per && e < E && (key[i] > bk)) {  // strict: the lowest index of a lane's equal keys wins (i ascending)
          bk = key[i];
          bi = e;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {  // wave arg-max, ties to the lower expert index
        const float ok = __shfl_xor(bk, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ok > bk || (ok == bk && oi < bi)) {
          bk = ok;
          bi = oi;
        }
      }
      be = bi;
      float mine = 0.f;
#pragma unroll
      for (int i = 0; i < kPer; ++i)
        if (i < per && i * 64 + lane == be) {
          mine = s[i];
          key[i] = -INFINITY;  // taken
        }
      bs = wave_sum(mine);
    }
    if (lane == j) {
      sel = bs;
      sel_e = be;
    }
    picked_sum += bs;
  }
  // selected mass (moe.py:634-641): softmax -> sum of the selected scores; sigmoid -> that sum over the token's total score
  if (lane == 0) confidence[tok] = sigmoid ? picked_sum / (total + 1e-20f) : picked_sum;
  if (lane < k) {
    float g = sel;
    if (route_norm) g = g / (picked_sum + 1e-20f);
    g *= route_scale;
    top_scores[tok * k + lane] = g;
    top_idx[tok * k + lane] = sel_e;
    if (counts) atomicAdd(&counts[sel_e], 1);  // 1k-way contended at 16k tokens x 8: callers on the hot path take the counts from pm_moe_sort's offsets
  }
}

// Router backward: d(gating values) -> d(gate logits), f32 [N, E].  With s_i the selected unbiased scores, S = sum s_i + 1e-20:
//   g_i = dtop_i * route_scale;  normalised: ds_i = (g_i - sum_j g_j s_j / S) / S, else ds_i = g_i;
//   softmax: dlogit_e = p_e * (ds_e - sum_i s_i ds_i);  sigmoid: dlogit_e = s_e (1 - s_e) ds_e   (ds_e = 0 for unselected experts).
__global__ __launch_bounds__(256) void moe_router_bwd_kernel(const float* __restrict__ scores, const int32_t* __restrict__ top_idx,
                                                            const float* __restrict__ grad_top, int64_t n, int E, int k, int sigmoid, int route_norm,
                                                            float route_scale, float* __restrict__ grad_logits) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n) return;
  int e_sel = -1;
  float s_sel = 0.f, g = 0.f;
  if (lane < k) {
    e_sel = top_idx[tok * k + lane];
    s_sel = scores[tok * E + e_sel];
    g = grad_top[tok * k + lane] * route_scale;
  }
  float ds = g;
  if (route_norm) {
    const float S = wave_sum(s_sel) + 1e-20f;
    const float dot = wave_sum(g * s_sel);
    ds = (g - dot / S) / S;
  }
  const float sds = wave_sum(s_sel * ds);  // sum_i s_i ds_i (softmax)
  for (int e = lane; e < E; e += 64) {
    float dse = 0.f;
    for (int j = 0; j < k; ++j) {
      const int ej = __shfl(e_sel, j, 64);
      const float dj = __shfl(ds, j, 64);
      if (ej == e) dse += dj;  // a forced routing may name an expert twice
    }
    const float p = scores[tok * E + e];
    grad_logits[tok * E + e] = sigmoid ? p * (1.0f - p) * dse : p * (dse - sds);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stable counting sort of the flattened (token, slot) pairs by expert (TokenReorderer, moe.py:769-809: argsort(stable=True)).
constexpr int kSortChunk = 256;  // slots per block (one round of 4 waves: 512 blocks for 16k tokens x 8 slots)

__global__ __launch_bounds__(256) void moe_sort_hist_kernel(const int32_t* __restrict__ idx, int64_t m, int E, int n_blocks, int32_t* __restrict__ block_hist) {
  extern __shared__ int32_t hist[];
  for (int e = threadIdx.x; e < E; e += 256) hist[e] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortChunk;
  for (int r = 0; r < kSortChunk / 256; ++r) {
    const int64_t j = base + r * 256 + threadIdx.x;
    if (j < m) atomicAdd(&hist[idx[j]], 1);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) block_hist[(int64_t)e * n_blocks + blockIdx.x] = hist[e];  // [E][n_blocks]: one expert's row is scanned by one wave
}

// one wave per expert: totals[e] = sum over blocks
__global__ __launch_bounds__(256) void moe_sort_totals_kernel(const int32_t* __restrict__ block_hist, int n_blocks, int E, int32_t* __restrict__ totals) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  int t = 0;
  for (int b = lane; b < n_blocks; b += 64) t += block_hist[(int64_t)e * n_blocks + b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if (lane == 0) totals[e] = t;
}

// one wave per expert: offsets[e] = sum of the totals before e; block_hist[e][b] becomes the first sorted row of (block b, expert e)
__global__ __launch_bounds__(256) void moe_sort_scan_kernel(int32_t* __restrict__ block_hist, int n_blocks, int E, const int32_t* __restrict__ totals,
                                                           int32_t* __restrict__ offsets) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  int before = 0, all = 0;
  for (int q = lane; q < E; q += 64) {
    const int t = totals[q];
    all += t;
    if (q < e) before += t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o, 64);
    all += __shfl_xor(all, o, 64);
  }
  if (lane == 0) {
    offsets[e] = before;
    if (e == E - 1) offsets[E] = all;
  }
  int run = before;
  for (int b0 = 0; b0 < n_blocks; b0 += 64) {
    const int b = b0 + lane;
    const int c = b < n_blocks ? block_hist[(int64_t)e * n_blocks + b] : 0;
    int incl = c;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (b < n_blocks) block_hist[(int64_t)e * n_blocks + b] = run + incl - c;
    run += __shfl(incl, 63, 64);
  }
}

__global__ __launch_bounds__(256) void moe_sort_place_kernel(const int32_t* __restrict__ idx, int64_t m, int E, int n_blocks, const int32_t* __restrict__ block_base,
                                                            int32_t* __restrict__ dest_of_slot, int32_t* __restrict__ slot_of_dest) {
  extern __shared__ int32_t next[];  // next free sorted row per expert for this block
  for (int e = threadIdx.x; e < E; e += 256) next[e] = block_base[(int64_t)e * n_blocks + blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kSortChunk;
  for (int r = 0; r < kSortChunk / 256; ++r) {
    const int64_t j = base + r * 256 + threadIdx.x;
    const bool valid = j < m;
    const int e = valid ? idx[j] : -1;
    for (int ww = 0; ww < 4; ++ww) {  // the four waves of a round go in slot order: stability
      if (w == ww) {
        unsigned long long remaining = __ballot(valid);
        int dest = -1;
        while (remaining) {
          const int leader = __ffsll((long long)remaining) - 1;
          const int le = __shfl(e, leader, 64);
          const unsigned long long same = __ballot(valid && e == le);
          if (valid && e == le) {
            const int rank = __popcll(same & ((1ull << lane) - 1ull));
            dest = next[le] + rank;
          }
          __builtin_amdgcn_wave_barrier();
          if (lane == leader) next[le] += __popcll(same);
          __builtin_amdgcn_wave_barrier();
          remaining &= ~same;
        }
        if (valid) {
          dest_of_slot[j] = dest;
          slot_of_dest[dest] = (int32_t)j;
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// out[r] = x[map[r] / div] (* scale[map[r]]), bf16 rows of H elements (H % 8 == 0).  div = top_k with map = slot_of_dest gives the
// expert-sorted gather of moe.py:891-895 (the optional f32 multiply rounded to bf16 is score_before_experts); div = 1 is a plain row
// permutation (expert-parallel regrouping).  map[r] < 0 writes a zero row.
__global__ __launch_bounds__(256) void moe_gather_rows_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ map, int div,
                                                             const float* __restrict__ scale, int64_t rows, int64_t H, bf16_t* __restrict__ out) {
  const int64_t r = blockIdx.x;
  const int slot = map[r];
  const float sc = (scale && slot >= 0) ? scale[slot] : 1.0f;
  const bf16_t* src = x + (int64_t)(slot < 0 ? 0 : slot / div) * H;
  bf16_t* dst = out + r * H;
  for (int64_t c = (int64_t)threadIdx.x * 8; c < H; c += 256 * 8) {
    u32x4 v = slot < 0 ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4*>(src + c);
    if (scale && slot >= 0) {
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = pack_bf16_hw(bf16lo(w[i]) * sc, bf16hi(w[i]) * sc);
      v = u32x4{w[0], w[1], w[2], w[3]};
    }
    *reinterpret_cast<u32x4*>(dst + c) = v;
  }
}

// out[n] = bf16(base[n] + sum_i bf16(y[dest[n*k+i]] * score[n*k+i])) over the token's k slots in ASCENDING sorted-row order, accumulated in
// f32 and rounded once — bit for bit what torch's CPU `out.scatter_add(0, rows, routed_output)` (moe.py:1050-1051) gives for bf16 (its
// expanded-index path accumulates each output row in f32), and a fixed order where the GPU scatter_add uses bf16 atomics in arrival
// order.  score = null: plain sum (the backward of the gather).
// KC = the top-k count at compile time (8: Qwen3-MoE) or 0 = the runtime value: with a constant count the k row loads of a thread are all in flight
// before the first is used (the runtime loop issued load, use, load, use: 3.4 TB/s at k = 8); same additions in the same order either way.
template <int KC>
__global__ __launch_bounds__(256) void moe_combine_kernel(const bf16_t* __restrict__ y, const int32_t* __restrict__ dest_of_slot, const float* __restrict__ score,
                                                         const bf16_t* __restrict__ base, int64_t n, int k_rt, int64_t H, bf16_t* __restrict__ out) {
  const int k = KC ? KC : k_rt;
  const int64_t tok = blockIdx.x;
  int d[kMaxTopK];
  float sc[kMaxTopK];
  for (int i = 0; i < k; ++i) {
    d[i] = dest_of_slot[tok * k + i];
    sc[i] = score ? score[tok * k + i] : 1.0f;
  }
  for (int i = 1; i < k; ++i) {  // insertion sort by sorted row (k <= 16)
    const int di = d[i];
    const float si = sc[i];
    int j = i - 1;
    while (j >= 0 && d[j] > di) {
      d[j + 1] = d[j];
      sc[j + 1] = sc[j];
      --j;
    }
    d[j + 1] = di;
    sc[j + 1] = si;
  }
  for (int64_t c = (int64_t)threadIdx.x * 8; c < H; c += 256 * 8) {
    float acc[8];
    if (base) {
      const u32x4 b = *reinterpret_cast<const u32x4*>(base + tok * H + c);
      const uint32_t w[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] = bf16lo(w[i]);
        acc[2 * i + 1] = bf16hi(w[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    }
    u32x4 rows[KC ? KC : 1];
    if (KC) {
#pragma unroll
      for (int i = 0; i < KC; ++i) rows[i] = *reinterpret_cast<const u32x4*>(y + (int64_t)d[i] * H + c);
    }
#pragma unroll KC ? KC : 1
    for (int i = 0; i < (KC ? KC : k); ++i) {
      const u32x4 v = KC ? rows[KC ? i : 0] : *reinterpret_cast<const u32x4*>(y + (int64_t)d[i] * H + c);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float lo = bf16lo(w[q]), hi = bf16hi(w[q]);
        if (score) {
          lo = round_bf16(lo * sc[i]);
          hi = round_bf16(hi * sc[i]);
        }
        acc[2 * q] += lo;
        acc[2 * q + 1] += hi;
      }
    }
    *reinterpret_cast<u32x4*>(out + tok * H + c) =
        u32x4{pack_bf16_hw(acc[0], acc[1]), pack_bf16_hw(acc[2], acc[3]), pack_bf16_hw(acc[4], acc[5]), pack_bf16_hw(acc[6], acc[7])};
  }
}

// Per sorted row r (token t = slot_of_dest[r] / k, slot s = slot_of_dest[r]):
//   rows_out[r] = bf16(a[t] * score[s])            (d routed_output of the score multiply, moe.py:899-900 backward)   [optional]
//   dscore[s]   = sum_h a[t][h] * b[r][h]  in f32  (d score of the same multiply)                                      [optional]
__global__ __launch_bounds__(256) void moe_slot_grads_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const int32_t* __restrict__ slot_of_dest,
                                                            int k, const float* __restrict__ score, int64_t rows, int64_t H, bf16_t* __restrict__ rows_out,
                                                            float* __restrict__ dscore) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const int slot = slot_of_dest[r];
  const int64_t tok = slot / k;
  const float sc = score ? score[slot] : 1.0f;
  float dot = 0.f;
  for (int64_t c = (int64_t)threadIdx.x * 8; c < H; c += 256 * 8) {
    const u32x4 va = *reinterpret_cast<const u32x4*>(a + tok * H + c);
    const uint32_t wa[4] = {va.x, va.y, va.z, va.w};
    if (dscore) {
      const u32x4 vb = *reinterpret_cast<const u32x4*>(b + r * H + c);
      const uint32_t wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) dot += bf16lo(wa[i]) * bf16lo(wb[i]) + bf16hi(wa[i]) * bf16hi(wb[i]);
    }
    if (rows_out) {
      uint32_t w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = pack_bf16_hw(bf16lo(wa[i]) * sc, bf16hi(wa[i]) * sc);
      *reinterpret_cast<u32x4*>(rows_out + r * H + c) = u32x4{w[0], w[1], w[2], w[3]};
    }
  }
  if (dscore) {
    dot = block_sum<4>(dot, red);
    if (threadIdx.x == 0) dscore[slot] = dot;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Expert-parallel regrouping map.  After the row all-to-all a rank holds its received rows in (source rank, local expert) order,
// recv_counts[src][le]; the local grouped GEMM wants (local expert, source rank) order.  One block per (le, src) segment:
//   row_map[dst + i] = src_off + i   and   inv_map[src_off + i] = dst + i,   local_offsets = exclusive scan of the per-expert totals.
__global__ __launch_bounds__(256) void moe_ep_row_map_kernel(const int32_t* __restrict__ recv_counts, int ep, int le, int32_t* __restrict__ row_map,
                                                            int32_t* __restrict__ inv_map, int32_t* __restrict__ local_offsets) {
  const int j = blockIdx.x / ep, s = blockIdx.x % ep;  // local expert j, source rank s
  int src_off = 0, dst_off = 0;
  for (int ss = 0; ss < ep; ++ss)
    for (int jj = 0; jj < le; ++jj) {
      const int c = recv_counts[ss * le + jj];
      if (ss < s || (ss == s && jj < j)) src_off += c;           // everything before (s, j) in source-major order
      if (jj < j || (jj == j && ss < s)) dst_off += c;           // everything before (j, s) in expert-major order
    }
  const int len = recv_counts[s * le + j];
  for (int i = threadIdx.x; i < len; i += 256) {
    row_map[dst_off + i] = src_off + i;
    inv_map[src_off + i] = dst_off + i;
  }
  if (s == 0 && threadIdx.x == 0) {
    local_offsets[j] = dst_off;
    if (j == le - 1) {
      int tot = 0;
      for (int q = 0; q < ep * le; ++q) tot += recv_counts[q];
      local_offsets[le] = tot;
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" int pm_moe_router_fwd(const void* gate_logits, int32_t logits_dtype, const float* expert_bias, const int32_t* forced_idx, int64_t n_tokens,
                                 int32_t n_experts, int32_t top_k, int32_t score_func, int32_t route_norm, float route_scale, float* scores,
                                 float* top_scores, int32_t* top_idx, int32_t* counts, float* confidence, void* stream) {
  PM_REQUIRE(gate_logits && scores && top_scores && top_idx && confidence, PM_ERR_INVALID_ARG, "pm_moe_router_fwd: null pointer");
  PM_REQUIRE(n_tokens >= 0 && n_experts > 0 && n_experts <= 64 * kMaxPerLane && top_k > 0 && top_k <= kMaxTopK && top_k <= n_experts, PM_ERR_UNSUPPORTED,
             "pm_moe_router_fwd: n_experts=%d (<= %d) top_k=%d (<= %d)", n_experts, 64 * kMaxPerLane, top_k, kMaxTopK);
  PM_REQUIRE(score_func == 0 || score_func == 1, PM_ERR_INVALID_ARG, "pm_moe_router_fwd: score_func must be 0 (softmax) or 1 (sigmoid)");
  PM_REQUIRE(logits_dtype == PM_DTYPE_BF16 || logits_dtype == PM_DTYPE_F32, PM_ERR_INVALID_ARG, "pm_moe_router_fwd: logits dtype");
  hipStream_t st = as_stream(stream);
  if (counts && hipMemsetAsync(counts, 0, (size_t)n_experts * 4, st) != hipSuccess) return PM_ERR_LAUNCH;
  if (n_tokens == 0) return PM_OK;
  const dim3 grid((unsigned)cdiv(n_tokens, 4));
#define PM_LAUNCH_ROUTER(T_, PER_)                                                                                                                 \
  hipLaunchKernelGGL((moe_router_fwd_kernel<T_, PER_>), grid, dim3(256), 0, st, (const T_*)gate_logits, expert_bias, forced_idx, n_tokens, n_experts, \
                     top_k, score_func, route_norm, route_scale, scores, top_scores, top_idx, counts, confidence)
#define PM_LAUNCH_ROUTER_T(T_)                                 \
  do {                                                        \
    if (n_experts <= 64) PM_LAUNCH_ROUTER(T_, 1);             \
    else if (n_experts <= 128) PM_LAUNCH_ROUTER(T_, 2);       \
    else if (n_experts <= 256) PM_LAUNCH_ROUTER(T_, 4);       \
    else if (n_experts <= 512) PM_LAUNCH_ROUTER(T_, 8);       \
    else PM_LAUNCH_ROUTER(T_, 16);                            \
  } while (0)
  if (logits_dtype == PM_DTYPE_BF16) PM_LAUNCH_ROUTER_T(bf16_t);
  else PM_LAUNCH_ROUTER_T(float);
#undef PM_LAUNCH_ROUTER_T
#undef PM_LAUNCH_ROUTER
  PM_CHECK_LAUNCH("pm_moe_router_fwd");
  return PM_OK;
}

extern "C" int pm_moe_router_bwd(const float* scores, const int32_t* top_idx, const float* grad_top_scores, int64_t n_tokens, int32_t n_experts,
                                 int32_t top_k, int32_t score_func, int32_t route_norm, float route_scale, float* grad_logits, void* stream) {
  PM_REQUIRE(scores && top_idx && grad_top_scores && grad_logits, PM_ERR_INVALID_ARG, "pm_moe_router_bwd: null pointer");
  PM_REQUIRE(n_tokens >= 0 && n_experts > 0 && top_k > 0 && top_k <= kMaxTopK && top_k <= 64, PM_ERR_UNSUPPORTED, "pm_moe_router_bwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  hipLaunchKernelGGL(moe_router_bwd_kernel, dim3((unsigned)cdiv(n_tokens, 4)), dim3(256), 0, as_stream(stream), scores, top_idx, grad_top_scores, n_tokens,
                     n_experts, top_k, score_func, route_norm, route_scale, grad_logits);
  PM_CHECK_LAUNCH("pm_moe_router_bwd");
  return PM_OK;
}

extern "C" int64_t pm_moe_sort_workspace_bytes(int64_t n_slots, int32_t n_experts) {
  return n_slots <= 0 ? 0 : (cdiv(n_slots, kSortChunk) + 1) * n_experts * 4;
}

extern "C" int pm_moe_sort(const int32_t* top_idx, int64_t n_slots, int32_t n_experts, int32_t* dest_of_slot, int32_t* slot_of_dest, int32_t* offsets,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(offsets && (n_slots == 0 || (dest_of_slot && slot_of_dest)), PM_ERR_INVALID_ARG, "pm_moe_sort: null pointer");
  PM_REQUIRE(n_slots >= 0 && n_slots < (1ll << 31) && n_experts > 0 && n_experts <= 8192, PM_ERR_UNSUPPORTED, "pm_moe_sort: bad sizes");
  hipStream_t st = as_stream(stream);
  if (n_slots == 0) {
    if (hipMemsetAsync(offsets, 0, (size_t)(n_experts + 1) * 4, st) != hipSuccess) return PM_ERR_LAUNCH;
    return PM_OK;
  }
  PM_REQUIRE(top_idx, PM_ERR_INVALID_ARG, "pm_moe_sort: null pointer");
  const int64_t need = pm_moe_sort_workspace_bytes(n_slots, n_experts);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_moe_sort: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
  const int nb = (int)cdiv(n_slots, kSortChunk);
  int32_t* bh = (int32_t*)workspace;
  int32_t* totals = bh + (int64_t)nb * n_experts;
  hipLaunchKernelGGL(moe_sort_hist_kernel, dim3(nb), dim3(256), (size_t)n_experts * 4, st, top_idx, n_slots, n_experts, nb, bh);
  hipLaunchKernelGGL(moe_sort_totals_kernel, dim3((unsigned)cdiv(n_experts, 4)), dim3(256), 0, st, bh, nb, n_experts, totals);
  hipLaunchKernelGGL(moe_sort_scan_kernel, dim3((unsigned)cdiv(n_experts, 4)), dim3(256), 0, st, bh, nb, n_experts, totals, offsets);
  hipLaunchKernelGGL(moe_sort_place_kernel, dim3(nb), dim3(256), (size_t)n_experts * 4, st, top_idx, n_slots, n_experts, nb, bh, dest_of_slot, slot_of_dest);
  PM_CHECK_LAUNCH("pm_moe_sort");
  return PM_OK;
}

extern "C" int pm_moe_gather_rows(const void* x_bf16, const int32_t* row_map, int32_t divisor, const float* scale, int64_t n_rows, int64_t row_elems,
                                  void* out_bf16, void* stream) {
  PM_REQUIRE(x_bf16 && row_map && out_bf16, PM_ERR_INVALID_ARG, "pm_moe_gather_rows: null pointer");
  PM_REQUIRE(n_rows >= 0 && row_elems > 0 && row_elems % 8 == 0 && divisor > 0, PM_ERR_INVALID_ARG, "pm_moe_gather_rows: bad sizes");
  if (n_rows == 0) return PM_OK;
  hipLaunchKernelGGL(moe_gather_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, as_stream(stream), (const bf16_t*)x_bf16, row_map, divisor, scale, n_rows,
                     row_elems, (bf16_t*)out_bf16);
  PM_CHECK_LAUNCH("pm_moe_gather_rows");
  return PM_OK;
}

extern "C" int pm_moe_combine(const void* y_bf16, const int32_t* dest_of_slot, const float* scores, const void* base_bf16, int64_t n_tokens, int32_t top_k,
                              int64_t row_elems, void* out_bf16, void* stream) {
  PM_REQUIRE(y_bf16 && dest_of_slot && out_bf16, PM_ERR_INVALID_ARG, "pm_moe_combine: null pointer");
  PM_REQUIRE(n_tokens >= 0 && top_k > 0 && top_k <= kMaxTopK && row_elems > 0 && row_elems % 8 == 0, PM_ERR_INVALID_ARG, "pm_moe_combine: bad sizes");
  if (n_tokens == 0) return PM_OK;
  if (top_k == 8)
    hipLaunchKernelGGL(moe_combine_kernel<8>, dim3((unsigned)n_tokens), dim3(256), 0, as_stream(stream), (const bf16_t*)y_bf16, dest_of_slot, scores,
                       (const bf16_t*)base_bf16, n_tokens, top_k, row_elems, (bf16_t*)out_bf16);
  else
    hipLaunchKernelGGL(moe_combine_kernel<0>, dim3((unsigned)n_tokens), dim3(256), 0, as_stream(stream), (const bf16_t*)y_bf16, dest_of_slot, scores,
                       (const bf16_t*)base_bf16, n_tokens, top_k, row_elems, (bf16_t*)out_bf16);
  PM_CHECK_LAUNCH("pm_moe_combine");
  return PM_OK;
}

extern "C" int pm_moe_slot_grads(const void* a_bf16, const void* b_bf16, const int32_t* slot_of_dest, int32_t top_k, const float* scores, int64_t n_rows,
                                 int64_t row_elems, void* rows_out_bf16, float* dscore, void* stream) {
  PM_REQUIRE(a_bf16 && slot_of_dest && (rows_out_bf16 || dscore) && (!dscore || b_bf16), PM_ERR_INVALID_ARG, "pm_moe_slot_grads: null pointer");
  PM_REQUIRE(n_rows >= 0 && top_k > 0 && row_elems > 0 && row_elems % 8 == 0, PM_ERR_INVALID_ARG, "pm_moe_slot_grads: bad sizes");
  if (n_rows == 0) return PM_OK;
  hipLaunchKernelGGL(moe_slot_grads_kernel, dim3((unsigned)n_rows), dim3(256), 0, as_stream(stream), (const bf16_t*)a_bf16, (const bf16_t*)b_bf16, slot_of_dest,
                     top_k, scores, n_rows, row_elems, (bf16_t*)rows_out_bf16, dscore);
  PM_CHECK_LAUNCH("pm_moe_slot_grads");
  return PM_OK;
}

extern "C" int pm_moe_ep_row_map(const int32_t* recv_counts, int32_t ep_size, int32_t n_local_experts, int32_t* row_map, int32_t* inv_map,
                                 int32_t* local_offsets, void* stream) {
  PM_REQUIRE(recv_counts && row_map && inv_map && local_offsets, PM_ERR_INVALID_ARG, "pm_moe_ep_row_map: null pointer");
  PM_REQUIRE(ep_size > 0 && n_local_experts > 0, PM_ERR_INVALID_ARG, "pm_moe_ep_row_map: bad sizes");
  hipLaunchKernelGGL(moe_ep_row_map_kernel, dim3((unsigned)(ep_size * n_local_experts)), dim3(256), 0, as_stream(stream), recv_counts, ep_size,
                     n_local_experts, row_map, inv_map, local_offsets);
  PM_CHECK_LAUNCH("pm_moe_ep_row_map");
  return PM_OK;
}
