// norm_rope.hip — RMSNorm (fwd/bwd, optional fused residual add), RoPE table, fused per-head q/k RMSNorm + RoPE
// (fwd/bwd), and token->document-start map.  All HBM-bound; 16-byte (8 x bf16) accesses per lane (guide G13).
//
// Reference arithmetic (src/prime_rl/trainer/models/layers/): RMSNorm.forward (norms.py:36-44),
// RotaryEmbedding.forward (rotary_emb.py:60-73), apply_rotary_pos_emb / rotate_half (rotary_emb.py:76-118),
// FlashAttention.attn_projections (attn.py:123-160); get_cu_seqlens_from_seq_lens (utils/sequence.py:38-52).
// bf16 rounding points follow the reference's eager bf16 ops so forward values match it bit-for-bit up to the
// reduction order of the sum of squares.
#include "common.hip.h"

namespace pm {

__device__ __forceinline__ void unpack8(const u32x4& v, float* o) {
  o[0] = bf16lo(v.x); o[1] = bf16hi(v.x); o[2] = bf16lo(v.y); o[3] = bf16hi(v.y);
  o[4] = bf16lo(v.z); o[5] = bf16hi(v.z); o[6] = bf16lo(v.w); o[7] = bf16hi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* o) {
  u32x4 v;
  v.x = pack_bf16_hw(o[0], o[1]); v.y = pack_bf16_hw(o[2], o[3]); v.z = pack_bf16_hw(o[4], o[5]); v.w = pack_bf16_hw(o[6], o[7]);  // v_cvt_pk_bf16_f32 (RNE)
  return v;
}
// round 8 f32 values to the nearest bf16 in place (a bf16 torch op's output rounding): 4 x v_cvt_pk_bf16_f32 + 8 bit ops, where the
// integer emulation costs ~6 VALU ops per value — the q/k-norm + RoPE forward has four such rounding points per element and was VALU-bound
__device__ __forceinline__ void round8_bf16(float* a) {
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const uint32_t w = pack_bf16_hw(a[k], a[k + 1]);
    a[k] = bf16lo(w);
    a[k + 1] = bf16hi(w);
  }
}

// ================================= RMSNorm =====================================================
// one wave64 per row; lane l owns 8-element vectors {v*64 + l}; MAXV*512 >= hidden.
template <int MAXV, bool HAS_RES>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                         const bf16_t* __restrict__ w, int64_t n_rows, int H, float eps,
                                                         bf16_t* __restrict__ y, bf16_t* __restrict__ sum_out, float* __restrict__ rstd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= n_rows) return;
  const int nvec = H >> 3;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * H);
  const u32x4* rr = HAS_RES ? reinterpret_cast<const u32x4*>(res + row * H) : nullptr;
  u32x4* sr = HAS_RES ? reinterpret_cast<u32x4*>(sum_out + row * H) : nullptr;
  float xf[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int idx = v * 64 + lane;
    if (idx < nvec) {
      unpack8(xr[idx], xf[v]);
      if (HAS_RES) {
        float rf[8];
        unpack8(rr[idx], rf);
#pragma unroll
        for (int k = 0; k < 8; ++k) xf[v][k] = xf[v][k] + rf[k];
        round8_bf16(xf[v]);  // bf16 residual add (modeling_qwen3.py:80,84)
        sr[idx] = pack8(xf[v]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) ss = fmaf(xf[v][k], xf[v][k], ss);
    }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)H + eps);
  if (lane == 0) rstd[row] = r;
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  u32x4* yr = reinterpret_cast<u32x4*>(y + row * H);
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int idx = v * 64 + lane;
    if (idx < nvec) {
      float wf[8], o[8];
      unpack8(wr[idx], wf);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = xf[v][k] * r;
      round8_bf16(o);  // hidden.to(bf16)
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = wf[k] * o[k];  // weight * hidden; pack8 rounds the product
      yr[idx] = pack8(o);
    }
  }
}

// backward.  A row is owned by a TEAM of two waves (128 lanes, lane l owns 8-element vectors {v*128 + l}), a workgroup holds two
// teams: at H = 4096 that is 4 vectors per lane, ~100 VGPRs, 4 workgroups (16 waves) resident per CU, and every wave has its whole
// x / dy / dresidual row slice in flight before the first use — memory-level parallelism comes from occupancy, nothing spills.
// The row's dot product crosses the team through two LDS floats (double-buffered: one barrier per row).  dweight partials stay in
// registers over the team's rows; at the end team 1 hands its partials to team 0 through LDS: ONE partial row per workgroup.
template <int MAXV, bool HAS_DRES>
__global__ __launch_bounds__(256, (MAXV <= 4 ? 4 : 2)) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x,
                                                                              const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                                              const bf16_t* __restrict__ dres, int64_t n_rows, int H,
                                                                              int64_t rows_per_wg, bf16_t* __restrict__ gx,
                                                                              float* __restrict__ dw_partials) {
  extern __shared__ float dw_lds[];  // [H] fold buffer, [2 buffers][2 teams][2 waves] dot partials, [H] bf16 weights
  float* red = dw_lds + H;
  const int team = threadIdx.x >> 7, tl = threadIdx.x & 127, wave_in_team = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
  const int nvec = H >> 3;
  const int64_t first = (int64_t)blockIdx.x * rows_per_wg;
  int64_t last = first + rows_per_wg;
  if (last > n_rows) last = n_rows;
  float dw[MAXV][8];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
#pragma unroll
    for (int k = 0; k < 8; ++k) dw[v][k] = 0.f;
  }
  u32x4* wl = reinterpret_cast<u32x4*>(red + 8);  // the weight row lives in LDS (read twice per row by every team), not in VGPRs
  for (int idx = threadIdx.x; idx < nvec; idx += 256) wl[idx] = reinterpret_cast<const u32x4*>(w)[idx];
  __syncthreads();
  const float inv_h = 1.0f / (float)H;
  int it = 0;
  for (int64_t base = first; base < last; base += 2, ++it) {  // uniform trip count over the workgroup (barrier inside)
    const int64_t row = base + team;
    const bool active = row < last;
    u32x4 xp[MAXV], gp[MAXV], dp[MAXV];
    float r = 0.f;
    if (active) {
      const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * H);
      const u32x4* gr = reinterpret_cast<const u32x4*>(gy + row * H);
      const u32x4* dr = HAS_DRES ? reinterpret_cast<const u32x4*>(dres + row * H) : nullptr;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int idx = v * 128 + tl;
        if (idx < nvec) {
          xp[v] = xr[idx];
          gp[v] = gr[idx];
          if (HAS_DRES) dp[v] = dr[idx];
        }
      }
      r = rstd[row];
    }
    float dot = 0.f;
    if (active) {
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        if (v * 128 + tl < nvec) {
          float xf[8], gf[8], wf[8];
          unpack8(xp[v], xf);
          unpack8(gp[v], gf);
          unpack8(wl[v * 128 + tl], wf);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float xh = xf[k] * r;
            dw[v][k] = fmaf(gf[k], xh, dw[v][k]);
            dot = fmaf(gf[k] * wf[k], xh, dot);
          }
        }
      }
    }
    dot = wave_sum(dot);
    float* slot = red + (it & 1) * 4 + team * 2;
    if (lane == 0) slot[wave_in_team] = dot;
    __syncthreads();
    if (active) {
      dot = (slot[0] + slot[1]) * inv_h;
      u32x4* ox = reinterpret_cast<u32x4*>(gx + row * H);
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int idx = v * 128 + tl;
        if (idx < nvec) {
          float xf[8], gf[8], wf[8], o[8];
          unpack8(xp[v], xf);
          unpack8(gp[v], gf);
          unpack8(wl[v * 128 + tl], wf);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = r * (gf[k] * wf[k] - (xf[k] * r) * dot);
          if (HAS_DRES) {
            float df[8];
            unpack8(dp[v], df);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += df[k];
          }
          ox[idx] = pack8(o);
        }
      }
    }
  }
  // fold: team 1 -> LDS, team 0 adds (fixed order) and writes the workgroup's partial row
  if (team == 1) {
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int idx = v * 128 + tl;
      if (idx < nvec) {
        float4* l4 = reinterpret_cast<float4*>(dw_lds + idx * 8);
        l4[0] = make_float4(dw[v][0], dw[v][1], dw[v][2], dw[v][3]);
        l4[1] = make_float4(dw[v][4], dw[v][5], dw[v][6], dw[v][7]);
      }
    }
  }
  __syncthreads();
  if (team == 0) {
    float* pw = dw_partials + (int64_t)blockIdx.x * H;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int idx = v * 128 + tl;
      if (idx < nvec) {
        const float4* l4 = reinterpret_cast<const float4*>(dw_lds + idx * 8);
        const float4 a = l4[0], b = l4[1];
        float4* p4 = reinterpret_cast<float4*>(pw + (int64_t)idx * 8);
        p4[0] = make_float4(dw[v][0] + a.x, dw[v][1] + a.y, dw[v][2] + a.z, dw[v][3] + a.w);
        p4[1] = make_float4(dw[v][4] + b.x, dw[v][5] + b.y, dw[v][6] + b.z, dw[v][7] + b.w);
      }
    }
  }
}

// Deterministic column sums of the per-workgroup partial rows, in two small launches so the reduction is spread over the chip:
// stage 1: slice s of kReduceSlices sums partial rows {s, s + kReduceSlices, ...} into stage[s][c]; stage 2: out[c] += sum_s stage[s][c].
constexpr int kReduceSlices = 32;

__global__ void column_reduce_stage1_kernel(const float* __restrict__ partials, int64_t n_partials, int64_t ncols, float* __restrict__ stage) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  float s0 = 0.f, s1 = 0.f;
  int64_t p = blockIdx.y;
  for (; p + kReduceSlices < n_partials; p += 2 * kReduceSlices) {
    s0 += partials[p * ncols + c];
    s1 += partials[(p + kReduceSlices) * ncols + c];
  }
  if (p < n_partials) s0 += partials[p * ncols + c];
  stage[(int64_t)blockIdx.y * ncols + c] = s0 + s1;
}

__global__ void column_reduce_stage2_kernel(const float* __restrict__ stage, int64_t ncols, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int s = 0; s < kReduceSlices; s += 4) {
    s0 += stage[(int64_t)(s + 0) * ncols + c];
    s1 += stage[(int64_t)(s + 1) * ncols + c];
    s2 += stage[(int64_t)(s + 2) * ncols + c];
    s3 += stage[(int64_t)(s + 3) * ncols + c];
  }
  out[c] += (s0 + s1) + (s2 + s3);
}

// Column sums of a bf16 matrix (the bias gradient of a linear layer: grad_bias = sum over tokens of grad_out, nn.Linear(bias=True) in
// models/layers/attn.py:64-72).  Row slice blockIdx.y of kColSumSlices sums its rows in f32 (8 columns per lane, 16-byte loads) into one
// partial row; the two column_reduce stages above add the partial rows into `out` in a fixed order.
constexpr int kColSumSlices = 256;

__global__ __launch_bounds__(256) void column_sum_bf16_kernel(const bf16_t* __restrict__ x, int64_t n_rows, int64_t n_cols, int64_t ld,
                                                             float* __restrict__ partials) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= n_cols) return;
  const int64_t per = (n_rows + kColSumSlices - 1) / kColSumSlices;
  const int64_t r0 = (int64_t)blockIdx.y * per;
  const int64_t r1 = r0 + per < n_rows ? r0 + per : n_rows;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  int64_t r = r0;
  for (; r + 4 <= r1; r += 4) {
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(x + (r + j) * ld + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f[8];
      unpack8(v[j], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f[k];
    }
  }
  for (; r < r1; ++r) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + r * ld + c), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += f[k];
  }
  float4* dst = reinterpret_cast<float4*>(partials + (int64_t)blockIdx.y * n_cols + c);
  dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// 4 workgroups (16 waves) per CU keep enough row loads in flight for HBM; 1024 = 4 x 256 CUs, so the grid is one full residency
static int64_t rms_bwd_num_wg(int64_t n_rows) {
  int64_t g = cdiv(n_rows, 8);
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  return g;
}

// ================================= RoPE table ==================================================
// angle = inv_freq[j] * pos in f32 (rotary_emb.py:66: a K=1 matmul, i.e. the exact product), cos/sin rounded to bf16 (:73).
__global__ void rope_table_kernel(const int64_t* __restrict__ pos, int64_t n_tokens, int half, const float* __restrict__ inv_freq,
                                  bf16_t* __restrict__ cosb, bf16_t* __restrict__ sinb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tokens * half) return;
  const int64_t t = i / half;
  const int j = (int)(i - t * half);
  const float ang = inv_freq[j] * (float)pos[t];
  cosb[i] = f32_to_bf16(cosf(ang));
  sinb[i] = f32_to_bf16(sinf(ang));
}

// ================================= q/k norm + RoPE =============================================
// 16 lanes per TOKEN, 8 contiguous elements per lane (head_dim = 128): the group walks the token's q and k heads in batches of
// kQkBatch with every load of a batch issued before its first use; cos / sin (64 values per token) and the two norm weights are
// loaded once per token instead of once per head, and nothing divides by the head count.  lanes 0-7 of a group hold elements
// [0,64), lanes 8-15 hold [64,128): the rotate_half partner is lane ^ 8 = DPP row_ror:8; row sums are DPP too (VALU, no LDS).
constexpr int kQkBatch = 4;

// NORM = false is the Llama / Qwen2 attention block (AttentionConfig.use_qk_norm = False, models/llama/modeling_llama.py:59-60): RoPE only, on the
// projection output as it is (norm weights, rstd unused).
template <bool NORM>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qw,
                                                             const bf16_t* __restrict__ kw, const bf16_t* __restrict__ cosb,
                                                             const bf16_t* __restrict__ sinb, int64_t n_tokens, int hq, int hkv, float eps,
                                                             bf16_t* __restrict__ qo, bf16_t* __restrict__ ko, float* __restrict__ rstd) {
  const int nh = hq + hkv;
  const int64_t tok = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (tok >= n_tokens) return;  // a whole 16-lane row leaves together; the DPP row ops never cross rows
  const int64_t row_stride = (int64_t)(hq + 2 * hkv) * 128;
  float wq[8], wk[8], cf[8], sf[8];
  if constexpr (NORM) {
    unpack8(*reinterpret_cast<const u32x4*>(qw + sub * 8), wq);
    unpack8(*reinterpret_cast<const u32x4*>(kw + sub * 8), wk);
  }
  const int cs = (sub & 7) * 8;
  unpack8(*reinterpret_cast<const u32x4*>(cosb + tok * 64 + cs), cf);
  unpack8(*reinterpret_cast<const u32x4*>(sinb + tok * 64 + cs), sf);
  const float sgn = (sub < 8) ? -1.0f : 1.0f;  // rotate_half: first half gets -x2, second half gets +x1
  const bf16_t* row = qkv + tok * row_stride + sub * 8;
  for (int h0 = 0; h0 < nh; h0 += kQkBatch) {
    u32x4 xv[kQkBatch];
#pragma unroll
    for (int j = 0; j < kQkBatch; ++j) xv[j] = *reinterpret_cast<const u32x4*>(row + (int64_t)(h0 + j < nh ? h0 + j : nh - 1) * 128);
#pragma unroll
    for (int j = 0; j < kQkBatch; ++j) {
      const int h = h0 + j;
      if (h >= nh) break;
      float xf[8];
      unpack8(xv[j], xf);
      float yv[8], o[8];
      float r = 0.f;
      if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) ss = fmaf(xf[k], xf[k], ss);
        ss = row16_sum(ss);
        r = rsqrtf(ss / 128.0f + eps);
        const float* wf = h < hq ? wq : wk;
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[k] = xf[k] * r;
        round8_bf16(yv);
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[k] = wf[k] * yv[k];
        round8_bf16(yv);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[k] = xf[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o[k] = yv[k] * cf[k];
        xf[k] = (sgn * dpp_f32<0x128>(yv[k])) * sf[k];
      }
      round8_bf16(o);
      round8_bf16(xf);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += xf[k];
      bf16_t* dst = (h < hq) ? (qo + (tok * hq + h) * 128 + sub * 8) : (ko + (tok * hkv + (h - hq)) * 128 + sub * 8);
      *reinterpret_cast<u32x4*>(dst) = pack8(o);
      if constexpr (NORM) {
        if (sub == 0) rstd[tok * nh + h] = r;
      }
    }
  }
}

template <bool NORM>
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const bf16_t* __restrict__ gq, const bf16_t* __restrict__ gk,
                                                             const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qw,
                                                             const bf16_t* __restrict__ kw, const bf16_t* __restrict__ cosb,
                                                             const bf16_t* __restrict__ sinb, const float* __restrict__ rstd,
                                                             int64_t n_tokens, int hq, int hkv, bf16_t* __restrict__ gqkv,
                                                             float* __restrict__ dw_partials /* [grid][2][128] */) {
  const int nh = hq + hkv;
  const int sub = threadIdx.x & 15;
  const int64_t row_stride = (int64_t)(hq + 2 * hkv) * 128;
  float dwq[8], dwk[8], wq[8], wk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) dwq[k] = dwk[k] = 0.f;
  if constexpr (NORM) {
    unpack8(*reinterpret_cast<const u32x4*>(qw + sub * 8), wq);
    unpack8(*reinterpret_cast<const u32x4*>(kw + sub * 8), wk);
  }
  const int cs = (sub & 7) * 8;
  const float sgn = (sub < 8) ? 1.0f : -1.0f;
  for (int64_t tok = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); tok < n_tokens; tok += (int64_t)gridDim.x * 16) {
    float cf[8], sf[8];
    unpack8(*reinterpret_cast<const u32x4*>(cosb + tok * 64 + cs), cf);
    unpack8(*reinterpret_cast<const u32x4*>(sinb + tok * 64 + cs), sf);
    const bf16_t* xrow = qkv + tok * row_stride + sub * 8;
    bf16_t* orow = gqkv + tok * row_stride + sub * 8;
    for (int h0 = 0; h0 < nh; h0 += kQkBatch) {
      u32x4 xv[kQkBatch], gv[kQkBatch];
      float rv[kQkBatch];
#pragma unroll
      for (int j = 0; j < kQkBatch; ++j) {
        const int h = h0 + j < nh ? h0 + j : nh - 1;
        gv[j] = *reinterpret_cast<const u32x4*>(h < hq ? (gq + (tok * hq + h) * 128 + sub * 8) : (gk + (tok * hkv + (h - hq)) * 128 + sub * 8));
        if constexpr (NORM) {
          xv[j] = *reinterpret_cast<const u32x4*>(xrow + (int64_t)h * 128);
          rv[j] = rstd[tok * nh + h];
        }
      }
#pragma unroll
      for (int j = 0; j < kQkBatch; ++j) {
        const int h = h0 + j;
        if (h >= nh) break;
        const bool is_q = h < hq;
        float g[8], xf[8], dn[8], nh_[8];
        unpack8(gv[j], g);
        if constexpr (!NORM) {  // transpose of the rotation alone
          float o[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = g[k] * cf[k] + (sgn * dpp_f32<0x128>(g[k])) * sf[k];
          *reinterpret_cast<u32x4*>(orow + (int64_t)h * 128) = pack8(o);
          continue;
        }
        unpack8(xv[j], xf);
        const float r = rv[j];
        const float* wv = is_q ? wq : wk;
        // transpose of out = y*cos + rotate_half(y)*sin :  dy[j] = g[j]*cos[j] + (j<64 ? g[j+64] : -g[j-64]) * sin[j mod 64]
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dy = g[k] * cf[k] + (sgn * dpp_f32<0x128>(g[k])) * sf[k];
          nh_[k] = xf[k] * r;
          if (is_q) dwq[k] = fmaf(dy, nh_[k], dwq[k]);
          else dwk[k] = fmaf(dy, nh_[k], dwk[k]);
          dn[k] = dy * wv[k];
          dot = fmaf(dn[k], nh_[k], dot);
        }
        dot = row16_sum(dot) / 128.0f;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = r * (dn[k] - nh_[k] * dot);
        *reinterpret_cast<u32x4*>(orow + (int64_t)h * 128) = pack8(o);
      }
    }
  }
  if constexpr (!NORM) return;
  // the 16 groups of the block through LDS (fixed order): one partial row [q 128 | k 128] per block
  __shared__ float red[16][256];
  const int grp = threadIdx.x >> 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[grp][sub * 8 + k] = dwq[k];
    red[grp][128 + sub * 8 + k] = dwk[k];
  }
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int gI = 0; gI < 16; gI += 2) {
    s0 += red[gI][threadIdx.x];
    s1 += red[gI + 1][threadIdx.x];
  }
  dw_partials[(int64_t)blockIdx.x * 256 + threadIdx.x] = s0 + s1;
}

__global__ void qk_dw_reduce_kernel(const float* __restrict__ partials, int64_t n_partials, float* __restrict__ gqw, float* __restrict__ gkw) {
  const int c = threadIdx.x;  // 256 threads: [0,128) q, [128,256) k; coalesced rows, 4 independent chains
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t p = 0;
  for (; p + 3 < n_partials; p += 4) {
    s0 += partials[(p + 0) * 256 + c];
    s1 += partials[(p + 1) * 256 + c];
    s2 += partials[(p + 2) * 256 + c];
    s3 += partials[(p + 3) * 256 + c];
  }
  for (; p < n_partials; ++p) s0 += partials[p * 256 + c];
  const float s = (s0 + s1) + (s2 + s3);
  if (c < 128) gqw[c] += s;
  else gkw[c - 128] += s;
}

static int64_t qk_bwd_grid(int64_t n_tokens, int /*nh*/) {
  int64_t g = cdiv(n_tokens, 16);  // one 16-lane group per token, 16 tokens per workgroup per sweep
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return g;
}

// ================================= token -> document start =====================================
__global__ void seq_start_kernel(const int32_t* __restrict__ cu, int n_seqs, int64_t n_tokens, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tokens) return;
  int lo = 0, hi = n_seqs;  // largest s in [0, n_seqs] with cu[s] <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int64_t)cu[mid] <= i) lo = mid;
    else hi = mid - 1;
  }
  out[i] = cu[lo];
}

}  // namespace pm

using namespace pm;

template <bool HAS_RES>
static int rms_fwd_launch(const bf16_t* x, const bf16_t* res, const bf16_t* w, int64_t n_rows, int H, float eps, bf16_t* y, bf16_t* so,
                          float* rstd, hipStream_t st) {
  const dim3 grid((unsigned)cdiv(n_rows, 4)), block(256);
  const int nv = (int)cdiv(H, 512);
  if (nv <= 2) hipLaunchKernelGGL((rmsnorm_fwd_kernel<2, HAS_RES>), grid, block, 0, st, x, res, w, n_rows, H, eps, y, so, rstd);
  else if (nv <= 4) hipLaunchKernelGGL((rmsnorm_fwd_kernel<4, HAS_RES>), grid, block, 0, st, x, res, w, n_rows, H, eps, y, so, rstd);
  else if (nv <= 8) hipLaunchKernelGGL((rmsnorm_fwd_kernel<8, HAS_RES>), grid, block, 0, st, x, res, w, n_rows, H, eps, y, so, rstd);
  else hipLaunchKernelGGL((rmsnorm_fwd_kernel<16, HAS_RES>), grid, block, 0, st, x, res, w, n_rows, H, eps, y, so, rstd);
  PM_CHECK_LAUNCH("pm_rmsnorm_fwd");
  return PM_OK;
}

extern "C" int pm_rmsnorm_fwd(const void* x, const void* residual, const void* weight, int64_t n_rows, int64_t hidden, float eps, void* y,
                              void* sum_out, float* rstd, void* stream) {
  PM_REQUIRE(x && weight && y && rstd, PM_ERR_INVALID_ARG, "pm_rmsnorm_fwd: null pointer");
  PM_REQUIRE(!(residual && !sum_out), PM_ERR_INVALID_ARG, "pm_rmsnorm_fwd: residual given without sum_out");
  PM_REQUIRE(n_rows >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 8192, PM_ERR_UNSUPPORTED,
             "pm_rmsnorm_fwd: hidden=%lld must be a multiple of 8 and <= 8192", (long long)hidden);
  if (n_rows == 0) return PM_OK;
  if (residual)
    return rms_fwd_launch<true>((const bf16_t*)x, (const bf16_t*)residual, (const bf16_t*)weight, n_rows, (int)hidden, eps, (bf16_t*)y,
                                (bf16_t*)sum_out, rstd, as_stream(stream));
  return rms_fwd_launch<false>((const bf16_t*)x, nullptr, (const bf16_t*)weight, n_rows, (int)hidden, eps, (bf16_t*)y, nullptr, rstd,
                               as_stream(stream));
}

extern "C" int64_t pm_rmsnorm_bwd_workspace_bytes(int64_t n_rows, int64_t hidden) {
  if (n_rows <= 0 || hidden <= 0) return 0;
  return (rms_bwd_num_wg(n_rows) + kReduceSlices) * hidden * (int64_t)sizeof(float);  // per-workgroup partial rows + the stage-1 rows
}

template <bool HAS_DRES>
static int rms_bwd_launch(const bf16_t* gy, const bf16_t* x, const bf16_t* w, const float* rstd, const bf16_t* dres, int64_t n_rows, int H,
                          bf16_t* gx, float* partials, int64_t num_wg, hipStream_t st) {
  const int64_t rows_per_wg = cdiv(n_rows, num_wg);
  const dim3 grid((unsigned)num_wg), block(256);
  const int nv = (int)cdiv(H, 1024);  // 8-element vectors per lane of a 128-lane team
  const size_t lds = (size_t)(H + 8) * sizeof(float) + (size_t)H * sizeof(bf16_t);
  if (nv <= 1) hipLaunchKernelGGL((rmsnorm_bwd_kernel<1, HAS_DRES>), grid, block, lds, st, gy, x, w, rstd, dres, n_rows, H, rows_per_wg, gx, partials);
  else if (nv <= 2) hipLaunchKernelGGL((rmsnorm_bwd_kernel<2, HAS_DRES>), grid, block, lds, st, gy, x, w, rstd, dres, n_rows, H, rows_per_wg, gx, partials);
  else if (nv <= 4) hipLaunchKernelGGL((rmsnorm_bwd_kernel<4, HAS_DRES>), grid, block, lds, st, gy, x, w, rstd, dres, n_rows, H, rows_per_wg, gx, partials);
  else hipLaunchKernelGGL((rmsnorm_bwd_kernel<8, HAS_DRES>), grid, block, lds, st, gy, x, w, rstd, dres, n_rows, H, rows_per_wg, gx, partials);
  PM_CHECK_LAUNCH("pm_rmsnorm_bwd");
  return PM_OK;
}

extern "C" int pm_rmsnorm_bwd(const void* grad_y, const void* x, const void* weight, const float* rstd, const void* dresidual, int64_t n_rows,
                              int64_t hidden, void* grad_x, float* grad_weight, void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(grad_y && x && weight && rstd && grad_x && grad_weight, PM_ERR_INVALID_ARG, "pm_rmsnorm_bwd: null pointer");
  PM_REQUIRE(n_rows >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 8192, PM_ERR_UNSUPPORTED, "pm_rmsnorm_bwd: hidden=%lld unsupported",
             (long long)hidden);
  if (n_rows == 0) return PM_OK;
  const int64_t need = pm_rmsnorm_bwd_workspace_bytes(n_rows, hidden);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_rmsnorm_bwd: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  const int64_t num_wg = rms_bwd_num_wg(n_rows);
  int rc;
  if (dresidual)
    rc = rms_bwd_launch<true>((const bf16_t*)grad_y, (const bf16_t*)x, (const bf16_t*)weight, rstd, (const bf16_t*)dresidual, n_rows, (int)hidden,
                              (bf16_t*)grad_x, (float*)workspace, num_wg, st);
  else
    rc = rms_bwd_launch<false>((const bf16_t*)grad_y, (const bf16_t*)x, (const bf16_t*)weight, rstd, nullptr, n_rows, (int)hidden,
                               (bf16_t*)grad_x, (float*)workspace, num_wg, st);
  if (rc != PM_OK) return rc;
  float* stage = (float*)workspace + num_wg * hidden;
  hipLaunchKernelGGL(column_reduce_stage1_kernel, dim3((unsigned)cdiv(hidden, 256), kReduceSlices), dim3(256), 0, st, (const float*)workspace, num_wg,
                     hidden, stage);
  hipLaunchKernelGGL(column_reduce_stage2_kernel, dim3((unsigned)cdiv(hidden, 256)), dim3(256), 0, st, (const float*)stage, hidden, grad_weight);
  PM_CHECK_LAUNCH("pm_rmsnorm_bwd(reduce)");
  return PM_OK;
}

extern "C" int64_t pm_column_sum_workspace_bytes(int64_t n_cols) {
  return n_cols <= 0 ? 0 : (int64_t)(kColSumSlices + kReduceSlices) * n_cols * (int64_t)sizeof(float);
}

extern "C" int pm_column_sum_bf16(const void* x, int64_t n_rows, int64_t n_cols, int64_t ld, float* out_accumulate, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(n_rows >= 0 && n_cols > 0 && n_cols % 8 == 0 && ld >= n_cols && ld % 8 == 0, PM_ERR_UNSUPPORTED,
             "pm_column_sum_bf16: n_cols and ld must be multiples of 8 (n_cols=%lld ld=%lld)", (long long)n_cols, (long long)ld);
  if (n_rows == 0) return PM_OK;
  PM_REQUIRE(x && out_accumulate, PM_ERR_INVALID_ARG, "pm_column_sum_bf16: null pointer");
  PM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, PM_ERR_INVALID_ARG, "pm_column_sum_bf16: x must be 16-byte aligned");
  const int64_t need = pm_column_sum_workspace_bytes(n_cols);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_column_sum_bf16: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  float* partials = (float*)workspace;
  float* stage = partials + (int64_t)kColSumSlices * n_cols;
  hipLaunchKernelGGL(column_sum_bf16_kernel, dim3((unsigned)cdiv(n_cols, (int64_t)2048), kColSumSlices), dim3(256), 0, st, (const bf16_t*)x, n_rows,
                     n_cols, ld, partials);
  hipLaunchKernelGGL(column_reduce_stage1_kernel, dim3((unsigned)cdiv(n_cols, (int64_t)256), kReduceSlices), dim3(256), 0, st, (const float*)partials,
                     (int64_t)kColSumSlices, n_cols, stage);
  hipLaunchKernelGGL(column_reduce_stage2_kernel, dim3((unsigned)cdiv(n_cols, (int64_t)256)), dim3(256), 0, st, (const float*)stage, n_cols,
                     out_accumulate);
  PM_CHECK_LAUNCH("pm_column_sum_bf16");
  return PM_OK;
}

extern "C" int pm_rope_table(const int64_t* position_ids, int64_t n_tokens, const float* inv_freq, int32_t half_dim, void* cos_bf16,
                             void* sin_bf16, void* stream) {
  PM_REQUIRE(position_ids && inv_freq && cos_bf16 && sin_bf16, PM_ERR_INVALID_ARG, "pm_rope_table: null pointer");
  PM_REQUIRE(n_tokens >= 0 && half_dim > 0, PM_ERR_INVALID_ARG, "pm_rope_table: bad sizes");
  if (n_tokens == 0) return PM_OK;
  const int half = half_dim;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)cdiv(n_tokens * half, 256)), dim3(256), 0, as_stream(stream), position_ids, n_tokens, half,
                     inv_freq, (bf16_t*)cos_bf16, (bf16_t*)sin_bf16);
  PM_CHECK_LAUNCH("pm_rope_table");
  return PM_OK;
}

extern "C" int pm_qknorm_rope_fwd(const void* qkv, const void* q_norm_w, const void* k_norm_w, const void* cosb, const void* sinb,
                                  int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float eps, void* q_out, void* k_out,
                                  float* rstd, void* stream) {
  const bool norm = q_norm_w || k_norm_w;  // both NULL: RoPE only (use_qk_norm = False); rstd is then not written and may be NULL
  PM_REQUIRE(qkv && cosb && sinb && q_out && k_out, PM_ERR_INVALID_ARG, "pm_qknorm_rope_fwd: null pointer");
  PM_REQUIRE(!norm || (q_norm_w && k_norm_w && rstd), PM_ERR_INVALID_ARG, "pm_qknorm_rope_fwd: q_norm_w, k_norm_w and rstd go together");
  PM_REQUIRE(head_dim == 128, PM_ERR_UNSUPPORTED, "pm_qknorm_rope_fwd: head_dim=%d (only 128 is built)", head_dim);
  PM_REQUIRE(n_tokens >= 0 && n_q_heads > 0 && n_kv_heads > 0, PM_ERR_INVALID_ARG, "pm_qknorm_rope_fwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  const dim3 grid((unsigned)cdiv(n_tokens, (int64_t)16));
  if (norm)
    hipLaunchKernelGGL(qknorm_rope_fwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)qkv, (const bf16_t*)q_norm_w,
                       (const bf16_t*)k_norm_w, (const bf16_t*)cosb, (const bf16_t*)sinb, n_tokens, n_q_heads, n_kv_heads, eps, (bf16_t*)q_out,
                       (bf16_t*)k_out, rstd);
  else
    hipLaunchKernelGGL(qknorm_rope_fwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)qkv, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, (const bf16_t*)cosb, (const bf16_t*)sinb, n_tokens, n_q_heads, n_kv_heads, eps, (bf16_t*)q_out,
                       (bf16_t*)k_out, (float*)nullptr);
  PM_CHECK_LAUNCH("pm_qknorm_rope_fwd");
  return PM_OK;
}

extern "C" int64_t pm_qknorm_rope_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads) {
  if (n_tokens <= 0) return 0;
  return (qk_bwd_grid(n_tokens, n_q_heads + n_kv_heads) + kReduceSlices) * 256 * (int64_t)sizeof(float);  // partial rows + stage-1 rows
}

extern "C" int pm_qknorm_rope_bwd(const void* grad_q, const void* grad_k, const void* qkv, const void* q_norm_w, const void* k_norm_w,
                                  const void* cosb, const void* sinb, const float* rstd, int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads,
                                  int32_t head_dim, void* grad_qkv, float* grad_q_norm_w, float* grad_k_norm_w, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  const bool norm = q_norm_w || k_norm_w;  // both NULL: transpose of the rotation only; qkv, rstd, the weight gradients and the workspace are unused
  PM_REQUIRE(grad_q && grad_k && cosb && sinb && grad_qkv, PM_ERR_INVALID_ARG, "pm_qknorm_rope_bwd: null pointer");
  PM_REQUIRE(!norm || (qkv && q_norm_w && k_norm_w && rstd && grad_q_norm_w && grad_k_norm_w), PM_ERR_INVALID_ARG,
             "pm_qknorm_rope_bwd: null pointer (the norm weights, rstd, qkv and the weight gradients go together)");
  PM_REQUIRE(head_dim == 128, PM_ERR_UNSUPPORTED, "pm_qknorm_rope_bwd: head_dim=%d (only 128 is built)", head_dim);
  if (n_tokens == 0) return PM_OK;
  if (!norm) {
    hipLaunchKernelGGL(qknorm_rope_bwd_kernel<false>, dim3((unsigned)qk_bwd_grid(n_tokens, n_q_heads + n_kv_heads)), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)grad_q, (const bf16_t*)grad_k, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                       (const bf16_t*)cosb, (const bf16_t*)sinb, (const float*)nullptr, n_tokens, n_q_heads, n_kv_heads, (bf16_t*)grad_qkv,
                       (float*)nullptr);
    PM_CHECK_LAUNCH("pm_qknorm_rope_bwd(rope only)");
    return PM_OK;
  }
  const int64_t need = pm_qknorm_rope_bwd_workspace_bytes(n_tokens, n_q_heads, n_kv_heads);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_qknorm_rope_bwd: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  const int64_t grid = qk_bwd_grid(n_tokens, n_q_heads + n_kv_heads);
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, (const bf16_t*)grad_q, (const bf16_t*)grad_k,
                     (const bf16_t*)qkv, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, (const bf16_t*)cosb, (const bf16_t*)sinb, rstd, n_tokens,
                     n_q_heads, n_kv_heads, (bf16_t*)grad_qkv, (float*)workspace);
  PM_CHECK_LAUNCH("pm_qknorm_rope_bwd");
  // [grid][256] partial rows -> 32 slice sums spread over the chip -> one 256-thread block adds them into the two weight gradients
  float* stage = (float*)workspace + grid * 256;
  hipLaunchKernelGGL(column_reduce_stage1_kernel, dim3(1, kReduceSlices), dim3(256), 0, st, (const float*)workspace, grid, (int64_t)256, stage);
  hipLaunchKernelGGL(qk_dw_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)stage, (int64_t)kReduceSlices, grad_q_norm_w, grad_k_norm_w);
  PM_CHECK_LAUNCH("pm_qknorm_rope_bwd(reduce)");
  return PM_OK;
}

extern "C" int pm_seq_start_from_cu(const int32_t* cu_seqlens, int32_t n_seqs, int64_t n_tokens, int32_t* seq_start, void* stream) {
  PM_REQUIRE(cu_seqlens && seq_start && n_seqs > 0 && n_tokens >= 0, PM_ERR_INVALID_ARG, "pm_seq_start_from_cu: bad args");
  if (n_tokens == 0) return PM_OK;
  hipLaunchKernelGGL(seq_start_kernel, dim3((unsigned)cdiv(n_tokens, 256)), dim3(256), 0, as_stream(stream), cu_seqlens, n_seqs, n_tokens, seq_start);
  PM_CHECK_LAUNCH("pm_seq_start_from_cu");
  return PM_OK;
}
