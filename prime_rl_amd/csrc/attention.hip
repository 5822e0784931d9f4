// attention.hip — varlen causal attention with GQA over packed documents (forward + deterministic backward), MFMA bf16.
//
// Reference: FlashAttention._compute_attention (src/prime_rl/trainer/models/layers/attn.py:94-110), i.e.
// flash_attn_varlen_func(q, k, v, cu_seqlens, cu_seqlens, max_seqlen, max_seqlen, causal=True) with GQA and
// scale head_dim^-0.5 (attn.py:60).  flash-attn itself is a third-party dependency (flash-attn 2.8.3, uv.lock:1737);
// the arithmetic restated here is its published algorithm: S = scale*Q K^T masked to (same document, j <= i),
// P = softmax_row(S), O = P V; backward dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dO o O)), dQ = scale dS K,
// dK = scale dS^T Q.
//
// MI355X design, v1 (correct-first; DESIGN.md lists what v2 changes):
//   * tiles live on the GLOBAL token grid of the packed row: row i may attend to j in [seq_start[i], i].  No per-document
//     tile scheduling on the host, no max_seqlen, no host sync: grid = ceil(N/BM) * heads.
//   * v_mfma_f32_16x16x32_bf16; every A/B fragment is a 16-byte load along the contraction dim (lane l: row l&15,
//     k = (l>>4)*8..+8), C layout row = (l>>4)*4 + r, col = l&15 (guide §3, verified by pm_mfma_probe_16x16x32).
//     Operands whose contraction dim is the token axis (V in PV; Q, dO in dK/dV; K in dQ) are read from
//     [head][128][N_pad] transposed copies written by a pre-pass into the caller's workspace.
//   * probabilities go C-layout -> per-wave LDS tile -> A-layout (one 16-byte ds_read per fragment, rows padded to
//     144 B so the 16 rows of a fragment hit 16 distinct 16-byte bank slots).  Waves never synchronise with each
//     other: each wave owns its rows end to end.
//   * blockIdx -> (kv head fastest): the workgroups that share K/V land on the same XCD (block b runs on XCD b%8),
//     q tiles are issued heaviest-first (causal imbalance).
//   * backward is two kernels (dK/dV per kv tile, dQ per q tile): no atomics, bitwise reproducible.
#include "common.cuh"
#include <stdlib.h>

namespace pm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 ldg16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

constexpr int kD = 128;       // head_dim
constexpr int kPadLd = 72;    // 64 + 8 bf16: 144-byte LDS rows

// ---------------------------------------------------------------------------------------------
// MFMA layout probe: c[i][j] = sum_k a[i][k] * b[j][k], a,b [16][32] bf16 row-major.  One wave.
__global__ void mfma_probe_kernel(const bf16_t* a, const bf16_t* b, float* c) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const u32x4 af = ldg16(a + lr * 32 + lq * 8);
  const u32x4 bf = ldg16(b + lr * 32 + lq * 8);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma16(af, bf, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[(lq * 4 + r) * 16 + lr] = acc[r];
}

// ---------------------------------------------------------------------------------------------
// ds_read_b64_tr_b16 probe (gfx950 LDS transpose-read): LDS holds lds[i] = i (u16, 4096 entries); lane l reads at element index
// elem_index[l] (multiple of 4 = 8-byte aligned) and returns its 4 result elements.  One wave.
typedef short v4i16_t __attribute__((ext_vector_type(4)));
__global__ void ds_read_tr16_probe_kernel(const int32_t* elem_index, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  const v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)(img + elem_index[threadIdx.x]));
#pragma unroll
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}

// ---------------------------------------------------------------------------------------------
// [n][row_stride] (head h at column h*128) -> [heads][128][n_pad], zero padding for tokens >= n.
__global__ __launch_bounds__(256) void transpose_heads_kernel(const bf16_t* __restrict__ src, int64_t row_stride, int64_t n, int64_t n_pad,
                                                             bf16_t* __restrict__ dst) {
  __shared__ bf16_t tile[64][66];
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  const int d0 = blockIdx.y * 64;
  const int h = blockIdx.z;
  const int t = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int tr = (t >> 3) + pass * 32;
    const int c8 = (t & 7) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t0 + tr < n) v = ldg16(src + (t0 + tr) * row_stride + (int64_t)h * kD + d0 + c8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tile[tr][c8 + 2 * k] = (bf16_t)(w[k] & 0xffffu);
      tile[tr][c8 + 2 * k + 1] = (bf16_t)(w[k] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int d = (t >> 3) + pass * 32;
    const int tk = (t & 7) * 8;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[tk + 2 * k][d] | ((uint32_t)tile[tk + 2 * k + 1][d] << 16);
    u32x4 v = {w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4*>(dst + ((int64_t)h * kD + d0 + d) * n_pad + t0 + tk) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// forward: 4 waves x 32 query rows; KV tiles of 64 on the global token grid.
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
                                                      const int32_t* __restrict__ seq_start, int64_t n, int64_t n_pad, int hq, int hkv,
                                                      float scale, bf16_t* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) bf16_t Ps[4][32][kPadLd];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  int b = blockIdx.x;
  const int kvh = b % hkv;
  b /= hkv;
  const int gi = b % group;
  b /= group;
  const int num_tiles = gridDim.x / hq;
  const int tile = num_tiles - 1 - b;
  const int qh = kvh * group + gi;
  const int64_t q0 = (int64_t)tile * 128 + w * 32;
  if (q0 >= n) return;

  u32x4 qf[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    int64_t row = q0 + rb * 16 + lr;
    if (row > n - 1) row = n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[rb][kk] = ldg16(q + (row * hq + qh) * kD + kk * 32 + lq * 8);
  }
  int lo[2][4];
  float m[2][4], l[2][4];
  f32x4 o[2][8];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t row = q0 + rb * 16 + lq * 4 + r;
      if (row > n - 1) row = n - 1;
      lo[rb][r] = seq_start[row];
      m[rb][r] = -INFINITY;
      l[rb][r] = 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) o[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int64_t kv_begin = ((int64_t)seq_start[q0] / 64) * 64;

  for (int64_t kv0 = kv_begin; kv0 <= wave_last; kv0 += 64) {
    f32x4 s[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) s[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      int64_t krow = kv0 + nb * 16 + lr;
      if (krow > n - 1) krow = n - 1;
      const bf16_t* kp = k + (krow * hkv + kvh) * kD + lq * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const u32x4 kf = ldg16(kp + kk * 32);
        s[0][nb] = mfma16(qf[0][kk], kf, s[0][nb]);
        s[1][nb] = mfma16(qf[1][kk], kf, s[1][nb]);
      }
    }
    // scale, mask, online softmax (rows live in 16-lane groups: reduce over lane bits 0..3)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = q0 + rb * 16 + lq * 4 + r;
        float mx = -INFINITY;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int64_t col = kv0 + nb * 16 + lr;
          const bool ok = (col <= row) && (col >= (int64_t)lo[rb][r]);
          const float v = ok ? s[rb][nb][r] * scale : -INFINITY;
          s[rb][nb][r] = v;
          mx = fmaxf(mx, v);
        }
        mx = group16_max(mx);
        const float m_new = fmaxf(m[rb][r], mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m[rb][r] - m_use);
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float p = __expf(s[rb][nb][r] - m_use);
          s[rb][nb][r] = p;
          sum += p;
        }
        sum = group16_sum(sum);
        l[rb][r] = l[rb][r] * alpha + sum;
        m[rb][r] = m_new;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) o[rb][nb][r] *= alpha;
      }
    }
    // P: C layout -> LDS -> A layout
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ps[w][rb * 16 + lq * 4 + r][nb * 16 + lr] = f32_to_bf16(s[rb][nb][r]);
    lds_fence();
#pragma unroll
    for (int kk2 = 0; kk2 < 2; ++kk2) {
      const u32x4 pa0 = *reinterpret_cast<const u32x4*>(&Ps[w][lr][kk2 * 32 + lq * 8]);
      const u32x4 pa1 = *reinterpret_cast<const u32x4*>(&Ps[w][16 + lr][kk2 * 32 + lq * 8]);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const u32x4 vf = ldg16(vt + ((int64_t)kvh * kD + nb * 16 + lr) * n_pad + kv0 + kk2 * 32 + lq * 8);
        o[0][nb] = mfma16(pa0, vf, o[0][nb]);
        o[1][nb] = mfma16(pa1, vf, o[1][nb]);
      }
    }
    lds_fence();
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = q0 + rb * 16 + lq * 4 + r;
      if (row < n) {
        const float inv = l[rb][r] > 0.f ? 1.0f / l[rb][r] : 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) out[(row * hq + qh) * kD + nb * 16 + lr] = f32_to_bf16(o[rb][nb][r] * inv);
        if (lr == 0) lse[(int64_t)qh * n + row] = m[rb][r] + logf(l[rb][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// delta[h][i] = sum_d dO[i,h,d] * O[i,h,d]   (16 lanes per (token, head))
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, int64_t n, int hq,
                                                        float* __restrict__ delta) {
  const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  const bool valid = item < n * hq;
  const int64_t it = valid ? item : 0;
  const u32x4 a = ldg16(dout + it * kD + sub * 8);
  const u32x4 b2 = ldg16(out + it * kD + sub * 8);
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b2.x, b2.y, b2.z, b2.w};
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s = fmaf(bf16lo(aw[k]), bf16lo(bw[k]), s);
    s = fmaf(bf16hi(aw[k]), bf16hi(bw[k]), s);
  }
  s = group16_sum(s);
  if (valid && sub == 0) {
    const int64_t tok = it / hq;
    const int h = (int)(it - tok * hq);
    delta[(int64_t)h * n + tok] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// dK / dV: one workgroup per (64 kv rows, kv head); wave w owns 16 kv rows; loops q tiles and the q heads of the group.
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                           int64_t v_stride, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qt,
                                                           const bf16_t* __restrict__ dot_, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, const int32_t* __restrict__ seq_start, int64_t n,
                                                           int64_t n_pad, int hq, int hkv, float scale, bf16_t* __restrict__ dk,
                                                           bf16_t* __restrict__ dv, int64_t dv_stride) {
  __shared__ __attribute__((aligned(16))) bf16_t Pt[4][16][kPadLd];
  __shared__ __attribute__((aligned(16))) bf16_t dSt[4][16][kPadLd];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  const int kvh = blockIdx.x % hkv;
  const int64_t kv0 = (int64_t)(blockIdx.x / hkv) * 64;
  const int64_t kvb = kv0 + w * 16;
  if (kvb >= n) return;
  u32x4 kf[4], vf[4];
  {
    int64_t row = kvb + lr;
    if (row > n - 1) row = n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ldg16(k + (row * hkv + kvh) * kD + kk * 32 + lq * 8);
      vf[kk] = ldg16(v + row * v_stride + (int64_t)kvh * kD + kk * 32 + lq * 8);
    }
  }
  f32x4 dkacc[8], dvacc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) dkacc[nb] = dvacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t kv_last = (kvb + 15 < n - 1) ? kvb + 15 : n - 1;  // last kv row this wave owns

  for (int64_t qi0 = kv0; qi0 < n && (int64_t)seq_start[qi0] <= kv_last; qi0 += 64) {
    for (int gi = 0; gi < group; ++gi) {
      const int qh = kvh * group + gi;
      f32x4 st[4], dpt[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) st[nb] = dpt[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        int64_t qrow = qi0 + nb * 16 + lr;
        if (qrow > n - 1) qrow = n - 1;
        const bf16_t* qp = q + (qrow * hq + qh) * kD + lq * 8;
        const bf16_t* dp = dout + (qrow * hq + qh) * kD + lq * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          st[nb] = mfma16(kf[kk], ldg16(qp + kk * 32), st[nb]);
          dpt[nb] = mfma16(vf[kk], ldg16(dp + kk * 32), dpt[nb]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int64_t qcol = qi0 + nb * 16 + lr;
        const int64_t qc = qcol < n ? qcol : n - 1;
        const float L = lse[(int64_t)qh * n + qc];
        const float Dl = delta[(int64_t)qh * n + qc];
        const int64_t lo = seq_start[qc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t kvr = kvb + lq * 4 + r;
          const bool ok = (qcol < n) && (kvr <= qcol) && (kvr >= lo);
          const float p = ok ? __expf(st[nb][r] * scale - L) : 0.f;
          const float ds = p * (dpt[nb][r] - Dl) * scale;
          Pt[w][lq * 4 + r][nb * 16 + lr] = f32_to_bf16(p);
          dSt[w][lq * 4 + r][nb * 16 + lr] = f32_to_bf16(ds);
        }
      }
      lds_fence();
#pragma unroll
      for (int kk2 = 0; kk2 < 2; ++kk2) {
        const u32x4 pa = *reinterpret_cast<const u32x4*>(&Pt[w][lr][kk2 * 32 + lq * 8]);
        const u32x4 dsa = *reinterpret_cast<const u32x4*>(&dSt[w][lr][kk2 * 32 + lq * 8]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const int64_t off = ((int64_t)qh * kD + nb * 16 + lr) * n_pad + qi0 + kk2 * 32 + lq * 8;
          dvacc[nb] = mfma16(pa, ldg16(dot_ + off), dvacc[nb]);
          dkacc[nb] = mfma16(dsa, ldg16(qt + off), dkacc[nb]);
        }
      }
      lds_fence();
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = kvb + lq * 4 + r;
    if (row < n) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        dk[(row * hkv + kvh) * kD + nb * 16 + lr] = f32_to_bf16(dkacc[nb][r]);
        dv[row * dv_stride + (int64_t)kvh * kD + nb * 16 + lr] = f32_to_bf16(dvacc[nb][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dQ: one workgroup per (64 q rows, q head); wave w owns 16 q rows.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                         int64_t v_stride, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ kt,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         const int32_t* __restrict__ seq_start, int64_t n, int64_t n_pad, int hq, int hkv,
                                                         float scale, bf16_t* __restrict__ dq) {
  __shared__ __attribute__((aligned(16))) bf16_t dSs[4][16][kPadLd];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  int b = blockIdx.x;
  const int kvh = b % hkv;
  b /= hkv;
  const int gi = b % group;
  b /= group;
  const int num_tiles = gridDim.x / hq;
  const int tile = num_tiles - 1 - b;
  const int qh = kvh * group + gi;
  const int64_t qb0 = (int64_t)tile * 64 + w * 16;
  if (qb0 >= n) return;
  u32x4 qf[4], dof[4];
  {
    int64_t row = qb0 + lr;
    if (row > n - 1) row = n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[kk] = ldg16(q + (row * hq + qh) * kD + kk * 32 + lq * 8);
      dof[kk] = ldg16(dout + (row * hq + qh) * kD + kk * 32 + lq * 8);
    }
  }
  float L[4], Dl[4];
  int lo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t row = qb0 + lq * 4 + r;
    if (row > n - 1) row = n - 1;
    L[r] = lse[(int64_t)qh * n + row];
    Dl[r] = delta[(int64_t)qh * n + row];
    lo[r] = seq_start[row];
  }
  f32x4 dqacc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) dqacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t wave_last = (qb0 + 15 < n - 1) ? qb0 + 15 : n - 1;
  const int64_t kv_begin = ((int64_t)seq_start[qb0] / 64) * 64;

  for (int64_t kv0 = kv_begin; kv0 <= wave_last; kv0 += 64) {
    f32x4 s[4], dp[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) s[nb] = dp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      int64_t krow = kv0 + nb * 16 + lr;
      if (krow > n - 1) krow = n - 1;
      const bf16_t* kp = k + (krow * hkv + kvh) * kD + lq * 8;
      const bf16_t* vp = v + krow * v_stride + (int64_t)kvh * kD + lq * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s[nb] = mfma16(qf[kk], ldg16(kp + kk * 32), s[nb]);
        dp[nb] = mfma16(dof[kk], ldg16(vp + kk * 32), dp[nb]);
      }
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int64_t col = kv0 + nb * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = qb0 + lq * 4 + r;
        const bool ok = (col <= row) && (col >= (int64_t)lo[r]) && (row < n);
        const float p = ok ? __expf(s[nb][r] * scale - L[r]) : 0.f;
        const float ds = p * (dp[nb][r] - Dl[r]) * scale;
        dSs[w][lq * 4 + r][nb * 16 + lr] = f32_to_bf16(ds);
      }
    }
    lds_fence();
#pragma unroll
    for (int kk2 = 0; kk2 < 2; ++kk2) {
      const u32x4 dsa = *reinterpret_cast<const u32x4*>(&dSs[w][lr][kk2 * 32 + lq * 8]);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const u32x4 kb = ldg16(kt + ((int64_t)kvh * kD + nb * 16 + lr) * n_pad + kv0 + kk2 * 32 + lq * 8);
        dqacc[nb] = mfma16(dsa, kb, dqacc[nb]);
      }
    }
    lds_fence();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = qb0 + lq * 4 + r;
    if (row < n) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) dq[(row * hq + qh) * kD + nb * 16 + lr] = f32_to_bf16(dqacc[nb][r]);
    }
  }
}

#include "attention_v2.cuh"
#include "attention_v3.cuh"
#include "attention_v4.cuh"

static inline int64_t pad64(int64_t n) { return cdiv(n, 64) * 64; }
static bool use_v2() {
  static const bool v2 = getenv("PM_ATTN_V2") != nullptr;  // A/B switch: LDS round-trip kernels
  return v2;
}
static bool occ1() {
  static const bool o = getenv("PM_ATTN_OCC1") != nullptr;  // A/B switch: 1 workgroup per CU (no register cap)
  return o;
}
static bool use_v1() {
  static const bool v1 = getenv("PM_ATTN_V1") != nullptr;  // A/B switch for the correct-first v1 kernels
  return v1;
}
static inline int64_t align256(int64_t b) { return cdiv(b, 256) * 256; }

static int launch_transpose(const bf16_t* src, int64_t row_stride, int64_t n, int64_t n_pad, int heads, bf16_t* dst, hipStream_t st) {
  hipLaunchKernelGGL(transpose_heads_kernel, dim3((unsigned)(n_pad / 64), 2, (unsigned)heads), dim3(256), 0, st, src, row_stride, n, n_pad, dst);
  PM_CHECK_LAUNCH("attention transpose");
  return PM_OK;
}

}  // namespace pm

using namespace pm;

extern "C" int pm_mfma_probe_16x16x32(const void* a, const void* b, float* c, void* stream) {
  PM_REQUIRE(a && b && c, PM_ERR_INVALID_ARG, "pm_mfma_probe_16x16x32: null pointer");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const bf16_t*)a, (const bf16_t*)b, c);
  PM_CHECK_LAUNCH("pm_mfma_probe_16x16x32");
  return PM_OK;
}

extern "C" int pm_ds_read_tr16_probe(const int32_t* elem_index, void* out_u16, void* stream) {
  PM_REQUIRE(elem_index && out_u16, PM_ERR_INVALID_ARG, "pm_ds_read_tr16_probe: null pointer");
  hipLaunchKernelGGL(ds_read_tr16_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), elem_index, (uint16_t*)out_u16);
  PM_CHECK_LAUNCH("pm_ds_read_tr16_probe");
  return PM_OK;
}

extern "C" int64_t pm_attn_fwd_workspace_bytes(int64_t n_tokens, int32_t n_kv_heads, int32_t head_dim) {
  if (n_tokens <= 0) return 0;
  return align256((int64_t)n_kv_heads * head_dim * pad64(n_tokens) * 2);
}

extern "C" int pm_attn_varlen_fwd(const void* q, const void* k, const void* v, int64_t v_row_stride, const int32_t* seq_start, int64_t n_tokens,
                                  int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale, void* out, float* lse,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(q && k && v && seq_start && out && lse, PM_ERR_INVALID_ARG, "pm_attn_varlen_fwd: null pointer");
  PM_REQUIRE(head_dim == kD, PM_ERR_UNSUPPORTED, "pm_attn_varlen_fwd: head_dim=%d (only 128 is built)", head_dim);
  PM_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, PM_ERR_INVALID_ARG, "pm_attn_varlen_fwd: bad head counts");
  PM_REQUIRE(n_tokens >= 0 && n_tokens < (1ll << 31) - 256 && v_row_stride >= (int64_t)n_kv_heads * kD && v_row_stride % 8 == 0, PM_ERR_INVALID_ARG,
             "pm_attn_varlen_fwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  const int64_t need = pm_attn_fwd_workspace_bytes(n_tokens, n_kv_heads, head_dim);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_attn_varlen_fwd: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  const int64_t n_pad = pad64(n_tokens);
  bf16_t* vt = (bf16_t*)workspace;
  const int64_t tiles = cdiv(n_tokens, 128);
  if (!use_v1() && !use_v2() && getenv("PM_ATTN_FWD_V3") == nullptr) {  // v4: V^T fragments are LDS transpose-reads, no transposed copy of V
    static const bool smem_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_v4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    kV4FwdSmemBytes) == hipSuccess;
    PM_REQUIRE(smem_ok, PM_ERR_LAUNCH, "pm_attn_varlen_fwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipLaunchKernelGGL((attn_fwd_v4_kernel<2>), dim3((unsigned)(tiles * n_q_heads)), dim3(256), kV4FwdSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, seq_start, n_tokens, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
    PM_CHECK_LAUNCH("pm_attn_varlen_fwd");
    return PM_OK;
  }
  int rc = launch_transpose((const bf16_t*)v, v_row_stride, n_tokens, n_pad, n_kv_heads, vt, st);
  if (rc != PM_OK) return rc;
  if (use_v1())
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(tiles * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, vt, seq_start,
                       n_tokens, n_pad, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  else if (!use_v2() && occ1())
    hipLaunchKernelGGL((attn_fwd_v3_kernel<1>), dim3((unsigned)(tiles * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, vt, seq_start,
                       n_tokens, n_pad, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  else if (!use_v2())
    hipLaunchKernelGGL((attn_fwd_v3_kernel<2>), dim3((unsigned)(tiles * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, vt, seq_start,
                       n_tokens, n_pad, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  else if (getenv("PM_ATTN_FWD_OCC2") != nullptr)  // A/B: 2 workgroups per CU with spills vs 1 without
    hipLaunchKernelGGL((attn_fwd_v2_kernel<2>), dim3((unsigned)(tiles * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, vt, seq_start,
                       n_tokens, n_pad, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  else
    hipLaunchKernelGGL((attn_fwd_v2_kernel<1>), dim3((unsigned)(tiles * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, vt, seq_start,
                       n_tokens, n_pad, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  PM_CHECK_LAUNCH("pm_attn_varlen_fwd");
  return PM_OK;
}

extern "C" int64_t pm_attn_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim) {
  if (n_tokens <= 0) return 0;
  const int64_t n_pad = pad64(n_tokens);
  const int64_t qt = align256((int64_t)n_q_heads * head_dim * n_pad * 2);
  const int64_t kt = align256((int64_t)n_kv_heads * head_dim * n_pad * 2);
  const int64_t dl = align256((int64_t)n_q_heads * n_tokens * 4);
  return qt * 2 + kt + dl;
}

extern "C" int pm_attn_varlen_bwd(const void* grad_out, const void* q, const void* k, const void* v, int64_t v_row_stride, const void* out,
                                  const float* lse, const int32_t* seq_start, int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads,
                                  int32_t head_dim, float softmax_scale, void* grad_q, void* grad_k, void* grad_v, int64_t grad_v_row_stride,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(grad_out && q && k && v && out && lse && seq_start && grad_q && grad_k && grad_v, PM_ERR_INVALID_ARG, "pm_attn_varlen_bwd: null pointer");
  PM_REQUIRE(head_dim == kD, PM_ERR_UNSUPPORTED, "pm_attn_varlen_bwd: head_dim=%d (only 128 is built)", head_dim);
  PM_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, PM_ERR_INVALID_ARG, "pm_attn_varlen_bwd: bad head counts");
  PM_REQUIRE(n_tokens >= 0 && v_row_stride % 8 == 0 && grad_v_row_stride >= (int64_t)n_kv_heads * kD, PM_ERR_INVALID_ARG, "pm_attn_varlen_bwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  const int64_t need = pm_attn_bwd_workspace_bytes(n_tokens, n_q_heads, n_kv_heads, head_dim);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_attn_varlen_bwd: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  const int64_t n = n_tokens, n_pad = pad64(n);
  const int64_t qt_b = align256((int64_t)n_q_heads * kD * n_pad * 2);
  const int64_t kt_b = align256((int64_t)n_kv_heads * kD * n_pad * 2);
  char* ws = (char*)workspace;
  bf16_t* qt = (bf16_t*)ws;
  bf16_t* dot_ = (bf16_t*)(ws + qt_b);
  bf16_t* kt = (bf16_t*)(ws + 2 * qt_b);
  float* delta = (float*)(ws + 2 * qt_b + kt_b);
  int rc;
  // dK/dV v4 reads its transposed fragments out of the row-major tiles (ds_read_b64_tr_b16): no transposed copies of Q / dO
  const bool dkdv_v4 = !use_v1() && !use_v2() && getenv("PM_ATTN_DKDV_V3") == nullptr;
  if (!dkdv_v4) {
    if ((rc = launch_transpose((const bf16_t*)q, (int64_t)n_q_heads * kD, n, n_pad, n_q_heads, qt, st)) != PM_OK) return rc;
    if ((rc = launch_transpose((const bf16_t*)grad_out, (int64_t)n_q_heads * kD, n, n_pad, n_q_heads, dot_, st)) != PM_OK) return rc;
  }
  const bool dq_v4 = dkdv_v4 && getenv("PM_ATTN_DQ_V3") == nullptr;  // same scheme for dQ: no transposed copy of K either
  if (!dq_v4 && (rc = launch_transpose((const bf16_t*)k, (int64_t)n_kv_heads * kD, n, n_pad, n_kv_heads, kt, st)) != PM_OK) return rc;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv(n * n_q_heads, 16)), dim3(256), 0, st, (const bf16_t*)grad_out, (const bf16_t*)out, n,
                     n_q_heads, delta);
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(delta)");
  if (use_v1()) {
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3((unsigned)(cdiv(n, 64) * n_kv_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                       softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dkdv)");
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(cdiv(n, 64) * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, kt, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                       softmax_scale, (bf16_t*)grad_q);
  } else if (!use_v2()) {
    static const bool smem_ok = [] {  // > 64 KB of dynamic LDS needs an explicit opt-in per kernel
      bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v3_kernel<8, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvSmemBytes) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v3_kernel<8, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvSmemBytes) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v3_kernel<8, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvSmemBytes) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v3_kernel<4, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvSmemBytes) == hipSuccess;
      return ok;
    }();
    PM_REQUIRE(smem_ok, PM_ERR_LAUNCH, "pm_attn_varlen_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    if (dkdv_v4) {
      // PM_ATTN_DKDV_V4_VAR (A/B; whole backward per 16k-token layer): 4 (default) = 64-row q tiles, 4 waves, 2 workgroups/CU (1.37 ms);
      // 0 = 32-row q tiles (1.47 ms); 1 / 2 = 3 workgroups/CU with 2- / 4-fragment batches (register spills: +9 % / +34 % over 0);
      // 3 = one 8-wave workgroup with 128 kv rows (+5 % over 0)
      static const int var = getenv("PM_ATTN_DKDV_V4_VAR") ? atoi(getenv("PM_ATTN_DKDV_V4_VAR")) : 4;
#define PM_LAUNCH_DKDV_V4(NW_, MINB_, FB_)                                                                                                       \
  hipLaunchKernelGGL((attn_bwd_dkdv_v4_kernel<NW_, MINB_, FB_>), dim3((unsigned)(cdiv(n, NW_ * 16) * n_kv_heads)), dim3(NW_ * 64), kV4SmemBytes, st, \
                     (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n,        \
                     n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride)
      if (var == 4) {  // 64-row q tiles
        constexpr int kSmemQ64 = 4 * 64 * kD * 2 + 6 * 64 * 4 + 16;
        static const bool q64_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v4q_kernel<4, 2, 4, 64>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSmemQ64) == hipSuccess;
        PM_REQUIRE(q64_ok, PM_ERR_LAUNCH, "pm_attn_varlen_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        hipLaunchKernelGGL((attn_bwd_dkdv_v4q_kernel<4, 2, 4, 64>), dim3((unsigned)(cdiv(n, 64) * n_kv_heads)), dim3(256), kSmemQ64, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n, n_q_heads, n_kv_heads,
                           softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
      } else if (var == 1) PM_LAUNCH_DKDV_V4(4, 3, 2);
      else if (var == 2) PM_LAUNCH_DKDV_V4(4, 3, 4);
      else if (var == 3) PM_LAUNCH_DKDV_V4(8, 1, 4);
      else PM_LAUNCH_DKDV_V4(4, 2, 4);
#undef PM_LAUNCH_DKDV_V4
    } else if (getenv("PM_ATTN_DKDV_CB2") != nullptr)  // A/B: 4 waves x 32 kv rows (register-starved) vs 8 waves x 16
      hipLaunchKernelGGL((attn_bwd_dkdv_v3_kernel<4, 2>), dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(256), kDkdvSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k,
                         (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    else if (getenv("PM_ATTN_ABL1") != nullptr)
      hipLaunchKernelGGL((attn_bwd_dkdv_v3_kernel<8, 1, 1>), dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(512), kDkdvSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k,
                         (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    else if (getenv("PM_ATTN_ABL2") != nullptr)
      hipLaunchKernelGGL((attn_bwd_dkdv_v3_kernel<8, 1, 2>), dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(512), kDkdvSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k,
                         (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    else
    hipLaunchKernelGGL((attn_bwd_dkdv_v3_kernel<8, 1>), dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(512), kDkdvSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                       softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dkdv)");
    if (dq_v4)
      hipLaunchKernelGGL((attn_bwd_dq_v4_kernel<2>), dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), kV4DqSmemBytes, st, (const bf16_t*)q,
                         (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_q);
    else if (occ1())
      hipLaunchKernelGGL((attn_bwd_dq_v3_kernel<1>), dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                         (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, kt, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_q);
    else
      hipLaunchKernelGGL((attn_bwd_dq_v3_kernel<2>), dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                         (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, kt, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_q);
  } else {
    hipLaunchKernelGGL(attn_bwd_dkdv_v2_kernel, dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                       softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dkdv)");
    hipLaunchKernelGGL(attn_bwd_dq_v2_kernel, dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, kt, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                       softmax_scale, (bf16_t*)grad_q);
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dq)");
  return PM_OK;
}
