// attention_v2.cuh — LDS-staged varlen attention kernels (included by attention.hip inside namespace pm).
//
// v1 (attention.hip) reads every MFMA operand fragment straight from global memory: correct, but each of the 4 waves of a
// workgroup re-reads the same K/V tile through the 64 B/clk vector L1, which caps it at ~10 % of the MFMA rate
// (profiles/r01_bench_mb8_v1.md: 6.0 ms dK/dV, 3.3 ms dQ, 1.4 ms forward per layer at 16k tokens).  v2 stages each shared tile
// ONCE per workgroup into LDS (16-byte global loads -> ds_write_b128 into padded rows, conflict-free ds_read_b128
// fragments), gives every wave two 16-row blocks so one B fragment feeds two MFMAs, and software-prefetches the next tile
// into registers while the current one is being multiplied (guide G15 "issue early / write late").
//
// LDS row strides: 272 B (= 17 x 16 B) for [rows][128] tiles, 144 B (= 9 x 16 B) for [rows][64], 80 B (= 5 x 16 B) for
// [rows][32]: 17, 9 and 5 are odd, so the 16 rows of a fragment land in 16 distinct 16-byte bank slots.
#pragma once

constexpr int kLd128 = 136;  // 128 + 8
constexpr int kLd64 = 72;    // 64 + 8
constexpr int kLd32 = 40;    // 32 + 8

__device__ __forceinline__ u32x4 lds16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void sts16(bf16_t* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------------------------
// forward v2: 4 waves x 32 query rows, KV tile 64 staged in LDS (K row-major, V transposed), register prefetch.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_fwd_v2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
                                                            const int32_t* __restrict__ seq_start, int64_t n, int64_t n_pad, int hq, int hkv,
                                                            float scale, bf16_t* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[kD][kLd64];
  __shared__ __attribute__((aligned(16))) bf16_t Ps[4][32][kLd64];
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
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
  const int64_t qt0 = (int64_t)tile * 128;
  const int64_t q0 = qt0 + w * 32;
  const bool active = q0 < n;

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
  const int64_t tile_last = (qt0 + 127 < n - 1) ? qt0 + 127 : n - 1;
  const int64_t wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int64_t kv_begin = ((int64_t)seq_start[qt0] / 64) * 64;
  const float sc2 = scale * 1.44269504088896340736f;  // exp(x) = exp2(x*log2e): scores are kept in the log2 domain

  // staging: K tile = 64 rows x 16 chunks, V^T tile = 128 rows x 8 chunks; 4 + 4 chunks per thread
  const int kr = t >> 4, kc = (t & 15) * 8;
  const int vr = t >> 3, vc = (t & 7) * 8;
  u32x4 kreg[4], vreg[4];
  auto issue = [&](int64_t kv0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t krow = kv0 + kr + 16 * i;
      if (krow > n - 1) krow = n - 1;
      kreg[i] = ldg16(k + (krow * hkv + kvh) * kD + kc);
      vreg[i] = ldg16(vt + ((int64_t)kvh * kD + vr + 32 * i) * n_pad + kv0 + vc);
    }
  };
  issue(kv_begin);
  for (int64_t kv0 = kv_begin; kv0 <= tile_last; kv0 += 64) {
    __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sts16(&Ks[kr + 16 * i][kc], kreg[i]);
      sts16(&Vs[vr + 32 * i][vc], vreg[i]);
    }
    __syncthreads();
    if (kv0 + 64 <= tile_last) issue(kv0 + 64);  // in flight while this tile is multiplied
    if (!active || kv0 > wave_last) continue;

    f32x4 s[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) s[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const u32x4 kf = lds16(&Ks[nb * 16 + lr][kk * 32 + lq * 8]);
        s[0][nb] = mfma16(qf[0][kk], kf, s[0][nb]);
        s[1][nb] = mfma16(qf[1][kk], kf, s[1][nb]);
      }
    }
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
          const float v = ok ? s[rb][nb][r] * sc2 : -INFINITY;
          s[rb][nb][r] = v;
          mx = fmaxf(mx, v);
        }
        mx = row16_max(mx);
        const float m_new = fmaxf(m[rb][r], mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m[rb][r] - m_use);
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float p = fast_exp2(s[rb][nb][r] - m_use);
          s[rb][nb][r] = p;
          sum += p;
        }
        sum = row16_sum(sum);
        l[rb][r] = l[rb][r] * alpha + sum;
        m[rb][r] = m_new;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) o[rb][nb][r] *= alpha;
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ps[w][rb * 16 + lq * 4 + r][nb * 16 + lr] = f32_to_bf16_hw(s[rb][nb][r]);
    lds_fence();
#pragma unroll
    for (int kk2 = 0; kk2 < 2; ++kk2) {
      const u32x4 pa0 = lds16(&Ps[w][lr][kk2 * 32 + lq * 8]);
      const u32x4 pa1 = lds16(&Ps[w][16 + lr][kk2 * 32 + lq * 8]);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const u32x4 vf = lds16(&Vs[nb * 16 + lr][kk2 * 32 + lq * 8]);
        o[0][nb] = mfma16(pa0, vf, o[0][nb]);
        o[1][nb] = mfma16(pa1, vf, o[1][nb]);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = q0 + rb * 16 + lq * 4 + r;
      if (row < n) {
        const float inv = l[rb][r] > 0.f ? 1.0f / l[rb][r] : 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) out[(row * hq + qh) * kD + nb * 16 + lr] = f32_to_bf16(o[rb][nb][r] * inv);
        // lse in natural log: m is in the log2 domain
        if (lr == 0) lse[(int64_t)qh * n + row] = m[rb][r] * 0.69314718055994530942f + logf(l[rb][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK/dV v2: workgroup = 128 kv rows (4 waves x 32), loops (q tile of 32) x (q heads of the group); the four B-operand tiles
// (Q, dO row-major for S^T / dP^T; Q^T, dO^T for dK / dV) are staged in LDS once per workgroup per iteration.
__global__ __launch_bounds__(256) void attn_bwd_dkdv_v2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                              int64_t v_stride, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qt,
                                                              const bf16_t* __restrict__ dot_, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, const int32_t* __restrict__ seq_start, int64_t n,
                                                              int64_t n_pad, int hq, int hkv, float scale, bf16_t* __restrict__ dk,
                                                              bf16_t* __restrict__ dv, int64_t dv_stride) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[32][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t dOs[32][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Qts[kD][kLd32];
  __shared__ __attribute__((aligned(16))) bf16_t dOts[kD][kLd32];
  __shared__ __attribute__((aligned(16))) bf16_t Pt[4][32][kLd32];
  __shared__ __attribute__((aligned(16))) bf16_t dSt[4][32][kLd32];
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  const int kvh = blockIdx.x % hkv;
  const int64_t kv0 = (int64_t)(blockIdx.x / hkv) * 128;
  const int64_t kvb = kv0 + w * 32;
  const bool active = kvb < n;
  u32x4 kf[2][4], vf[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    int64_t row = kvb + rb * 16 + lr;
    if (row > n - 1) row = n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[rb][kk] = ldg16(k + (row * hkv + kvh) * kD + kk * 32 + lq * 8);
      vf[rb][kk] = ldg16(v + row * v_stride + (int64_t)kvh * kD + kk * 32 + lq * 8);
    }
  }
  f32x4 dkacc[2][8], dvacc[2][8];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dkacc[rb][nb] = dvacc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t wg_last = (kv0 + 127 < n - 1) ? kv0 + 127 : n - 1;
  const float sc2 = scale * 1.44269504088896340736f;

  // staging maps: Q/dO tiles 32 rows x 16 chunks (2 per thread each); Q^T/dO^T tiles 128 rows x 4 chunks (2 per thread each)
  const int rr = t >> 4, rc = (t & 15) * 8;
  const int tr = t >> 2, tc = (t & 3) * 8;
  u32x4 rq[2], rdo[2], rqt[2], rdot[2];
  auto issue = [&](int64_t qi0, int qh) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t qrow = qi0 + rr + 16 * i;
      if (qrow > n - 1) qrow = n - 1;
      rq[i] = ldg16(q + (qrow * hq + qh) * kD + rc);
      rdo[i] = ldg16(dout + (qrow * hq + qh) * kD + rc);
      const int64_t off = ((int64_t)qh * kD + tr + 64 * i) * n_pad + qi0 + tc;
      rqt[i] = ldg16(qt + off);
      rdot[i] = ldg16(dot_ + off);
    }
  };
  // iteration space: it = (q tile index) * group + gi
  int64_t qi0 = kv0;
  int gi = 0;
  bool more = (qi0 < n) && ((int64_t)seq_start[qi0] <= wg_last);
  if (more) issue(qi0, kvh * group);
  while (more) {
    const int qh = kvh * group + gi;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sts16(&Qs[rr + 16 * i][rc], rq[i]);
      sts16(&dOs[rr + 16 * i][rc], rdo[i]);
      sts16(&Qts[tr + 64 * i][tc], rqt[i]);
      sts16(&dOts[tr + 64 * i][tc], rdot[i]);
    }
    __syncthreads();
    // next iteration (uniform across the workgroup)
    int64_t nqi0 = qi0;
    int ngi = gi + 1;
    if (ngi == group) {
      ngi = 0;
      nqi0 = qi0 + 32;
    }
    const bool nmore = (nqi0 < n) && ((int64_t)seq_start[nqi0] <= wg_last);
    if (nmore) issue(nqi0, kvh * group + ngi);

    if (active && qi0 + 31 >= kvb) {  // some q row of this tile can see this wave's kv rows
      f32x4 st[2][2], dpt[2][2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) st[rb][nb] = dpt[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const u32x4 qb = lds16(&Qs[nb * 16 + lr][kk * 32 + lq * 8]);
          const u32x4 dob = lds16(&dOs[nb * 16 + lr][kk * 32 + lq * 8]);
          st[0][nb] = mfma16(kf[0][kk], qb, st[0][nb]);
          st[1][nb] = mfma16(kf[1][kk], qb, st[1][nb]);
          dpt[0][nb] = mfma16(vf[0][kk], dob, dpt[0][nb]);
          dpt[1][nb] = mfma16(vf[1][kk], dob, dpt[1][nb]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int64_t qcol = qi0 + nb * 16 + lr;
        const int64_t qc = qcol < n ? qcol : n - 1;
        const float L2 = lse[(int64_t)qh * n + qc] * 1.44269504088896340736f;
        const float Dl = delta[(int64_t)qh * n + qc];
        const int64_t lo = seq_start[qc];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t kvr = kvb + rb * 16 + lq * 4 + r;
            const bool ok = (qcol < n) && (kvr <= qcol) && (kvr >= lo);
            const float p = ok ? fast_exp2(st[rb][nb][r] * sc2 - L2) : 0.f;
            const float ds = p * (dpt[rb][nb][r] - Dl) * scale;
            Pt[w][rb * 16 + lq * 4 + r][nb * 16 + lr] = f32_to_bf16_hw(p);
            dSt[w][rb * 16 + lq * 4 + r][nb * 16 + lr] = f32_to_bf16_hw(ds);
          }
        }
      }
      lds_fence();
      const u32x4 pa0 = lds16(&Pt[w][lr][lq * 8]);
      const u32x4 pa1 = lds16(&Pt[w][16 + lr][lq * 8]);
      const u32x4 da0 = lds16(&dSt[w][lr][lq * 8]);
      const u32x4 da1 = lds16(&dSt[w][16 + lr][lq * 8]);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const u32x4 dob = lds16(&dOts[nb * 16 + lr][lq * 8]);
        const u32x4 qb = lds16(&Qts[nb * 16 + lr][lq * 8]);
        dvacc[0][nb] = mfma16(pa0, dob, dvacc[0][nb]);
        dvacc[1][nb] = mfma16(pa1, dob, dvacc[1][nb]);
        dkacc[0][nb] = mfma16(da0, qb, dkacc[0][nb]);
        dkacc[1][nb] = mfma16(da1, qb, dkacc[1][nb]);
      }
    }
    qi0 = nqi0;
    gi = ngi;
    more = nmore;
  }
  if (!active) return;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = kvb + rb * 16 + lq * 4 + r;
      if (row < n) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          dk[(row * hkv + kvh) * kD + nb * 16 + lr] = f32_to_bf16(dkacc[rb][nb][r]);
          dv[row * dv_stride + (int64_t)kvh * kD + nb * 16 + lr] = f32_to_bf16(dvacc[rb][nb][r]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ v2: workgroup = 128 q rows (4 waves x 32), KV tile 32: K, V row-major (S, dP) and K^T (dQ) staged in LDS (38 KB -> 3-4
// workgroups per CU hide each other's staging).
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_v2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               int64_t v_stride, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ kt,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const int32_t* __restrict__ seq_start, int64_t n, int64_t n_pad, int hq, int hkv,
                                                               float scale, bf16_t* __restrict__ dq) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Vr[32][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Kts[kD][kLd32];
  __shared__ __attribute__((aligned(16))) bf16_t dSs[4][32][kLd32];
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
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
  const int64_t qt0 = (int64_t)tile * 128;
  const int64_t q0 = qt0 + w * 32;
  const bool active = q0 < n;
  u32x4 qf[2][4], dof[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    int64_t row = q0 + rb * 16 + lr;
    if (row > n - 1) row = n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[rb][kk] = ldg16(q + (row * hq + qh) * kD + kk * 32 + lq * 8);
      dof[rb][kk] = ldg16(dout + (row * hq + qh) * kD + kk * 32 + lq * 8);
    }
  }
  float L2[2][4], Dl[2][4];
  int lo[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t row = q0 + rb * 16 + lq * 4 + r;
      if (row > n - 1) row = n - 1;
      L2[rb][r] = lse[(int64_t)qh * n + row] * 1.44269504088896340736f;
      Dl[rb][r] = delta[(int64_t)qh * n + row];
      lo[rb][r] = seq_start[row];
    }
  f32x4 dqacc[2][8];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dqacc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t tile_last = (qt0 + 127 < n - 1) ? qt0 + 127 : n - 1;
  const int64_t wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int64_t kv_begin = ((int64_t)seq_start[qt0] / 32) * 32;
  const float sc2 = scale * 1.44269504088896340736f;
  const int rr = t >> 4, rc = (t & 15) * 8;  // K / V tiles: 32 rows x 16 chunks
  const int tr = t >> 2, tc = (t & 3) * 8;   // K^T tile: 128 rows x 4 chunks

  for (int64_t kv0 = kv_begin; kv0 <= tile_last; kv0 += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t krow = kv0 + rr + 16 * i;
      if (krow > n - 1) krow = n - 1;
      sts16(&Ks[rr + 16 * i][rc], ldg16(k + (krow * hkv + kvh) * kD + rc));
      sts16(&Vr[rr + 16 * i][rc], ldg16(v + krow * v_stride + (int64_t)kvh * kD + rc));
      sts16(&Kts[tr + 64 * i][tc], ldg16(kt + ((int64_t)kvh * kD + tr + 64 * i) * n_pad + kv0 + tc));
    }
    __syncthreads();
    if (!active || kv0 > wave_last) continue;
    f32x4 s[2][2], dp[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) s[rb][nb] = dp[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const u32x4 kb = lds16(&Ks[nb * 16 + lr][kk * 32 + lq * 8]);
        const u32x4 vb = lds16(&Vr[nb * 16 + lr][kk * 32 + lq * 8]);
        s[0][nb] = mfma16(qf[0][kk], kb, s[0][nb]);
        s[1][nb] = mfma16(qf[1][kk], kb, s[1][nb]);
        dp[0][nb] = mfma16(dof[0][kk], vb, dp[0][nb]);
        dp[1][nb] = mfma16(dof[1][kk], vb, dp[1][nb]);
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = kv0 + nb * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = q0 + rb * 16 + lq * 4 + r;
          const bool ok = (col <= row) && (col >= (int64_t)lo[rb][r]) && (row < n);
          const float p = ok ? fast_exp2(s[rb][nb][r] * sc2 - L2[rb][r]) : 0.f;
          const float ds = p * (dp[rb][nb][r] - Dl[rb][r]) * scale;
          dSs[w][rb * 16 + lq * 4 + r][nb * 16 + lr] = f32_to_bf16_hw(ds);
        }
      }
    lds_fence();
    const u32x4 a0 = lds16(&dSs[w][lr][lq * 8]);
    const u32x4 a1 = lds16(&dSs[w][16 + lr][lq * 8]);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const u32x4 kb = lds16(&Kts[nb * 16 + lr][lq * 8]);
      dqacc[0][nb] = mfma16(a0, kb, dqacc[0][nb]);
      dqacc[1][nb] = mfma16(a1, kb, dqacc[1][nb]);
    }
  }
  if (!active) return;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = q0 + rb * 16 + lq * 4 + r;
      if (row < n) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) dq[(row * hq + qh) * kD + nb * 16 + lr] = f32_to_bf16(dqacc[rb][nb][r]);
      }
    }
}
