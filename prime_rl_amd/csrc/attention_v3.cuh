// attention_v3.cuh — transposed ("swapped QK^T") varlen attention kernels: probabilities never leave registers.
//
// v2 staged tiles in LDS but still (a) moved P / dS from the MFMA C layout to the A layout through a per-wave LDS tile and
// (b) reduced 8 softmax rows per lane with 16-lane shuffles per KV tile.  v3 computes the TRANSPOSED products instead:
//     S^T[kv][q]  = K[kv][:]   . Q[q][:]^T          A = K fragment (LDS),  B = Q fragment (registers)
//     O^T[d][q]  += V^T[d][kv] . P^T[kv][q]         A = V^T fragment (LDS), B = P^T (registers)
// With v_mfma_f32_16x16x32_bf16 the C layout of S^T puts, in lane (lr = q, lq), the 4 consecutive kv rows lq*4..lq*4+3 of each
// 16-row block — and a B fragment wants, in the same lane (n = lr = q), 8 values along the contraction (kv) axis.  The
// contraction index may be permuted freely as long as A and B agree, so the B fragment of k-step kk2 is built IN REGISTERS from
// the C registers of blocks 2*kk2 and 2*kk2+1 (j<4 -> block 2kk2 row lq*4+j, j>=4 -> block 2kk2+1 row lq*4+j-4) and the
// matching A fragment of V^T is two 8-byte LDS reads at columns (2kk2)*16+lq*4 and (2kk2+1)*16+lq*4.  Consequences:
//   * no P / dS LDS round trip, no 2-byte LDS stores;
//   * softmax statistics are per LANE (q = lr): the row max is 15 in-lane v_max + 2 cross-row exchanges, the row sum is kept as
//     a per-lane partial and reduced once at the end; the O^T rescale is a per-lane scalar multiply;
//   * outputs leave as 8-byte stores (4 consecutive d per lane) instead of 2-byte stores.
// The same trick gives dQ^T = K^T . dS^T (dq kernel) and dV^T = dO^T . P, dK^T = Q^T . dS (dk/dv kernel, which computes
// S = Q . K^T with the kv rows of the wave as the B operand).
#pragma once

// Column permutation of the transposed tiles ([d][token] with 32-token groups): the in-register B fragment of lane (lr, lq)
// carries tokens {lq*4..lq*4+3} u {16+lq*4..16+lq*4+3} of a group, so the tile is STORED with those 8 tokens contiguous at
// position lq*8 and the matching A fragment is ONE conflict-free ds_read_b128 (the unpermuted layout needed two 8-byte reads
// that hipcc fuses into ds_read2_b64, which is served in 16-lane groups on 32 banks: 2-way conflicts at any 16-byte-aligned
// row stride).  A 16-byte global chunk c8 (tokens 8*c8..8*c8+7) therefore lands as two 8-byte LDS stores.
__device__ __forceinline__ void sts_perm(bf16_t* row_base, int c8, const u32x4& v) {
  const int c = c8 & 3;
  const int p0 = 32 * (c8 >> 2) + 16 * (c & 1) + 4 * (c >> 1);
  *reinterpret_cast<u32x2*>(row_base + p0) = u32x2{v.x, v.y};
  *reinterpret_cast<u32x2*>(row_base + p0 + 8) = u32x2{v.z, v.w};
}
__device__ __forceinline__ u32x4 pack_c_pair(const f32x4& c0, const f32x4& c1) {
  return u32x4{pack_bf16_hw(c0[0], c0[1]), pack_bf16_hw(c0[2], c0[3]), pack_bf16_hw(c1[0], c1[1]), pack_bf16_hw(c1[2], c1[3])};
}
__device__ __forceinline__ float xrow_max(float v) {  // across the 4 sixteen-lane rows of a wave (lanes lr, lr+16, lr+32, lr+48)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float xrow_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void st8_bf16(bf16_t* p, const f32x4& c, float mul) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16_hw(c[0] * mul, c[1] * mul), pack_bf16_hw(c[2] * mul, c[3] * mul)};
}

// ---------------------------------------------------------------------------------------------------------------------
// forward v3: 4 waves x 32 q (two 16-wide column blocks per wave), KV tile 64.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_fwd_v3_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
                                                               const int32_t* __restrict__ seq_start, int64_t n64, int64_t n_pad, int hq, int hkv,
                                                               float scale, bf16_t* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[kD][kLd64];
  const int n = (int)n64;
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
  const int qt0 = tile * 128;
  const int q0 = qt0 + w * 32;
  const bool active = q0 < n;

  u32x4 qf[2][4];
  int qi[2], lo[2];
  float m[2], l[2];
  f32x4 ot[2][8];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    qi[cb] = q0 + cb * 16 + lr;
    const int row = qi[cb] < n ? qi[cb] : n - 1;
    lo[cb] = seq_start[row];
    m[cb] = -INFINITY;
    l[cb] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[cb][kk] = ldg16(q + ((int64_t)row * hq + qh) * kD + kk * 32 + lq * 8);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) ot[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tile_last = (qt0 + 127 < n - 1) ? qt0 + 127 : n - 1;
  const int wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int kv_begin = (seq_start[qt0] / 64) * 64;
  const float sc2 = scale * 1.44269504088896340736f;

  const int kr = t >> 4, kc = (t & 15) * 8;
  const int vr = t >> 3, vc = (t & 7) * 8;
  u32x4 kreg[4], vreg[4];
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int krow = kv0 + kr + 16 * i;
      if (krow > n - 1) krow = n - 1;
      kreg[i] = ldg16(k + ((int64_t)krow * hkv + kvh) * kD + kc);
      vreg[i] = ldg16(vt + ((int64_t)kvh * kD + vr + 32 * i) * n_pad + kv0 + vc);
    }
  };
  issue(kv_begin);
  for (int kv0 = kv_begin; kv0 <= tile_last; kv0 += 64) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sts16(&Ks[kr + 16 * i][kc], kreg[i]);
      sts_perm(&Vs[vr + 32 * i][0], t & 7, vreg[i]);
    }
    __syncthreads();
    if (kv0 + 64 <= tile_last) issue(kv0 + 64);
    if (!active || kv0 > wave_last) continue;

    // S^T[kv][q]: rows kv = nb*16 + lq*4 + r, column q = lr of block cb
    f32x4 st[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) st[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      u32x4 ka[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) ka[0][kk] = lds16(&Ks[lr][kk * 32 + lq * 8]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        if (nb < 3) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) ka[(nb + 1) & 1][kk] = lds16(&Ks[(nb + 1) * 16 + lr][kk * 32 + lq * 8]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          st[0][nb] = mfma16(ka[nb & 1][kk], qf[0][kk], st[0][nb]);
          st[1][nb] = mfma16(ka[nb & 1][kk], qf[1][kk], st[1][nb]);
        }
      }
    }
    const bool need_mask = (kv0 + 63 > q0) || (kv0 < lo[0]) || (kv0 < lo[1]);  // per lane; cheap and exact either way
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float mx = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kv = kv0 + nb * 16 + lq * 4 + r;
          float v = st[cb][nb][r] * sc2;
          if (need_mask) v = ((kv <= qi[cb]) && (kv >= lo[cb])) ? v : -INFINITY;
          st[cb][nb][r] = v;
          mx = fmaxf(mx, v);
        }
      }
      mx = xrow_max(mx);
      const float m_new = fmaxf(m[cb], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m[cb] - m_use);
      float sum = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(st[cb][nb][r] - m_use);
          st[cb][nb][r] = p;
          sum += p;
        }
      }
      l[cb] = l[cb] * alpha + sum;  // per-lane partial row sum (own kv subset); reduced across lq once at the end
      m[cb] = m_new;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        ot[cb][nb][0] *= alpha;
        ot[cb][nb][1] *= alpha;
        ot[cb][nb][2] *= alpha;
        ot[cb][nb][3] *= alpha;
      }
    }
    // O^T[d][q] += V^T[d][kv] . P^T[kv][q]
#pragma unroll
    for (int kk2 = 0; kk2 < 2; ++kk2) {
      const u32x4 pb0 = pack_c_pair(st[0][2 * kk2], st[0][2 * kk2 + 1]);
      const u32x4 pb1 = pack_c_pair(st[1][2 * kk2], st[1][2 * kk2 + 1]);
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        u32x4 va[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nb = h4 * 4 + i;
          va[i] = lds16(&Vs[nb * 16 + lr][kk2 * 32 + lq * 8]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nb = h4 * 4 + i;
          ot[0][nb] = mfma16(va[i], pb0, ot[0][nb]);
          ot[1][nb] = mfma16(va[i], pb1, ot[1][nb]);
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const float lsum = xrow_sum(l[cb]);
    if (qi[cb] < n) {
      const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
      bf16_t* op = out + ((int64_t)qi[cb] * hq + qh) * kD + lq * 4;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) st8_bf16(op + nb * 16, ot[cb][nb], inv);
      if (lq == 0) lse[(int64_t)qh * n + qi[cb]] = m[cb] * 0.69314718055994530942f + logf(lsum);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ v3: 4 waves x 32 q, KV tile 32.  dQ^T[d][q] += K^T[d][kv] . dS^T[kv][q], dS^T built in registers.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dq_v3_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  int64_t v_stride, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ kt,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  const int32_t* __restrict__ seq_start, int64_t n64, int64_t n_pad, int hq, int hkv,
                                                                  float scale, bf16_t* __restrict__ dq) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Vr[32][kLd128];
  __shared__ __attribute__((aligned(16))) bf16_t Kts[kD][kLd32];
  const int n = (int)n64;
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
  const int qt0 = tile * 128;
  const int q0 = qt0 + w * 32;
  const bool active = q0 < n;
  u32x4 qf[2][4], dof[2][4];
  int qi[2], lo[2];
  float L2[2], Dl[2];
  f32x4 dqt[2][8];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    qi[cb] = q0 + cb * 16 + lr;
    const int row = qi[cb] < n ? qi[cb] : n - 1;
    lo[cb] = seq_start[row];
    L2[cb] = lse[(int64_t)qh * n + row] * 1.44269504088896340736f;
    Dl[cb] = delta[(int64_t)qh * n + row];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[cb][kk] = ldg16(q + ((int64_t)row * hq + qh) * kD + kk * 32 + lq * 8);
      dof[cb][kk] = ldg16(dout + ((int64_t)row * hq + qh) * kD + kk * 32 + lq * 8);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dqt[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tile_last = (qt0 + 127 < n - 1) ? qt0 + 127 : n - 1;
  const int wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int kv_begin = (seq_start[qt0] / 32) * 32;
  const float sc2 = scale * 1.44269504088896340736f;
  const int rr = t >> 4, rc = (t & 15) * 8;
  const int tr = t >> 2, tc = (t & 3) * 8;
  u32x4 rk[2], rv[2], rkt[2];
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int krow = kv0 + rr + 16 * i;
      if (krow > n - 1) krow = n - 1;
      rk[i] = ldg16(k + ((int64_t)krow * hkv + kvh) * kD + rc);
      rv[i] = ldg16(v + (int64_t)krow * v_stride + (int64_t)kvh * kD + rc);
      rkt[i] = ldg16(kt + ((int64_t)kvh * kD + tr + 64 * i) * n_pad + kv0 + tc);
    }
  };
  issue(kv_begin);
  for (int kv0 = kv_begin; kv0 <= tile_last; kv0 += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sts16(&Ks[rr + 16 * i][rc], rk[i]);
      sts16(&Vr[rr + 16 * i][rc], rv[i]);
      sts_perm(&Kts[tr + 64 * i][0], t & 3, rkt[i]);
    }
    __syncthreads();
    if (kv0 + 32 <= tile_last) issue(kv0 + 32);
    if (!active || kv0 > wave_last) continue;
    f32x4 st[2][2], dpt[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) st[cb][nb] = dpt[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      u32x4 ka[4], va[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        ka[kk] = lds16(&Ks[nb * 16 + lr][kk * 32 + lq * 8]);
        va[kk] = lds16(&Vr[nb * 16 + lr][kk * 32 + lq * 8]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        st[0][nb] = mfma16(ka[kk], qf[0][kk], st[0][nb]);
        st[1][nb] = mfma16(ka[kk], qf[1][kk], st[1][nb]);
        dpt[0][nb] = mfma16(va[kk], dof[0][kk], dpt[0][nb]);
        dpt[1][nb] = mfma16(va[kk], dof[1][kk], dpt[1][nb]);
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kv = kv0 + nb * 16 + lq * 4 + r;
          const bool ok = (kv <= qi[cb]) && (kv >= lo[cb]) && (qi[cb] < n);
          const float p = ok ? fast_exp2(st[cb][nb][r] * sc2 - L2[cb]) : 0.f;
          st[cb][nb][r] = p * (dpt[cb][nb][r] - Dl[cb]) * scale;  // dS^T
        }
      }
    }
    const u32x4 sb0 = pack_c_pair(st[0][0], st[0][1]);
    const u32x4 sb1 = pack_c_pair(st[1][0], st[1][1]);
#pragma unroll
    for (int h4 = 0; h4 < 2; ++h4) {
      u32x4 ka[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ka[i] = lds16(&Kts[(h4 * 4 + i) * 16 + lr][lq * 8]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dqt[0][h4 * 4 + i] = mfma16(ka[i], sb0, dqt[0][h4 * 4 + i]);
        dqt[1][h4 * 4 + i] = mfma16(ka[i], sb1, dqt[1][h4 * 4 + i]);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    if (qi[cb] < n) {
      bf16_t* op = dq + ((int64_t)qi[cb] * hq + qh) * kD + lq * 4;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) st8_bf16(op + nb * 16, dqt[cb][nb], 1.0f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK/dV v3: NW waves x CB 16-wide kv column blocks (B operand), loops (q tile 32) x (q heads of the group).
//   S[q][kv] = Q . K^T, dP[q][kv] = dO . V^T  (A = Q / dO fragments from LDS);  dV^T += dO^T . P,  dK^T += Q^T . dS.
// Default NW=8, CB=1 (128 kv rows per workgroup, 16 per wave): with CB=2 the two 64-register accumulator sets push the kernel
// past 256 architectural VGPRs and hipcc serialises every LDS fragment read behind a single register quad (measured 2.5 ms per
// 16k-token layer); CB=1 keeps ~150 registers, 2 workgroups (16 waves) per CU and lets the other waves hide LDS latency.
constexpr int kDkdvOffDO = 2 * 32 * kLd128 * 2;
constexpr int kDkdvOffQt = 2 * kDkdvOffDO;
constexpr int kDkdvOffDOt = kDkdvOffQt + 2 * kD * kLd32 * 2;
constexpr int kDkdvOffL = kDkdvOffDOt + 2 * kD * kLd32 * 2;
constexpr int kDkdvSmemBytes = kDkdvOffL + 768;

template <int NW, int CB, int ABL = 0>  // ABL: ablation switch for profiling (0 = real kernel, 1 = no compute, 2 = no staging)
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkdv_v3_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  int64_t v_stride, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qt,
                                                                  const bf16_t* __restrict__ dot_, const float* __restrict__ lse,
                                                                  const float* __restrict__ delta, const int32_t* __restrict__ seq_start, int64_t n64,
                                                                  int64_t n_pad, int hq, int hkv, float scale, bf16_t* __restrict__ dk,
                                                                  bf16_t* __restrict__ dv, int64_t dv_stride) {
  constexpr int NT = NW * 64;
  constexpr int ROWS = NW * CB * 16;  // kv rows per workgroup
  constexpr int NCH = 512 / NT;       // 16-byte chunks per thread per staged tile
  static_assert(512 % NT == 0, "tile chunks must divide evenly");
  // double-buffered tiles in dynamic LDS (2 x 38.3 KB > the 64 KB static limit); one barrier per iteration
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  typedef bf16_t (*tile128_t)[32][kLd128];
  typedef bf16_t (*tile32_t)[kD][kLd32];
  tile128_t Qs = reinterpret_cast<tile128_t>(dyn_smem);
  tile128_t dOs = reinterpret_cast<tile128_t>(dyn_smem + kDkdvOffDO);
  tile32_t Qts = reinterpret_cast<tile32_t>(dyn_smem + kDkdvOffQt);
  tile32_t dOts = reinterpret_cast<tile32_t>(dyn_smem + kDkdvOffDOt);
  float(*Ls)[32] = reinterpret_cast<float(*)[32]>(dyn_smem + kDkdvOffL);
  float(*Ds)[32] = reinterpret_cast<float(*)[32]>(dyn_smem + kDkdvOffL + 256);
  int(*Los)[32] = reinterpret_cast<int(*)[32]>(dyn_smem + kDkdvOffL + 512);
  __shared__ int s_ntiles;
  const int n = (int)n64;
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  const int kvh = blockIdx.x % hkv;
  const int kv0 = (blockIdx.x / hkv) * ROWS;
  const int kvb = kv0 + w * CB * 16;
  const bool active = kvb < n;
  u32x4 kf[CB][4], vf[CB][4];
  int kvi[CB];
  f32x4 dkt[CB][8], dvt[CB][8];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    kvi[cb] = kvb + cb * 16 + lr;
    const int row = kvi[cb] < n ? kvi[cb] : n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[cb][kk] = ldg16(k + ((int64_t)row * hkv + kvh) * kD + kk * 32 + lq * 8);
      vf[cb][kk] = ldg16(v + (int64_t)row * v_stride + (int64_t)kvh * kD + kk * 32 + lq * 8);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dkt[cb][nb] = dvt[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int wg_last = (kv0 + ROWS - 1 < n - 1) ? kv0 + ROWS - 1 : n - 1;
  const float sc2 = scale * 1.44269504088896340736f;
  // number of 32-row q tiles, starting at kv0, that can see this workgroup's kv rows: seq_start is non-decreasing, so count the
  // tile starts whose document begins at or before the workgroup's last kv row (all candidates tested in parallel, no chain)
  if (t == 0) s_ntiles = 0;
  __syncthreads();
  {
    int mine = 0;
    for (int c = t; kv0 + 32 * c < n; c += NT)
      if (seq_start[kv0 + 32 * c] <= wg_last) mine = c + 1;
    if (mine) atomicMax(&s_ntiles, mine);
  }
  __syncthreads();
  const int total_iters = s_ntiles * group;

  // Two register stages of prefetch (2 tiles = 64 KB per workgroup in flight) and DESCENDING q order: every workgroup of a
  // (document, kv head) starts at the document's last q tile, so the workgroups that run together on one XCD read the same Q / dO
  // tiles at about the same time and the XCD's L2 serves them once (ascending order had each workgroup at a different q tile:
  // measured 2.0 ms per 16k-token layer, memory-latency bound).
  struct Stage {
    u32x4 rq[NCH], rdo[NCH], rqt[NCH], rdot[NCH];
    float rL, rD;
    int rLo;
  };
  Stage sa, sb_;
  sa.rL = sa.rD = sb_.rL = sb_.rD = 0.f;
  sa.rLo = sb_.rLo = 0;
  auto issue = [&](Stage& st_, int j) {  // j-th iteration in execution order = iteration (total_iters-1-j) in (q tile, head) order
    const int it = total_iters - 1 - j;
    const int qi0 = kv0 + 32 * (it / group);
    const int qh = kvh * group + (it % group);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = t + i * NT;
      int qrow = qi0 + (c >> 4);
      if (qrow > n - 1) qrow = n - 1;
      st_.rq[i] = ldg16(q + ((int64_t)qrow * hq + qh) * kD + (c & 15) * 8);
      st_.rdo[i] = ldg16(dout + ((int64_t)qrow * hq + qh) * kD + (c & 15) * 8);
      const int64_t off = ((int64_t)qh * kD + (c >> 2)) * n_pad + qi0 + (c & 3) * 8;
      st_.rqt[i] = ldg16(qt + off);
      st_.rdot[i] = ldg16(dot_ + off);
    }
    if (t < 32) {
      const int qrow = (qi0 + t < n) ? qi0 + t : n - 1;
      // RAW loads only: any arithmetic on a just-loaded value makes hipcc wait for it here, and vmcnt retires in order, so the
      // wait would drain the whole prefetch (measured: this alone cost 4x on the kernel)
      st_.rL = lse[(int64_t)qh * n + qrow];
      st_.rD = delta[(int64_t)qh * n + qrow];
      st_.rLo = seq_start[qrow];
    }
  };
  auto stage = [&](const Stage& st_, int jj, int buf) {  // write the prefetched tile of execution-order iteration jj into LDS buffer buf
    const int qi0s = kv0 + 32 * ((total_iters - 1 - jj) / group);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = t + i * NT;
      sts16(&Qs[buf][c >> 4][(c & 15) * 8], st_.rq[i]);
      sts16(&dOs[buf][c >> 4][(c & 15) * 8], st_.rdo[i]);
      sts_perm(&Qts[buf][c >> 2][0], c & 3, st_.rqt[i]);
      sts_perm(&dOts[buf][c >> 2][0], c & 3, st_.rdot[i]);
    }
    if (t < 32) {
      Ls[buf][t] = st_.rL * 1.44269504088896340736f;
      Ds[buf][t] = st_.rD;
      Los[buf][t] = (qi0s + t < n) ? st_.rLo : 0x7fffffff;  // out-of-range q rows can see nothing
    }
  };
  // prologue: tile 0 -> LDS[0]; tiles 1 and 2 in flight in registers
  if (total_iters > 0) issue(sa, 0);
  if (total_iters > 1) issue(sb_, 1);
  if (total_iters > 0) stage(sa, 0, 0);
  if (total_iters > 2) issue(sa, 2);
  for (int j = 0; j < total_iters; ++j) {
    const int qi0 = kv0 + 32 * ((total_iters - 1 - j) / group);
    const int buf = j & 1;
    __syncthreads();  // compute(j-1) is finished everywhere (its buffer may be overwritten) and tile j is visible
    if (ABL != 2 || j < 2) {
      if (j + 1 < total_iters) {
        if ((j & 1) == 0) {
          stage(sb_, j + 1, buf ^ 1);
          if (j + 3 < total_iters) issue(sb_, j + 3);
        } else {
          stage(sa, j + 1, buf ^ 1);
          if (j + 3 < total_iters) issue(sa, j + 3);
        }
      }
    }

    if ((ABL != 1 || j < 2) && active && qi0 + 31 >= kvb) {
      // S[q][kv], dP[q][kv]: rows q = nb*16 + lq*4 + r, column kv = lr of block cb
      f32x4 s[CB][2], dp[CB][2];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) s[cb][nb] = dp[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        u32x4 qa[4], da[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          qa[kk] = lds16(&Qs[buf][nb * 16 + lr][kk * 32 + lq * 8]);
          da[kk] = lds16(&dOs[buf][nb * 16 + lr][kk * 32 + lq * 8]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            s[cb][nb] = mfma16(qa[kk], kf[cb][kk], s[cb][nb]);
            dp[cb][nb] = mfma16(da[kk], vf[cb][kk], dp[cb][nb]);
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = nb * 16 + lq * 4 + r;
          const int qg = qi0 + ql;
          const float L2 = Ls[buf][ql], Dl = Ds[buf][ql];
          const int lo = Los[buf][ql];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            const bool ok = (kvi[cb] <= qg) && (kvi[cb] >= lo);
            const float p = ok ? fast_exp2(s[cb][nb][r] * sc2 - L2) : 0.f;
            s[cb][nb][r] = p;
            dp[cb][nb][r] = p * (dp[cb][nb][r] - Dl) * scale;  // dS
          }
        }
      }
      u32x4 pb[CB], sb[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        pb[cb] = pack_c_pair(s[cb][0], s[cb][1]);
        sb[cb] = pack_c_pair(dp[cb][0], dp[cb][1]);
      }
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        u32x4 doa[4], qa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nb = h4 * 4 + i;
          doa[i] = lds16(&dOts[buf][nb * 16 + lr][lq * 8]);
          qa[i] = lds16(&Qts[buf][nb * 16 + lr][lq * 8]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nb = h4 * 4 + i;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            dvt[cb][nb] = mfma16(doa[i], pb[cb], dvt[cb][nb]);
            dkt[cb][nb] = mfma16(qa[i], sb[cb], dkt[cb][nb]);
          }
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    if (kvi[cb] < n) {
      bf16_t* pk_ = dk + ((int64_t)kvi[cb] * hkv + kvh) * kD + lq * 4;
      bf16_t* pv_ = dv + (int64_t)kvi[cb] * dv_stride + (int64_t)kvh * kD + lq * 4;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        st8_bf16(pk_ + nb * 16, dkt[cb][nb], 1.0f);
        st8_bf16(pv_ + nb * 16, dvt[cb][nb], 1.0f);
      }
    }
  }
}
