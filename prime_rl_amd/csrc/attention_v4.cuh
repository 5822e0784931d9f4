// attention_v4.cuh — dK/dV with gfx950 LDS transpose-reads: ONE swizzled row-major image of each Q / dO tile serves both operand
// orientations.
//
// v3's dK/dV kernel staged every Q and dO tile twice — row-major (A fragments of S = Q.K^T, dP = dO.V^T) and as a transposed copy
// (A fragments of dV^T += dO^T.P, dK^T += Q^T.dS) that a pre-pass had to write to global memory first.  PMC counters on that kernel
// (rocprofv3, MI355X): 58 % of wave cycles parked (one 8-wave workgroup per CU: 218 VGPRs, 76 KB LDS; every wave waits at the same
// barrier), 41 % of LDS cycles are bank conflicts (ds_read_b128 is serviced in NON-contiguous 16-lane groups, which the padded
// layouts did not account for), MFMA busy 14 %.  v4:
//   * the transposed fragments come from the row-major tile through `ds_read_b64_tr_b16` (semantics pinned by
//     tests/test_kernels_gpu.py::test_ds_read_tr16_semantics): no transposed global copies, no pre-pass, half the staging traffic
//     and LDS footprint (32 KB double-buffered instead of 76 KB);
//   * tiles are stored unpadded ([32][128] bf16, 256-byte rows) with a 16-byte-chunk XOR swizzle chunk' = chunk ^ f(row & 15),
//     f(r) = ((r3^r2)<<3 | r1<<2 | r0<<1 | r2), chosen so that BOTH access patterns are conflict-free: the ds_read_b128 row reads
//     (lane groups {0-3,12-15,20-27}, ...: two lanes collide iff f(a)^f(b) = 1 iff a^b = 12, which never pairs lanes of one group)
//     and the transpose-reads (a 32-lane half touches 8 rows x 32 bytes; (f>>1) is injective over those rows, so the halves tile
//     all 64 banks exactly), and the 16-byte staging stores (8-lane groups: 8 chunks of one row, distinct mod 8);
//   * NW = 4 waves x 16 kv rows per workgroup, one register stage of prefetch (global loads of tile j+2 are in flight while tile j
//     is consumed and tile j+1 sits in the other LDS buffer): ~130 VGPRs less pressure, two or more workgroups per CU, i.e.
//     independent barrier domains that overlap each other's staging / LDS / MFMA phases.
// Math, masking, accumulation order within a wave and the output path are those of v3 (S, dP in C layout -> P, dS packed in
// registers as the B operand with the permuted contraction index q = (i>>2)*16 + lq*4 + (i&3), i = 0..7).
#pragma once

// ---- the transposed-product formulation shared by all three kernels (introduced with the v3 generation, now in the git history) -----
// Every product is computed TRANSPOSED:
//     S^T[kv][q]  = K[kv][:]   . Q[q][:]^T          A = K fragment (LDS),  B = Q fragment (registers)
//     O^T[d][q]  += V^T[d][kv] . P^T[kv][q]         A = V^T fragment (LDS), B = P^T (registers)
// With v_mfma_f32_16x16x32_bf16 the C layout of S^T puts, in lane (lr = q, lq), the 4 consecutive kv rows lq*4..lq*4+3 of each
// 16-row block — and a B fragment wants, in the same lane (n = lr = q), 8 values along the contraction (kv) axis.  The
// contraction index may be permuted freely as long as A and B agree, so the B fragment of k-step kk2 is built IN REGISTERS from
// the C registers of blocks 2*kk2 and 2*kk2+1 (j<4 -> block 2kk2 row lq*4+j, j>=4 -> block 2kk2+1 row lq*4+j-4) and the
// matching A fragment of V^T is read with the same permutation.  Consequences: no P / dS LDS round trip, no 2-byte LDS stores;
// softmax statistics are per LANE (q = lr): the row max is 15 in-lane v_max + 2 cross-row exchanges, the row sum is kept as a
// per-lane partial and reduced once at the end; the O^T rescale is a per-lane scalar multiply; outputs leave as 8-byte stores.
// The same trick gives dQ^T = K^T . dS^T (dq kernel) and dV^T = dO^T . P, dK^T = Q^T . dS (dk/dv kernel).
__device__ __forceinline__ u32x4 pack_c_pair(const f32x4& c0, const f32x4& c1) {
  return u32x4{pack_bf16_hw(c0[0], c0[1]), pack_bf16_hw(c0[2], c0[3]), pack_bf16_hw(c1[0], c1[1]), pack_bf16_hw(c1[2], c1[3])};
}
__device__ __forceinline__ float xrow_sum(float v) {  // across the 4 sixteen-lane rows of a wave (lanes lr, lr+16, lr+32, lr+48)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void st8_bf16(bf16_t* p, const f32x4& c, float mul) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16_hw(c[0] * mul, c[1] * mul), pack_bf16_hw(c[2] * mul, c[3] * mul)};
}

// Pin a loop-invariant register operand that was loaded from global memory in the prologue: the empty asm takes the value as a read-write
// operand, so the load must have COMPLETED here and the compiler cannot sink it towards its first use inside the main loop.  Without it
// hipcc placed the Q-fragment loads after the first K/V prefetch and left them pending at loop entry; its waitcnt pass then guarded every
// use of those registers inside the loop with `s_waitcnt vmcnt(3..0)` — and vmcnt counts in order, so in steady state each of those waits
// drained the K/V PREFETCH issued a few instructions earlier: the memory latency the prefetch exists to hide was paid in every iteration.
__device__ __forceinline__ void pin_loaded(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_loaded(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_loaded(int& v) { asm volatile("" : "+v"(v)); }

typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t* lds_v4i16_ptr;

__device__ __forceinline__ int swz_f(int r) {  // r in [0, 16)
  return (((r >> 3) ^ (r >> 2)) & 1) << 3 | (r & 2) << 1 | (r & 1) << 1 | ((r >> 2) & 1);
}
// element offset of (row, 16-byte chunk) in a swizzled [rows][128] bf16 tile
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * kD + ((chunk ^ swz_f(row & 15)) << 3); }

// A fragment (16 x 32, M = 16 consecutive tile COLUMNS starting at col0, K = the 32 tile rows in the permuted order
// q = (i>>2)*16 + lq*4 + (i&3)) of the TRANSPOSED tile, read from the row-major swizzled image with two transpose-reads.
// Lane (lr, lq) addresses 4 contiguous elements of tile row r0 = lq*4 + (lr>>2) at column col0 + 4*(lr&3) (`p`, already swizzled);
// the second read takes the same columns of row 16 + r0 (same row & 15, hence the same swizzle: a constant 16-row offset).
__device__ __forceinline__ u32x4 lds_tr_pair(const bf16_t* p) {
  const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)p);
  const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(p + 16 * kD));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  return u32x4{l2.x, l2.y, h2.x, h2.y};
}

// max over the four 16-lane rows of a wave (lanes lr, lr+16, lr+32, lr+48) on the VALU: v_permlane32_swap exchanges the wave halves,
// v_permlane16_swap the odd/even rows (gfx950); with both operands holding the same value the two results are "mine" and "the other
// half's / row's".  Replaces two __shfl_xor = ds_bpermute round trips through the LDS pipe per call.
// Written as inline asm on purpose: with this toolchain (ROCm 7.2 hipcc) the builtins `__builtin_amdgcn_permlane{16,32}_swap` return
// their FIRST result twice (the v_max of the two results is folded away; tools/probes/swap_probe.hip shows it), which made every lane
// adopt row 0's partial max — outputs stayed right, lse went to -inf for rows whose lq = 0 lane saw only masked scores.  Both
// registers are read-write operands of the instruction; the s_nops keep the VALU-write -> permlane hazard distance explicit because
// the hazard recognizer does not look inside inline asm.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fmax_raw(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float xrow_max_swap(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);  // a = [v.lo, v.lo], b = [v.hi, v.hi]
  a = fmaxf(a, b);
  b = a;
  permlane16_swap(a, b);  // a = [r0, r0, r2, r2], b = [r1, r1, r3, r3]
  return fmaxf(a, b);
}

constexpr int kV4TileElems = 32 * kD;                      // one [32][128] tile
constexpr int kV4OffDO = 2 * kV4TileElems * 2;             // bytes: Q[2] then dO[2]
constexpr int kV4OffL = 2 * kV4OffDO;                      // then lse / delta / lo rows [2][32] each
constexpr int kV4SmemBytes = kV4OffL + 3 * 2 * 32 * 4 + 16;  // + the q-tile count (no static LDS in front of the dynamic region)

template <int NW, int MINB, int FB>  // waves per workgroup, workgroups per CU the register budget is sized for, fragment batch
__global__ __launch_bounds__(NW * 64, MINB * NW / 4) void attn_bwd_dkdv_v4_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t v_stride, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, const float* __restrict__ delta, const int32_t* __restrict__ seq_start, int64_t n64, int hq, int hkv,
    float scale, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dv_stride) {
  constexpr int NT = NW * 64;
  constexpr int ROWS = NW * 16;      // kv rows per workgroup
  constexpr int NCH = 512 / NT;      // 16-byte chunks per thread per staged tile
  static_assert(512 % NT == 0, "tile chunks must divide evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(dyn_smem);
  bf16_t* dOs = reinterpret_cast<bf16_t*>(dyn_smem + kV4OffDO);
  float(*Ls)[32] = reinterpret_cast<float(*)[32]>(dyn_smem + kV4OffL);
  float(*Ds)[32] = reinterpret_cast<float(*)[32]>(dyn_smem + kV4OffL + 256);
  int(*Los)[32] = reinterpret_cast<int(*)[32]>(dyn_smem + kV4OffL + 512);
  int& s_ntiles = *reinterpret_cast<int*>(dyn_smem + kV4OffL + 768);
  const int n = (int)n64;
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  const int kvh = blockIdx.x % hkv;
  const int kv0 = (blockIdx.x / hkv) * ROWS;
  const int kvb = kv0 + w * 16;
  const bool active = kvb < n;
  const int kvi = kvb + lr;
  u32x4 kf[4], vf[4];
  f32x4 dkt[8], dvt[8];
  {
    const int row = kvi < n ? kvi : n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ldg16(k + ((int64_t)row * hkv + kvh) * kD + kk * 32 + lq * 8);
      vf[kk] = ldg16(v + (int64_t)row * v_stride + (int64_t)kvh * kD + kk * 32 + lq * 8);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dkt[nb] = dvt[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int wg_last = (kv0 + ROWS - 1 < n - 1) ? kv0 + ROWS - 1 : n - 1;
  const float sc2 = scale * 1.44269504088896340736f;
  // number of 32-row q tiles, starting at kv0, that can see this workgroup's kv rows (seq_start is non-decreasing)
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

  // execution order j = 0.. walks (q tile, head) pairs from the LAST q tile down (workgroups of one document stay in step on L2).
  // Three cursors (issue runs two tiles ahead, stage one) step through that order with a decrement instead of a divide per use.
  struct Cursor {
    int tile, head;
  };
  auto advance = [&](Cursor& c) {
    if (--c.head < 0) {
      c.head = group - 1;
      --c.tile;
    }
  };
  Cursor c_issue{s_ntiles - 1, group - 1}, c_stage = c_issue, c_cur = c_issue;
  u32x4 rq[NCH], rdo[NCH];
  float rL = 0.f, rD = 0.f;
  int rLo = 0;
  auto issue = [&]() {
    const int qi0 = kv0 + 32 * c_issue.tile;
    const int qh = kvh * group + c_issue.head;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = t + i * NT;
      int qrow = qi0 + (c >> 4);
      if (qrow > n - 1) qrow = n - 1;
      rq[i] = ldg16(q + ((int64_t)qrow * hq + qh) * kD + (c & 15) * 8);
      rdo[i] = ldg16(dout + ((int64_t)qrow * hq + qh) * kD + (c & 15) * 8);
    }
    if (t < 32) {  // RAW loads only (arithmetic on a just-loaded value would drain the prefetch queue: vmcnt retires in order)
      const int qrow = (qi0 + t < n) ? qi0 + t : n - 1;
      rL = lse[(int64_t)qh * n + qrow];
      rD = delta[(int64_t)qh * n + qrow];
      rLo = seq_start[qrow];
    }
    advance(c_issue);
  };
  int st_off[NCH];  // loop-invariant swizzled store offsets of this thread's chunks
#pragma unroll
  for (int i = 0; i < NCH; ++i) st_off[i] = swz_off((t + i * NT) >> 4, (t + i * NT) & 15);
  auto stage = [&](int buf) {
    const int qi0s = kv0 + 32 * c_stage.tile;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      sts16(Qs + buf * kV4TileElems + st_off[i], rq[i]);
      sts16(dOs + buf * kV4TileElems + st_off[i], rdo[i]);
    }
    if (t < 32) {
      Ls[buf][t] = rL * 1.44269504088896340736f;
      Ds[buf][t] = rD;
      Los[buf][t] = (qi0s + t < n) ? rLo : 0x7fffffff;  // out-of-range q rows can see nothing
    }
    advance(c_stage);
  };
  // Lane-invariant parts of the fragment addresses (elements).  The swizzle XORs only bits 3..6 of the element offset, so the
  // per-fragment part is ONE v_xor with a literal and everything else (row block, Q vs dO tile, second transpose-read) is an
  // immediate offset of the LDS instruction.
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);                                      // row read: ^ (kk << 5), + nb * 16 * kD
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));     // transpose-read: ^ (col block << 4), + 16 * kD
  if (total_iters > 0) {
    issue();
    stage(0);
  }
  if (total_iters > 1) issue();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    pin_loaded(kf[kk]);
    pin_loaded(vf[kk]);
  }
  for (int j = 0; j < total_iters; ++j) {
    const int qi0 = kv0 + 32 * c_cur.tile;
    advance(c_cur);
    const int buf = j & 1;
    __syncthreads();  // compute(j-1) is finished everywhere (buffer buf^1 may be overwritten) and tile j is visible
    if (j + 1 < total_iters) {
      stage(buf ^ 1);
      if (j + 2 < total_iters) issue();
    }
    if (!active || qi0 + 31 < kvb) continue;
    const bf16_t* Qt = Qs + buf * kV4TileElems;  // the dO tile of the same buffer sits kV4OffDO bytes further
    constexpr int kDO = kV4OffDO / 2;
    // S[q][kv], dP[q][kv]: rows q = nb*16 + lq*4 + r, column kv = lr
    f32x4 s[2], dp[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      s[nb] = dp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k0 = 0; k0 < 4; k0 += FB) {  // FB fragments of each operand in flight per batch (register pressure vs LDS latency hiding)
        u32x4 qa[FB], da[FB];
#pragma unroll
        for (int kk = 0; kk < FB; ++kk) {
          const bf16_t* pa = Qt + (bq ^ ((k0 + kk) << 5)) + nb * 16 * kD;
          qa[kk] = lds16(pa);
          da[kk] = lds16(pa + kDO);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < FB; ++kk) {
          s[nb] = mfma16(qa[kk], kf[k0 + kk], s[nb]);
          dp[nb] = mfma16(da[kk], vf[k0 + kk], dp[nb]);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const f32x4 L4 = *reinterpret_cast<const f32x4*>(&Ls[buf][nb * 16 + lq * 4]);
      const f32x4 D4 = *reinterpret_cast<const f32x4*>(&Ds[buf][nb * 16 + lq * 4]);
      const u32x4 lo4 = *reinterpret_cast<const u32x4*>(&Los[buf][nb * 16 + lq * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qg = qi0 + nb * 16 + lq * 4 + r;
        const bool ok = (kvi <= qg) && (kvi >= (int)lo4[r]);
        const float p = ok ? fast_exp2(s[nb][r] * sc2 - L4[r]) : 0.f;
        s[nb][r] = p;
        dp[nb][r] = p * (dp[nb][r] - D4[r]) * scale;  // dS
      }
    }
    const u32x4 pb = pack_c_pair(s[0], s[1]);
    const u32x4 sb = pack_c_pair(dp[0], dp[1]);
#pragma unroll
    for (int n0 = 0; n0 < 8; n0 += FB) {
      u32x4 doa[FB], qa[FB];
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const bf16_t* pt = Qt + (bt ^ ((n0 + i) << 4));
        qa[i] = lds_tr_pair(pt);
        doa[i] = lds_tr_pair(pt + kDO);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        dvt[n0 + i] = mfma16(doa[i], pb, dvt[n0 + i]);
        dkt[n0 + i] = mfma16(qa[i], sb, dkt[n0 + i]);
      }
    }
  }
  if (!active || kvi >= n) return;
  bf16_t* pk_ = dk + ((int64_t)kvi * hkv + kvh) * kD + lq * 4;
  bf16_t* pv_ = dv + (int64_t)kvi * dv_stride + (int64_t)kvh * kD + lq * 4;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    st8_bf16(pk_ + nb * 16, dkt[nb], 1.0f);
    st8_bf16(pv_ + nb * 16, dvt[nb], 1.0f);
  }
}

// The same kernel with a QT-row q tile per iteration (QT = 64: half the barriers, cursor steps and softmax-statistics loads per flop;
// LDS 2 x 2 x QT x 256 B).  Kept next to the 32-row version for A/B (PM_ATTN_DKDV_V4_VAR=4).
template <int NW, int MINB, int FB, int QT>
__global__ __launch_bounds__(NW * 64, MINB * NW / 4) void attn_bwd_dkdv_v4q_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t v_stride, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, const float* __restrict__ delta, const int32_t* __restrict__ seq_start, int64_t n64, int hq, int hkv,
    float scale, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dv_stride) {
  constexpr int NT = NW * 64;
  constexpr int ROWS = NW * 16;      // kv rows per workgroup
  constexpr int NCH = QT * 16 / NT;  // 16-byte chunks per thread per staged tile
  static_assert((QT * 16) % NT == 0 && QT % 32 == 0, "tile chunks must divide evenly");
  constexpr int TILE = QT * kD;                 // elements of one [QT][128] tile
  constexpr int OFF_DO = 2 * TILE * 2;          // bytes: Q[2] then dO[2]
  constexpr int OFF_L = 2 * OFF_DO;             // then lse / delta / lo rows [2][QT] each, then the q-tile count
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(dyn_smem);
  bf16_t* dOs = reinterpret_cast<bf16_t*>(dyn_smem + OFF_DO);
  float(*Ls)[QT] = reinterpret_cast<float(*)[QT]>(dyn_smem + OFF_L);
  float(*Ds)[QT] = reinterpret_cast<float(*)[QT]>(dyn_smem + OFF_L + 2 * QT * 4);
  int(*Los)[QT] = reinterpret_cast<int(*)[QT]>(dyn_smem + OFF_L + 4 * QT * 4);
  int& s_ntiles = *reinterpret_cast<int*>(dyn_smem + OFF_L + 6 * QT * 4);
  const int n = (int)n64;
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  const int kvh = blockIdx.x % hkv;
  const int kv0 = (blockIdx.x / hkv) * ROWS;
  const int kvb = kv0 + w * 16;
  const bool active = kvb < n;
  const int kvi = kvb + lr;
  u32x4 kf[4], vf[4];
  f32x4 dkt[8], dvt[8];
  {
    const int row = kvi < n ? kvi : n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ldg16(k + ((int64_t)row * hkv + kvh) * kD + kk * 32 + lq * 8);
      vf[kk] = ldg16(v + (int64_t)row * v_stride + (int64_t)kvh * kD + kk * 32 + lq * 8);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dkt[nb] = dvt[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int wg_last = (kv0 + ROWS - 1 < n - 1) ? kv0 + ROWS - 1 : n - 1;
  const float sc2 = scale * 1.44269504088896340736f;
  // number of QT-row q tiles, starting at kv0, that can see this workgroup's kv rows (seq_start is non-decreasing)
  if (t == 0) s_ntiles = 0;
  __syncthreads();
  {
    int mine = 0;
    for (int c = t; kv0 + QT * c < n; c += NT)
      if (seq_start[kv0 + QT * c] <= wg_last) mine = c + 1;
    if (mine) atomicMax(&s_ntiles, mine);
  }
  __syncthreads();
  const int total_iters = s_ntiles * group;

  // execution order j = 0.. walks (q tile, head) pairs from the LAST q tile down (workgroups of one document stay in step on L2).
  // Three cursors (issue runs two tiles ahead, stage one) step through that order with a decrement instead of a divide per use.
  struct Cursor {
    int tile, head;
  };
  auto advance = [&](Cursor& c) {
    if (--c.head < 0) {
      c.head = group - 1;
      --c.tile;
    }
  };
  Cursor c_issue{s_ntiles - 1, group - 1}, c_stage = c_issue, c_cur = c_issue;
  u32x4 rq[NCH], rdo[NCH];
  float rL = 0.f, rD = 0.f;
  int rLo = 0;
  auto issue = [&]() {
    const int qi0 = kv0 + QT * c_issue.tile;
    const int qh = kvh * group + c_issue.head;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = t + i * NT;
      int qrow = qi0 + (c >> 4);
      if (qrow > n - 1) qrow = n - 1;
      rq[i] = ldg16(q + ((int64_t)qrow * hq + qh) * kD + (c & 15) * 8);
      rdo[i] = ldg16(dout + ((int64_t)qrow * hq + qh) * kD + (c & 15) * 8);
    }
    if (t < QT) {  // RAW loads only (arithmetic on a just-loaded value would drain the prefetch queue: vmcnt retires in order)
      const int qrow = (qi0 + t < n) ? qi0 + t : n - 1;
      rL = lse[(int64_t)qh * n + qrow];
      rD = delta[(int64_t)qh * n + qrow];
      rLo = seq_start[qrow];
    }
    advance(c_issue);
  };
  int st_off[NCH];  // loop-invariant swizzled store offsets of this thread's chunks
#pragma unroll
  for (int i = 0; i < NCH; ++i) st_off[i] = swz_off((t + i * NT) >> 4, (t + i * NT) & 15);
  auto stage = [&](int buf) {
    const int qi0s = kv0 + QT * c_stage.tile;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      sts16(Qs + buf * TILE + st_off[i], rq[i]);
      sts16(dOs + buf * TILE + st_off[i], rdo[i]);
    }
    if (t < QT) {
      Ls[buf][t] = rL * 1.44269504088896340736f;
      Ds[buf][t] = rD;
      Los[buf][t] = (qi0s + t < n) ? rLo : 0x7fffffff;  // out-of-range q rows can see nothing
    }
    advance(c_stage);
  };
  // Lane-invariant parts of the fragment addresses (elements).  The swizzle XORs only bits 3..6 of the element offset, so the
  // per-fragment part is ONE v_xor with a literal and everything else (row block, Q vs dO tile, second transpose-read) is an
  // immediate offset of the LDS instruction.
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);                                      // row read: ^ (kk << 5), + nb * 16 * kD
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));     // transpose-read: ^ (col block << 4), + 16 * kD
  if (total_iters > 0) {
    issue();
    stage(0);
  }
  if (total_iters > 1) issue();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    pin_loaded(kf[kk]);
    pin_loaded(vf[kk]);
  }
  for (int j = 0; j < total_iters; ++j) {
    const int qi0 = kv0 + QT * c_cur.tile;
    advance(c_cur);
    const int buf = j & 1;
    __syncthreads();  // compute(j-1) is finished everywhere (buffer buf^1 may be overwritten) and tile j is visible
    if (j + 1 < total_iters) {
      stage(buf ^ 1);
      if (j + 2 < total_iters) issue();
    }
    if (!active || qi0 + QT - 1 < kvb) continue;
    const bf16_t* Qt = Qs + buf * TILE;  // the dO tile of the same buffer sits OFF_DO bytes further
    constexpr int kDO = OFF_DO / 2;
    // S[q][kv], dP[q][kv]: rows q = nb*16 + lq*4 + r, column kv = lr
    f32x4 s[QT / 16], dp[QT / 16];
#pragma unroll
    for (int nb = 0; nb < QT / 16; ++nb) s[nb] = dp[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // k-step outermost: the 2 * QT/16 accumulators of one batch are independent, so consecutive MFMAs never wait for each other
    // (with the row block outermost every second MFMA depended on the one two slots earlier)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 qa[QT / 16], da[QT / 16];
#pragma unroll
      for (int nb = 0; nb < QT / 16; ++nb) {
        const bf16_t* pa = Qt + (bq ^ (kk << 5)) + nb * 16 * kD;
        qa[nb] = lds16(pa);
        da[nb] = lds16(pa + kDO);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < QT / 16; ++nb) {
        s[nb] = mfma16(qa[nb], kf[kk], s[nb]);
        dp[nb] = mfma16(da[nb], vf[kk], dp[nb]);
      }
    }
#pragma unroll
    for (int nb = 0; nb < QT / 16; ++nb) {
      const f32x4 L4 = *reinterpret_cast<const f32x4*>(&Ls[buf][nb * 16 + lq * 4]);
      const f32x4 D4 = *reinterpret_cast<const f32x4*>(&Ds[buf][nb * 16 + lq * 4]);
      const u32x4 lo4 = *reinterpret_cast<const u32x4*>(&Los[buf][nb * 16 + lq * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qg = qi0 + nb * 16 + lq * 4 + r;
        const bool ok = (kvi <= qg) && (kvi >= (int)lo4[r]);
        const float p = ok ? fast_exp2(s[nb][r] * sc2 - L4[r]) : 0.f;
        s[nb][r] = p;
        dp[nb][r] = p * (dp[nb][r] - D4[r]) * scale;  // dS
      }
    }
#pragma unroll
    for (int h = 0; h < QT / 32; ++h) {  // 32-row halves of the q tile: one packed P / dS operand each
      const u32x4 pb = pack_c_pair(s[2 * h], s[2 * h + 1]);
      const u32x4 sb = pack_c_pair(dp[2 * h], dp[2 * h + 1]);
#pragma unroll
      for (int n0 = 0; n0 < 8; n0 += FB) {
        u32x4 doa[FB], qa[FB];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
          const bf16_t* pt = Qt + (bt ^ ((n0 + i) << 4)) + h * 32 * kD;
          qa[i] = lds_tr_pair(pt);
          doa[i] = lds_tr_pair(pt + kDO);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FB; ++i) {
          dvt[n0 + i] = mfma16(doa[i], pb, dvt[n0 + i]);
          dkt[n0 + i] = mfma16(qa[i], sb, dkt[n0 + i]);
        }
      }
    }
  }
  if (!active || kvi >= n) return;
  bf16_t* pk_ = dk + ((int64_t)kvi * hkv + kvh) * kD + lq * 4;
  bf16_t* pv_ = dv + (int64_t)kvi * dv_stride + (int64_t)kvh * kD + lq * 4;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    st8_bf16(pk_ + nb * 16, dkt[nb], 1.0f);
    st8_bf16(pv_ + nb * 16, dvt[nb], 1.0f);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// dQ v4: 4 waves x 32 q rows (two 16-wide column blocks per wave), KV tile 32 — the v3 kernel on the v4 LDS scheme: K and V tiles are
// stored once, row-major and swizzled (A fragments of S^T = K.Q^T and dP^T = V.dO^T by ds_read_b128), and the A fragments of
// dQ^T[d][q] += K^T[d][kv] . dS^T[kv][q] come from the SAME K tile through ds_read_b64_tr_b16 — no [head][128][N] transposed copy of K,
// no pre-pass, no bank conflicts (v3: 43 % of its LDS cycles), LDS double-buffered with one barrier per KV tile.
// KT = kv rows per barrier: the tile is consumed in 32-row halves (same registers), so KT = 64 halves the barriers per flop at the
// price of 16 more staging registers and 64 KB of LDS.
template <int KT>
constexpr int kV4DqSmemBytes = 2 * 2 * KT * kD * 2;  // K[2] then V[2]

template <int MINW, int KT>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dq_v4_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  int64_t v_stride, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                  const float* __restrict__ delta, const int32_t* __restrict__ seq_start, int64_t n64,
                                                                  int hq, int hkv, float scale, bf16_t* __restrict__ dq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(dyn_smem);
  constexpr int TILE = KT * kD;      // elements of one [KT][128] tile
  constexpr int kV = 2 * TILE;       // element offset of the V tiles
  constexpr int NCH = KT / 16;       // 16-byte chunks per thread per staged tile
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
  const int kv_begin = (seq_start[qt0] / KT) * KT;
  const int n_iters = (tile_last - kv_begin) / KT + 1;
  const float sc2 = scale * 1.44269504088896340736f;
  const int rr = t >> 4, rc = (t & 15) * 8;
  u32x4 rk[NCH], rv[NCH];
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int krow = kv0 + rr + 16 * i;
      if (krow > n - 1) krow = n - 1;
      rk[i] = ldg16(k + ((int64_t)krow * hkv + kvh) * kD + rc);
      rv[i] = ldg16(v + (int64_t)krow * v_stride + (int64_t)kvh * kD + rc);
    }
  };
  const int st_off = swz_off(rr, t & 15);  // rows rr and rr + 16 share row & 15, hence the swizzle
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      sts16(Ks + buf * TILE + st_off + i * 16 * kD, rk[i]);
      sts16(Ks + kV + buf * TILE + st_off + i * 16 * kD, rv[i]);
    }
  };
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));
  issue(kv_begin);
  stage(0);
  if (n_iters > 1) issue(kv_begin + KT);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    pin_loaded(lo[cb]);
    pin_loaded(L2[cb]);
    pin_loaded(Dl[cb]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      pin_loaded(qf[cb][kk]);
      pin_loaded(dof[cb][kk]);
    }
  }
  for (int j = 0; j < n_iters; ++j) {
    const int kvt = kv_begin + KT * j;
    const int buf = j & 1;
    __syncthreads();
    if (j + 1 < n_iters) {
      stage(buf ^ 1);
      if (j + 2 < n_iters) issue(kvt + 2 * KT);
    }
    if (!active) continue;
#pragma unroll
    for (int half = 0; half < KT / 32; ++half) {
    const int kv0 = kvt + 32 * half;
    if (kv0 > wave_last) break;
    const bf16_t* Kt = Ks + buf * TILE + half * 32 * kD;
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
        const bf16_t* pa = Kt + (bq ^ (kk << 5)) + nb * 16 * kD;
        ka[kk] = lds16(pa);
        va[kk] = lds16(pa + kV);
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
      for (int i = 0; i < 4; ++i) ka[i] = lds_tr_pair(Kt + (bt ^ ((h4 * 4 + i) << 4)));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dqt[0][h4 * 4 + i] = mfma16(ka[i], sb0, dqt[0][h4 * 4 + i]);
        dqt[1][h4 * 4 + i] = mfma16(ka[i], sb1, dqt[1][h4 * 4 + i]);
      }
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
// forward v4: 4 waves x 32 q rows, KV tile 64 — v3's online-softmax kernel on the v4 LDS scheme.  K and V tiles ([64][128], swizzled)
// are staged straight from the row-major k / v tensors (v may be a strided view of the fused qkv buffer); the A fragments of
// O^T[d][q] += V^T[d][kv] . P^T[kv][q] are transpose-reads of the row-major V tile, so the [head][128][N] copy of V and its pre-pass
// are gone; double-buffered LDS, one barrier per KV tile.
constexpr int kV4FwdTile = 64 * kD;              // elements of one [64][128] tile
constexpr int kV4FwdOffV = 2 * kV4FwdTile * 2;   // bytes: K[2] then V[2]
constexpr int kV4FwdSmemBytes = 2 * kV4FwdOffV;  // 64 KB

// NW waves x 32 q rows per workgroup (4 -> 128-row q tiles, two workgroups per CU; 8 -> 256-row q tiles: every staged K/V tile feeds twice the
// MFMA work).  OPT bits (A/B through PM_ATTN_FWD_VAR = 100 * NW + OPT, measured in ONE process per comparison: boxes differ by ~5 %):
//   1  s_setprio(1) around the MFMA clusters;
//   2  skip the O-accumulator rescale when no lane's running max moved (alpha == 1 for the whole wave: bit-identical results);
//   4  K / V staging through buffer loads with 32-bit offsets (no 64-bit address arithmetic, no row clamps in the loop);
//   8  wave-uniform mask branch (one scalar branch per tile instead of per-lane exec-mask sequences);
//  16  row max as 16 inline-asm v_max3_f32;   32  precomputed transpose-read offsets.
template <int MINW, int NW = 4, int OPT = 0>
__global__ __launch_bounds__(NW * 64, MINW) void attn_fwd_v4_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  int64_t v_stride, const int32_t* __restrict__ seq_start, int64_t n64, int hq, int hkv,
                                                                  float scale, bf16_t* __restrict__ out, float* __restrict__ lse) {
  constexpr int NT = NW * 64, QR = NW * 32, SROWS = NT / 16, SPASS = 64 / SROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(dyn_smem);
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
  const int qt0 = tile * QR;
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
  const int tile_last = (qt0 + QR - 1 < n - 1) ? qt0 + QR - 1 : n - 1;
  const int wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int kv_begin = (seq_start[qt0] / 64) * 64;
  const int n_iters = (tile_last - kv_begin) / 64 + 1;
  const float sc2 = scale * 1.44269504088896340736f;

  // K / V staging through buffer loads: a wave-uniform descriptor per tensor + ONE 32-bit byte offset per thread and pass that advances by
  // a scalar per tile (the flat-pointer form cost ~45 VALU per tile in 64-bit address arithmetic and row clamps — on a kernel whose VALU,
  // not its MFMA pipe, is the busy unit).  Rows past the end of the tensor read as zeros (hardware range check), which the causal /
  // document mask discards like any other kv > q.
  const int kr = t >> 4, kc = (t & 15) * 8;
  u32x4 kreg[SPASS], vreg[SPASS];
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(k), 0, (int)((int64_t)n * hkv * kD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(v), 0, (int)(((int64_t)(n - 1) * v_stride + (int64_t)hkv * kD) * 2), 0x00020000);
  int koff[SPASS], voff[SPASS];
#pragma unroll
  for (int i = 0; i < SPASS; ++i) {
    koff[i] = (((kr + SROWS * i) * hkv + kvh) * kD + kc) * 2;
    voff[i] = (int)(((int64_t)(kr + SROWS * i) * v_stride + (int64_t)kvh * kD + kc) * 2);
  }
  const int ktile = hkv * kD * 2, vtile = (int)(v_stride * 2);  // bytes per kv row
  auto issue = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < SPASS; ++i) {
      if (OPT & 4) {
        kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(krsrc, koff[i] + kv0 * ktile, 0, 0);
        vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, voff[i] + kv0 * vtile, 0, 0);
      } else {
        int krow = kv0 + kr + SROWS * i;
        if (krow > n - 1) krow = n - 1;
        kreg[i] = ldg16(k + ((int64_t)krow * hkv + kvh) * kD + kc);
        vreg[i] = ldg16(v + (int64_t)krow * v_stride + (int64_t)kvh * kD + kc);
      }
    }
  };
  const int st_off = swz_off(kr, t & 15);  // rows kr + SROWS i share row & 15 (SROWS is a multiple of 16), hence the swizzle
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < SPASS; ++i) {
      sts16(Ks + buf * kV4FwdTile + st_off + i * SROWS * kD, kreg[i]);
      sts16(Ks + kV4FwdOffV / 2 + buf * kV4FwdTile + st_off + i * SROWS * kD, vreg[i]);
    }
  };
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));
  int bt8[8];  // the 8 column-block offsets of the V^T transpose-reads, XOR-swizzled once instead of once per read per tile
#pragma unroll
  for (int i = 0; i < 8; ++i) bt8[i] = bt ^ (i << 4);
  issue(kv_begin);
  stage(0);
  if (n_iters > 1) issue(kv_begin + 64);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    pin_loaded(lo[cb]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pin_loaded(qf[cb][kk]);
  }
  for (int j = 0; j < n_iters; ++j) {
    const int kv0 = kv_begin + 64 * j;
    const int buf = j & 1;
    __syncthreads();
    if (j + 1 < n_iters) {
      stage(buf ^ 1);
      if (j + 2 < n_iters) issue(kv0 + 128);
    }
    if (!active || kv0 > wave_last) continue;
    const bf16_t* Kt = Ks + buf * kV4FwdTile;
    const bf16_t* Vt = Kt + kV4FwdOffV / 2;

    // S^T[kv][q]: rows kv = nb*16 + lq*4 + r, column q = lr of block cb
    f32x4 st[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) st[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    {  // k-step outermost: 8 independent accumulators per batch (no back-to-back dependent MFMAs), next batch's fragments prefetched
      u32x4 ka[2][4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) ka[0][nb] = lds16(Kt + bq + nb * 16 * kD);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) ka[(kk + 1) & 1][nb] = lds16(Kt + (bq ^ ((kk + 1) << 5)) + nb * 16 * kD);
        }
        if (OPT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          st[0][nb] = mfma16(ka[kk & 1][nb], qf[0][kk], st[0][nb]);
          st[1][nb] = mfma16(ka[kk & 1][nb], qf[1][kk], st[1][nb]);
        }
        if (OPT & 1) __builtin_amdgcn_s_setprio(0);
      }
    }
    // Online softmax, written for the VALU (the unit that bounds this kernel: 32 scores per lane per tile against 64 MFMAs): the row
    // max is taken on the RAW scores (scale > 0 commutes with max), scale and subtraction are one fma feeding v_exp_f32, the f32x4
    // accumulators are rescaled with vector multiplies (v_pk_mul_f32), and the max crosses the four 16-lane rows of the wave through
    // v_permlane32_swap / v_permlane16_swap (VALU) instead of two ds_bpermute round trips through the LDS pipe.
    // wave-uniform: one scalar branch per tile (a per-lane condition made hipcc emit an exec-mask branch sequence per pair of scores — 16 of
    // them per tile — plus unconditional compare / select code); masking a score that needs no mask is exact, so "any lane" is enough
    const bool lane_mask = (kv0 + 63 > q0) || (kv0 < lo[0]) || (kv0 < lo[1]);
    const bool need_mask = (OPT & 8) ? (bool)__any(lane_mask) : lane_mask;
    if (need_mask) {  // a separate pass, so that the common (interior-tile) path below carries no mask code at all
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kv = kv0 + nb * 16 + lq * 4 + r;
            st[cb][nb][r] = ((kv <= qi[cb]) && (kv >= lo[cb])) ? st[cb][nb][r] : -INFINITY;
          }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float mx;
      if (OPT & 16) {  // 16 single-instruction v_max3 for the 32 scores (plain fmaxf on MFMA outputs gets a canonicalising v_max each)
        mx = st[cb][0][0];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if (nb == 0) {
            mx = max3_f32(mx, st[cb][0][1], st[cb][0][2]);
            mx = fmax_raw(mx, st[cb][0][3]);
          } else {
            mx = max3_f32(mx, st[cb][nb][0], st[cb][nb][1]);
            mx = max3_f32(mx, st[cb][nb][2], st[cb][nb][3]);
          }
        }
      } else {
        mx = -INFINITY;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[cb][nb][r]);
        }
      }
      mx = xrow_max_swap(mx) * sc2;
      const float m_new = fmaxf(m[cb], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m[cb] - m_use);
      f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const f32x4 e = st[cb][nb] * sc2 - m_use;  // contracted to (packed) fma
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = fast_exp2(e[r]);
        st[cb][nb] = p;
        sum4 += p;
      }
      l[cb] = l[cb] * alpha + ((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));  // per-lane partial row sum; reduced across lq at the end
      m[cb] = m_new;
      if (!(OPT & 2) || !__all(alpha == 1.0f)) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) ot[cb][nb] *= alpha;
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
        for (int i = 0; i < 4; ++i)
          va[i] = (OPT & 32) ? lds_tr_pair(Vt + kk2 * 32 * kD + bt8[h4 * 4 + i]) : lds_tr_pair(Vt + kk2 * 32 * kD + (bt ^ ((h4 * 4 + i) << 4)));
        __builtin_amdgcn_sched_barrier(0);
        if (OPT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nb = h4 * 4 + i;
          ot[0][nb] = mfma16(va[i], pb0, ot[0][nb]);
          ot[1][nb] = mfma16(va[i], pb1, ot[1][nb]);
        }
        if (OPT & 1) __builtin_amdgcn_s_setprio(0);
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
