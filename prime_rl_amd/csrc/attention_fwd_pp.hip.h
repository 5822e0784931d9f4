// attention_fwd_pp.hip.h — EXPERIMENT, not the default: the forward attention kernel as an 8-wave ping-pong with a software-pipelined softmax
// (included by attention.hip after attention_kernels.hip.h, whose LDS image, fragment maps and softmax arithmetic it reuses UNCHANGED: outputs and lse
// are bit-identical to attn_fwd_kernel).  Reached only through pm_attn_debug_set_variant(0, 64 + OPT); measured SLOWER than the shipped kernel
// (0.42 vs 0.34 ms at the bench shape) — profiles/r03_attention.md has the phase-by-phase cycle counts and the reasons (the forward pass is bounded
// by VALU issue, and barrier-coupled phases leave only one wave per SIMD issuing it).  Kept in the tree as the A/B partner those numbers come from.
//
// Structure:
//   * a workgroup is 8 waves = two groups of four (one wave of each group per SIMD); a 64-row kv tile is EIGHT barrier-delimited phases per wave,
//         L1a/b  K fragments of tile j+1, one half of d each: LDS -> registers (8 ds_read_b128), one LDS-DMA piece of K_{j+3}
//         C1a/b  16 MFMAs  S_{j+1}^T += K_{j+1} . Q^T     with the FINISH of tile j's softmax (scale, exp2, row sums, bf16 pack, O rescale), row block a/b
//         L2a/b  V fragments of tile j, kv rows 0..31 / 32..63: LDS -> registers (16 transpose-reads), one LDS-DMA piece of V_{j+2}
//         C2a/b  16 MFMAs  O^T += V_j^T . P_j^T          with the START of tile j+1's softmax (row max, new running max, rescale factor), row block a/b
//     and group 1 runs ONE phase behind group 0: while one wave of a SIMD is in a matrix phase its partner is in a load phase
//     (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md T15 / T16);
//   * two score tiles are live per wave (S_j being exponentiated while S_{j+1} accumulates), statically named (the tile loop is unrolled by two);
//   * the two groups take two q HEADS of the same kv head over the same 128 q rows (GQA groups of even size: both waves of a SIMD have identical
//     work, tile for tile, including the causal early exit), or two consecutive 128-row blocks of one head otherwise;
//   * K and V tiles live in two 3-stage LDS rings (16 KiB per stage), filled by LDS-DMA: every wave waits for ITS pieces with a counted
//     `s_waitcnt vmcnt(6)` one phase before the first wave reads them.
#pragma once

constexpr int kPPTile = 64 * kD * 2;        // one K or V tile: [64][128] bf16
constexpr int kPPSmem = 6 * kPPTile;        // K ring [3] then V ring [3]

template <int IMM>
__device__ __forceinline__ u32x4 pp_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}
template <int IMM>
__device__ __forceinline__ u32x2 pp_tr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}
// The asm reads above are invisible to the compiler's waitcnt pass: after the explicit `s_waitcnt lgkmcnt(0)` every destination passes through an
// empty asm as a read-write operand, so no use can be scheduled above the wait.
__device__ __forceinline__ void pp_tie(u32x4& a, u32x4& b, u32x4& c, u32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void pp_tie(u32x2& a, u32x2& b, u32x2& c, u32x2& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

// Results of a softmax chunk pass through an empty volatile asm at the END of the chunk: without it the compiler sinks the whole chunk (pure arithmetic
// whose results are first used a phase later) below the fences and out of the matrix phase it is meant to hide in.
__device__ __forceinline__ void pp_pin(f32x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pp_pin(u32x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pp_pin(float& a) { asm volatile("" : "+v"(a)); }

__device__ __forceinline__ void pp_dma2(__amdgpu_buffer_rsrc_t r, unsigned char* dst, const int* off, int base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)dst, 16, off[0] + base, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(dst + 1024), 16, off[1] + base, 0, 0, 0);
}

// per-wave softmax state of one 16-row block of q rows (cb): running max m (scaled), partial row sum l, and what the START step hands to the FINISH step
struct PPRow {
  float m, l, m_use, alpha, mx;
};

// The softmax of a score tile in two steps of four chunks each, one chunk per group of 4 MFMAs of the matrix phase that hides it (the arithmetic of
// fwd_tile<3>, operation for operation).  START: row max of the raw scores, new running max, rescale factor.
template <int C>
__device__ __forceinline__ void pp_sm_start(const f32x4 (&s)[4], PPRow& r, float sc2) {
  if (C == 0) {
    float mx = max3_f32(s[0][0], s[0][1], s[0][2]);
    mx = fmax_raw(mx, s[0][3]);
    mx = max3_f32(mx, s[1][0], s[1][1]);
    r.mx = max3_f32(mx, s[1][2], s[1][3]);
    pp_pin(r.mx);
  } else if (C == 1) {
    float mx = max3_f32(r.mx, s[2][0], s[2][1]);
    mx = max3_f32(mx, s[2][2], s[2][3]);
    mx = max3_f32(mx, s[3][0], s[3][1]);
    r.mx = max3_f32(mx, s[3][2], s[3][3]);
    pp_pin(r.mx);
  } else if (C == 2) {
    r.mx = xrow_max_swap(r.mx) * sc2;
    pp_pin(r.mx);
  } else {
    const float m_new = fmaxf(r.m, r.mx);
    r.m_use = (m_new == -INFINITY) ? 0.f : m_new;
    r.alpha = fast_exp2(r.m - r.m_use);
    r.m = m_new;
    pp_pin(r.alpha);
    pp_pin(r.m_use);
    pp_pin(r.m);
  }
}
// FINISH: p = exp2(s * sc2 - m_use) in place, row-sum partial, O^T rescale, P packed to the B fragments of the two k-steps of P.V
template <int C, int OPT>
__device__ __forceinline__ void pp_sm_finish(f32x4 (&s)[4], PPRow& r, float sc2, f32x4& sum4, f32x4 (&o)[8], u32x4 (&pb)[2]) {
  if (C == 0) sum4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 e = s[C] * sc2 - r.m_use;
  f32x4 p;
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = (OPT & 4) ? e[i] : fast_exp2(e[i]);
  s[C] = p;
  sum4 += p;
  if (!(OPT & 8)) {
    o[2 * C] *= r.alpha;  // alpha == 1 is the common case after the first tiles: multiplying by one is exact
    o[2 * C + 1] *= r.alpha;
  }
  pp_pin(o[2 * C]);
  pp_pin(o[2 * C + 1]);
  if (C == 0 || C == 2) {
    pp_pin(s[C]);
    pp_pin(sum4);
  }
  if (C == 1) {
    pb[0] = pack_c_pair(s[0], s[1]);
    pp_pin(pb[0]);
    pp_pin(sum4);
  }
  if (C == 3) {
    pb[1] = pack_c_pair(s[2], s[3]);
    r.l = r.l * r.alpha + ((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));
    pp_pin(pb[1]);
    pp_pin(r.l);
  }
}

// OPT bits: 1 = s_setprio 1 around the matrix phases; timing ablations (garbage results): 4 = no v_exp, 8 = no O rescale, 16 = no QK^T MFMAs,
// 64 = no P.V MFMAs; 32 = phase timer: the waves
// of block 0 write, instead of their first lse values, the s_memtime cycles they spent in phases L1, C1, L2, C2 (each up to and including its closing
// barrier), their total, the s_memrealtime total and the tile count
template <int OPT>
__global__ __launch_bounds__(512, 1) void attn_fwd_pp_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                              int64_t v_stride, const int32_t* __restrict__ seq_start, int64_t n64, int hq, int hkv,
                                                              float scale, bf16_t* __restrict__ out, float* __restrict__ lse, int q_row0, int by_rows) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char pp_smem[];
  const int n = (int)n64;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = w >> 2, wl = w & 3;
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  int b = blockIdx.x;
  const int kvh = b % hkv;  // block b runs on XCD b % 8: with 8 kv heads every XCD keeps ONE head's K / V in its L2
  b /= hkv;
  int qh, qt0, wg_first, wg_rows;
  if (!by_rows) {  // the two wave groups: two q heads of this kv head over the same 128 rows
    const int pairs = group >> 1;
    const int pi = b % pairs;
    b /= pairs;
    const int num_tiles = gridDim.x / (hkv * pairs);
    wg_first = q_row0 + (num_tiles - 1 - b) * 128;  // the longest tiles first
    wg_rows = 128;
    qh = kvh * group + pi * 2 + g;
    qt0 = wg_first;
  } else {  // two consecutive 128-row blocks of one head
    const int gi = b % group;
    b /= group;
    const int num_tiles = gridDim.x / hq;
    wg_first = q_row0 + (num_tiles - 1 - b) * 256;
    wg_rows = 256;
    qh = kvh * group + gi;
    qt0 = wg_first + g * 128;
  }
  const int nq = n - q_row0;
  const int q0 = qt0 + wl * 32;
  const bool active = q0 < n;

  u32x4 qf[2][4];
  int qi[2], lo[2];
  PPRow row[2];
  f32x4 ot[2][8];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    qi[cb] = q0 + cb * 16 + lr;
    const int r = qi[cb] < n ? qi[cb] : n - 1;
    lo[cb] = seq_start[r];
    row[cb].m = -INFINITY;
    row[cb].l = 0.f;
    row[cb].m_use = 0.f;
    row[cb].alpha = 1.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[cb][kk] = ldg16(q + ((int64_t)(r - q_row0) * hq + qh) * kD + kk * 32 + lq * 8);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) ot[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int wg_last = (wg_first + wg_rows - 1 < n - 1) ? wg_first + wg_rows - 1 : n - 1;
  const int wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int kv_begin = (seq_start[wg_first] / 64) * 64;
  const int n_iters = (wg_last - kv_begin) / 64 + 1;
  const int j_hi = active ? (wave_last - kv_begin) / 64 : -1;  // the last tile any row of this wave sees (tiles before a row's document are masked)
  const float sc2 = scale * 1.44269504088896340736f;
  // rows past the end of the tensor read as zeros (hardware range check), which the causal / document mask discards like any other kv > q
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(k), 0, n * hkv * (kD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(v), 0, (int)(((int64_t)(n - 1) * v_stride + (int64_t)hkv * kD) * 2), 0x00020000);
  const int krow = hkv * (kD * 2), vrow = (int)v_stride * 2;
  int koff[2], voff[2];  // this wave's two 1 KiB pieces (4 tile rows each) of a K tile and of a V tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (w * 2 + i) * 4 + (lane >> 4);
    const int ch = ((lane & 15) ^ swz_f(r & 15)) << 4;
    koff[i] = r * krow + kvh * (kD * 2) + ch;
    voff[i] = r * vrow + kvh * (kD * 2) + ch;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)pp_smem;
  const unsigned bq2 = (unsigned)(lr * kD + ((lq ^ swz_f(lr)) << 3)) * 2;  // byte offset of this lane's K fragment chunk (k-step 0, block 0)
  const int r0 = lq * 4 + (lr >> 2);
  const unsigned bt2 = (unsigned)(r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2))) * 2;  // same for the transposed V fragments

  // Tiles past the last one are requested like any other (rows beyond the tensor read as zeros; nobody reads those stages): the ring needs no tail
  // cases and the counted waits stay constant.
  auto dma_k = [&](int stage, int kv0) { pp_dma2(krsrc, pp_smem + stage * kPPTile + w * 2048, koff, kv0 * krow); };
  auto dma_v = [&](int stage, int kv0) { pp_dma2(vrsrc, pp_smem + (3 + stage) * kPPTile + w * 2048, voff, kv0 * vrow); };
  auto read_k = [&](int stage, u32x4 (&kf)[4][4]) {
    const unsigned sa = lds0 + stage * kPPTile;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const unsigned ak = sa + (bq2 ^ (kk << 6));
      kf[kk][0] = pp_b128<0>(ak);
      kf[kk][1] = pp_b128<4096>(ak);
      kf[kk][2] = pp_b128<8192>(ak);
      kf[kk][3] = pp_b128<12288>(ak);
    }
  };
  auto mask_tile = [&](f32x4 (&s)[2][4], int kv0) {  // only for tiles some lane of the wave needs it for (diagonal / document boundary)
    if (__any((kv0 + 63 > q0) || (kv0 < lo[0]) || (kv0 < lo[1]))) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kv = kv0 + nb * 16 + lq * 4 + r;
            s[cb][nb][r] = ((kv <= qi[cb]) && (kv >= lo[cb])) ? s[cb][nb][r] : -INFINITY;
          }
    }
  };
  auto qk_mfmas = [&](const u32x4 (&kf)[4][4], f32x4 (&s)[2][4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        s[0][nb] = mfma16(kf[kk][nb], qf[0][kk], kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : s[0][nb]);
        s[1][nb] = mfma16(kf[kk][nb], qf[1][kk], kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : s[1][nb]);
      }
  };

  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = 0, t_begin = 0, r_begin = 0;
#define PP_TICK(i)                                               \
  if (OPT & 32) {                                                \
    const unsigned long long now = __builtin_readcyclecounter(); \
    tacc[i] += now - tlast;                                      \
    tlast = now;                                                 \
  }
#define PP_BARRIER()                     \
  __builtin_amdgcn_sched_barrier(0);     \
  __builtin_amdgcn_s_barrier();          \
  __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: the ring's first requests in the order the loop continues (K_{j+3} in L1 of tile j, V_{j+2} in L2), K_0's fragments, S_0, its row max
  dma_k(0, kv_begin);
  dma_k(1, kv_begin + 64);
  dma_v(0, kv_begin);
  dma_k(2, kv_begin + 128);
  dma_v(1, kv_begin + 64);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    pin_loaded(lo[cb]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pin_loaded(qf[cb][kk]);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // K_0
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one phase behind the first
  f32x4 sA[2][4], sB[2][4];
  {
    u32x4 kf[4][4];
    if (j_hi >= 0) read_k(0, kf);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K_1 (read by the first group in L1 of tile 0, one phase after the second group passes this wait)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pp_tie(kf[kk][0], kf[kk][1], kf[kk][2], kf[kk][3]);
    PP_BARRIER();
    if (j_hi >= 0) {
      qk_mfmas(kf, sA);
      mask_tile(sA, kv_begin);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        pp_sm_start<0>(sA[cb], row[cb], sc2);
        pp_sm_start<1>(sA[cb], row[cb], sc2);
        pp_sm_start<2>(sA[cb], row[cb], sc2);
        pp_sm_start<3>(sA[cb], row[cb], sc2);
      }
    }
    PP_BARRIER();
  }
  if (OPT & 32) {
    t_begin = tlast = __builtin_readcyclecounter();
    r_begin = __builtin_amdgcn_s_memrealtime();
  }

  int a = 0;  // j % 3
  // One kv tile = EIGHT phases per wave: each matrix phase is 16 MFMAs on HALF of the contraction (d 0..63 / 64..127 of K, kv rows 0..31 / 32..63 of V)
  // with 32 fragment registers, so that two score tiles, the O^T accumulators and the Q fragments fit the 256-register budget of two waves per SIMD.
  auto tile = [&](f32x4 (&sC)[2][4], f32x4 (&sN)[2][4], int j) {
    const int kv0 = kv_begin + 64 * j;
    // A wave runs ALL FOUR steps for every tile up to its last visible one: the score tile it computes ahead of the last one lies entirely above
    // the diagonal, so the mask turns it into -inf and its softmax start leaves (m, alpha) unchanged — one path, no tail case (cost: 32 MFMAs).
    const bool do_pv = j <= j_hi, do_qk = do_pv;
    const int a1 = a == 2 ? 0 : a + 1, a2 = a == 0 ? 2 : a - 1;  // (j + 1) % 3, (j + 2) % 3
    u32x4 pb[2][2];
    const unsigned ka = lds0 + a1 * kPPTile, va = lds0 + (3 + a) * kPPTile;
    unsigned char* kdst = pp_smem + a * kPPTile + w * 2048;         // K_{j+3}: the stage that held K_j (last read one tile ago)
    unsigned char* vdst = pp_smem + (3 + a2) * kPPTile + w * 2048;  // V_{j+2}: the stage that held V_{j-1}
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // ---------------- S_{j+1}^T = K_{j+1} . Q^T over d half h  ||  finish the softmax of tile j, row block h
      u32x4 kf[2][4];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, (lds_void_ptr)(kdst + h * 1024), 16, koff[h] + (kv0 + 192) * krow, 0, 0, 0);
      if (do_qk) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const unsigned ak = ka + (bq2 ^ ((h * 2 + k2) << 6));
          kf[k2][0] = pp_b128<0>(ak);
          kf[k2][1] = pp_b128<4096>(ak);
          kf[k2][2] = pp_b128<8192>(ak);
          kf[k2][3] = pp_b128<12288>(ak);
        }
      }
      if (h == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // this wave's pieces of V_j have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_tie(kf[0][0], kf[0][1], kf[0][2], kf[0][3]);
      pp_tie(kf[1][0], kf[1][1], kf[1][2], kf[1][3]);
      PP_BARRIER();
      PP_TICK(0)
      f32x4 sum4;
#define PP_QK(C_)                                                                                                             \
  {                                                                                                                           \
    const int k2 = (C_) >> 1, kk = h * 2 + k2;                                                                                \
    _Pragma("unroll") for (int nb = 2 * ((C_) & 1); nb < 2 * ((C_) & 1) + 2; ++nb) {                                         \
      if (OPT & 16) { if (kk == 0) sN[0][nb] = sN[1][nb] = f32x4{0.f, 1.f, 2.f, 3.f}; asm volatile("" :: "v"(kf[k2][nb])); continue; }     \
      sN[0][nb] = mfma16(kf[k2][nb], qf[0][kk], kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sN[0][nb]);                             \
      sN[1][nb] = mfma16(kf[k2][nb], qf[1][kk], kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sN[1][nb]);                             \
    }                                                                                                                         \
  }
      if (do_qk) {  // four chunks of [4 MFMAs][a quarter of the softmax finish], fenced: the VALU work issues while the matrix pipe runs the MFMAs
        if (OPT & 1) __builtin_amdgcn_s_setprio(1);
        PP_QK(0) pp_sm_finish<0, OPT>(sC[h], row[h], sc2, sum4, ot[h], pb[h]);
        __builtin_amdgcn_sched_barrier(0);
        PP_QK(1) pp_sm_finish<1, OPT>(sC[h], row[h], sc2, sum4, ot[h], pb[h]);
        __builtin_amdgcn_sched_barrier(0);
        PP_QK(2) pp_sm_finish<2, OPT>(sC[h], row[h], sc2, sum4, ot[h], pb[h]);
        __builtin_amdgcn_sched_barrier(0);
        PP_QK(3) pp_sm_finish<3, OPT>(sC[h], row[h], sc2, sum4, ot[h], pb[h]);
        if (OPT & 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (h == 1) mask_tile(sN, kv0 + 64);
      }
#undef PP_QK
      PP_BARRIER();
      PP_TICK(1)
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // ---------------- O^T += V_j^T . P_j^T over kv rows half h  ||  start the softmax of tile j+1, row block h
      u32x2 vlo[8], vhi[8];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void_ptr)(vdst + h * 1024), 16, voff[h] + (kv0 + 128) * vrow, 0, 0, 0);
      if (do_pv) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const unsigned av = va + (bt2 ^ (nb << 5));
          if (h == 0) {
            vlo[nb] = pp_tr<0>(av);
            vhi[nb] = pp_tr<4096>(av);
          } else {
            vlo[nb] = pp_tr<8192>(av);
            vhi[nb] = pp_tr<12288>(av);
          }
        }
      }
      if (h == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // this wave's pieces of K_{j+2} have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int nb = 0; nb < 8; nb += 2) pp_tie(vlo[nb], vhi[nb], vlo[nb + 1], vhi[nb + 1]);
      PP_BARRIER();
      PP_TICK(2)
#define PP_PV(C_)                                                                                        \
  _Pragma("unroll") for (int nb = 2 * (C_); nb < 2 * (C_) + 2; ++nb) {                                  \
    const u32x4 vfrag = u32x4{vlo[nb].x, vlo[nb].y, vhi[nb].x, vhi[nb].y};                               \
    if (OPT & 64) { asm volatile("" :: "v"(vfrag), "v"(pb[0][h]), "v"(pb[1][h])); continue; }            \
    ot[0][nb] = mfma16(vfrag, pb[0][h], ot[0][nb]);                                                      \
    ot[1][nb] = mfma16(vfrag, pb[1][h], ot[1][nb]);                                                      \
  }
      if (do_qk) {
        if (OPT & 1) __builtin_amdgcn_s_setprio(1);
        PP_PV(0) pp_sm_start<0>(sN[h], row[h], sc2);
        __builtin_amdgcn_sched_barrier(0);
        PP_PV(1) pp_sm_start<1>(sN[h], row[h], sc2);
        __builtin_amdgcn_sched_barrier(0);
        PP_PV(2) pp_sm_start<2>(sN[h], row[h], sc2);
        __builtin_amdgcn_sched_barrier(0);
        PP_PV(3) pp_sm_start<3>(sN[h], row[h], sc2);
        if (OPT & 1) __builtin_amdgcn_s_setprio(0);
      }
#undef PP_PV
      PP_BARRIER();
      PP_TICK(3)
    }
    a = a1;
  };
  for (int j = 0; j < n_iters; j += 2) {
    tile(sA, sB, j);
    if (j + 1 < n_iters) tile(sB, sA, j + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing requests must have landed before this workgroup's LDS is handed on
  if (g == 0) __builtin_amdgcn_s_barrier();
  if (!active) return;
  if ((OPT & 32) && blockIdx.x == 0) {
    const unsigned long long t_end = __builtin_readcyclecounter(), r_end = __builtin_amdgcn_s_memrealtime();
    if (lane < 7) {
      const unsigned long long val = lane == 0 ? tacc[0] : lane == 1 ? tacc[1] : lane == 2 ? tacc[2] : lane == 3 ? tacc[3] : lane == 4 ? t_end - t_begin
                                     : lane == 5 ? r_end - r_begin : (unsigned long long)n_iters;
      lse[(int64_t)qh * nq + q0 - q_row0 + lane] = (float)val;
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const float lsum = xrow_sum(row[cb].l);
    if (qi[cb] < n) {
      const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
      bf16_t* op = out + ((int64_t)(qi[cb] - q_row0) * hq + qh) * kD + lq * 4;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) st8_bf16(op + nb * 16, ot[cb][nb], inv);
      if (lq == 0) lse[(int64_t)qh * nq + qi[cb] - q_row0] = row[cb].m * 0.69314718055994530942f + logf(lsum);
    }
  }
#undef PP_TICK
#undef PP_BARRIER
}
