// attention_kernels.hip.h — the three varlen GQA attention kernels (forward, dK/dV, dQ) for gfx950.  Earlier generations (v1 correct-first,
// v2 LDS round trip, v3 padded tiles + transposed global copies, v4 register-staged tiles) are in the git history; what each step bought
// is recorded in DESIGN.md §3 and profiles/r01_attention_pmc.md, profiles/r02_attention.md.
//
// LDS scheme shared by all three:
//   * every tile (K, V, Q, dO) is staged ONCE, row-major, unpadded ([rows][128] bf16, 256-byte rows) with a 16-byte-chunk XOR swizzle
//     chunk' = chunk ^ f(row & 15), f(r) = ((r3^r2)<<3 | r1<<2 | r0<<1 | r2), chosen so that all access patterns are conflict-free:
//     the ds_read_b128 row reads (lane groups {0-3,12-15,20-27}, ...: two lanes collide iff f(a)^f(b) = 1 iff a^b = 12, which never
//     pairs lanes of one group) and the transpose-reads (a 32-lane half touches 8 rows x 32 bytes; (f>>1) is injective over those
//     rows, so the halves tile all 64 banks exactly) — SQ_LDS_BANK_CONFLICT = 0 on all three kernels;
//   * operands whose contraction runs along the token axis (V^T in P.V, Q^T / dO^T in dK/dV, K^T in dQ) are read from that same image
//     with `ds_read_b64_tr_b16` (semantics pinned by tests/test_kernels_gpu.py::test_ds_read_tr16_semantics): no transposed copies;
//   * tiles arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write_b128 pass (13 LDS-path cycles per
//     wave-instruction), no 64-bit address arithmetic, and rows past the end of a tensor read as zero through the buffer range check
//     instead of per-row clamps.  A 1 KiB DMA piece is lane-linear (LDS byte = base + lane * 16 = 4 tile rows x 16 chunks), so the
//     swizzle moves to the SOURCE side: lane (row, chunk c') fetches chunk c' ^ f(row & 15);
//   * the two LDS buffers of each kernel are two distinct static arrays and the tile loop is unrolled by two, on purpose: hipcc guards
//     every `ds_read_b64_tr_b16` that MAY alias a pending LDS-DMA with `s_waitcnt vmcnt(0)` (plain ds_reads are not guarded), which
//     with one dynamically indexed array made every iteration wait for the NEXT tile's DMA in the middle of its own compute.  With the
//     read buffer and the DMA target in different arrays the alias query is answered at compile time and the prefetch stays in flight.
//     The wait that IS needed — this wave's pieces of the current tile, before the barrier — hipcc does not emit for LDS-DMA, hence the
//     explicit `s_waitcnt vmcnt(0)` in front of every __syncthreads().
#pragma once

// ---- the transposed-product formulation shared by all three kernels ----
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
// lse = m ln 2 + ln(sum) from the base-2 running maximum and the row sum, as ONE explicit fma (left to the compiler's contraction choice the two forward
// kernels — attn_fwd_kernel and attn_fwd_p3_kernel — came out one ulp apart)
__device__ __forceinline__ float lse_of(float m2, float lsum) { return __builtin_fmaf(m2, 0.69314718055994530942f, logf(lsum)); }
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
typedef __attribute__((address_space(3))) void* lds_void_ptr;

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
// v_max3_f32 / v_max_f32 as inline asm: a plain fmaxf on an MFMA output gets a canonicalising v_max_f32 in front of it (32 extra VALU per tile)
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
// Epilogue of the transposed products: a lane (lr = row, lq) holds, of every 16-wide block nb of its row, the 4 values d = nb*16 + lq*4 .. +3 — stored
// directly that is 8 x 8-byte stores per row (st8_bf16).  16-byte form: the lanes lq = 2a and 2a + 1 hold d = 8a .. 8a+3 and 8a+4 .. 8a+7 of a block, so for
// a PAIR of blocks the even lane hands its part of the second block to its partner and receives the partner's part of the first (v_permlane16_swap: odd
// 16-lane rows of one register <-> even rows of the other): 4 x 16-byte stores per lane instead of 8 x 8 — the store tail of a workgroup is store-ISSUE
// bound (MI355X_MICROARCH.md).  ALL lanes must call it (cross-lane swaps); `ok` = this lane's row exists; row0 = address of d = 0 of the row.
__device__ __forceinline__ void store_row_blocks16(bf16_t* row0, const f32x4 (&acc)[8], float mul, int lq, bool ok) {
  bf16_t* op = row0 + (lq >> 1) * 8 + (lq & 1) * 16;
#pragma unroll
  for (int nb = 0; nb < 8; nb += 2) {
    const f32x4 &a = acc[nb], &b = acc[nb + 1];
    float x0 = __uint_as_float(pack_bf16_hw(a[0] * mul, a[1] * mul)), x1 = __uint_as_float(pack_bf16_hw(a[2] * mul, a[3] * mul));
    float y0 = __uint_as_float(pack_bf16_hw(b[0] * mul, b[1] * mul)), y1 = __uint_as_float(pack_bf16_hw(b[2] * mul, b[3] * mul));
    permlane16_swap(x0, y0);  // even lanes: (x, y) = (own part of block nb, partner's part of it); odd lanes: (partner's part of block nb + 1, own part)
    permlane16_swap(x1, y1);
    if (ok) *reinterpret_cast<u32x4*>(op + nb * 16) = u32x4{__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(y0), __float_as_uint(y1)};
  }
}

// The whole row-max reduction of one 16-row block of a 64-row score tile as ONE asm statement: max over the lane's 16 raw scores (v_max3 chain), then
// across the four 16-lane rows of the wave (v_permlane32_swap / v_permlane16_swap).  As separate statements (max3_f32 / fmax_raw / xrow_max_swap) the same
// work costs about twice the issue slots: hipcc pads every inline-asm statement with an `s_nop`, and the fmaxf calls between the swaps get canonicalising
// `v_max_f32 x, x` in front of them — on a kernel that is bounded by VALU issue (profiles/r03_attention.md).  Same values, operation for operation.
// Also in the block: mx * sc2, the new running maximum max(m_old, mx * sc2), and the value the exponent subtracts — m_new clamped to -FLT_MAX, which
// stands in for the "(m_new == -inf) ? 0 : m_new" select (two compares' worth of issue slots and a vcc hazard less): while every score so far is -inf,
// exp2(-inf - (-FLT_MAX)) = exp2(-inf) = 0 exactly like exp2(-inf - 0), for the scores and for the rescale factor alike; a finite maximum is unchanged.
__device__ __forceinline__ void row_max16_xrow(const f32x4& s0, const f32x4& s1, const f32x4& s2, const f32x4& s3, float m_old, float sc2, float& m_new,
                                               float& m_use) {
  float mx, t;
  asm volatile(
      "v_max3_f32 %0, %2, %3, %4\n\t"
      "v_max_f32 %0, %0, %5\n\t"
      "v_max3_f32 %0, %0, %6, %7\n\t"
      "v_max3_f32 %0, %0, %8, %9\n\t"
      "v_max3_f32 %0, %0, %10, %11\n\t"
      "v_max3_f32 %0, %0, %12, %13\n\t"
      "v_max3_f32 %0, %0, %14, %15\n\t"
      "v_max3_f32 %0, %0, %16, %17\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "s_nop 1\n\t"
      "v_max_f32 %0, %0, %1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %1\n\t"
      "s_nop 1\n\t"
      "v_max_f32 %0, %0, %1\n\t"
      "v_mul_f32 %0, %0, %19\n\t"
      "v_max_f32 %0, %18, %0\n\t"
      "v_max_f32 %1, 0xff7fffff, %0"
      : "=&v"(mx), "=&v"(t)
      : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]), "v"(s2[0]), "v"(s2[1]), "v"(s2[2]), "v"(s2[3]),
        "v"(s3[0]), "v"(s3[1]), "v"(s3[2]), "v"(s3[3]), "v"(m_old), "v"(sc2));
  m_new = mx;
  m_use = t;
}

__device__ __forceinline__ float xrow_max_swap(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);  // a = [v.lo, v.lo], b = [v.hi, v.hi]
  a = fmaxf(a, b);
  b = a;
  permlane16_swap(a, b);  // a = [r0, r0, r2, r2], b = [r1, r1, r3, r3]
  return fmaxf(a, b);
}

// ---------------------------------------------------------------------------------------------------------------------
// dK/dV: 4 waves x 16 kv rows per workgroup (two workgroups per CU), 64-row q tiles walked per (q tile, head) from the LAST tile down.
//   * K / V fragments of the wave's 16 kv rows live in registers for the whole kernel; Q / dO tiles (and the lse / delta / seq_start
//     rows of the tile: one wave-wide dword DMA each, issued by waves 0 / 1 / 2) are DMA'd one tile ahead;
//   * the S accumulators START at -lse / scale and the dP accumulators at -delta (rows prepared by attn_delta_kernel), so that
//     P = exp2(acc_S * scale * log2 e) and dS / scale = P * acc_dP: no zero-fill, no per-score subtraction of the row statistics;
//     the softmax scale of dS is applied once, in the final dK store;
//   * a tile is fully visible to a wave when every q row is at or past the wave's last kv row, inside the tensor, and its document
//     started at or before the wave's first kv row (seq_start is non-decreasing: the tile's last row decides) — then the two
//     compares, the AND and the select per score are skipped as well.  VALU instructions per (64 x 16) tile: 260 (round 1) -> ~100.
// One operand pair of a tile by LDS-DMA: this wave's PPW one-KiB pieces of tensors A and B (byte offsets off[i] + base into each buffer)
// land at dst + i * 1024 and dst + b_shift + i * 1024.  (A free function with plain value / pointer parameters on purpose: with the builtin
// inside a lambda of the kernel template, hipcc's host pass fails the substitution silently and drops the kernel stub.)
template <int PPW>
__device__ __forceinline__ void dma_pieces(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, unsigned char* dst, int b_shift, const int* off_a,
                                           const int* off_b, int base_a, int base_b) {
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    // the tile's base offset rides in the instruction's SCALAR offset: no per-piece vector add (the kernels are bounded by VALU issue)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(dst + i * 1024), 16, off_a[i], base_a, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(dst + b_shift + i * 1024), 16, off_b[i], base_b, 0, 0);
  }
}

// LDS image of one q tile: Q [QT][128] bf16, dO [QT][128] bf16, then lse / delta / seq_start rows (64 each: a wave-wide dword DMA fills 64).
template <int QT>
constexpr int kV5DkdvBufBytes = 2 * QT * kD * 2 + 3 * 64 * 4;

// One (q tile, head) step of the dK/dV kernel for one wave: S and dP from the row-major fragments, P and dS in registers, then
// dV^T += dO^T . P and dK^T += Q^T . dS through transpose-reads of the same tiles.  The S accumulators start at -lse / scale and the dP
// accumulators at -delta (rows prepared by attn_delta_kernel and DMA'd next to the tiles), so that P = exp2(acc_S * scale * log2 e) and
// dS / scale = P * acc_dP: no zero-fill, no per-score subtraction of the row statistics.
template <int FB, int QT, int KVB, int ABL = 0>  // ABL (make VARIANTS=1 only; WRONG results): 1 = no exp / mask VALU, 2 = no fragment reads, 4 = no LDS-DMA, 8 = no MFMAs
__device__ __forceinline__ void dkdv_tile(const unsigned char* sm, int qi0, int kvb, int kvi, unsigned nk_u, int n, float sc2, int bq, int bt, int lq,
                                          const u32x4 (&kf)[KVB][4], const u32x4 (&vf)[KVB][4], f32x4 (&dkt)[KVB][8], f32x4 (&dvt)[KVB][8]) {
  // KVB = 16-row kv blocks per wave: every Q / dO fragment read from LDS feeds KVB MFMAs (the kernel is bounded by the LDS port at KVB = 1:
  // 1 KiB of fragment reads per 16-cycle MFMA and wave = twice what the port delivers to four SIMDs).  kvi / nk_u are those of block 0.
  constexpr int TILE = QT * kD;
  constexpr int kDO = TILE;  // element offset of the dO tile
  const bf16_t* Qt = reinterpret_cast<const bf16_t*>(sm);
  const float* nLs = reinterpret_cast<const float*>(sm + 2 * TILE * 2);
  const float* nDs = nLs + 64;
  const int* Los = reinterpret_cast<const int*>(nLs + 128);
  // fully visible tile (wave-uniform; read first so that its LDS wait does not sit behind the fragment reads)
  const int lo_last = __builtin_amdgcn_readfirstlane(Los[QT - 1]);
  const bool full = (qi0 >= kvb + 16 * KVB - 1) && (qi0 + QT <= n) && (lo_last <= kvb);
  f32x4 s[KVB][QT / 16], dp[KVB][QT / 16];
#pragma unroll
  for (int nb = 0; nb < QT / 16; ++nb) {
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(nLs + nb * 16 + lq * 4);
    const f32x4 d0 = *reinterpret_cast<const f32x4*>(nDs + nb * 16 + lq * 4);
#pragma unroll
    for (int b = 0; b < KVB; ++b) {
      s[b][nb] = s0;
      dp[b][nb] = d0;
    }
  }
  // k-step outermost: the 2 * KVB * QT/16 accumulators of one batch are independent, so consecutive MFMAs never wait for each other
  __builtin_amdgcn_s_setprio(1);  // (as in fwd_tile: the matrix phases win the issue slot over the partner wave's exp stream; backward 1.017 -> 0.994 ms)
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    u32x4 qa[QT / 16], da[QT / 16];
#pragma unroll
    for (int nb = 0; nb < QT / 16; ++nb) {
      const bf16_t* pa = Qt + (bq ^ (kk << 5)) + nb * 16 * kD;
      if constexpr (ABL & 2) {
        qa[nb] = kf[0][kk];
        da[nb] = vf[0][kk];
      } else {
        qa[nb] = lds16(pa);
        da[nb] = lds16(pa + kDO);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < QT / 16; ++nb)
#pragma unroll
      for (int b = 0; b < KVB; ++b) {
        if constexpr (ABL & 8) {
          s[b][nb][0] += __uint_as_float(qa[nb].x ^ kf[b][kk].y);
          dp[b][nb][1] += __uint_as_float(da[nb].z ^ vf[b][kk].w);
        } else {
          s[b][nb] = mfma16(qa[nb], kf[b][kk], s[b][nb]);
          dp[b][nb] = mfma16(da[nb], vf[b][kk], dp[b][nb]);
        }
      }
  }
  __builtin_amdgcn_s_setprio(0);
  if constexpr (ABL & 1) {
  } else if (full) {
#pragma unroll
    for (int b = 0; b < KVB; ++b)
#pragma unroll
      for (int nb = 0; nb < QT / 16; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(s[b][nb][r] * sc2);
          s[b][nb][r] = p;
          dp[b][nb][r] *= p;  // dS / scale
        }
  } else {
#pragma unroll
    for (int nb = 0; nb < QT / 16; ++nb) {
      const u32x4 lo4 = *reinterpret_cast<const u32x4*>(Los + nb * 16 + lq * 4);
#pragma unroll
      for (int b = 0; b < KVB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qg = qi0 + nb * 16 + lq * 4 + r;
          const int kvi_b = kvi + 16 * b;
          const bool ok = ((unsigned)(qg - kvi_b) < nk_u - 16u * b) && (kvi_b >= (int)lo4[r]);
          const float p = ok ? fast_exp2(s[b][nb][r] * sc2) : 0.f;
          s[b][nb][r] = p;
          dp[b][nb][r] *= p;
        }
    }
  }
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int h = 0; h < QT / 32; ++h) {  // 32-row halves of the q tile: one packed P / dS operand per kv block each
    u32x4 pb[KVB], sb[KVB];
#pragma unroll
    for (int b = 0; b < KVB; ++b) {
      pb[b] = pack_c_pair(s[b][2 * h], s[b][2 * h + 1]);
      sb[b] = pack_c_pair(dp[b][2 * h], dp[b][2 * h + 1]);
    }
#pragma unroll
    for (int n0 = 0; n0 < 8; n0 += FB) {
      u32x4 doa[FB], qa[FB];
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const bf16_t* pt = Qt + (bt ^ ((n0 + i) << 4)) + h * 32 * kD;
        if constexpr (ABL & 2) {
          qa[i] = kf[0][i & 3];
          doa[i] = vf[0][i & 3];
        } else {
          qa[i] = lds_tr_pair(pt);
          doa[i] = lds_tr_pair(pt + kDO);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FB; ++i)
#pragma unroll
        for (int b = 0; b < KVB; ++b) {
          if constexpr (ABL & 8) {
            dvt[b][n0 + i][0] += __uint_as_float(doa[i].x ^ pb[b].y);
            dkt[b][n0 + i][1] += __uint_as_float(qa[i].z ^ sb[b].w);
          } else {
            dvt[b][n0 + i] = mfma16(doa[i], pb[b], dvt[b][n0 + i]);
            dkt[b][n0 + i] = mfma16(qa[i], sb[b], dkt[b][n0 + i]);
          }
        }
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// The two LDS buffers are two distinct static arrays and the tile loop is unrolled by two, on purpose: hipcc guards every
// `ds_read_b64_tr_b16` that MAY alias a pending LDS-DMA with `s_waitcnt vmcnt(0)` (plain ds_reads are not guarded), which with one dynamic
// array made every iteration wait for the NEXT tile's DMA in the middle of its own compute.  With the read buffer and the DMA target in
// different arrays the alias query is answered at compile time and the prefetch stays in flight.
template <int MINB, int FB, int QT, int KVB = 1, int ABL = 0>
__global__ __launch_bounds__(256, MINB) void attn_bwd_dkdv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t v_stride, const bf16_t* __restrict__ dout,
    const float* __restrict__ nlse, const float* __restrict__ ndelta, const int32_t* __restrict__ seq_start, int64_t n64, int hq, int hkv,
    float scale, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dv_stride, int q_row0, int narrow_stores) {
  // nlse = -lse / scale, ndelta = -delta, both [hq][nq] (attn_delta_kernel)
  // Row window (context parallelism with gathered K / V, pm_attn_varlen_*_rows): k, v, seq_start, dk, dv hold rows [0, n) of the pass, the q-side
  // arrays (q, dout, nlse, ndelta) only the rows [q_row0, n) this rank owns, addressed as row - q_row0; q_row0 is a multiple of 128, so no q
  // tile straddles it.  q_row0 = 0 is the ordinary call.
  constexpr int NT = 256;
  constexpr int ROWS = 64 * KVB;     // kv rows per workgroup (4 waves x KVB blocks of 16)
  constexpr int PPW = QT / 16;       // 1 KiB DMA pieces per wave per operand tile
  static_assert(QT % 32 == 0, "q tile is consumed in 32-row halves");
  constexpr int TILE = QT * kD;      // elements of one [QT][128] tile
  __shared__ __attribute__((aligned(16))) unsigned char smA[kV5DkdvBufBytes<QT>];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kV5DkdvBufBytes<QT>];
  __shared__ int s_ntiles;
  const int n = (int)n64;
  const int t = threadIdx.x;
  const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform: LDS-DMA destinations (m0) stay on the scalar unit
  const int lr = lane & 15, lq = lane >> 4;
  const int group = hq / hkv;
  const int kvh = blockIdx.x % hkv;
  const int kv0 = (blockIdx.x / hkv) * ROWS;
  const int kvb = kv0 + w * 16 * KVB;
  const bool active = kvb < n;
  const int kvi = kvb + lr;
  u32x4 kf[KVB][4], vf[KVB][4];
  f32x4 dkt[KVB][8], dvt[KVB][8];
#pragma unroll
  for (int b = 0; b < KVB; ++b) {
    const int row = kvi + 16 * b < n ? kvi + 16 * b : n - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[b][kk] = ldg16(k + ((int64_t)row * hkv + kvh) * kD + kk * 32 + lq * 8);
      vf[b][kk] = ldg16(v + (int64_t)row * v_stride + (int64_t)kvh * kD + kk * 32 + lq * 8);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dkt[b][nb] = dvt[b][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int wg_last = (kv0 + ROWS - 1 < n - 1) ? kv0 + ROWS - 1 : n - 1;
  const float sc2 = scale * 1.44269504088896340736f;
  const int nq = n - q_row0;
  const int tile_lo = q_row0 > kv0 ? (q_row0 - kv0) / QT : 0;  // q tiles below the row window belong to other ranks
  if (t == 0) s_ntiles = 0;
  __syncthreads();
  {
    int mine = 0;
    for (int c = tile_lo + t; kv0 + QT * c < n; c += NT)
      if (seq_start[kv0 + QT * c] <= wg_last) mine = c + 1;
    if (mine) atomicMax(&s_ntiles, mine);
  }
  __syncthreads();
  const int ntiles = __builtin_amdgcn_readfirstlane(s_ntiles);
  const int total_iters = ntiles > tile_lo ? (ntiles - tile_lo) * group : 0;

  // execution order j = 0.. walks (q tile, head) pairs from the LAST q tile down (workgroups of one document stay in step on L2)
  struct Cursor {
    int tile, head;
  };
  auto advance = [&](Cursor& c) {
    if (--c.head < 0) {
      c.head = group - 1;
      --c.tile;
    }
  };
  Cursor c_issue{ntiles - 1, group - 1}, c_cur = c_issue;
  const int qbytes = nq * hq * (kD * 2);  // host guarantees < 2 GiB
  const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(q), 0, qbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dorsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dout), 0, qbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nlse), 0, hq * nq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ndelta), 0, hq * nq * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(seq_start), 0, n * 4, 0x00020000);
  int voff[PPW];  // loop-invariant byte offset of this lane's chunk inside a q tile (source-side swizzle)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (w * PPW + i) * 4 + (lane >> 4);
    voff[i] = row * hq * (kD * 2) + (((lane & 15) ^ swz_f(row & 15)) << 4);
  }
  auto dma = [&](unsigned char* sm) {
    const int qi0 = kv0 + QT * c_issue.tile;
    const int qh = kvh * group + c_issue.head;
    const int base = ((qi0 - q_row0) * hq + qh) * (kD * 2);
    if constexpr ((ABL & 4) == 0) dma_pieces<PPW>(qrsrc, dorsrc, sm + w * PPW * 1024, TILE * 2, voff, voff, base, base);
    const int roff = (qh * nq + qi0 - q_row0 + lane) * 4;
    unsigned char* rows = sm + 2 * TILE * 2;
    if (w == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(lrsrc, (lds_void_ptr)rows, 4, roff, 0, 0, 0);
    else if (w == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void_ptr)(rows + 256), 4, roff, 0, 0, 0);
    else if (w == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_void_ptr)(rows + 512), 4, (qi0 + lane) * 4, 0, 0, 0);
    advance(c_issue);
  };
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));
  const unsigned nk_u = (unsigned)(n - kvi);  // q row qg is inside the tensor and at or past this lane's kv row iff (unsigned)(qg - kvi) < nk_u
  if (total_iters > 0) dma(smA);
#pragma unroll
  for (int b = 0; b < KVB; ++b)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      pin_loaded(kf[b][kk]);
      pin_loaded(vf[b][kk]);
    }
  auto step = [&](const unsigned char* rd, unsigned char* wr, int j) {
    const int qi0 = kv0 + QT * c_cur.tile;
    advance(c_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile j have landed ...
    __syncthreads();                                  // ... so have everyone's, and compute(j-1) is finished everywhere
    if (j + 1 < total_iters) dma(wr);
    if (active && qi0 + QT - 1 >= kvb) dkdv_tile<FB, QT, KVB, ABL>(rd, qi0, kvb, kvi, nk_u, n, sc2, bq, bt, lq, kf, vf, dkt, dvt);
  };
  for (int j = 0; j < total_iters; j += 2) {
    step(smA, smB, j);
    if (j + 1 < total_iters) step(smB, smA, j + 1);
  }
  if (!active) return;  // wave-uniform
#pragma unroll
  for (int b = 0; b < KVB; ++b) {
    const int kvr = kvi + 16 * b;
    if (!narrow_stores) {
      const int row = kvr < n ? kvr : n - 1;
      store_row_blocks16(dk + ((int64_t)row * hkv + kvh) * kD, dkt[b], scale, lq, kvr < n);
      store_row_blocks16(dv + (int64_t)row * dv_stride + (int64_t)kvh * kD, dvt[b], 1.0f, lq, kvr < n);
      continue;
    }
    if (kvr >= n) continue;
    bf16_t* pk_ = dk + ((int64_t)kvr * hkv + kvh) * kD + lq * 4;
    bf16_t* pv_ = dv + (int64_t)kvr * dv_stride + (int64_t)kvh * kD + lq * 4;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      st8_bf16(pk_ + nb * 16, dkt[b][nb], scale);
      st8_bf16(pv_ + nb * 16, dvt[b][nb], 1.0f);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// dQ: 4 waves x 32 q rows (two 16-wide column blocks per wave), K / V tiles of KT rows DMA'd one tile ahead and consumed in 32-row halves.
// A fragments of S^T = K.Q^T and dP^T = V.dO^T are ds_read_b128 row reads, the A fragments of dQ^T[d][q] += K^T[d][kv] . dS^T[kv][q]
// transpose-reads of the SAME K tile.  Accumulator starts (-lse / scale, -delta of the lane's q column), the wave-uniform fast path
// for fully visible halves and the scale folded into the final store are those of the dK/dV kernel.
template <int KT>
constexpr int kV5DqBufBytes = 2 * KT * kD * 2;  // K [KT][128] then V [KT][128]

template <int KT>
__device__ __forceinline__ void dq_tile(const unsigned char* sm, int kvt, int q0, int wave_last, int lo_max, bool whole, float sc2, int bq, int bt, int lq,
                                        const u32x4 (&qf)[2][4], const u32x4 (&dof)[2][4], const int (&hi)[2], const int (&lo)[2],
                                        const float (&nL)[2], const float (&nD)[2], f32x4 (&dqt)[2][8]) {
  constexpr int kV = KT * kD;  // element offset of the V tile
#pragma unroll
  for (int half = 0; half < KT / 32; ++half) {
    const int kv0 = kvt + 32 * half;
    if (kv0 > wave_last) break;
    const bf16_t* Kt = reinterpret_cast<const bf16_t*>(sm) + half * 32 * kD;
    f32x4 st[2][2], dpt[2][2];  // S^T and dP^T accumulators start at -lse / scale and -delta of the lane's q column (see dkdv_tile)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        st[cb][nb] = f32x4{nL[cb], nL[cb], nL[cb], nL[cb]};
        dpt[cb][nb] = f32x4{nD[cb], nD[cb], nD[cb], nD[cb]};
      }
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
    if (whole && kv0 + 31 <= q0 && kv0 >= lo_max) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[cb][nb][r] = fast_exp2(st[cb][nb][r] * sc2) * dpt[cb][nb][r];  // dS^T / scale
    } else {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kv = kv0 + nb * 16 + lq * 4 + r;
            const bool ok = (kv <= hi[cb]) && (kv >= lo[cb]);
            st[cb][nb][r] = ok ? fast_exp2(st[cb][nb][r] * sc2) * dpt[cb][nb][r] : 0.f;
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

template <int MINW, int KT>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  int64_t v_stride, const bf16_t* __restrict__ dout, const float* __restrict__ nlse,
                                                                  const float* __restrict__ ndelta, const int32_t* __restrict__ seq_start, int64_t n64,
                                                                  int hq, int hkv, float scale, bf16_t* __restrict__ dq, int q_row0, int narrow_stores) {
  // q_row0: see attn_bwd_dkdv_kernel — q, dout, nlse, ndelta, dq hold rows [q_row0, n) only
  __shared__ __attribute__((aligned(16))) unsigned char smA[kV5DqBufBytes<KT>];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kV5DqBufBytes<KT>];
  constexpr int TILE = KT * kD;      // elements of one [KT][128] tile
  constexpr int PPW = KT / 16;       // 1 KiB DMA pieces per wave per operand tile
  const int n = (int)n64;
  const int t = threadIdx.x;
  const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform: LDS-DMA destinations (m0) stay on the scalar unit
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
  const int qt0 = q_row0 + tile * 128;
  const int nq = n - q_row0;
  const int q0 = qt0 + w * 32;
  const bool active = q0 < n;
  u32x4 qf[2][4], dof[2][4];
  int qi[2], lo[2], hi[2];
  float nL[2], nD[2];  // -lse / scale and -delta of the lane's q rows
  f32x4 dqt[2][8];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    qi[cb] = q0 + cb * 16 + lr;
    const int row = qi[cb] < n ? qi[cb] : n - 1;
    hi[cb] = qi[cb] < n ? qi[cb] : -1;  // rows past the end see nothing
    lo[cb] = seq_start[row];
    nL[cb] = nlse[(int64_t)qh * nq + row - q_row0];
    nD[cb] = ndelta[(int64_t)qh * nq + row - q_row0];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[cb][kk] = ldg16(q + ((int64_t)(row - q_row0) * hq + qh) * kD + kk * 32 + lq * 8);
      dof[cb][kk] = ldg16(dout + ((int64_t)(row - q_row0) * hq + qh) * kD + kk * 32 + lq * 8);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) dqt[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tile_last = (qt0 + 127 < n - 1) ? qt0 + 127 : n - 1;
  const int wave_last = (q0 + 31 < n - 1) ? q0 + 31 : n - 1;
  const int kv_begin = (seq_start[qt0] / KT) * KT;
  const int lo_max = active ? seq_start[wave_last] : 0;  // the latest document start among this wave's q rows
  const bool whole = q0 + 31 < n;
  const int n_iters = (tile_last - kv_begin) / KT + 1;
  const float sc2 = scale * 1.44269504088896340736f;
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(k), 0, n * hkv * (kD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(v), 0, (int)(((int64_t)(n - 1) * v_stride + (int64_t)hkv * kD) * 2), 0x00020000);
  const int krow = hkv * (kD * 2), vrow = (int)v_stride * 2;
  int koff[PPW], voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (w * PPW + i) * 4 + (lane >> 4);
    const int ch = ((lane & 15) ^ swz_f(row & 15)) << 4;
    koff[i] = row * krow + kvh * (kD * 2) + ch;
    voff[i] = row * vrow + kvh * (kD * 2) + ch;
  }
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));
  dma_pieces<PPW>(krsrc, vrsrc, smA + w * PPW * 1024, TILE * 2, koff, voff, kv_begin * krow, kv_begin * vrow);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    pin_loaded(lo[cb]);
    pin_loaded(nL[cb]);
    pin_loaded(nD[cb]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      pin_loaded(qf[cb][kk]);
      pin_loaded(dof[cb][kk]);
    }
  }
  auto step = [&](const unsigned char* rd, unsigned char* wr, int j) {
    const int kvt = kv_begin + KT * j;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < n_iters) dma_pieces<PPW>(krsrc, vrsrc, wr + w * PPW * 1024, TILE * 2, koff, voff, (kvt + KT) * krow, (kvt + KT) * vrow);
    if (active) dq_tile<KT>(rd, kvt, q0, wave_last, lo_max, whole, sc2, bq, bt, lq, qf, dof, hi, lo, nL, nD, dqt);
  };
  for (int j = 0; j < n_iters; j += 2) {
    step(smA, smB, j);
    if (j + 1 < n_iters) step(smB, smA, j + 1);
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    if (!narrow_stores) {
      const int row = qi[cb] < n ? qi[cb] : n - 1;
      store_row_blocks16(dq + ((int64_t)(row - q_row0) * hq + qh) * kD, dqt[cb], scale, lq, qi[cb] < n);
    } else if (qi[cb] < n) {
      bf16_t* op = dq + ((int64_t)(qi[cb] - q_row0) * hq + qh) * kD + lq * 4;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) st8_bf16(op + nb * 16, dqt[cb][nb], scale);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: 4 waves x 32 q rows, K / V tiles of 64 rows DMA'd one tile ahead; online softmax per lane (column q = lr of S^T); the A
// fragments of O^T[d][q] += V^T[d][kv] . P^T[kv][q] are transpose-reads of the row-major V tile; the mask pass runs only for tiles some
// lane of the wave needs it for (diagonal / document-boundary tiles), so interior tiles carry no compare / select code.
constexpr int kV5FwdBufBytes = 2 * 64 * kD * 2;  // K [64][128] then V [64][128]

// The three parts of one (64-row kv tile, wave) step of the forward pass; fwd_tile runs them back to back.  (Round 5 tried fwd_qk of tile j + 1 in front of
// fwd_softmax of tile j with a second score register set: hipcc keeps the 32 MFMAs clustered ahead of the softmax's branches instead of interleaving, the
// build needs 472 registers = one wave per SIMD, and runs 0.58 vs 0.34 ms, bit-identical — profiles/r05_attention.md.  An in-wave overlap of QK^T and the
// softmax needs a hand-placed instruction stream, not a source-level reordering.)
template <int CB>
__device__ __forceinline__ void fwd_qk(const unsigned char* smK, int bq, const u32x4 (&qf)[CB][4], f32x4 (&st)[CB][4]) {
  const bf16_t* Kt = reinterpret_cast<const bf16_t*>(smK);
  // S^T[kv][q]: rows kv = nb*16 + lq*4 + r, column q = lr of block cb
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) st[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // k-step outermost: 8 independent accumulators per batch (no back-to-back dependent MFMAs), next batch's fragments prefetched
  u32x4 ka[2][4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) ka[0][nb] = lds16(Kt + bq + nb * 16 * kD);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (kk < 3) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) ka[(kk + 1) & 1][nb] = lds16(Kt + (bq ^ ((kk + 1) << 5)) + nb * 16 * kD);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) st[cb][nb] = mfma16(ka[kk & 1][nb], qf[cb][kk], st[cb][nb]);
  }
}

template <int OPT, int CB>  // A/B bits (pm_attn_debug_set_variant(0, OPT)): 1 = asm max3 row max, 2 = skip the O rescale when no row max moved, 4 = the row-max reduction as ONE
                    // asm statement + clamp instead of the -inf select, 8 = 16-byte epilogue stores (default 15; profiles/r03_attention.md section 4); (bits 1 | 2:
                    // 0.3314 -> 0.3257 ms, bit-identical; an explicit v_pk_fma_f32 for the scale-and-subtract was 4 % SLOWER: pair-building moves)
__device__ __forceinline__ void fwd_softmax(int kv0, int q0, float sc2, int lq, const int (&qi)[CB], const int (&lo)[CB], float (&m)[CB], float (&l)[CB],
                                            f32x4 (&st)[CB][4], f32x4 (&ot)[CB][8]) {
  // wave-uniform mask decision (masking a score that needs no mask is exact, so "any lane" is enough); a separate pass, so that the
  // interior-tile path carries no mask code at all
  bool need_mask = kv0 + 63 > q0;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) need_mask = need_mask || (kv0 < lo[cb]);
  if (__any(need_mask)) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kv = kv0 + nb * 16 + lq * 4 + r;
          st[cb][nb][r] = ((kv <= qi[cb]) && (kv >= lo[cb])) ? st[cb][nb][r] : -INFINITY;
        }
  }
  // Online softmax, written for the VALU: the row max is taken on the RAW scores (scale > 0 commutes with max), scale and subtraction are
  // one fma feeding v_exp_f32, the f32x4 accumulators are rescaled with vector multiplies (v_pk_mul_f32), and the max crosses the four
  // 16-lane rows of the wave through v_permlane32_swap / v_permlane16_swap (VALU) instead of ds_bpermute round trips.
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    float mx = 0.f, m_new4 = 0.f, m_use4 = 0.f;
    if constexpr ((OPT & 4) != 0) {
      row_max16_xrow(st[cb][0], st[cb][1], st[cb][2], st[cb][3], m[cb], sc2, m_new4, m_use4);
    } else if constexpr (OPT & 1) {
      mx = max3_f32(st[cb][0][0], st[cb][0][1], st[cb][0][2]);
      mx = fmax_raw(mx, st[cb][0][3]);
#pragma unroll
      for (int nb = 1; nb < 4; ++nb) {
        mx = max3_f32(mx, st[cb][nb][0], st[cb][nb][1]);
        mx = max3_f32(mx, st[cb][nb][2], st[cb][nb][3]);
      }
    } else {
      mx = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[cb][nb][r]);
    }
    if constexpr ((OPT & 4) == 0) mx = xrow_max_swap(mx) * sc2;
    const float m_new = (OPT & 4) ? m_new4 : fmaxf(m[cb], mx);
    const float m_use = (OPT & 4) ? m_use4 : ((m_new == -INFINITY) ? 0.f : m_new);
    const float alpha = fast_exp2(m[cb] - m_use);
    f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const f32x4 e = st[cb][nb] * sc2 - m_use;  // contracted to four fmas
      f32x4 p;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = fast_exp2(e[r]);
      st[cb][nb] = p;
      if ((OPT & 4) && nb == 0) sum4 = p;  // 0 + p is p: two packed adds less per block
      else sum4 += p;
    }
    l[cb] = l[cb] * alpha + ((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));  // per-lane partial row sum; reduced across lq at the end
    m[cb] = m_new;
    if (!(OPT & 2) || !__all(alpha == 1.0f)) {  // alpha == 1 for the whole wave: the rescale would multiply by one (bit-identical to skip it)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) ot[cb][nb] *= alpha;
    }
  }
}

template <int CB>
__device__ __forceinline__ void fwd_pv(const unsigned char* smV, int bt, const f32x4 (&st)[CB][4], f32x4 (&ot)[CB][8]) {
  const bf16_t* Vt = reinterpret_cast<const bf16_t*>(smV);
  // O^T[d][q] += V^T[d][kv] . P^T[kv][q]
#pragma unroll
  for (int kk2 = 0; kk2 < 2; ++kk2) {
    u32x4 pb[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) pb[cb] = pack_c_pair(st[cb][2 * kk2], st[cb][2 * kk2 + 1]);
#pragma unroll
    for (int h4 = 0; h4 < 2; ++h4) {
      u32x4 va[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) va[i] = lds_tr_pair(Vt + kk2 * 32 * kD + (bt ^ ((h4 * 4 + i) << 4)));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nb = h4 * 4 + i;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) ot[cb][nb] = mfma16(va[i], pb[cb], ot[cb][nb]);
      }
    }
  }
}

template <int OPT, int CB = 2>  // CB = 16-row q blocks per wave (2: the shipped 4-wave kernel; 1: the 8-wave build)
__device__ __forceinline__ void fwd_tile(const unsigned char* sm, int kv0, int q0, float sc2, int bq, int bt, int lq, const u32x4 (&qf)[CB][4],
                                         const int (&qi)[CB], const int (&lo)[CB], float (&m)[CB], float (&l)[CB], f32x4 (&ot)[CB][8]) {
  // s_setprio 1 around the two matrix phases: with two waves per SIMD, the wave that is issuing MFMAs wins the issue slot over the partner's exp / pack
  // stream, so the matrix pipe is not left waiting behind VALU work (forward 0.354 -> 0.333 ms at 16k tokens, bit-identical; profiles/r05_attention.md)
  f32x4 st[CB][4];
  __builtin_amdgcn_s_setprio(1);
  fwd_qk<CB>(sm, bq, qf, st);
  __builtin_amdgcn_s_setprio(0);
  fwd_softmax<OPT, CB>(kv0, q0, sc2, lq, qi, lo, m, l, st, ot);
  __builtin_amdgcn_s_setprio(1);
  fwd_pv<CB>(sm + 64 * kD * 2, bt, st, ot);
  __builtin_amdgcn_s_setprio(0);
}

template <int MINW, int OPT = 0, int CB = 2, int NW = 4>
__global__ __launch_bounds__(64 * NW, MINW) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               int64_t v_stride, const int32_t* __restrict__ seq_start, int64_t n64, int hq, int hkv,
                                                               float scale, bf16_t* __restrict__ out, float* __restrict__ lse, int q_row0) {
  // q_row0 (a multiple of 128; 0 = the ordinary call): q, out, lse hold rows [q_row0, n) only — addressed as row - q_row0 — while k, v, seq_start
  // hold rows [0, n): one rank's share of a pass under context parallelism with gathered K / V (pm_attn_varlen_fwd_rows)
  __shared__ __attribute__((aligned(16))) unsigned char smA[kV5FwdBufBytes];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kV5FwdBufBytes];
  static_assert(NW * CB * 16 == 128 && 16 % NW == 0, "a workgroup covers one 128-row q tile");
  constexpr int PPW = 16 / NW;  // 1 KiB DMA pieces per wave per operand tile (64 rows x 256 B over NW waves)
  const int n = (int)n64;
  const int t = threadIdx.x;
  const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform: LDS-DMA destinations (m0) stay on the scalar unit
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
  const int qt0 = q_row0 + tile * 128;
  const int nq = n - q_row0;
  const int q0 = qt0 + w * 16 * CB;
  const bool active = q0 < n;

  u32x4 qf[CB][4];
  int qi[CB], lo[CB];
  float m[CB], l[CB];
  f32x4 ot[CB][8];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qi[cb] = q0 + cb * 16 + lr;
    const int row = qi[cb] < n ? qi[cb] : n - 1;
    lo[cb] = seq_start[row];
    m[cb] = -INFINITY;
    l[cb] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[cb][kk] = ldg16(q + ((int64_t)(row - q_row0) * hq + qh) * kD + kk * 32 + lq * 8);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) ot[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tile_last = (qt0 + 127 < n - 1) ? qt0 + 127 : n - 1;
  const int wave_last = (q0 + 16 * CB - 1 < n - 1) ? q0 + 16 * CB - 1 : n - 1;
  const int kv_begin = (seq_start[qt0] / 64) * 64;
  const int n_iters = (tile_last - kv_begin) / 64 + 1;
  const float sc2 = scale * 1.44269504088896340736f;
  // rows past the end of the tensor read as zeros (hardware range check), which the causal / document mask discards like any other kv > q
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(k), 0, n * hkv * (kD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(v), 0, (int)(((int64_t)(n - 1) * v_stride + (int64_t)hkv * kD) * 2), 0x00020000);
  const int krow = hkv * (kD * 2), vrow = (int)v_stride * 2;
  int koff[PPW], voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (w * PPW + i) * 4 + (lane >> 4);
    const int ch = ((lane & 15) ^ swz_f(row & 15)) << 4;
    koff[i] = row * krow + kvh * (kD * 2) + ch;
    voff[i] = row * vrow + kvh * (kD * 2) + ch;
  }
  const int bq = lr * kD + ((lq ^ swz_f(lr)) << 3);
  const int r0 = lq * 4 + (lr >> 2);
  const int bt = r0 * kD + (((((lr & 3) >> 1) ^ swz_f(r0)) << 3) | ((lr & 1) << 2));
  dma_pieces<PPW>(krsrc, vrsrc, smA + w * PPW * 1024, 64 * kD * 2, koff, voff, kv_begin * krow, kv_begin * vrow);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    pin_loaded(lo[cb]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pin_loaded(qf[cb][kk]);
  }
  auto step = [&](const unsigned char* rd, unsigned char* wr, int j) {
    const int kv0 = kv_begin + 64 * j;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < n_iters) dma_pieces<PPW>(krsrc, vrsrc, wr + w * PPW * 1024, 64 * kD * 2, koff, voff, (kv0 + 64) * krow, (kv0 + 64) * vrow);
    if (active && kv0 <= wave_last) fwd_tile<OPT, CB>(rd, kv0, q0, sc2, bq, bt, lq, qf, qi, lo, m, l, ot);
  };
  for (int j = 0; j < n_iters; j += 2) {
    step(smA, smB, j);
    if (j + 1 < n_iters) step(smB, smA, j + 1);
  }
  if (!active) return;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const float lsum = xrow_sum(l[cb]);
    if (qi[cb] < n) {
      const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
      bf16_t* op = out + ((int64_t)(qi[cb] - q_row0) * hq + qh) * kD + lq * 4;
      if constexpr ((OPT & 8) == 0) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) st8_bf16(op + nb * 16, ot[cb][nb], inv);
      }
      if (lq == 0) lse[(int64_t)qh * nq + qi[cb] - q_row0] = lse_of(m[cb], lsum);
    }
    if constexpr ((OPT & 8) != 0) {
      const int row = qi[cb] < n ? qi[cb] : n - 1;
      store_row_blocks16(out + ((int64_t)(row - q_row0) * hq + qh) * kD, ot[cb], lsum > 0.f ? 1.0f / lsum : 0.f, lq, qi[cb] < n);
    }
  }
}
