// gemm.hip — the scoring model's dense GEMMs as ONE hand-written MFMA core for gfx950 (north_star: "MFMA only in the scoring-model GEMMs ...
// hand-written CDNA4 HIP kernels").
//
// Reference ops this replaces (all `F.linear` / autograd-derived matmuls executed by cuBLAS in the reference):
//   forward   y[m][n]  = sum_k x[m][k] * w[n][k] (+ bias[n])      src/prime_rl/trainer/models/layers/mlp.py:27-29, attn.py:60-84 (q/k/v/o projections),
//                                                                 lm_head.py:118-148 (hidden @ w_chunk.t())
//   dgrad     dx[m][k] = sum_n dy[m][n] * w[n][k]                  (autograd of the above)
//   wgrad     dw[n][k] = sum_m dy[m][n] * x[m][k]                  (autograd; accumulated in f32 — FSDP `reduce_dtype=float32`, configs/trainer.py:233-237)
//
// One kernel template, three operand layouts.  Naming inside this file: the product is always OUT[M, N] = sum_K  X[M, K] * W[N, K]; an operand is
// either "K-contiguous" (its rows are indexed by its M / N coordinate, K runs along the row: read from LDS with ds_read_b128) or "K-major" (rows
// are indexed by K, its M / N coordinate runs along the row: read from LDS with ds_read_b64_tr_b16 pairs).  Forward = (X Kc, W Kc); dgrad with the
// native [out, in] weight = (X Kc, W Km); wgrad = (X Km, W Km) with X = the layer input, W = dy, K = tokens.  No transposed copy of any weight or
// activation exists in HBM.
//
// MI355X design (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md §5 "256² 8-phase template"):
//   * 256 x 256 output tile per workgroup, contraction step 64, 8 waves = 2 (W rows) x 4 (X rows), 128 x 64 outputs per wave held as 2 x 2 x 2
//     accumulator fragments of v_mfma_f32_32x32x16_bf16 (128 registers).  TRANSPOSED product: the MFMA A operand is the W fragment, so a lane
//     owns ONE X row (token) and 4 x 4 consecutive W rows (output columns): 8 / 16-byte stores along the contiguous axis of OUT, and per-lane
//     softmax state for the fused log-prob epilogue.
//   * PING-PONG: waves 0-3 (one per SIMD) and waves 4-7 (their SIMD partners) run the same phase stream one barrier apart.  A phase =
//     [load segment: LDS fragment reads + 2 LDS-DMA pieces] s_barrier [compute segment: 8 MFMAs = one 64 x 32 quadrant x K 64] s_barrier, so on
//     every SIMD one wave is in its compute segment while its partner is in its load segment; the matrix pipe never waits for an LDS read and
//     no wave mixes LDS / VMEM issue into its MFMA stream.
//   * LDS = a ring of 8 half-tile slots of 16 KB (two K-tiles): per K-tile W0 | W1 | X0 | X1, where W^i holds every wave's i-th 64-row W sub-tile
//     and X^j every wave's j-th 32-row X sub-tile, so that ALL waves consume the slots in the same phase: phase 1 reads W0 + X0, 2 reads X1,
//     3 reads W1, 4 reads nothing (X0 stays in registers).  One half-tile (2 KiB per wave) is staged per phase, 1.25-1.5 K-tiles ahead of its
//     use:  ph1(t): X1(t+1)   ph2(t): W1(t+1)   ph3(t): W0(t+2)   ph4(t): X0(t+2).
//     A slot is re-staged >= 2 phases after its last read (WAR across the two staggered wave groups) and read >= 1 phase after the counted
//     `s_waitcnt vmcnt(6)` + barrier that retires it (RAW; the wait leaves the 3 newest half-tiles in flight, never 0 inside the loop).
//   * tiles arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write.  The DMA writes lane-linear, so the
//     bank-conflict swizzle is applied to the per-lane SOURCE address: 16-byte chunk c of tile row r (128-byte rows) lives at chunk
//     c ^ ((r >> 1) & 7) — conflict-free for the 32-row fragments of the 32x32x16 MFMA against the real ds_read_b128 service groups
//     {0-3,12-15,20-27} {4-11,16-19,28-31} (+32).  Rows past the end of a tensor read as zeros through the buffer range check.
//   * blockIdx -> tile: block b runs on XCD b % 8; the 32 tiles an XCD runs concurrently form an 8 (X) x 4 (W) patch that walks K in step,
//     so 12 row panels feed 32 tiles through one L2, and the 8 XCDs work on patches that share their W panels.
#include "common.cuh"
#include <algorithm>

namespace pm {
namespace gm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int kTile = 256;            // output tile edge (both ways)
constexpr int kBK = 64;               // contraction per K-tile
constexpr int kSlot = 128 * 128;      // one half-tile slot: 128 rows x 64 bf16
constexpr int kSmem = 8 * kSlot;      // 128 KiB

enum { SLOT_W0 = 0, SLOT_W1 = 1, SLOT_X0 = 2, SLOT_X1 = 3 };

struct Params {
  const bf16_t* x;    // M-side operand: [M, K] (K-contiguous) or [K, M] (K-major), row stride ldx (elements)
  const bf16_t* w;    // N-side operand: [N, K] or [K, N], row stride ldw
  void* out;          // [M, N] bf16 or f32, row stride ldo
  const bf16_t* bias; // [N] or null: added to the f32 accumulator before the rounding (F.linear)
  int64_t ldx, ldw, ldo;
  int m, n, k;
  int tiles_m, tiles_n;
  int out_f32;        // 0: bf16 out; 1: f32 out
  int accumulate;     // f32 out only: out += product
};

__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// blockIdx.x -> (tile_m, tile_n); false for the padding blocks of the last patch round
__device__ __forceinline__ bool tile_of_block(int b, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int xcd = b & 7, s = b >> 3;
  const int patch = (s >> 5) * 8 + xcd, within = s & 31;
  const int patches_m = (tiles_m + 7) >> 3;
  const int patches_n = (tiles_n + 3) >> 2;
  if (patch >= patches_m * patches_n) return false;
  tm = (patch % patches_m) * 8 + (within & 7);
  tn = (patch / patches_m) * 4 + (within >> 3);
  return tm < tiles_m && tn < tiles_n;
}
static inline int grid_blocks(int tiles_m, int tiles_n) {
  const int patches = ((tiles_m + 7) / 8) * ((tiles_n + 3) / 4);
  return ((patches + 7) / 8) * 8 * 32;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Everything a wave carries through the K loop.  After inlining the struct is scalar-replaced (all indices are compile-time).
//
// Half-tile slots.  W^i = tile rows [128 i, 128 i + 128) of the N-side operand, X^j = rows [128 j, 128 j + 128) of the M-side operand; wave
// (wv, wt) owns W rows 128 i + 64 wv + [0, 64) and X rows 128 j + 32 wt + [0, 32) of every half, i.e. the output elements
//   OUT[m0 + 128 j + 32 wt + l31][n0 + 128 i + 64 wv + 32 a + 8 (r >> 2) + 4 hi + (r & 3)]  =  acc[i][a][j][r].
// LDS image of a slot:
//   K-contiguous operand: [128 rows][64 k] bf16, 128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7); a fragment (32 rows x 16 k)
//     is ONE ds_read_b128 per lane: row l31, chunk 2 kk + hi.
//   K-major operand: [64 k][128 rows] bf16, 256-byte LDS rows, chunk c (8 operand rows) of k-row q at chunk c ^ ((q & 3) << 2); a fragment is
//     TWO ds_read_b64_tr_b16 per lane (k-rows 16 kk + 8 hi + {0..3}, then {4..7}): the 16 lanes of a transpose group read a 4 (k) x 16 (rows)
//     block, and the four k-rows of a half-wave land in four different 64-byte quads of the bank row (conflict-free by construction).
struct Wave {
  f32x16 acc[2][2][2];  // [i: W half][a: 32-row fragment][j: X half]
  u32x4 wf[2][2][4];    // W fragments [i][a][kk]      (K-contiguous operand)
  u32x4 xf[2][4];       // X fragments [j][kk]
  u32x2 wlo[2][2][4], whi[2][2][4];  // K-major operand: the two transpose-read halves of a fragment (k 0..3 | 4..7 of the lane's 8)
  u32x2 xlo[2][4], xhi[2][4];
  unsigned aw[2][2], ax[2];          // K-major operand: per-lane LDS byte address of the fragment read for (kk = 0, slot 0 of buffer BUF) [BUF][a]
  __amdgpu_buffer_rsrc_t wr, xr;
  int vw[2], vx[2];     // per-lane byte offsets of this wave's two DMA pieces (half-tile 0; half-tile 1 adds wstep / xstep)
  int wstep, xstep;     // byte distance between the two halves of an operand tile (128 rows, or 128 columns of a K-major operand)
  int wk, xk;           // byte distance between consecutive K-tiles
  int piece_lds;        // wave-uniform: byte offset of this wave's first piece inside a slot
  int rd_w[4], rd_x[4]; // per-lane LDS byte offsets of the fragment reads per kk (inside a slot; W fragment a adds a constant)
};

__device__ __forceinline__ unsigned char* lds_base() {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  return smem;
}

typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t* lds_v4i16_ptr;

template <int SLOT, int BUF>
__device__ __forceinline__ void stage(const Wave& s, unsigned char* smem, int ktile) {
  constexpr int slot_off = (BUF * 4 + SLOT) * kSlot;
  constexpr bool is_w = SLOT == SLOT_W0 || SLOT == SLOT_W1;
  constexpr int half = (SLOT == SLOT_W1 || SLOT == SLOT_X1) ? 1 : 0;
  const int soff = ktile * (is_w ? s.wk : s.xk) + half * (is_w ? s.wstep : s.xstep);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    unsigned char* dst = smem + slot_off + s.piece_lds + q * 1024;
    if (is_w) __builtin_amdgcn_raw_ptr_buffer_load_lds(s.wr, (lds_void_ptr)dst, 16, s.vw[q], soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(s.xr, (lds_void_ptr)dst, 16, s.vx[q], soff, 0, 0);
  }
}

// K-major fragments are read with INLINE-ASM transpose reads: hipcc guards `__builtin_amdgcn_ds_read_tr16_b64` against every LDS-DMA in flight
// with `s_waitcnt vmcnt(0)` (it cannot tell the ring slots apart), which would drain the whole prefetch in every phase.  The compiler does not
// track asm results on lgkmcnt, so the consumer side is explicit: `s_waitcnt lgkmcnt(0)` after the barrier, then `tie()` — an empty asm that
// takes the registers in and out — so that nothing that uses them can be scheduled above the wait.
template <int IMM>
__device__ __forceinline__ u32x2 tr_read(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
__device__ __forceinline__ void tie(u32x2& a, u32x2& b, u32x2& c, u32x2& d, u32x2& e, u32x2& f, u32x2& g, u32x2& h) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ u32x4 join(const u32x2& lo, const u32x2& hi) { return u32x4{lo.x, lo.y, hi.x, hi.y}; }

template <int I, int BUF, bool KM>
__device__ __forceinline__ void read_w(Wave& s, const unsigned char* smem) {
  constexpr int slot = I ? SLOT_W1 : SLOT_W0;
  if constexpr (!KM) {
    constexpr int slot_off = (BUF * 4 + slot) * kSlot;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s.wf[I][a][kk] = *reinterpret_cast<const u32x4*>(smem + slot_off + a * 32 * 128 + s.rd_w[kk]);
  } else {
    // k-rows 16 kk + {0..3} then {4..7}: 4 KiB per kk, 1 KiB between the two reads; slot offsets inside the 64 KiB of buffer BUF as immediates
#define PM_TRW(A, KK)                                                                 \
    s.wlo[I][A][KK] = tr_read<slot * kSlot + KK * 4096>(s.aw[BUF][A]);                \
    s.whi[I][A][KK] = tr_read<slot * kSlot + KK * 4096 + 1024>(s.aw[BUF][A]);
    PM_TRW(0, 0) PM_TRW(0, 1) PM_TRW(0, 2) PM_TRW(0, 3) PM_TRW(1, 0) PM_TRW(1, 1) PM_TRW(1, 2) PM_TRW(1, 3)
#undef PM_TRW
  }
}
template <int J, int BUF, bool KM>
__device__ __forceinline__ void read_x(Wave& s, const unsigned char* smem) {
  constexpr int slot = J ? SLOT_X1 : SLOT_X0;
  if constexpr (!KM) {
    constexpr int slot_off = (BUF * 4 + slot) * kSlot;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) s.xf[J][kk] = *reinterpret_cast<const u32x4*>(smem + slot_off + s.rd_x[kk]);
  } else {
#define PM_TRX(KK)                                                              \
    s.xlo[J][KK] = tr_read<slot * kSlot + KK * 4096>(s.ax[BUF]);                \
    s.xhi[J][KK] = tr_read<slot * kSlot + KK * 4096 + 1024>(s.ax[BUF]);
    PM_TRX(0) PM_TRX(1) PM_TRX(2) PM_TRX(3)
#undef PM_TRX
  }
}
template <int I>
__device__ __forceinline__ void tie_w(Wave& s) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
    tie(s.wlo[I][a][0], s.whi[I][a][0], s.wlo[I][a][1], s.whi[I][a][1], s.wlo[I][a][2], s.whi[I][a][2], s.wlo[I][a][3], s.whi[I][a][3]);
}
template <int J>
__device__ __forceinline__ void tie_x(Wave& s) {
  tie(s.xlo[J][0], s.xhi[J][0], s.xlo[J][1], s.xhi[J][1], s.xlo[J][2], s.xhi[J][2], s.xlo[J][3], s.xhi[J][3]);
}
template <int I, int J, bool XK, bool WK>
__device__ __forceinline__ void quadrant(Wave& s) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const u32x4 wfrag = WK ? join(s.wlo[I][a][kk], s.whi[I][a][kk]) : s.wf[I][a][kk];
      const u32x4 xfrag = XK ? join(s.xlo[J][kk], s.xhi[J][kk]) : s.xf[J][kk];
      s.acc[I][a][J] = mfma32(wfrag, xfrag, s.acc[I][a][J]);
    }
}

// One phase.  P = phase inside the K-tile (0..3), BUF = parity of the K-tile.  `more1` / `more2`: K-tiles t+1 / t+2 exist (wave-uniform).
// V bit 0: s_setprio(1) around the MFMA cluster.  V bit 1: balanced reads (W0 of the NEXT K-tile is read in phase 4 instead of phase 1).
template <int P, int BUF, int V, bool XK, bool WK>
__device__ __forceinline__ void phase(Wave& s, unsigned char* smem, int t, bool more1, bool more2) {
  constexpr bool kBalanced = (V & 2) != 0;
  // ---- load segment
  if (P == 0) {
    if (!kBalanced) read_w<0, BUF, WK>(s, smem);
    read_x<0, BUF, XK>(s, smem);
  } else if (P == 1) {
    read_x<1, BUF, XK>(s, smem);
  } else if (P == 2) {
    read_w<1, BUF, WK>(s, smem);
  } else if (kBalanced) {
    if (more1) read_w<0, BUF ^ 1, WK>(s, smem);
  }
  const bool issue = (P < 2) ? more1 : more2;
  if (issue) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (P == 0) stage<SLOT_X1, BUF ^ 1>(s, smem, t + 1);
    else if (P == 1) stage<SLOT_W1, BUF ^ 1>(s, smem, t + 1);
    else if (P == 2) stage<SLOT_W0, BUF>(s, smem, t + 2);
    else stage<SLOT_X0, BUF>(s, smem, t + 2);
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // asm transpose reads of this phase: their registers become usable only here
  if (P == 0) {
    if (WK) tie_w<0>(s);  // also in balanced mode: the prologue read of K-tile 0
    if (XK) tie_x<0>(s);
  } else if (P == 1) {
    if (XK) tie_x<1>(s);
  } else if (P == 2) {
    if (WK) tie_w<1>(s);
  } else if (kBalanced) {
    if (WK) tie_w<0>(s);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- compute segment
  if (V & 1) __builtin_amdgcn_s_setprio(1);
  if (P == 0) quadrant<0, 0, XK, WK>(s);
  else if (P == 1) quadrant<0, 1, XK, WK>(s);
  else if (P == 2) quadrant<1, 1, XK, WK>(s);
  else quadrant<1, 0, XK, WK>(s);
  if (V & 1) __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_s_barrier();
}

// per-lane DMA source offsets + fragment read offsets of one operand.  `rows_valid`: operand rows (M or N coordinate) left from the tile's first
// row; `wsel`: this wave's first row inside a half (64 wv for W, 32 wt for X).
template <bool KM>
__device__ __forceinline__ void setup_operand(const bf16_t* base, int64_t ld, int64_t row0, int64_t rows_total, int k, int wave, int lane, int wsel,
                                              __amdgpu_buffer_rsrc_t& rsrc, int (&voff)[2], int& half_step, int& k_step, int (&rd)[4]) {
  const int l31 = lane & 31, hi = lane >> 5;
  if (!KM) {
    const int64_t rows = rows_total - row0 < kTile ? rows_total - row0 : kTile;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + row0 * ld), 0, (int)(((rows - 1) * ld + k) * 2), 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int rho = (wave * 2 + q) * 8 + (lane >> 3);  // slot row of this lane's 16 bytes (piece = 8 rows x 128 bytes)
      voff[q] = (int)(rho * ld * 2) + (((lane & 7) ^ ((rho >> 1) & 7)) << 4);
    }
    half_step = (int)(128 * ld * 2);
    k_step = kBK * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) rd[kk] = (wsel + l31) * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
  } else {
    // [K, rows] tensor: the descriptor starts at column row0 and ends with the tensor, so k-rows past K read as zeros
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + row0), 0, (int)((((int64_t)k - 1) * ld + (rows_total - row0)) * 2), 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int krow = (wave * 2 + q) * 4 + (lane >> 4);  // piece = 4 k-rows x 256 bytes
      voff[q] = (int)(krow * ld * 2) + (((lane & 15) ^ ((krow & 3) << 2)) << 4);
    }
    half_step = 128 * 2;
    k_step = (int)(kBK * ld * 2);
    const int lam = lane & 15, grp = (lane >> 4) & 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int krow = kk * 16 + hi * 8 + (lam >> 2);
      const int chunk = ((wsel + grp * 16) >> 3) + ((lam & 3) >> 1);
      rd[kk] = krow * 256 + ((chunk ^ ((krow & 3) << 2)) << 4) + (lam & 1) * 8;
    }
  }
}

template <int V, bool XK, bool WK>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const Params p) {
  unsigned char* smem = lds_base();
  int tm, tn;
  if (!tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = wave >> 2, wt = wave & 3;  // wv: which 64 W rows of a half (also the ping-pong group), wt: which 32 X rows of a half
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = tm * kTile, n0 = tn * kTile;

  Wave s;
  setup_operand<WK>(p.w, p.ldw, n0, p.n, p.k, wave, lane, wv * 64, s.wr, s.vw, s.wstep, s.wk, s.rd_w);
  setup_operand<XK>(p.x, p.ldx, m0, p.m, p.k, wave, lane, wt * 32, s.xr, s.vx, s.xstep, s.xk, s.rd_x);
  s.piece_lds = wave * 2048;
  {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      s.aw[b][0] = lds0 + b * 4 * kSlot + s.rd_w[0];
      s.aw[b][1] = lds0 + b * 4 * kSlot + (s.rd_w[0] ^ 64);  // fragment a = 1: chunk bit 2 (an XOR: the swizzle flips that bit)
      s.ax[b] = lds0 + b * 4 * kSlot + s.rd_x[0];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s.acc[i][a][j][r] = 0.f;

  const int nk = ((p.k + 2 * kBK - 1) / (2 * kBK)) * 2;  // even; K-tiles past K read zeros (K-major operands) or do not exist (K % 128 == 0)
  // ---- prologue: the issue order continues into the loop's (X1, W1, W0, X0) rotation
  stage<SLOT_W0, 0>(s, smem, 0);
  stage<SLOT_X0, 0>(s, smem, 0);
  stage<SLOT_X1, 0>(s, smem, 0);
  stage<SLOT_W1, 0>(s, smem, 0);
  stage<SLOT_W0, 1>(s, smem, 1);
  stage<SLOT_X0, 1>(s, smem, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // W0(0) and X0(0) of this wave have landed
  __builtin_amdgcn_s_barrier();
  if ((V & 2) != 0) {  // balanced reads: W0 of K-tile 0 is read here (tied in phase 1 like the later ones, which are read in phase 4 of the previous K-tile)
    read_w<0, 0, WK>(s, smem);
  }
  if (wv == 1) __builtin_amdgcn_s_barrier();    // the second wave group runs one barrier behind the first

  for (int t = 0; t < nk; t += 2) {
    const bool m2 = t + 2 < nk, m3 = t + 3 < nk;
    phase<0, 0, V, XK, WK>(s, smem, t, true, m2);
    phase<1, 0, V, XK, WK>(s, smem, t, true, m2);
    phase<2, 0, V, XK, WK>(s, smem, t, true, m2);
    phase<3, 0, V, XK, WK>(s, smem, t, true, m2);
    phase<0, 1, V, XK, WK>(s, smem, t + 1, m2, m3);
    phase<1, 1, V, XK, WK>(s, smem, t + 1, m2, m3);
    phase<2, 1, V, XK, WK>(s, smem, t + 1, m2, m3);
    phase<3, 1, V, XK, WK>(s, smem, t + 1, m2, m3);
  }
  if (wv == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + j * 128 + wt * 32 + l31;
    if (m >= p.m) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = n0 + i * 128 + wv * 64 + a * 32 + g * 8 + hi * 4;
          if (nb >= p.n) continue;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = s.acc[i][a][j][g * 4 + r];
          const bool full = nb + 3 < p.n;
          if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (full || nb + r < p.n) v[r] += bf16_to_f32(p.bias[nb + r]);
          }
          if (p.out_f32) {
            float* dst = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + nb;
            if (full) {
              f32x4 o = {v[0], v[1], v[2], v[3]};
              if (p.accumulate) {
                const f32x4 old = *reinterpret_cast<const f32x4*>(dst);
                o += old;
              }
              *reinterpret_cast<f32x4*>(dst) = o;
            } else {
              for (int r = 0; r < 4; ++r)
                if (nb + r < p.n) dst[r] = p.accumulate ? dst[r] + v[r] : v[r];
            }
          } else {
            bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (int64_t)m * p.ldo + nb;
            if (full) {
              *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16_hw(v[0], v[1]), pack_bf16_hw(v[2], v[3])};
            } else {
              for (int r = 0; r < 4; ++r)
                if (nb + r < p.n) dst[r] = f32_to_bf16(v[r]);
            }
          }
        }
  }
}

static int g_variant = 1;

template <bool XK, bool WK>
static void launch(const Params& p, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<0, XK, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<1, XK, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<2, XK, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<3, XK, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    attr_done = true;
  }
  const dim3 grid(grid_blocks(p.tiles_m, p.tiles_n)), block(512);
  switch (g_variant & 3) {
    case 0: hipLaunchKernelGGL((gemm_kernel<0, XK, WK>), grid, block, kSmem, st, p); break;
    case 1: hipLaunchKernelGGL((gemm_kernel<1, XK, WK>), grid, block, kSmem, st, p); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<2, XK, WK>), grid, block, kSmem, st, p); break;
    default: hipLaunchKernelGGL((gemm_kernel<3, XK, WK>), grid, block, kSmem, st, p); break;
  }
}

}  // namespace gm
}  // namespace pm

using namespace pm;

extern "C" int pm_gemm_debug_set_variant(int32_t v) {
  gm::g_variant = v;
  return PM_OK;
}

// K-contiguous operands need K % 128 == 0 (a K-tile past the end of a row would read the next row); two K-major operands take any K
extern "C" int pm_gemm_supported(int64_t m, int64_t n, int64_t k, int32_t x_kmajor, int32_t w_kmajor) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  if (x_kmajor && !w_kmajor) return 0;  // no caller has that pair
  if (!(x_kmajor && w_kmajor) && k % 128 != 0) return 0;
  if (x_kmajor && m % 8 != 0) return 0;  // 16-byte chunks along the rows of a K-major operand
  if (w_kmajor && n % 8 != 0) return 0;
  return 1;
}

extern "C" int pm_gemm_bf16(const void* x, int64_t ldx, int32_t x_kmajor, const void* w, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo,
                            int32_t out_dtype, const void* bias, int32_t accumulate, int64_t m, int64_t n, int64_t k, void* stream) {
  if (m == 0 || n == 0) return PM_OK;
  PM_REQUIRE(x && w && out, PM_ERR_INVALID_ARG, "pm_gemm_bf16: null pointer");
  PM_REQUIRE(pm_gemm_supported(m, n, k, x_kmajor, w_kmajor), PM_ERR_UNSUPPORTED,
             "pm_gemm_bf16: unsupported shape / layout (m %lld, n %lld, k %lld, x_kmajor %d, w_kmajor %d)", (long long)m, (long long)n, (long long)k, x_kmajor, w_kmajor);
  PM_REQUIRE(out_dtype == PM_DTYPE_BF16 || out_dtype == PM_DTYPE_F32, PM_ERR_INVALID_ARG, "pm_gemm_bf16: out dtype must be bf16 or f32");
  PM_REQUIRE(!accumulate || out_dtype == PM_DTYPE_F32, PM_ERR_INVALID_ARG, "pm_gemm_bf16: accumulate needs an f32 output");
  PM_REQUIRE(ldx >= (x_kmajor ? m : k) && ldw >= (w_kmajor ? n : k) && ldo >= n, PM_ERR_INVALID_ARG, "pm_gemm_bf16: leading dimension too small");
  PM_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0, PM_ERR_INVALID_ARG,
             "pm_gemm_bf16: rows must start 16-byte aligned (ldx, ldw multiples of 8; ldo of 4; 16-byte aligned bases)");
  // 32-bit byte offsets inside a tile's buffer descriptor: a K-contiguous operand spans 256 rows, a K-major operand all K rows of the tensor.  When
  // a K-major span does not fit (the LM-head weight gradient: 16384 tokens x 151936 columns of dlogits = 5 GB), the contraction is cut into K-chunks
  // that accumulate into the f32 output.
  const int64_t kLimit = (1ll << 31) - 4096;
  PM_REQUIRE((x_kmajor || 256 * ldx * 2 < kLimit) && (w_kmajor || 256 * ldw * 2 < kLimit) && m < (1ll << 31) && n < (1ll << 31), PM_ERR_UNSUPPORTED,
             "pm_gemm_bf16: sizes beyond int32 offsets");
  int64_t k_chunk = k;
  if (x_kmajor && k * ldx * 2 >= kLimit) k_chunk = std::min(k_chunk, kLimit / (ldx * 2) / 128 * 128);
  if (w_kmajor && k * ldw * 2 >= kLimit) k_chunk = std::min(k_chunk, kLimit / (ldw * 2) / 128 * 128);
  PM_REQUIRE(k_chunk >= 128, PM_ERR_UNSUPPORTED, "pm_gemm_bf16: a K-major operand with rows beyond 8M elements");
  PM_REQUIRE(k_chunk == k || (out_dtype == PM_DTYPE_F32 && x_kmajor && w_kmajor && !bias), PM_ERR_UNSUPPORTED,
             "pm_gemm_bf16: a K-major operand beyond 2 GB needs the f32-accumulating wgrad form");
  gm::Params p;
  p.out = out; p.bias = (const bf16_t*)bias;
  p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
  p.m = (int)m; p.n = (int)n;
  p.tiles_m = (int)cdiv(m, gm::kTile); p.tiles_n = (int)cdiv(n, gm::kTile);
  p.out_f32 = out_dtype == PM_DTYPE_F32;
  hipStream_t st = as_stream(stream);
  for (int64_t k0 = 0; k0 < k; k0 += k_chunk) {
    p.x = (const bf16_t*)x + (x_kmajor ? k0 * ldx : k0);
    p.w = (const bf16_t*)w + (w_kmajor ? k0 * ldw : k0);
    p.k = (int)std::min(k_chunk, k - k0);
    p.accumulate = (accumulate || k0 > 0) ? 1 : 0;
    if (!x_kmajor && !w_kmajor) gm::launch<false, false>(p, st);
    else if (!x_kmajor && w_kmajor) gm::launch<false, true>(p, st);
    else gm::launch<true, true>(p, st);
  }
  PM_CHECK_LAUNCH("gemm_kernel");
  return PM_OK;
}
