// grouped_gemm.hip — the expert GEMMs of the MoE layer as hand-written MFMA kernels over the expert-sorted rows (SURVEY §8 row f4:
// "MoE EP grouped-GEMM").
//
// Reference: _run_experts_grouped_mm_impl (src/prime_rl/trainer/models/layers/moe.py:161-188): three `torch._grouped_mm(x, w.transpose(-2, -1),
// offs=offsets)` per layer — rows [offs[e-1], offs[e]) of x are multiplied with expert e's matrix — and the two gradients autograd derives
// from each of them.  Measured on the MI355X at Qwen3-30B-A3B layer shapes (131 072 sorted rows, 128 experts, 2048 <-> 768): the library
// grouped GEMM runs at 0.17 PFLOP/s, a per-expert loop of library GEMMs at 0.12 (tools/microbench.py moe), i.e. 7 % of the MFMA peak; these
// kernels: 0.72 / 0.57 / 0.78 PFLOP/s (nt / nn / tn), bit-identical to the per-expert loop on the forward (profiles/r02_moe.md).
//
// Three products, all bf16 x bf16 -> f32 accumulate -> bf16, group boundaries read on the device (no host sync):
//   NT  y[m][n]  = sum_k x[m][k]  * w[e][n][k]     forward                       x, w K-contiguous: both operands by ds_read_b128
//   NN  dx[m][k] = sum_n dy[m][n] * w[e][n][k]     gradient w.r.t. the rows      w tile read TRANSPOSED out of LDS (ds_read_b64_tr_b16)
//   TN  dw[e][n][k] = sum_{m in group e} dy[m][n] * x[m][k]                      both tiles read transposed; one workgroup per (e, tile)
// so no transposed copy of the expert weights and no transposed activations ever exist in HBM.
//
// MI355X design (the GEMM core of lmhead.hip, re-targeted):
//   * v_mfma_f32_16x16x32_bf16, TRANSPOSED product (the MFMA A operand is the operand whose rows become the output's CONTIGUOUS axis), so
//     a lane of the C fragment holds 4 consecutive output columns of one output row: 8-byte stores.
//   * 128 x 128 output tile per workgroup, 4 waves as 2 x 2 (64 x 64 = 4 x 4 accumulator fragments each), contraction step 64, LDS
//     2 x 32 KB -> two workgroups per CU.
//   * tiles arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write pass; rows past the end of a tensor / of a
//     group read as zeros through the buffer range check (the TN kernel bounds its descriptor at the END OF THE GROUP, which is what
//     makes a ragged last contraction step exact).  XOR swizzles are applied on the source side of the DMA:
//       [128 rows][64 cols] tiles (operand K-contiguous): chunk ^ (row & 7), read with ds_read_b128;
//       [64 rows][128 cols] tiles (contraction along the rows): the attention kernels' chunk ^ f(row & 15), read with pairs of
//       ds_read_b64_tr_b16 addressed so that a fragment holds its 8 contraction values in NATURAL order (rows lq*8 .. lq*8+7: first
//       read rows +0..3, second +4..7) — the partner operand can then be a plain ds_read_b128 fragment.
//   * the two LDS buffers are two static arrays and the stage loop is unrolled by two (hipcc guards transpose-reads that may alias a
//     pending LDS-DMA with vmcnt(0); see attention_kernels.hip.h), explicit `s_waitcnt vmcnt(0)` before each barrier.
//   * NT / NN: 128-row tiles of the SORTED rows x 128-column tiles, block order XCD-aware (tile_of_block: 0.44 -> 0.72 PFLOP/s on the forward
//     product); a row tile never straddles two experts (per expert ceil(rows / 128) tiles), the tile -> expert search is a wave-parallel
//     prefix scan over the offsets, rows past the group's end are computed and not stored.
#include "common.hip.h"

namespace pm {
namespace gg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t* lds_v4i16_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int kT = 128;                 // output tile edge
constexpr int kBK = 64;                 // contraction per stage
constexpr int kTileBytes = 128 * 64 * 2;  // 16 KB: either [128][64] or [64][128] bf16
constexpr int kStageBytes = 2 * kTileBytes;

__device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ int swz_f(int r) {  // r in [0, 16): the attention kernels' swizzle for 256-byte rows
  return (((r >> 3) ^ (r >> 2)) & 1) << 3 | (r & 2) << 1 | (r & 1) << 1 | ((r >> 2) & 1);
}

// [128 rows][64 cols] tile (128-byte rows), rows row0.. of a row-major tensor with `row_bytes` per row, columns starting at byte col_byte0:
// 16 one-KiB DMA pieces of 8 rows; wave w issues pieces 4w..4w+3; lane L lands at (row 8p + L/8, chunk L%8) and fetches chunk (L%8)^(row&7).
__device__ __forceinline__ void dma_tile_k64(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int64_t row0, int row_bytes, int col_byte0, int wave, int lane) {
  const int sub = lane >> 3, ch = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave * 4 + i;
    const int row = p * 8 + sub;
    const int voff = (int)((row0 + row) * row_bytes) + col_byte0 + ((ch ^ (row & 7)) << 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(dst + p * 1024), 16, voff, 0, 0, 0);
  }
}
// [64 rows][128 cols] tile (256-byte rows): 16 pieces of 4 rows; lane L lands at (row 4p + L/16, chunk L%16), fetches chunk (L%16)^f(row&15).
__device__ __forceinline__ void dma_tile_r64(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int64_t row0, int row_bytes, int col_byte0, int wave, int lane) {
  const int sub = lane >> 4, ch = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave * 4 + i;
    const int row = p * 4 + sub;
    const int voff = (int)((row0 + row) * row_bytes) + col_byte0 + ((ch ^ swz_f(row & 15)) << 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(dst + p * 1024), 16, voff, 0, 0, 0);
  }
}
// [64 rows][256 cols] tile (512-byte rows) for the 256 x 256 kernels: 32 pieces of 2 rows; lane L lands at (row 2p + L/32, chunk L%32) and fetches
// chunk (L%32) ^ f(row & 15) (the 4-bit XOR stays inside a 256-byte half row, so the bank picture is the one of the 256-byte rows).  Wave w of 8
// issues pieces 4w..4w+3.
__device__ __forceinline__ void dma_tile_r64w(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int64_t row0, int row_bytes, int col_byte0, int wave, int lane) {
  const int sub = lane >> 5, ch = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave * 4 + i;
    const int row = p * 2 + sub;
    const int voff = (int)((row0 + row) * row_bytes) + col_byte0 + ((ch ^ swz_f(row & 15)) << 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(dst + p * 1024), 16, voff, 0, 0, 0);
  }
}
// [256 rows][64 cols] K-contiguous tile: 32 pieces of 8 rows, wave w of 8 issues pieces 4w..4w+3
__device__ __forceinline__ void dma_tile_k64w(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int64_t row0, int row_bytes, int col_byte0, int wave, int lane) {
  const int sub = lane >> 3, ch = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave * 4 + i;
    const int row = p * 8 + sub;
    const int voff = (int)((row0 + row) * row_bytes) + col_byte0 + ((ch ^ (row & 7)) << 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(dst + p * 1024), 16, voff, 0, 0, 0);
  }
}
__device__ __forceinline__ void tr_bases_w(int kk, int lr, int lq, int (&base)[2]) {  // tr_bases for 512-byte rows
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = kk * 32 + lq * 8 + (lr >> 2) + 4 * h;
    base[h] = row * 512 + (((((lr & 3) >> 1) ^ swz_f(row & 15)) << 4) | ((lr & 1) << 3));
  }
}

// K-contiguous fragment (16 rows x 32 k) of a [128][64] tile: lane (lr, lq) reads row r0 + lr, 16-byte chunk kk*4 + lq
__device__ __forceinline__ u32x4 frag_k(const unsigned char* tile, int r0, int kk, int lr, int lq) {
  const int row = r0 + lr;
  return *reinterpret_cast<const u32x4*>(tile + row * 128 + (((kk * 4 + lq) ^ (row & 7)) << 4));
}
// Transposed fragment of a [64][128] tile: 16 consecutive tile COLUMNS (column block cb) x 32 contraction rows (k-step kk), natural order.
// `base` = byte offsets of this lane's two reads for column block 0 (precomputed per k-step by tr_bases); a column block is an XOR.
__device__ __forceinline__ void tr_bases(int kk, int lr, int lq, int (&base)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = kk * 32 + lq * 8 + (lr >> 2) + 4 * h;
    base[h] = row * 256 + (((((lr & 3) >> 1) ^ swz_f(row & 15)) << 4) | ((lr & 1) << 3));
  }
}
__device__ __forceinline__ u32x4 frag_tr(const unsigned char* tile, const int (&base)[2], int cb) {
  const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(tile + (base[0] ^ (cb << 5))));
  const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(tile + (base[1] ^ (cb << 5))));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  return u32x4{l2.x, l2.y, h2.x, h2.y};
}

// ROWS-row tile `t` of the sorted rows -> (expert, first row, end of the expert's rows); false past the last tile.  Every wave runs the
// same prefix scan over ceil(rows_e / ROWS).
template <int ROWS = kT>
__device__ __forceinline__ bool locate_tile(const int32_t* __restrict__ offsets, int n_experts, int t, int lane, int& expert, int& row0, int& row_end) {
  int base = 0;
  for (int e0 = 0; e0 < n_experts; e0 += 64) {
    const int e = e0 + lane;
    int lo = 0, hi = 0;
    if (e < n_experts) {
      lo = offsets[e];
      hi = offsets[e + 1];
    }
    const int tiles = (hi - lo + ROWS - 1) / ROWS;
    int incl = tiles;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const unsigned long long hit = __ballot(base + incl > t);
    if (hit) {
      const int src = __ffsll((long long)hit) - 1;
      const int first_tile = base + __shfl(incl, src, 64) - __shfl(tiles, src, 64);
      expert = e0 + src;
      row0 = __shfl(lo, src, 64) + (t - first_tile) * ROWS;
      row_end = __shfl(hi, src, 64);
      return true;
    }
    base += __shfl(incl, 63, 64);
  }
  return false;
}

// blockIdx.x -> (row tile, column tile).  Block b runs on XCD b % 8 (its own L2): the `ncol` column tiles of one row tile — which all
// stream the same 128 rows of x / dy — are consecutive blocks of ONE XCD, so the rows leave HBM once, not ncol times.
__device__ __forceinline__ void tile_of_block(int b, int ncol, int& row_tile, int& col_tile) {
  const int xcd = b & 7, q = b >> 3;
  row_tile = (q / ncol) * 8 + xcd;
  col_tile = q % ncol;
}

// 4 consecutive output columns of one output row, bf16
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& c) { *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16_hw(c[0], c[1]), pack_bf16_hw(c[2], c[3])}; }

// ---------------------------------------------------------------------------------------------------------------------
// NT: y[m][n0 + ..] = x[m][:] . w[e][n][:]   (A = w rows, B = x rows; both [128][64] tiles, contraction k)
// Two weight sets: output columns [0, n_a) come from w_a [E, n_a, K], columns [n_a, n_out) from w_b [E, n_out - n_a, K] (the gate and up
// projections of the experts in ONE launch, written side by side as the gate|up layout pm_swiglu_fwd consumes); w_b unused when n_a == n_out.
__global__ __launch_bounds__(256, 2) void grouped_nt_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w_a, const bf16_t* __restrict__ w_b,
                                                           const int32_t* __restrict__ offsets, int n_experts, int64_t m_total, int n_a, int n_out, int k_dim,
                                                           bf16_t* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) unsigned char smA[kStageBytes];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kStageBytes];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave >> 1, wt = wave & 1;
  int expert, row0, row_end, row_tile, col_tile;
  tile_of_block(blockIdx.x, n_out / kT, row_tile, col_tile);
  if (!locate_tile(offsets, n_experts, row_tile, lane, expert, row0, row_end)) return;
  const int n0 = col_tile * kT;                    // output column
  const bool second = n0 >= n_a;
  const int wn = second ? n_out - n_a : n_a;        // rows of the selected weight set
  const int nw0 = second ? n0 - n_a : n0;           // first row inside it
  const bf16_t* w = second ? w_b : w_a;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, (int)(m_total * k_dim * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(w + (int64_t)expert * wn * k_dim), 0, wn * k_dim * 2, 0x00020000);
  const int row_bytes = k_dim * 2;
  const int nk = k_dim / kBK;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](unsigned char* sm, int s) {
    dma_tile_k64(wr, sm, nw0, row_bytes, s * kBK * 2, wave, lane);
    dma_tile_k64(xr, sm + kTileBytes, row0, row_bytes, s * kBK * 2, wave, lane);
  };
  auto compute = [&](const unsigned char* sm) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = frag_k(sm, wv * 64 + i * 16, kk, lr, lq);
        bfr[i] = frag_k(sm + kTileBytes, wt * 64 + i * 16, kk, lr, lq);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma(af[a], bfr[b], acc[a][b]);
    }
  };
  stage(smA, 0);
  for (int s = 0; s < nk; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nk) stage(smB, s + 1);
    compute(smA);
    if (s + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < nk) stage(smA, s + 2);
      compute(smB);
    }
  }
  // acc[a][b][r] = y[row0 + wt*64 + b*16 + lr][n0 + wv*64 + a*16 + lq*4 + r]
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int row = row0 + wt * 64 + b * 16 + lr;
    if (row < row_end) {
#pragma unroll
      for (int a = 0; a < 4; ++a) store4(y + (int64_t)row * n_out + n0 + wv * 64 + a * 16 + lq * 4, acc[a][b]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NT on 256 x 256 tiles: 8 waves (2 along the weight rows x 4 along the tokens), 128 x 64 accumulators per wave (8 x 4 fragments = 128
// VGPRs), one workgroup per CU (two waves per SIMD), LDS 2 x 64 KB.  Against the 128 x 128 kernel: half the L2 -> LDS bytes per flop (the
// 128 x 128 tile saturates a CU's 64 B/clk L2 port at the full MFMA rate), 12 instead of 16 fragment reads per 32 MFMAs, 64 MFMAs per
// wave between barriers instead of 32, and a DMA stage has 2048 MFMA cycles to land.
constexpr int kBigTile = 256;
constexpr int kBigOperand = kBigTile * kBK * 2;  // 32 KB
constexpr int kBigStage = 2 * kBigOperand;       // 64 KB

__global__ __launch_bounds__(512, 1) void grouped_nt_big_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w_a, const bf16_t* __restrict__ w_b,
                                                               const int32_t* __restrict__ offsets, int n_experts, int64_t m_total, int n_a, int n_out, int k_dim,
                                                               bf16_t* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) unsigned char smA[kBigStage];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kBigStage];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave >> 2, wt = wave & 3;  // wv: 128 weight rows, wt: 64 tokens
  int expert, row0, row_end, row_tile, col_tile;
  tile_of_block(blockIdx.x, n_out / kBigTile, row_tile, col_tile);
  if (!locate_tile<kBigTile>(offsets, n_experts, row_tile, lane, expert, row0, row_end)) return;
  const int n0 = col_tile * kBigTile;
  const bool second = n0 >= n_a;
  const int wn = second ? n_out - n_a : n_a;
  const int nw0 = second ? n0 - n_a : n0;
  const bf16_t* w = second ? w_b : w_a;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, (int)(m_total * k_dim * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(w + (int64_t)expert * wn * k_dim), 0, wn * k_dim * 2, 0x00020000);
  const int row_bytes = k_dim * 2;
  const int nk = k_dim / kBK;
  const int sub = lane >> 3, ch = lane & 7;
  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](unsigned char* sm, int s) {  // 32 + 32 one-KiB pieces (8 rows x 8 chunks): wave w issues pieces 4w..4w+3 of each operand
    const int kb = s * kBK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave * 4 + i, row = p * 8 + sub;
      const int swz = (ch ^ (row & 7)) << 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(sm + p * 1024), 16, (nw0 + row) * row_bytes + kb + swz, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(sm + kBigOperand + p * 1024), 16, (int)((int64_t)(row0 + row) * row_bytes) + kb + swz, 0, 0, 0);
    }
  };
  auto compute = [&](const unsigned char* sm) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[8], bfr[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = frag_k(sm, wv * 128 + i * 16, kk, lr, lq);
#pragma unroll
      for (int i = 0; i < 4; ++i) bfr[i] = frag_k(sm + kBigOperand, wt * 64 + i * 16, kk, lr, lq);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma(af[a], bfr[b], acc[a][b]);
    }
  };
  stage(smA, 0);
  for (int s = 0; s < nk; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nk) stage(smB, s + 1);
    compute(smA);
    if (s + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < nk) stage(smA, s + 2);
      compute(smB);
    }
  }
  // acc[a][b][r] = y[row0 + wt*64 + b*16 + lr][n0 + wv*128 + a*16 + lq*4 + r]
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int row = row0 + wt * 64 + b * 16 + lr;
    if (row < row_end) {
#pragma unroll
      for (int a = 0; a < 8; ++a) store4(y + (int64_t)row * n_out + n0 + wv * 128 + a * 16 + lq * 4, acc[a][b]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NN: dx[m][k0 + ..] = dy[m][:] . w[e][:][k]   (A = w^T: transposed fragments of the [64 n][128 k] tile; B = dy rows: [128 m][64 n] tile)
// Two weight sets: the contraction runs over dy's n_dim columns, the first n_a against w_a, the rest against w_b (dx = d_gate . w1 + d_up . w3).
__global__ __launch_bounds__(256, 2) void grouped_nn_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w_a, const bf16_t* __restrict__ w_b,
                                                           const int32_t* __restrict__ offsets, int n_experts, int64_t m_total, int n_a, int n_dim, int k_out,
                                                           bf16_t* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) unsigned char smA[kStageBytes];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kStageBytes];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave >> 1, wt = wave & 1;
  int expert, row0, row_end, row_tile, col_tile;
  tile_of_block(blockIdx.x, k_out / kT, row_tile, col_tile);
  if (!locate_tile(offsets, n_experts, row_tile, lane, expert, row0, row_end)) return;
  const int k0 = col_tile * kT;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, (int)(m_total * n_dim * 2), 0x00020000);
  const int n_b = n_dim - n_a;
  const __amdgpu_buffer_rsrc_t wra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(w_a + (int64_t)expert * n_a * k_out), 0, n_a * k_out * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(n_b ? w_b + (int64_t)expert * n_b * k_out : w_a), 0, n_b * k_out * 2, 0x00020000);
  const int ns = n_dim / kBK;
  int tb[2][2];
  tr_bases(0, lr, lq, tb[0]);
  tr_bases(1, lr, lq, tb[1]);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](unsigned char* sm, int s) {
    if (s * kBK < n_a) dma_tile_r64(wra, sm, (int64_t)s * kBK, k_out * 2, k0 * 2, wave, lane);        // w_a[e][n = s*64 ..][k0 ..]
    else dma_tile_r64(wrb, sm, (int64_t)s * kBK - n_a, k_out * 2, k0 * 2, wave, lane);
    dma_tile_k64(yr, sm + kTileBytes, row0, n_dim * 2, s * kBK * 2, wave, lane);         // dy[row0 ..][n = s*64 ..]
  };
  auto compute = [&](const unsigned char* sm) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = frag_tr(sm, tb[kk], wv * 4 + i);
        bfr[i] = frag_k(sm + kTileBytes, wt * 64 + i * 16, kk, lr, lq);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma(af[a], bfr[b], acc[a][b]);
    }
  };
  stage(smA, 0);
  for (int s = 0; s < ns; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < ns) stage(smB, s + 1);
    compute(smA);
    if (s + 1 < ns) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < ns) stage(smA, s + 2);
      compute(smB);
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int row = row0 + wt * 64 + b * 16 + lr;
    if (row < row_end) {
#pragma unroll
      for (int a = 0; a < 4; ++a) store4(dx + (int64_t)row * k_out + k0 + wv * 64 + a * 16 + lq * 4, acc[a][b]);
    }
  }
}

// NN on 256 x 256 tiles (see grouped_nt_big_kernel): A = transposed fragments of the [64 n][256 k] weight tile, B = [256 m][64 n] rows of dy.
__global__ __launch_bounds__(512, 1) void grouped_nn_big_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w_a, const bf16_t* __restrict__ w_b,
                                                               const int32_t* __restrict__ offsets, int n_experts, int64_t m_total, int n_a, int n_dim, int k_out,
                                                               bf16_t* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) unsigned char smA[kBigStage];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kBigStage];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave >> 2, wt = wave & 3;  // wv: 128 output columns k, wt: 64 rows m
  int expert, row0, row_end, row_tile, col_tile;
  tile_of_block(blockIdx.x, k_out / kBigTile, row_tile, col_tile);
  if (!locate_tile<kBigTile>(offsets, n_experts, row_tile, lane, expert, row0, row_end)) return;
  const int k0 = col_tile * kBigTile;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, (int)(m_total * n_dim * 2), 0x00020000);
  const int n_b = n_dim - n_a;
  const __amdgpu_buffer_rsrc_t wra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(w_a + (int64_t)expert * n_a * k_out), 0, n_a * k_out * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(n_b ? w_b + (int64_t)expert * n_b * k_out : w_a), 0, n_b * k_out * 2, 0x00020000);
  const int ns = n_dim / kBK;
  int tb[2][2];
  tr_bases_w(0, lr, lq, tb[0]);
  tr_bases_w(1, lr, lq, tb[1]);
  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](unsigned char* sm, int s) {
    if (s * kBK < n_a) dma_tile_r64w(wra, sm, (int64_t)s * kBK, k_out * 2, k0 * 2, wave, lane);
    else dma_tile_r64w(wrb, sm, (int64_t)s * kBK - n_a, k_out * 2, k0 * 2, wave, lane);
    dma_tile_k64w(yr, sm + kBigOperand, row0, n_dim * 2, s * kBK * 2, wave, lane);
  };
  auto compute = [&](const unsigned char* sm) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[8], bfr[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = frag_tr(sm, tb[kk], wv * 8 + i);
#pragma unroll
      for (int i = 0; i < 4; ++i) bfr[i] = frag_k(sm + kBigOperand, wt * 64 + i * 16, kk, lr, lq);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma(af[a], bfr[b], acc[a][b]);
    }
  };
  stage(smA, 0);
  for (int s = 0; s < ns; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < ns) stage(smB, s + 1);
    compute(smA);
    if (s + 1 < ns) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < ns) stage(smA, s + 2);
      compute(smB);
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int row = row0 + wt * 64 + b * 16 + lr;
    if (row < row_end) {
#pragma unroll
      for (int a = 0; a < 8; ++a) store4(dx + (int64_t)row * k_out + k0 + wv * 128 + a * 16 + lq * 4, acc[a][b]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN: dw[e][n0 + ..][k0 + ..] = sum over the group's rows of dy[m][n] * x[m][k]   (A = x^T, B = dy^T: both transposed fragments of [64 m][128] tiles)
// Two outputs: dy columns [0, n_a) give dw_a [E, n_a, K], columns [n_a, n_dim) give dw_b [E, n_dim - n_a, K].
__global__ __launch_bounds__(256, 2) void grouped_tn_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const int32_t* __restrict__ offsets,
                                                           int n_a, int n_dim, int k_dim, bf16_t* __restrict__ dw_a, bf16_t* __restrict__ dw_b) {
  __shared__ __attribute__((aligned(16))) unsigned char smA[kStageBytes];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kStageBytes];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave >> 1, wt = wave & 1;
  const int expert = blockIdx.y;
  const int tiles_k = k_dim / kT;
  const int n0 = (blockIdx.x / tiles_k) * kT, k0 = (blockIdx.x % tiles_k) * kT;
  const int lo = offsets[expert], hi = offsets[expert + 1];
  // descriptors end at the END OF THE GROUP: rows past it (the next expert's) read as zeros, so the last, ragged stage adds nothing wrong
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, hi * n_dim * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, hi * k_dim * 2, 0x00020000);
  const int ns = (hi - lo + kBK - 1) / kBK;
  int tb[2][2];
  tr_bases(0, lr, lq, tb[0]);
  tr_bases(1, lr, lq, tb[1]);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](unsigned char* sm, int s) {
    dma_tile_r64(xr, sm, (int64_t)lo + s * kBK, k_dim * 2, k0 * 2, wave, lane);               // x[m = lo + s*64 ..][k0 ..]
    dma_tile_r64(yr, sm + kTileBytes, (int64_t)lo + s * kBK, n_dim * 2, n0 * 2, wave, lane);  // dy[m ..][n0 ..]
  };
  auto compute = [&](const unsigned char* sm) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = frag_tr(sm, tb[kk], wv * 4 + i);                 // rows k = k0 + wv*64 + i*16 + ..
        bfr[i] = frag_tr(sm + kTileBytes, tb[kk], wt * 4 + i);    // cols n = n0 + wt*64 + i*16 + lr
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma(af[a], bfr[b], acc[a][b]);
    }
  };
  if (ns > 0) stage(smA, 0);
  for (int s = 0; s < ns; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < ns) stage(smB, s + 1);
    compute(smA);
    if (s + 1 < ns) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < ns) stage(smA, s + 2);
      compute(smB);
    }
  }
  // acc[a][b][r] = dw[e][n0 + wt*64 + b*16 + lr][k0 + wv*64 + a*16 + lq*4 + r]; an empty group writes zeros
  const bool second = n0 >= n_a;
  bf16_t* out = second ? dw_b + (int64_t)expert * (n_dim - n_a) * k_dim : dw_a + (int64_t)expert * n_a * k_dim;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = (second ? n0 - n_a : n0) + wt * 64 + b * 16 + lr;
#pragma unroll
    for (int a = 0; a < 4; ++a) store4(out + (int64_t)n * k_dim + k0 + wv * 64 + a * 16 + lq * 4, acc[a][b]);
  }
}

// TN on 256 x 256 tiles: A = x^T, B = dy^T, both transposed fragments of [64 m][256] tiles; one workgroup per (expert, 256 n x 256 k tile).
__global__ __launch_bounds__(512, 1) void grouped_tn_big_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const int32_t* __restrict__ offsets,
                                                               int n_a, int n_dim, int k_dim, bf16_t* __restrict__ dw_a, bf16_t* __restrict__ dw_b) {
  __shared__ __attribute__((aligned(16))) unsigned char smA[kBigStage];
  __shared__ __attribute__((aligned(16))) unsigned char smB[kBigStage];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave >> 2, wt = wave & 3;  // wv: 128 columns k, wt: 64 rows n
  const int expert = blockIdx.y;
  const int tiles_k = k_dim / kBigTile;
  const int n0 = (blockIdx.x / tiles_k) * kBigTile, k0 = (blockIdx.x % tiles_k) * kBigTile;
  const int lo = offsets[expert], hi = offsets[expert + 1];
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, hi * n_dim * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, hi * k_dim * 2, 0x00020000);
  const int ns = (hi - lo + kBK - 1) / kBK;
  int tb[2][2];
  tr_bases_w(0, lr, lq, tb[0]);
  tr_bases_w(1, lr, lq, tb[1]);
  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](unsigned char* sm, int s) {
    dma_tile_r64w(xr, sm, (int64_t)lo + s * kBK, k_dim * 2, k0 * 2, wave, lane);
    dma_tile_r64w(yr, sm + kBigOperand, (int64_t)lo + s * kBK, n_dim * 2, n0 * 2, wave, lane);
  };
  auto compute = [&](const unsigned char* sm) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[8], bfr[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = frag_tr(sm, tb[kk], wv * 8 + i);                  // rows k = k0 + wv*128 + i*16 + ..
#pragma unroll
      for (int i = 0; i < 4; ++i) bfr[i] = frag_tr(sm + kBigOperand, tb[kk], wt * 4 + i);   // cols n = n0 + wt*64 + i*16 + lr
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma(af[a], bfr[b], acc[a][b]);
    }
  };
  if (ns > 0) stage(smA, 0);
  for (int s = 0; s < ns; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < ns) stage(smB, s + 1);
    compute(smA);
    if (s + 1 < ns) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 2 < ns) stage(smA, s + 2);
      compute(smB);
    }
  }
  const bool second = n0 >= n_a;
  bf16_t* out = second ? dw_b + (int64_t)expert * (n_dim - n_a) * k_dim : dw_a + (int64_t)expert * n_a * k_dim;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = (second ? n0 - n_a : n0) + wt * 64 + b * 16 + lr;
#pragma unroll
    for (int a = 0; a < 8; ++a) store4(out + (int64_t)n * k_dim + k0 + wv * 128 + a * 16 + lq * 4, acc[a][b]);
  }
}

}  // namespace gg
}  // namespace pm

using namespace pm;

// Round 3: the three products run on the ping-pong core of gemm.hip (a problem list in its persistent tile stream: gm::grouped_nt / nn / tn) whenever the
// shape is made of whole 256-wide tiles; the kernels of this file remain for the other shapes (and as the A/B reference).
namespace pm {
namespace gm {
bool grouped_shapes_ok(int64_t n_first, int64_t n, int64_t k_or_second);
int grouped_nt(const bf16_t* x, const bf16_t* w, const bf16_t* w2, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t n_first, int64_t n_out,
               int64_t k_dim, bf16_t* y, hipStream_t st, int sched);
int grouped_nn(const bf16_t* dy, const bf16_t* w, const bf16_t* w2, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t n_first, int64_t n_dim,
               int64_t k_out, bf16_t* dx, hipStream_t st, int sched);
int grouped_tn(const bf16_t* dy, const bf16_t* x, const int32_t* offsets, int n_experts, int64_t n_first, int64_t n_dim, int64_t k_dim, void* dw, void* dw2,
               int out_f32, int accumulate, float* tile_sumsq, hipStream_t st, int sched);
int grouped_nn_swiglu_bwd(const bf16_t* dy, const bf16_t* w, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t n_dim, int64_t inter,
                          const bf16_t* gate_up, bf16_t* d_gate_up, hipStream_t st, int sched);
int grouped_nt_swiglu(const bf16_t* x, const bf16_t* w, const bf16_t* w2, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t inter, int64_t k_dim,
                      bf16_t* gate_up, bf16_t* act, hipStream_t st, int sched);
}  // namespace gm
}  // namespace pm

// Which kernels run a grouped product travels in the launch's `sched` word (no process-wide state): bits 16-17 = 0: the gemm.hip core where the shape allows,
// 2 (PM_GROUPED_SCHED_BIG): this file's 256 x 256 kernels, 1 (PM_GROUPED_SCHED_SMALL): 128 x 128 tiles everywhere — the A/B references of the GPU tests;
// the low bits are the dense core's schedule word (-1 = default).  pm_grouped_gemm_debug_set_variant keeps a process-wide override in `make VARIANTS=1`
// builds only.
#if PM_VARIANTS
static int g_nt_override = -1;
#endif
static inline int nt_variant_of(int sched) {
#if PM_VARIANTS
  if (g_nt_override >= 0) return g_nt_override;
#endif
  return sched < 0 ? 0 : (sched >> 16) & 3;
}
static inline int core_sched_of(int sched) { return sched < 0 ? -1 : (sched & 0xffff); }
extern "C" int pm_grouped_gemm_debug_set_variant(int32_t value) {
#if PM_VARIANTS
  g_nt_override = value;
  return PM_OK;
#else
  PM_REQUIRE(value < 0, PM_ERR_UNSUPPORTED, "pm_grouped_gemm_debug_set_variant(%d): the product library holds no variant state — pass PM_GROUPED_SCHED_* in the launch's "
             "`sched` word", (int)value);
  return PM_OK;
#endif
}

static bool gg_shapes_ok(int64_t m, int64_t n, int64_t k) { return n % 128 == 0 && k % 128 == 0 && m * (n > k ? n : k) * 2 < ((int64_t)1 << 31); }

extern "C" int pm_grouped_gemm_supported(int64_t n_rows, int64_t n_out, int64_t k_dim) { return gg_shapes_ok(n_rows, n_out, k_dim) ? 1 : 0; }

extern "C" int pm_grouped_gemm_nt(const void* x_bf16, const void* w_bf16, const void* w2_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows,
                                  int64_t n_first, int64_t n_out, int64_t k_dim, void* y_bf16, int32_t sched, void* stream) {
  const int g_nt_variant = nt_variant_of(sched);
  (void)g_nt_variant;
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_nt: bad sizes");
  PM_REQUIRE(w_bf16 && offsets && (n_rows == 0 || (x_bf16 && y_bf16)) && (n_first == n_out || w2_bf16), PM_ERR_INVALID_ARG, "pm_grouped_gemm_nt: null pointer");
  PM_REQUIRE(n_first > 0 && n_first <= n_out && n_first % 128 == 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_nt: n_first must be a multiple of 128 in (0, n_out]");
  PM_REQUIRE(gg_shapes_ok(n_rows, n_out, k_dim), PM_ERR_UNSUPPORTED, "pm_grouped_gemm_nt: n_out=%lld k=%lld must be multiples of 128 (and the row tensors < 2 GiB)",
             (long long)n_out, (long long)k_dim);
  if (n_rows == 0) return PM_OK;
  if (g_nt_variant == 0 && gm::grouped_shapes_ok(n_first, n_out, k_dim) && (n_out - n_first) % 256 == 0) {
    gm::grouped_nt((const bf16_t*)x_bf16, (const bf16_t*)w_bf16, (const bf16_t*)w2_bf16, offsets, n_experts, n_rows, n_first, n_out, k_dim, (bf16_t*)y_bf16, as_stream(stream), core_sched_of(sched));
    PM_CHECK_LAUNCH("pm_grouped_gemm_nt");
    return PM_OK;
  }
  if (g_nt_variant != 1 && n_first % 256 == 0 && n_out % 256 == 0) {
    const dim3 gridb((unsigned)(cdiv(cdiv(n_rows, 256) + n_experts, 8) * 8 * (n_out / 256)));
    hipLaunchKernelGGL(gg::grouped_nt_big_kernel, gridb, dim3(512), 0, as_stream(stream), (const bf16_t*)x_bf16, (const bf16_t*)w_bf16, (const bf16_t*)w2_bf16,
                       offsets, n_experts, n_rows, (int)n_first, (int)n_out, (int)k_dim, (bf16_t*)y_bf16);
    PM_CHECK_LAUNCH("pm_grouped_gemm_nt");
    return PM_OK;
  }
  const dim3 grid((unsigned)(cdiv(cdiv(n_rows, 128) + n_experts, 8) * 8 * (n_out / 128)));
  hipLaunchKernelGGL(gg::grouped_nt_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x_bf16, (const bf16_t*)w_bf16, (const bf16_t*)w2_bf16, offsets,
                     n_experts, n_rows, (int)n_first, (int)n_out, (int)k_dim, (bf16_t*)y_bf16);
  PM_CHECK_LAUNCH("pm_grouped_gemm_nt");
  return PM_OK;
}

extern "C" int pm_grouped_gemm_nn(const void* dy_bf16, const void* w_bf16, const void* w2_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows,
                                  int64_t n_first, int64_t n_dim, int64_t k_out, void* dx_bf16, int32_t sched, void* stream) {
  const int g_nt_variant = nt_variant_of(sched);
  (void)g_nt_variant;
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_nn: bad sizes");
  PM_REQUIRE(w_bf16 && offsets && (n_rows == 0 || (dy_bf16 && dx_bf16)) && (n_first == n_dim || w2_bf16), PM_ERR_INVALID_ARG, "pm_grouped_gemm_nn: null pointer");
  PM_REQUIRE(n_first > 0 && n_first <= n_dim && n_first % 128 == 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_nn: n_first must be a multiple of 128 in (0, n_dim]");
  PM_REQUIRE(gg_shapes_ok(n_rows, n_dim, k_out), PM_ERR_UNSUPPORTED, "pm_grouped_gemm_nn: n=%lld k=%lld must be multiples of 128 (and the row tensors < 2 GiB)",
             (long long)n_dim, (long long)k_out);
  if (n_rows == 0) return PM_OK;
  if (g_nt_variant == 0 && k_out % 256 == 0 && n_first % 128 == 0 && (n_dim - n_first) % 128 == 0) {
    gm::grouped_nn((const bf16_t*)dy_bf16, (const bf16_t*)w_bf16, (const bf16_t*)w2_bf16, offsets, n_experts, n_rows, n_first, n_dim, k_out, (bf16_t*)dx_bf16, as_stream(stream), core_sched_of(sched));
    PM_CHECK_LAUNCH("pm_grouped_gemm_nn");
    return PM_OK;
  }
  if (g_nt_variant != 1 && k_out % 256 == 0) {
    const dim3 gridb((unsigned)(cdiv(cdiv(n_rows, 256) + n_experts, 8) * 8 * (k_out / 256)));
    hipLaunchKernelGGL(gg::grouped_nn_big_kernel, gridb, dim3(512), 0, as_stream(stream), (const bf16_t*)dy_bf16, (const bf16_t*)w_bf16, (const bf16_t*)w2_bf16,
                       offsets, n_experts, n_rows, (int)n_first, (int)n_dim, (int)k_out, (bf16_t*)dx_bf16);
    PM_CHECK_LAUNCH("pm_grouped_gemm_nn");
    return PM_OK;
  }
  const dim3 grid((unsigned)(cdiv(cdiv(n_rows, 128) + n_experts, 8) * 8 * (k_out / 128)));
  hipLaunchKernelGGL(gg::grouped_nn_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)dy_bf16, (const bf16_t*)w_bf16, (const bf16_t*)w2_bf16, offsets,
                     n_experts, n_rows, (int)n_first, (int)n_dim, (int)k_out, (bf16_t*)dx_bf16);
  PM_CHECK_LAUNCH("pm_grouped_gemm_nn");
  return PM_OK;
}

// gate_up[m] = x[m] . [w_gate | w_up][e(m)]^T and act[m] = silu(gate) * up in one launch (moe.py:181-182: F.silu(x @ w1^T) * (x @ w3^T)): the grouped
// gate|up forward with pm_swiglu_fwd in its epilogue — gate_up equals pm_grouped_gemm_nt(w, w2), act equals pm_swiglu_fwd(gate_up), bit for bit.  The two weight
// tensors are read through ONE buffer descriptor (the up half-tile = the gate half-tile's address + the distance between the tensors), so w_up must lie
// above w_gate within 2 GiB (parameters of one flat buffer do; PM_ERR_UNSUPPORTED otherwise — the caller keeps the two-launch form).
extern "C" int pm_grouped_gemm_nt_swiglu_supported(int64_t n_rows, int64_t inter, int64_t k_dim) {
  return inter > 0 && k_dim > 0 && inter % 128 == 0 && k_dim % 128 == 0 && gg_shapes_ok(n_rows, 2 * inter, k_dim) ? 1 : 0;
}

extern "C" int pm_grouped_gemm_nt_swiglu(const void* x_bf16, const void* w_gate_bf16, const void* w_up_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows,
                                         int64_t inter, int64_t k_dim, void* gate_up_bf16, void* act_bf16, int32_t sched, void* stream) {
  const int g_nt_variant = nt_variant_of(sched);
  (void)g_nt_variant;
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_nt_swiglu: bad sizes");
  PM_REQUIRE(w_gate_bf16 && w_up_bf16 && offsets && (n_rows == 0 || (x_bf16 && gate_up_bf16 && act_bf16)), PM_ERR_INVALID_ARG, "pm_grouped_gemm_nt_swiglu: null pointer");
  PM_REQUIRE(pm_grouped_gemm_nt_swiglu_supported(n_rows, inter, k_dim), PM_ERR_UNSUPPORTED,
             "pm_grouped_gemm_nt_swiglu: needs inter %% 128 == 0 and k %% 128 == 0 (inter %lld, k %lld; row tensors < 2 GiB)", (long long)inter, (long long)k_dim);
  const int64_t delta = (const unsigned char*)w_up_bf16 - (const unsigned char*)w_gate_bf16;
  const int64_t span = delta + ((int64_t)n_experts * inter) * k_dim * 2;  // from w_gate[0] to the end of w_up: what one descriptor must be able to address from any panel base
  // a whole number of weight ROWS apart: the tile's descriptor is sized in rows ((w_up - w_gate) / k_dim + 128), so a distance that is not a multiple of the row
  // pitch would cut the K tail of the last up row of every tile short (separately allocated w1 / w3 can sit at such a distance)
  PM_REQUIRE(delta > 0 && delta % (k_dim * 2) == 0 && delta + 128 * k_dim * 2 < ((int64_t)1 << 31) - 4096, PM_ERR_UNSUPPORTED,
             "pm_grouped_gemm_nt_swiglu: w_up must lie above w_gate, a whole number of %lld-byte weight rows away and within 2 GiB (distance %lld bytes)",
             (long long)(k_dim * 2), (long long)delta);
  (void)span;
  if (n_rows == 0) return PM_OK;
  gm::grouped_nt_swiglu((const bf16_t*)x_bf16, (const bf16_t*)w_gate_bf16, (const bf16_t*)w_up_bf16, offsets, n_experts, n_rows, inter, k_dim, (bf16_t*)gate_up_bf16,
                        (bf16_t*)act_bf16, as_stream(stream), core_sched_of(sched));
  PM_CHECK_LAUNCH("pm_grouped_gemm_nt_swiglu");
  return PM_OK;
}

// [d_gate | d_up][m] = SwiGLU-backward of (dy[m] . w_down[e(m)]) against the saved gate_up[m]: the experts' down-projection row gradient with pm_swiglu_bwd
// in its epilogue (moe.py:181-183 backward: the gradient of h = silu(x w1^T) * (x w3^T) never reaches HBM).  grad_gate_up may be gate_up (in place).
extern "C" int pm_grouped_gemm_nn_swiglu_bwd_supported(int64_t n_rows, int64_t n_dim, int64_t inter) {
  return inter > 0 && n_dim > 0 && inter % 256 == 0 && n_dim % 128 == 0 && gg_shapes_ok(n_rows, n_dim, 2 * inter) ? 1 : 0;
}

extern "C" int pm_grouped_gemm_nn_swiglu_bwd(const void* dy_bf16, const void* w_down_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_dim,
                                             int64_t inter, const void* gate_up_bf16, void* grad_gate_up_bf16, int32_t sched, void* stream) {
  const int g_nt_variant = nt_variant_of(sched);
  (void)g_nt_variant;
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_nn_swiglu_bwd: bad sizes");
  PM_REQUIRE(w_down_bf16 && offsets && (n_rows == 0 || (dy_bf16 && gate_up_bf16 && grad_gate_up_bf16)), PM_ERR_INVALID_ARG, "pm_grouped_gemm_nn_swiglu_bwd: null pointer");
  PM_REQUIRE(pm_grouped_gemm_nn_swiglu_bwd_supported(n_rows, n_dim, inter), PM_ERR_UNSUPPORTED,
             "pm_grouped_gemm_nn_swiglu_bwd: needs inter %% 256 == 0 and n_dim %% 128 == 0 (n_dim %lld, inter %lld; row tensors < 2 GiB)", (long long)n_dim, (long long)inter);
  if (n_rows == 0) return PM_OK;
  gm::grouped_nn_swiglu_bwd((const bf16_t*)dy_bf16, (const bf16_t*)w_down_bf16, offsets, n_experts, n_rows, n_dim, inter, (const bf16_t*)gate_up_bf16,
                            (bf16_t*)grad_gate_up_bf16, as_stream(stream), core_sched_of(sched));
  PM_CHECK_LAUNCH("pm_grouped_gemm_nn_swiglu_bwd");
  return PM_OK;
}

extern "C" int pm_grouped_gemm_tn(const void* dy_bf16, const void* x_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_first,
                                  int64_t n_dim, int64_t k_dim, void* dw_bf16, void* dw2_bf16, int32_t sched, void* stream) {
  const int g_nt_variant = nt_variant_of(sched);
  (void)g_nt_variant;
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn: bad sizes");
  PM_REQUIRE(offsets && dw_bf16 && (n_rows == 0 || (dy_bf16 && x_bf16)) && (n_first == n_dim || dw2_bf16), PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn: null pointer");
  PM_REQUIRE(n_first > 0 && n_first <= n_dim && n_first % 128 == 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn: n_first must be a multiple of 128 in (0, n_dim]");
  PM_REQUIRE(gg_shapes_ok(n_rows, n_dim, k_dim), PM_ERR_UNSUPPORTED, "pm_grouped_gemm_tn: n=%lld k=%lld must be multiples of 128 (and the row tensors < 2 GiB)",
             (long long)n_dim, (long long)k_dim);
  if (g_nt_variant == 0 && n_first % 256 == 0 && n_dim % 256 == 0 && k_dim % 256 == 0) {
    gm::grouped_tn((const bf16_t*)dy_bf16, (const bf16_t*)x_bf16, offsets, n_experts, n_first, n_dim, k_dim, dw_bf16, dw2_bf16, 0, 0, nullptr, as_stream(stream), core_sched_of(sched));
    PM_CHECK_LAUNCH("pm_grouped_gemm_tn");
    return PM_OK;
  }
  if (g_nt_variant != 1 && n_first % 256 == 0 && n_dim % 256 == 0 && k_dim % 256 == 0) {
    const dim3 gridb((unsigned)((n_dim / 256) * (k_dim / 256)), (unsigned)n_experts);
    hipLaunchKernelGGL(gg::grouped_tn_big_kernel, gridb, dim3(512), 0, as_stream(stream), (const bf16_t*)dy_bf16, (const bf16_t*)x_bf16, offsets, (int)n_first,
                       (int)n_dim, (int)k_dim, (bf16_t*)dw_bf16, (bf16_t*)dw2_bf16);
    PM_CHECK_LAUNCH("pm_grouped_gemm_tn");
    return PM_OK;
  }
  const dim3 grid((unsigned)((n_dim / 128) * (k_dim / 128)), (unsigned)n_experts);
  hipLaunchKernelGGL(gg::grouped_tn_kernel, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)dy_bf16, (const bf16_t*)x_bf16, offsets, (int)n_first, (int)n_dim,
                     (int)k_dim, (bf16_t*)dw_bf16, (bf16_t*)dw2_bf16);
  PM_CHECK_LAUNCH("pm_grouped_gemm_tn");
  return PM_OK;
}

// The weight gradient written (or accumulated) in f32 straight into a main-gradient buffer — what the dense engine's wgrad GEMMs do (pm_gemm_bf16 with an
// f32 output): no bf16 gradient tensor, no separate bf16 -> f32 accumulation pass over the expert parameters.  Runs on the gemm.hip core only.
extern "C" int pm_grouped_gemm_tn_f32(const void* dy_bf16, const void* x_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_first,
                                      int64_t n_dim, int64_t k_dim, float* dw_f32, float* dw2_f32, int32_t accumulate, int32_t sched, void* stream) {
  const int g_nt_variant = nt_variant_of(sched);
  (void)g_nt_variant;
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn_f32: bad sizes");
  PM_REQUIRE(offsets && dw_f32 && (n_rows == 0 || (dy_bf16 && x_bf16)) && (n_first == n_dim || dw2_f32), PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn_f32: null pointer");
  PM_REQUIRE(n_first > 0 && n_first <= n_dim && gg_shapes_ok(n_rows, n_dim, k_dim) && n_first % 256 == 0 && n_dim % 256 == 0 && k_dim % 256 == 0, PM_ERR_UNSUPPORTED,
             "pm_grouped_gemm_tn_f32: n_first=%lld n=%lld k=%lld must be multiples of 256 (and the row tensors < 2 GiB)", (long long)n_first, (long long)n_dim,
             (long long)k_dim);
  gm::grouped_tn((const bf16_t*)dy_bf16, (const bf16_t*)x_bf16, offsets, n_experts, n_first, n_dim, k_dim, dw_f32, dw2_f32, 1, accumulate ? 1 : 0, nullptr, as_stream(stream), core_sched_of(sched));
  PM_CHECK_LAUNCH("pm_grouped_gemm_tn_f32");
  return PM_OK;
}

// ... and with the gradient-norm pass fused into the epilogue (what pm_gemm_bf16_sumsq is to the dense weight gradients): every wave also writes the sum of
// squares of the FINAL f32 values it stored for output tile (expert e, tile t of its n_dim / 256 x k_dim / 256) to tile_sumsq[(e * tiles + t) * 8 + wave].
// Every tile of every expert is visited by every launch (an expert without rows stores zeros, or with accumulate = 1 what was there), so after the LAST
// accumulating launch of a step sum(tile_sumsq) is the squared norm of both gradients.
extern "C" int64_t pm_grouped_gemm_tn_sumsq_slots(int32_t n_experts, int64_t n_dim, int64_t k_dim) {
  return n_experts <= 0 || n_dim <= 0 || k_dim <= 0 ? 0 : (int64_t)n_experts * cdiv(n_dim, 256) * cdiv(k_dim, 256) * 8;
}
extern "C" int pm_grouped_gemm_tn_f32_sumsq(const void* dy_bf16, const void* x_bf16, const int32_t* offsets, int32_t n_experts, int64_t n_rows, int64_t n_first,
                                            int64_t n_dim, int64_t k_dim, float* dw_f32, float* dw2_f32, int32_t accumulate, float* tile_sumsq,
                                            int64_t tile_sumsq_len, int32_t sched, void* stream) {
  PM_REQUIRE(n_experts > 0 && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn_f32_sumsq: bad sizes");
  PM_REQUIRE(offsets && dw_f32 && (dy_bf16 && x_bf16) && (n_first == n_dim || dw2_f32), PM_ERR_INVALID_ARG, "pm_grouped_gemm_tn_f32_sumsq: null pointer");
  PM_REQUIRE(n_first > 0 && n_first <= n_dim && gg_shapes_ok(n_rows, n_dim, k_dim) && n_first % 256 == 0 && n_dim % 256 == 0 && k_dim % 256 == 0, PM_ERR_UNSUPPORTED,
             "pm_grouped_gemm_tn_f32_sumsq: n_first=%lld n=%lld k=%lld must be multiples of 256 (and the row tensors < 2 GiB)", (long long)n_first, (long long)n_dim,
             (long long)k_dim);
  PM_REQUIRE(tile_sumsq && tile_sumsq_len >= pm_grouped_gemm_tn_sumsq_slots(n_experts, n_dim, k_dim), PM_ERR_WORKSPACE,
             "pm_grouped_gemm_tn_f32_sumsq: tile_sumsq needs pm_grouped_gemm_tn_sumsq_slots(n_experts, n_dim, k_dim) floats");
  gm::grouped_tn((const bf16_t*)dy_bf16, (const bf16_t*)x_bf16, offsets, n_experts, n_first, n_dim, k_dim, dw_f32, dw2_f32, 1, accumulate ? 1 : 0, tile_sumsq,
                 as_stream(stream), core_sched_of(sched));
  PM_CHECK_LAUNCH("pm_grouped_gemm_tn_f32_sumsq");
  return PM_OK;
}

