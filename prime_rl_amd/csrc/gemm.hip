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
//   * 256 x 256 output tile per workgroup, contraction step 64, 8 waves = 2 (W rows) x 4 (X rows), 128 x 64 outputs per wave held as 2
//     x 4 x 2 x 2 accumulator fragments of v_mfma_f32_16x16x32_bf16 (128 registers; the first version used 32x32x16, which the chip's power limit
//     makes 8-13 % more expensive per flop: profiles/r03_gemm.md section 3b).  TRANSPOSED product: the MFMA A operand is the W fragment, so a lane
//     owns ONE X row (token) and 4 consecutive W rows (output columns) of every 16-column block: 8 / 16-byte stores along the contiguous axis of
//     OUT, and per-lane softmax state for the fused log-prob epilogue.
//   * PING-PONG: waves 0-3 (one per SIMD) and waves 4-7 (their SIMD partners) run the same phase stream one barrier apart.  A phase =
//     [load segment: LDS fragment reads + 2 LDS-DMA pieces] s_barrier [compute segment: 16 MFMAs = one 64 x 32 quadrant x K 64] s_barrier, so on
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
//     c ^ ((r >> 1) & 7) — conflict-free for the 16-row fragments of the 16x16x32 MFMA (and the 32-row ones of 32x32x16) against the real ds_read_b128 service groups
//     {0-3,12-15,20-27} {4-11,16-19,28-31} (+32).  Rows past the end of a tensor read as zeros through the buffer range check.
//   * blockIdx -> tile: block b runs on XCD b % 8; the 32 tiles an XCD runs concurrently form an 8 (X) x 4 (W) patch that walks K in step,
//     so 12 row panels feed 32 tiles through one L2, and the 8 XCDs work on patches that share their W panels.
#include "common.hip.h"
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
constexpr int kRing = 8 * kSlot;      // 128 KiB: the staging ring (two K-tiles)
constexpr int kSmem = kRing + 8 * 4096;  // + a 4 KiB epilogue block per wave = all 160 KiB of the CU

enum { SLOT_W0 = 0, SLOT_W1 = 1, SLOT_X0 = 2, SLOT_X1 = 3 };
// Ring layout: the four W slots (two buffers x two halves) fill the first 64 KiB, the four X slots the second, so that ONE per-lane address register
// per k-step reaches every slot of its operand through the 16-bit immediate offset of the LDS read.
__host__ __device__ constexpr int slot_in_window(int buf, int slot) { return (buf * 2 + (slot & 1)) * kSlot; }            // offset inside the operand's 64 KiB window
__host__ __device__ constexpr int slot_offset(int buf, int slot) { return (slot >= SLOT_X0 ? 4 * kSlot : 0) + slot_in_window(buf, slot); }

// The schedule word travels WITH EVERY CALL (the `sched` argument of the pm_gemm_* / pm_grouped_gemm_* entry points; include/prime_mi355x.h): the product
// library holds no mutable schedule state.  Only a `make VARIANTS=1` library keeps a process-wide override for its A/B builds (pm_gemm_debug_set_variant).
constexpr int kDefaultSched = 7;  // bit 2: persistent grid (one workgroup per CU walking its tiles)
#if PM_VARIANTS
static int g_variant_override = -1;
#endif
static inline int eff_sched(int sched) {
#if PM_VARIANTS
  if (g_variant_override >= 0) return g_variant_override;
#endif
  return sched < 0 ? kDefaultSched : sched;
}

struct Params {
  const bf16_t* x;    // M-side operand: [M, K] (K-contiguous) or [K, M] (K-major), row stride ldx (elements)
  const bf16_t* w;    // N-side operand: [N, K] or [K, N], row stride ldw
  void* out;          // [M, N] bf16 or f32, row stride ldo
  const bf16_t* bias; // [N] or null: added to the f32 accumulator before the rounding (F.linear)
  int64_t ldx, ldw, ldo;
  int m, n, k;
  int tiles_m, tiles_n;
  int n_blocks;       // virtual block ids (grid_blocks): a workgroup walks ids blockIdx.x, + gridDim.x, ...
  int out_f32;        // 0: bf16 out; 1: f32 out
  int nt_out;         // (VARIANTS=1 builds: the one-wave-per-SIMD kernel's non-temporal result stores, A/B)
  int sweep_n;        // patch order of the dense product (tile_of_block): 1 = consecutive patches share their X block and cycle through W
  int accumulate;     // f32 out only: out += product
  // Split-K of the LAST round of tiles (dense f32 products): when the tile count leaves 1-4 of the 8 XCD patch slots of the last round empty, each tail
  // patch is computed by split_s workgroups on different XCDs, every one over 1 / split_s of the contraction; part 0 writes `out` as usual, parts >= 1 write
  // whole f32 tiles into `split_ws` ([tail tile][part - 1][256 x 256]) and splitk_reduce_kernel adds them in part order (deterministic).
  int k_parts, k_chunk;  // dense: the contraction of every tile as k_parts CHAINED parts of k_chunk (one launch; a K-major operand's 32-bit offsets reach 2 GB)
  int split_s, split_rp, split_v0, split_k;
  float* split_ws;
  float* sumsq;       // f32 out only, optional: [n_blocks][8] — every wave writes the sum of squares of the FINAL values it stored for tile v to [v][wave]
  // grouped products over expert-sorted rows (MODE 1-3 of gemm_kernel; see the tile sources below)
  const int32_t* offsets;  // [n_experts + 1] on the device: rows of expert e = [offsets[e], offsets[e + 1])
  const bf16_t* w2;        // second weight set (the experts' up projection next to the gate projection), or null
  void* out2;              // second output (MODE 3: its weight gradient; MODE 4: act = silu(gate) * up, [M, n_first] bf16, row stride ldo2)
  int64_t ldo2;
  const bf16_t* gu;        // MODE 5 / 6: the saved [gate | up] of the SwiGLU whose output gradient this product is ([M, 2 n], row stride ldgu); OUT = its gradient
  int64_t ldgu;
  int n_experts;
  int n_first;             // columns / contraction rows served by the first weight set
  // MODE 8 / 9: the fused LM head (lmhead.hip, pm_lmhead_logprob_fwd / _bwd) on this tile stream — X = hidden [tokens, H], W = head weight [vocab, H];
  // per-token inputs, and (MODE 8) the per-(vocab tile, wave group) partial softmax states [2 tiles_n][m] x {max, sum, sum p l, target} merged by lmhead.hip
  const int64_t* lm_labels;
  const float* lm_inv_temp;
  const float* lm_logz;    // MODE 9
  const float* lm_grad;    // MODE 9: d loss / d log-prob
  float* lm_partial;       // MODE 8
  int lm_round;            // logits rounded to bf16 before the softmax (the reference's bf16 logits chunk, lm_head.py:133)
};

// v_mfma_f32_16x16x32_bf16, not 32x32x16: the chip is POWER-limited on dense GEMMs and the 16x16x32 form moves less register data per MAC (its
// accumulator tile is half as large per unit of work): tools/probes/mfma_power_probe.hip sustains 1.83-1.92 PFLOP/s with it on random operands against
// 1.69-1.70 for 32x32x16 (zero-filled operands: 2.41-2.45 vs 2.48-2.49 — the bigger shape only wins where power is not the limit).
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// virtual block id -> (tile_m, tile_n); false for the padding ids of the last patch round.
// `sweep_n`: the order in which consecutive patches (the 8 XCDs of a round, then the next round) walk the patch grid.  A patch (i, j) reads X block i and
// W block j through its XCD's L2; across XCDs and rounds the blocks are shared through the 256 MB Infinity Cache.  Sweeping i fastest (0) holds W block j
// while the whole X operand cycles past it once per j: right when X fits the Infinity Cache (forward: X = 134 MB of activations).  Sweeping j fastest (1)
// holds X block i and cycles W: right when X is the big operand (the gate|up weight gradient: X = dy, 805 MB, otherwise re-read from HBM once per column
// block of patches: 4 x; the LM-head weight gradient: X = dlogits, 5 GB).  gemm_impl picks the order that cycles the SMALLER operand.
__device__ __forceinline__ bool tile_of_block(int b, int tiles_m, int tiles_n, int sweep_n, int& tm, int& tn) {
  const int xcd = b & 7, s = b >> 3;
  const int patch = (s >> 5) * 8 + xcd, within = s & 31;
  const int patches_m = (tiles_m + 7) >> 3;
  const int patches_n = (tiles_n + 3) >> 2;
  if (patch >= patches_m * patches_n) return false;
  const int pi = sweep_n ? patch / patches_n : patch % patches_m;
  const int pj = sweep_n ? patch % patches_n : patch / patches_m;
  tm = pi * 8 + (within & 7);
  tn = pj * 4 + (within >> 3);
  return tm < tiles_m && tn < tiles_n;
}
struct SplitPlan {
  int s, rp, k_part;
  int64_t ws_bytes;
};
// How the last round of tiles is split along K: rp = patches of the last round (of 8 XCD slots), s = parts per patch (2 or 4), k_part = contraction
// length of a part (a multiple of 128).  s = 1: no split (full last round, or more than half of it occupied, or a short contraction).
static inline SplitPlan plan_split(int tiles_m, int tiles_n, int64_t k) {
  const int patches = ((tiles_m + 7) / 8) * ((tiles_n + 3) / 4);
  const int rp = patches % 8;
  SplitPlan sp{1, rp, 0, 0};
  if (rp == 0 || rp > 4) return sp;
  int s = rp == 1 ? 4 : rp == 2 ? 4 : 2;
  if (k < 8192) return sp;                         // measured: below ~8k the per-unit prologue / epilogue and the reduction eat the gain (profiles/r03_gemm.md)
  while (s > 1 && k < (int64_t)s * 4096) s >>= 1;
  if (s == 1) return sp;
  const int64_t k_part = (k + s - 1) / s;
  sp.k_part = (int)((k_part + 127) / 128 * 128);
  sp.s = (int)((k + sp.k_part - 1) / sp.k_part);
  if (sp.s < 2) { sp.s = 1; return sp; }
  sp.ws_bytes = (int64_t)rp * 32 * (sp.s - 1) * kTile * kTile * 4;
  return sp;
}
static inline int grid_blocks(int tiles_m, int tiles_n) {
  const int patches = ((tiles_m + 7) / 8) * ((tiles_n + 3) / 4);
  return ((patches + 7) / 8) * 8 * 32;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Everything a wave carries through the K loop.  After inlining the struct is scalar-replaced (all indices are compile-time).
//
// Half-tile slots.  W^i = tile rows [128 i, 128 i + 128) of the N-side operand, X^j = rows [128 j, 128 j + 128) of the M-side operand; wave
// (wv, wt) owns W rows 128 i + 64 wv + [0, 64) and X rows 128 j + 32 wt + [0, 32) of every half, i.e. with lr = lane & 15, lq = lane >> 4 the output elements
//   OUT[m0 + 128 j + 32 wt + 16 g + lr][n0 + 128 i + 64 wv + 16 f + 4 lq + r]  =  acc[i][f][j][g][r]        (f = 0..3, g = 0..1, r = 0..3).
// LDS image of a slot:
//   K-contiguous operand: [128 rows][64 k] bf16, 128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7); a fragment (16 rows x 32 k) is
//     ONE ds_read_b128 per lane: row lr, chunk 4 kk + lq — conflict-free against the real lane groups of ds_read_b128 ({0-3,12-15,20-27} reads rows
//     0-3, 12-15 at chunk c and rows 4-11 at chunk c + 1: the eight XOR values {0,1,6,7} and {2,3,4,5}^1 are distinct).
//   K-major operand: [64 k][128 rows] bf16, 256-byte LDS rows, chunk c (8 operand rows) of k-row q at chunk c ^ ((q & 3) << 2 | ((q >> 3) & 1) << 1); a
//     fragment is TWO ds_read_b64_tr_b16 per lane (k-rows 32 kk + 8 lq + {0..3}, then {4..7}): the 16 lanes of a transpose group read a 4 (k) x 16
//     (rows) block, and the 8 k-rows x 32 bytes a half-wave touches land on 16 different 16-byte slots of the bank row.
//
// The LOAD SEGMENT of a phase is on the critical path of the ping-pong (it runs beside the partner's 256-cycle MFMA segment and every instruction
// of it costs ~5 cycles of issue; measured: two extra scalar branches per phase = -4 %), so it is kept to the minimum: fragment reads with
// precomputed per-lane addresses + immediate offsets (inline asm, so that hipcc can neither re-materialise the addresses with VALU adds nor guard
// the reads with `vmcnt(0)` against the LDS-DMA in flight), one counted wait, ONE m0 write for the two DMA pieces (the second piece uses the
// instruction offset, which moves the LDS and the memory address together, on a source offset pre-decremented by the same 1 KiB), one scalar add.
struct Wave {
  f32x4 acc[2][4][2][2];  // [i: W half][f: 16-row W fragment][j: X half][g: 16-row X fragment]
  u32x4 wf[2][4][2];      // W fragments [i][f][kk]     (K-contiguous operand; kk = 32-deep k-step)
  u32x4 xf[2][2][2];      // X fragments [j][g][kk]
  u32x2 wlo[2][4][2], whi[2][4][2];  // K-major operand: the two transpose-read halves of a fragment (k 0..3 | 4..7 of the lane's 8)
  u32x2 xlo[2][2][2], xhi[2][2][2];
  __amdgpu_buffer_rsrc_t wr, xr;     // buffer descriptors of the operand panels being STAGED: the current output tile's, and from the third phase of its
                                     // last K-tile pair on the next tile's (persistent tile stream)
  int vw[2], vx[2];     // per-lane byte offsets of this wave's two DMA pieces (half-tile 0); the second one minus the 1 KiB instruction offset
  int off[4];           // wave-uniform running source offsets per slot kind (K-tile advance + half-tile step), see stage()
  int wk, xk;           // byte distance between consecutive K-tiles
  int wstep, xstep;     // byte distance between the two halves of an operand tile (128 rows, or 128 columns of a K-major operand)
  int piece_lds;        // wave-uniform: byte offset of this wave's first piece inside a slot
  unsigned aw[4], ax[4];  // per-lane LDS byte addresses of the fragment reads (slot 0 of the operand's window): K-contiguous [kk] (fragment = immediate);
                          // K-major [f] (W) / [g] (X) (k-step = immediate; a fragment is an XOR on the address because the swizzle flips its chunk bits)
};

__device__ __forceinline__ unsigned char* lds_base() {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  return smem;
}

// stage one half-tile (this wave's 2 KiB of it) of slot kind SLOT into buffer BUF from the source offset `s.off[SLOT]`, then advance that offset by
// one K-tile.
template <int SLOT, int BUF, int PIECES = 3>  // PIECES: bit 0 = the first 1 KiB piece, bit 1 = the second one (+ the K-tile advance of the source offset)
__device__ __forceinline__ void stage(Wave& s, unsigned char* smem) {
  constexpr int slot_off = slot_offset(BUF, SLOT);
  constexpr bool is_w = SLOT == SLOT_W0 || SLOT == SLOT_W1;
  const __amdgpu_buffer_rsrc_t rsrc = is_w ? s.wr : s.xr;
  unsigned char* dst = smem + slot_off + s.piece_lds;
  if constexpr ((PIECES & 1) != 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)dst, 16, is_w ? s.vw[0] : s.vx[0], s.off[SLOT], 0, 0);
  if constexpr ((PIECES & 2) != 0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)dst, 16, is_w ? s.vw[1] : s.vx[1], s.off[SLOT], 1024, 0);
    s.off[SLOT] += is_w ? s.wk : s.xk;
  }
}
// source offsets of K-tile 0 (slot kinds of buffer 0: W0 X0 X1 W1 of K-tile 0; the loop then alternates buffers, so every kind advances one K-tile
// per use)
__device__ __forceinline__ void reset_offsets(Wave& s) {
  s.off[SLOT_W0] = 0;
  s.off[SLOT_W1] = s.wstep;
  s.off[SLOT_X0] = 0;
  s.off[SLOT_X1] = s.xstep;
}

// Fragment reads are INLINE ASM: (a) hipcc guards `__builtin_amdgcn_ds_read_tr16_b64` against every LDS-DMA in flight with `s_waitcnt vmcnt(0)` (it
// cannot tell the ring slots apart), which would drain the whole prefetch in every phase; (b) it re-materialises plain-load addresses with one VALU
// add per read once registers are tight.  The compiler does not track asm results on lgkmcnt, so the consumer side is explicit: `s_waitcnt
// lgkmcnt(0)` after the barrier, then `tie()` — an empty asm that takes the registers in and out — so that nothing that uses them can be scheduled
// above the wait.
template <int IMM>
__device__ __forceinline__ u32x2 tr_read(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
template <int IMM>
__device__ __forceinline__ u32x4 b128_read(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
__device__ __forceinline__ void tie(u32x2& a, u32x2& b, u32x2& c, u32x2& d, u32x2& e, u32x2& f, u32x2& g, u32x2& h) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ void tie(u32x4& a, u32x4& b, u32x4& c, u32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ u32x4 join(const u32x2& lo, const u32x2& hi) { return u32x4{lo.x, lo.y, hi.x, hi.y}; }

template <int I, int BUF, bool KM>
__device__ __forceinline__ void read_w(Wave& s) {
  constexpr int slot = slot_in_window(BUF, I ? SLOT_W1 : SLOT_W0);  // an immediate
  if constexpr (!KM) {
#define PM_RDW(F, KK) s.wf[I][F][KK] = b128_read<slot + F * 16 * 128>(s.aw[KK]);
    PM_RDW(0, 0) PM_RDW(1, 0) PM_RDW(2, 0) PM_RDW(3, 0) PM_RDW(0, 1) PM_RDW(1, 1) PM_RDW(2, 1) PM_RDW(3, 1)
#undef PM_RDW
  } else {
    // k-rows 32 kk + 8 lq + {0..3} then {4..7}: 8 KiB per kk, 1 KiB between the two reads
#define PM_TRW(F, KK)                                               \
    s.wlo[I][F][KK] = tr_read<slot + KK * 8192>(s.aw[F]);            \
    s.whi[I][F][KK] = tr_read<slot + KK * 8192 + 1024>(s.aw[F]);
    PM_TRW(0, 0) PM_TRW(1, 0) PM_TRW(2, 0) PM_TRW(3, 0) PM_TRW(0, 1) PM_TRW(1, 1) PM_TRW(2, 1) PM_TRW(3, 1)
#undef PM_TRW
  }
}
// a quarter (PART 0..3) of read_w: fragments PART / 2 * 2 .. + 1 of k-step... in issue order of read_w
template <int I, int BUF, bool KM, int PART>
__device__ __forceinline__ void read_w_part(Wave& s) {
  constexpr int slot = slot_in_window(BUF, I ? SLOT_W1 : SLOT_W0);
  constexpr int KK = PART >> 1, F0 = (PART & 1) * 2;
  if constexpr (!KM) {
    s.wf[I][F0][KK] = b128_read<slot + F0 * 16 * 128>(s.aw[KK]);
    s.wf[I][F0 + 1][KK] = b128_read<slot + (F0 + 1) * 16 * 128>(s.aw[KK]);
  } else {
    s.wlo[I][F0][KK] = tr_read<slot + KK * 8192>(s.aw[F0]);
    s.whi[I][F0][KK] = tr_read<slot + KK * 8192 + 1024>(s.aw[F0]);
    s.wlo[I][F0 + 1][KK] = tr_read<slot + KK * 8192>(s.aw[F0 + 1]);
    s.whi[I][F0 + 1][KK] = tr_read<slot + KK * 8192 + 1024>(s.aw[F0 + 1]);
  }
}
template <int J, int BUF, bool KM>
__device__ __forceinline__ void read_x(Wave& s) {
  constexpr int slot = slot_in_window(BUF, J ? SLOT_X1 : SLOT_X0);
  if constexpr (!KM) {
#define PM_RDX(G, KK) s.xf[J][G][KK] = b128_read<slot + G * 16 * 128>(s.ax[KK]);
    PM_RDX(0, 0) PM_RDX(1, 0) PM_RDX(0, 1) PM_RDX(1, 1)
#undef PM_RDX
  } else {
#define PM_TRX(G, KK)                                            \
    s.xlo[J][G][KK] = tr_read<slot + KK * 8192>(s.ax[G]);          \
    s.xhi[J][G][KK] = tr_read<slot + KK * 8192 + 1024>(s.ax[G]);
    PM_TRX(0, 0) PM_TRX(1, 0) PM_TRX(0, 1) PM_TRX(1, 1)
#undef PM_TRX
  }
}
template <int I, bool KM>
__device__ __forceinline__ void tie_w(Wave& s) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    if constexpr (KM) tie(s.wlo[I][0][kk], s.whi[I][0][kk], s.wlo[I][1][kk], s.whi[I][1][kk], s.wlo[I][2][kk], s.whi[I][2][kk], s.wlo[I][3][kk], s.whi[I][3][kk]);
    else tie(s.wf[I][0][kk], s.wf[I][1][kk], s.wf[I][2][kk], s.wf[I][3][kk]);
  }
}
template <int J, bool KM>
__device__ __forceinline__ void tie_x(Wave& s) {
  if constexpr (KM) tie(s.xlo[J][0][0], s.xhi[J][0][0], s.xlo[J][1][0], s.xhi[J][1][0], s.xlo[J][0][1], s.xhi[J][0][1], s.xlo[J][1][1], s.xhi[J][1][1]);
  else tie(s.xf[J][0][0], s.xf[J][1][0], s.xf[J][0][1], s.xf[J][1][1]);
}
template <int I, int J, bool XK, bool WK>
__device__ __forceinline__ void quadrant(Wave& s) {  // 64 W rows x 32 X rows x 64 k: 16 MFMAs on 8 independent accumulators per k-step
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const u32x4 wfrag = WK ? join(s.wlo[I][f][kk], s.whi[I][f][kk]) : s.wf[I][f][kk];
        const u32x4 xfrag = XK ? join(s.xlo[J][g][kk], s.xhi[J][g][kk]) : s.xf[J][g][kk];
        s.acc[I][f][J][g] = mfma16(wfrag, xfrag, s.acc[I][f][J][g]);
      }
}

// One phase.  P = phase inside the K-tile (0..3), BUF = parity of the K-tile.  Phases 1-2 stage the OTHER buffer's X1 / W1 half-tiles (K-tile + 1),
// phases 3-4 THIS buffer's W0 / X0 (K-tile + 2) — unconditionally, from whatever panels `s.wr / s.xr / s.off[]` currently describe: the stream of
// K-tiles runs across output-tile boundaries (the last K-tile pair of a tile stages the first six half-tiles of the next one, its "prologue"; after
// the workgroup's last tile it re-stages that tile's first K-tiles, which nobody reads), so that the phase body has no branch and ONE copy.
// V bit 0: s_setprio(1) around the MFMA cluster.  V bit 1: balanced reads (W0 of the NEXT K-tile is read in phase 4 instead of phase 1).
// V bits 3 / 4: TIMING ABLATIONS (wrong results): no LDS-DMA inside the loop / no fragment reads.
template <int P, int BUF, int V, bool XK, bool WK>
__device__ __forceinline__ void phase(Wave& s, unsigned char* smem) {
  constexpr bool kBalanced = (V & 2) != 0, kNoDma = (V & 8) != 0, kNoReads = (V & 16) != 0;
  // V bits 5 / 6 (round 5): WHERE the two LDS-DMA pieces of the phase sit in the load segment.  The four loading waves of a group run in lockstep, so with
  // both pieces after the reads (the default) eight 1 KiB requests reach the CU's one texture addresser at the same instant at the END of the segment and
  // every wave sits in its second `buffer_load` until the queue has room.  32: piece 0 BEFORE the fragment reads, piece 1 after them (two bursts of four,
  // the first one accepted while the waves issue their LDS reads).  64: both pieces before the reads.  The counted wait moves with them: `vmcnt` counts the
  // pieces issued in this phase too.
  constexpr int kDmaEarly = (V & 32) ? 1 : (V & 64) ? 2 : 0;
#define PM_STAGE_PHASE(PIECES)                                        \
  if (!kNoDma) {                                                      \
    if (P == 0) stage<SLOT_X1, BUF ^ 1, PIECES>(s, smem);             \
    else if (P == 1) stage<SLOT_W1, BUF ^ 1, PIECES>(s, smem);        \
    else if (P == 2) stage<SLOT_W0, BUF, PIECES>(s, smem);            \
    else stage<SLOT_X0, BUF, PIECES>(s, smem);                        \
  }
  if (kDmaEarly == 1) { PM_STAGE_PHASE(1) }
  if (kDmaEarly == 2) { PM_STAGE_PHASE(3) }
  // ---- load segment
  if (!kNoReads) {
    if (P == 0) {
      if (!kBalanced) read_w<0, BUF, WK>(s);
      read_x<0, BUF, XK>(s);
    } else if (P == 1) {
      read_x<1, BUF, XK>(s);
    } else if (P == 2) {
      read_w<1, BUF, WK>(s);
    } else if (kBalanced) {
      read_w<0, BUF ^ 1, WK>(s);
    }
  }
  if (kDmaEarly == 0) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    PM_STAGE_PHASE(3)
  } else if (kDmaEarly == 1) {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    PM_STAGE_PHASE(2)
  } else {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
#undef PM_STAGE_PHASE
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // the asm reads of this phase: their registers become usable only here
  if (!kNoReads) {
    if (P == 0) {
      tie_w<0, WK>(s);  // also in balanced mode: the prologue read of the first K-tile
      tie_x<0, XK>(s);
    } else if (P == 1) {
      tie_x<1, XK>(s);
    } else if (P == 2) {
      tie_w<1, WK>(s);
    } else if (kBalanced) {
      tie_w<0, WK>(s);
    }
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

// Round 5: the TWO-PHASE K-tile (V bit 8).  Same ping-pong, same ring, same fragments — but a compute segment is TWO quadrants (32 MFMAs = 512 matrix-pipe
// cycles) instead of one, so a K-tile has 2 role swaps per wave group instead of 4 (each swap idles the matrix pipe for a barrier latency: the no-DMA
// ablation of the four-phase loop runs at 89 % of the MFMA rate) and a load segment has 512 cycles to get its reads and its DMA pieces accepted.
// Plain form (V bit 1 clear): W0 and W1 are never live together, 64 fragment registers instead of 96:
//     A(t):  stage W1(t+1) -> b^1 | read W0(t), X0(t), X1(t) <- b                        | vmcnt(8) lgkmcnt(0) BARRIER | (W0,X0)(W0,X1) 32 MFMAs | BARRIER
//     B(t):  stage W0(t+2) -> b   | read W1(t) <- b | stage X0(t+2), X1(t+2) -> b        | vmcnt(8) lgkmcnt(0) BARRIER | (W1,X1)(W1,X0) 32 MFMAs | BARRIER
// Pre-read form (V bit 1): W0(t+1) is read in B(t), the stages are two half-tiles per segment:
//     A(t):  stage W1(t+1) -> b^1 | read X0(t), X1(t) <- b | stage W0(t+2) -> b          | ...
//     B(t):  stage X0(t+2) -> b   | read W1(t) <- b, W0(t+1) <- b^1 | stage X1(t+2) -> b | ...
// Hazards (b = t & 1; group 1 runs one barrier behind group 0):
//   RAW  a half-tile is read in a load segment AFTER the one whose counted wait + barrier retired it: every half-tile is staged three segments before its
//        read (W1(t+1): A(t) -> B(t+1); W0 / X0 / X1(t+2): B(t) -> A(t+2)); both forms issue 8 pieces per wave per two segments, and `vmcnt(8)` after a
//        segment's own pieces retires everything older than the two newest segments — exactly what the NEXT segment reads.
//   WAR  a slot is re-staged in the segment after the one that read it: legal because the reads are RETIRED (`lgkmcnt(0)`) before the reading segment's
//        barrier and the other group's DMA is issued after that barrier (guide, 8-phase template: "1 phase after when an lgkmcnt before the reading
//        phase's first barrier retired those reads").
// The unit switch of the persistent stream sits right after A's first stage of the last K-tile pair: W1(t+1) still belongs to this unit, everything
// staged after it to the next one.
template <int S, int BUF, int V, bool XK, bool WK>
__device__ __forceinline__ void superphase(Wave& s, unsigned char* smem) {
  constexpr bool kPre = (V & 2) != 0, kNoDma = (V & 8) != 0, kNoReads = (V & 16) != 0, kSplit = (V & 1024) != 0;
  if (S == 0) {
    if (!kNoReads) {
      if (!kPre) read_w<0, BUF, WK>(s);
      read_x<0, BUF, XK>(s);
      read_x<1, BUF, XK>(s);
    }
    if (kPre && !kNoDma) stage<SLOT_W0, BUF>(s, smem);
  } else {
    if (kPre) {
      if (!kNoDma) stage<SLOT_X0, BUF>(s, smem);
      if (!kNoReads) {
        read_w<1, BUF, WK>(s);
        read_w<0, BUF ^ 1, WK>(s);
      }
      if (!kNoDma) stage<SLOT_X1, BUF>(s, smem);
    } else if (kSplit) {  // one piece, two fragment reads, one piece, ...: six bursts of four requests instead of one of eight and one of sixteen
      if (!kNoDma) stage<SLOT_W0, BUF, 1>(s, smem);
      if (!kNoReads) read_w_part<1, BUF, WK, 0>(s);
      if (!kNoDma) stage<SLOT_W0, BUF, 2>(s, smem);
      if (!kNoReads) read_w_part<1, BUF, WK, 1>(s);
      if (!kNoDma) stage<SLOT_X0, BUF, 1>(s, smem);
      if (!kNoReads) read_w_part<1, BUF, WK, 2>(s);
      if (!kNoDma) stage<SLOT_X0, BUF, 2>(s, smem);
      if (!kNoReads) read_w_part<1, BUF, WK, 3>(s);
      if (!kNoDma) stage<SLOT_X1, BUF>(s, smem);
    } else {
      if (!kNoDma) stage<SLOT_W0, BUF>(s, smem);
      if (!kNoReads) read_w<1, BUF, WK>(s);
      if (!kNoDma) {
        stage<SLOT_X0, BUF>(s, smem);
        stage<SLOT_X1, BUF>(s, smem);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (S == 0) {
    tie_w<0, WK>(s);  // pre-read form: read in the previous B segment (or the prologue)
    tie_x<0, XK>(s);
    tie_x<1, XK>(s);
  } else {
    tie_w<1, WK>(s);
    if (kPre) tie_w<0, WK>(s);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (V & 1) __builtin_amdgcn_s_setprio(1);
  if (S == 0) {
    quadrant<0, 0, XK, WK>(s);
    quadrant<0, 1, XK, WK>(s);
  } else {
    quadrant<1, 1, XK, WK>(s);
    quadrant<1, 0, XK, WK>(s);
  }
  if (V & 1) __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Tile sources.  A workgroup walks virtual ids v = blockIdx.x, + gridDim.x, ...; a source turns (v, part) into one unit of the K-tile stream:
// operand panels, contraction length, output block.  A unit is a whole output tile, or one PART of an accumulation chain (`first` / `last`: zero the
// accumulators before / run the epilogue after) — how the grouped row gradient contracts over two weight sets without a descriptor switch inside
// the K loop.  Everything in a Tile is wave-uniform.
struct Tile {
  const bf16_t *xb, *wb;   // panel bases: first row (K-contiguous) or first column (K-major) of the tile's panel
  int64_t x_rows, w_rows;  // rows / columns of the operand tensor left from the base (range checks: what lies beyond reads as zero / is never stored)
  int k;                   // contraction length (rows past it read as zeros for K-major operands)
  unsigned char* out;      // output block origin (element (0, 0) of the tile)
  int64_t ldo;
  int m_valid, n_valid;    // rows / columns of the block that exist
  int nk;                  // K-tiles: even, >= 2
  bool first, last;
  int slot;                // p.sumsq: the tile's 8 floats start at slot * 8 (MODE 0: the virtual block id; MODE 3: expert * tiles per expert + tile — both stable across launches)
};
template <bool KM, bool CLAMP = true>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t panel_rsrc(const bf16_t* base, int64_t ld, int64_t rows_left, int k) {
  // K-contiguous [rows, K]: rows past the tensor's (or the group's) end are out of range -> zeros.  K-major [K, rows]: the descriptor ends with the
  // last k-row, so k-rows past K read as zeros.
  // Every input is wave-uniform, but the compiler cannot always prove it (the grouped sources go through cross-lane scans): a descriptor it believes
  // divergent lives in VGPRs and EVERY DMA becomes a waterfall loop.  readfirstlane pins the three words to scalar registers.
  const uint64_t addr = reinterpret_cast<uint64_t>(base);
  const uint64_t addr_u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)addr);
  int bytes;
  if (!KM) {
    const int64_t rows = (rows_left < kTile || !CLAMP) ? rows_left : kTile;
    bytes = rows > 0 ? (int)(((rows - 1) * ld + k) * 2) : 0;
  } else {
    bytes = k > 0 ? (int)((((int64_t)k - 1) * ld + rows_left) * 2) : 0;
  }
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(addr_u), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// descriptor of an output block for the epilogue's buffer stores: every word pinned to scalar registers (a descriptor hipcc believes divergent turns each store
// into a waterfall loop, guide T20)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(const void* base, int bytes) {
  const uint64_t addr = reinterpret_cast<uint64_t>(base);
  const uint64_t addr_u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)addr);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(addr_u), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ int k_tiles(int k) {
  const int nk = ((k + 2 * kBK - 1) / (2 * kBK)) * 2;  // even; K-tiles past K read zeros (K-major operands) or do not exist (K % 128 == 0)
  return nk < 2 ? 2 : nk;
}

// ROWS-row tile `t` of the expert-sorted rows -> (expert, first row, end of the expert's rows); false past the last tile.  Every wave runs the same
// prefix scan over ceil(rows_e / 256) (a row tile never straddles two experts).
__device__ __forceinline__ bool locate_row_tile(const int32_t* __restrict__ offsets, int n_experts, int t, int lane, int& expert, int& row0, int& row_end) {
  int base = 0;
  for (int e0 = 0; e0 < n_experts; e0 += 64) {
    const int e = e0 + lane;
    int lo = 0, hi = 0;
    if (e < n_experts) {
      lo = offsets[e];
      hi = offsets[e + 1];
    }
    const int tiles = (hi - lo + kTile - 1) / kTile;
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
      expert = __builtin_amdgcn_readfirstlane(e0 + src);
      row0 = __builtin_amdgcn_readfirstlane(__shfl(lo, src, 64) + (t - first_tile) * kTile);
      row_end = __builtin_amdgcn_readfirstlane(__shfl(hi, src, 64));
      return true;
    }
    base += __shfl(incl, 63, 64);
  }
  return false;
}

// The same lookup from REGISTERS.  A unit switch happens inside the K loop, where the wave has LDS-DMA loads in flight: a lookup that loads `offsets`
// from memory waits on vmcnt(0) — i.e. for the whole staging queue — and does so twice per unit (as the next unit when the staging switches to it, as the
// current one at its own turn); with the short contractions of expert products (12-32 K-tiles per unit) that was 1-2 us out of every 10-27.  So every wave
// reads the offsets ONCE, when the workgroup starts: lane l of set s keeps expert 64 s + l's rows and the inclusive tile count up to it, and a lookup
// is a ballot and three v_readlane.  Up to 64 * kTabSets experts (12 registers); more fall back to the memory form.
constexpr int kTabSets = 4;
struct RowTab {
  int lo[kTabSets], hi[kTabSets], incl[kTabSets];
  bool use;
  int g0, g1;  // MODE 3: this workgroup's XCD owns tiles [g0, g1) of the tile line (expert-major, tiles_m * tiles_n per expert): build_wgrad_range
};
__device__ __forceinline__ void build_row_tab(const int32_t* __restrict__ offsets, int n_experts, int lane, RowTab& rt) {
  rt.use = n_experts <= 64 * kTabSets;
  int base = 0;
#pragma unroll
  for (int s = 0; s < kTabSets; ++s) {
    const int e = s * 64 + lane;
    int lo = 0, hi = 0;
    if (rt.use && e < n_experts) {
      lo = offsets[e];
      hi = offsets[e + 1];
    }
    int incl = (hi - lo + kTile - 1) / kTile;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    rt.lo[s] = lo;
    rt.hi[s] = hi;
    rt.incl[s] = base + incl;
    base += __builtin_amdgcn_readlane(incl, 63);
  }
}
// MODE 3 (grouped weight gradient): the contraction length of a tile is its expert's ROW COUNT, so tiles differ in cost by the routing's imbalance (a hot
// expert of a fresh model: 10x the mean).  "Expert e on XCD e % 8" then makes one XCD finish long after the others.  Instead the tile line (expert-major,
// `per` tiles per expert, cost of a tile = its K-tile count) is cut into 8 contiguous pieces of equal COST, one per XCD: an expert's tiles stay together
// (its two row panels stay in one L2) except where a cut falls, and with balanced routing the pieces are the 16-expert blocks they always were.  g(B) = the
// first tile that starts at or after cost position B; computed once per wave when the workgroup starts (one scan over the offsets, nothing kept but g0, g1).
template <int S>
__device__ __forceinline__ void first_tile_at(const int (&incl)[kTabSets], const int (&nkv)[kTabSets], int per, long long pos, int n_experts, int& g) {
  if constexpr (S >= kTabSets) {
    g = per * n_experts;
  } else {
    const unsigned long long hit = __ballot((long long)per * incl[S] > pos);
    if (!hit) {
      first_tile_at<S + 1>(incl, nkv, per, pos, n_experts, g);
    } else {
      const int src = __ffsll((long long)hit) - 1;
      const int nk = __builtin_amdgcn_readlane(nkv[S], src);
      const long long start = (long long)per * (__builtin_amdgcn_readlane(incl[S], src) - nk);  // cost position of the expert's first tile
      g = per * (S * 64 + src) + (int)((pos - start + nk - 1) / nk);
    }
  }
}
__device__ __forceinline__ void build_wgrad_range(const int32_t* __restrict__ offsets, int n_experts, int per, int lane, int xcd, RowTab& rt) {
  rt.use = n_experts <= 64 * kTabSets;
  rt.g0 = rt.g1 = 0;
  if (!rt.use) return;
  int incl[kTabSets], nkv[kTabSets];
  int base = 0;
#pragma unroll
  for (int s = 0; s < kTabSets; ++s) {
    const int e = s * 64 + lane;
    int nk = 0;
    if (e < n_experts) nk = k_tiles(offsets[e + 1] - offsets[e]);
    int sum = nk;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(sum, o, 64);
      if (lane >= o) sum += up;
    }
    nkv[s] = nk;
    incl[s] = base + sum;
    base += __builtin_amdgcn_readlane(sum, 63);
  }
  const long long total = (long long)per * base;
  first_tile_at<0>(incl, nkv, per, total * xcd / 8, n_experts, rt.g0);
  first_tile_at<0>(incl, nkv, per, total * (xcd + 1) / 8, n_experts, rt.g1);
  if (xcd == 7) rt.g1 = per * n_experts;
}
template <int S>
__device__ __forceinline__ bool locate_in_set(const RowTab& rt, int t, int& expert, int& row0, int& row_end) {
  if constexpr (S >= kTabSets) {
    return false;
  } else {
    const unsigned long long hit = __ballot(rt.incl[S] > t);
    if (!hit) return locate_in_set<S + 1>(rt, t, expert, row0, row_end);
    const int src = __ffsll((long long)hit) - 1;
    const int lo = __builtin_amdgcn_readlane(rt.lo[S], src), hi = __builtin_amdgcn_readlane(rt.hi[S], src);
    const int first_tile = __builtin_amdgcn_readlane(rt.incl[S], src) - (hi - lo + kTile - 1) / kTile;
    expert = S * 64 + src;
    row0 = lo + (t - first_tile) * kTile;
    row_end = hi;
    return true;
  }
}
__device__ __forceinline__ bool locate_row_tile(const RowTab& rt, int t, int& expert, int& row0, int& row_end) { return locate_in_set<0>(rt, t, expert, row0, row_end); }

// MODE 0: the dense product.  1: grouped forward  y[m] = x[m] . [w | w2][e(m)]^T      (x, w K-contiguous; column tiles >= n_first come from w2)
// 2: grouped row gradient  dx[m] = dy[m, :n_first] . w[e] + dy[m, n_first:] . w2[e]   (dy K-contiguous, weights K-major; two chained parts)
// 3: grouped weight gradient  dw[e] = dy[rows of e]^T . x[rows of e]                  (both K-major; K = the expert's row count; rows < n_first -> out, else out2)
// 5 / 6: the tile streams of 0 / 2 (a dense / grouped row gradient through a down projection) with the SwiGLU-backward epilogue: the product is the
//        gradient of act = silu(gate) * up, and the epilogue turns each value into d_gate and d_up at once (OUT = [d_gate | d_up], gate / up read from p.gu)
// 8 / 9: the tile stream of 0 on (hidden, head weight) with the log-prob epilogues of the fused LM head (lm_head.py:89-202): 8 = forward, the tile's logits
//        leave as per-token partial softmax states and never reach HBM; 9 = backward, the recomputed tile leaves as dlogits = g inv_T (onehot - p)
template <int MODE>
__device__ __forceinline__ constexpr int parts_of() { return MODE == 2 ? 2 : 1; }
template <int MODE>
__device__ __forceinline__ constexpr int tile_mode() { return (MODE == 5 || MODE == 8 || MODE == 9) ? 0 : MODE == 6 ? 2 : MODE; }

template <int MODE, bool XK, bool WK>
__device__ __forceinline__ bool tile_at(const Params& p, int v, int part, int lane, const RowTab& rtab, Tile& t) {
  t.first = t.last = true;
  t.slot = v;
  if (MODE == 0) {
    int tm, tn, kpart = 0, vv = v;
    if (p.split_s > 1 && v >= p.split_v0) {  // last round, split: XCD slot x computes K-part x / rp of tail patch x % rp
      const int x = v & 7;
      if (x >= p.split_rp * p.split_s) return false;
      kpart = x / p.split_rp;
      vv = (v & ~7) | (x - kpart * p.split_rp);
    }
    if (!tile_of_block(vv, p.tiles_m, p.tiles_n, p.sweep_n, tm, tn)) return false;
    const int64_t m0 = (int64_t)tm * kTile, n0 = (int64_t)tn * kTile;
    int64_t k0 = (int64_t)kpart * p.split_k;
    int klen = p.split_s > 1 && v >= p.split_v0 ? (int)(p.k - k0 < p.split_k ? p.k - k0 : p.split_k) : p.k;
    if (p.k_parts > 1) {  // chained parts of one accumulation (never together with the split: gemm_impl)
      k0 = (int64_t)part * p.k_chunk;
      klen = (int)(p.k - k0 < p.k_chunk ? p.k - k0 : p.k_chunk);
      t.first = part == 0;
      t.last = part == p.k_parts - 1;
    }
    t.xb = p.x + (XK ? m0 + k0 * p.ldx : m0 * p.ldx + k0);
    t.wb = p.w + (WK ? n0 + k0 * p.ldw : n0 * p.ldw + k0);
    t.x_rows = p.m - m0;
    t.w_rows = p.n - n0;
    t.k = klen;
    if (kpart == 0) {
      t.out = reinterpret_cast<unsigned char*>(p.out) + (m0 * p.ldo + n0) * (p.out_f32 ? 4 : 2);
      t.ldo = p.ldo;
    } else {  // a private f32 tile of the workspace: tail tile = (patch slot, position inside the patch)
      const int tail_tile = (vv & 7) * 32 + ((vv >> 3) & 31);
      t.out = reinterpret_cast<unsigned char*>(p.split_ws + ((int64_t)tail_tile * (p.split_s - 1) + (kpart - 1)) * (kTile * kTile));
      t.ldo = kTile;
    }
    t.m_valid = (int)(p.m - m0 < kTile ? p.m - m0 : kTile);
    t.n_valid = (int)(p.n - n0 < kTile ? p.n - n0 : kTile);
    t.nk = k_tiles(klen);
    return true;
  }
  if (MODE == 4) {
    // dense forward of a fused gate|up projection with the SwiGLU epilogue: OUT = [gate | up] = X . [Wg ; Wu]^T with Wg = rows [0, n_first) and Wu = rows
    // [n_first, 2 n_first) of W.  A tile is 128 FEATURES wide: its W^0 half-tile is 128 rows of Wg, its W^1 half-tile the SAME 128 rows of Wu (the half-tile
    // step of the DMA is n_first rows instead of 128), so a lane holds gate and up of one (token, feature) in acc[0][..] and acc[1][..] and the epilogue
    // writes act = silu(gate) * up next to the two halves of OUT (mlp.py:27-29: down_proj(act_fn(gate_proj(x)) * up_proj(x))).
    int tm, tn;
    if (!tile_of_block(v, p.tiles_m, p.tiles_n, p.sweep_n, tm, tn)) return false;
    const int64_t m0 = (int64_t)tm * kTile, f0 = (int64_t)tn * 128;
    t.xb = p.x + m0 * p.ldx;
    t.wb = p.w + f0 * p.ldw;
    t.x_rows = p.m - m0;
    t.w_rows = p.n_first + 128;  // rows from the panel base to the end of the up half (panel_rsrc<.., false>: not clamped to the tile)
    t.k = p.k;
    t.out = reinterpret_cast<unsigned char*>(p.out) + (m0 * p.ldo + f0) * 2;
    t.ldo = p.ldo;
    t.m_valid = (int)(p.m - m0 < kTile ? p.m - m0 : kTile);
    t.n_valid = p.n_first + 128;
    t.nk = k_tiles(p.k);
    return true;
  }
  // grouped: block v runs on XCD v % 8; the `ncol` column tiles of one row tile (MODE 1, 2) / the tiles of one expert (MODE 3) are consecutive ids of ONE
  // XCD, so the rows they share leave HBM once
  const int xcd = v & 7, q = v >> 3;
  if (MODE == 7) {
    // grouped forward of the experts' gate and up projections with the SwiGLU epilogue (MODE 4's tile on MODE 1's rows): a tile = one 256-row tile of the
    // expert-sorted rows x 128 FEATURES; W^0 = 128 rows of w[e] (gate), W^1 = the same rows of w2[e] (up) — two tensors of the same shape, so the half-tile
    // step of the DMA is the byte distance between them (the entry point requires 0 < w2 - w < 2 GiB)
    const int ncol = p.tiles_n;
    const int rt = (q / ncol) * 8 + xcd, ct = q % ncol;
    int e, row0, row_end;
    if (!(rtab.use ? locate_row_tile(rtab, rt, e, row0, row_end) : locate_row_tile(p.offsets, p.n_experts, rt, lane, e, row0, row_end))) return false;
    const int f0 = ct * 128;
    t.xb = p.x + (int64_t)row0 * p.ldx;
    t.x_rows = row_end - row0;
    t.wb = p.w + ((int64_t)e * p.n_first + f0) * p.ldw;
    t.w_rows = (int)((p.w2 - p.w) / p.ldw) + 128;  // rows from the panel base to the end of the up half-tile (panel_rsrc<.., false>)
    t.k = p.k;
    t.out = reinterpret_cast<unsigned char*>(p.out) + ((int64_t)row0 * p.ldo + f0) * 2;
    t.ldo = p.ldo;
    t.m_valid = row_end - row0 < kTile ? row_end - row0 : kTile;
    t.n_valid = p.n_first + 128;
    t.nk = k_tiles(p.k);
    return true;
  }
  if (MODE == 1 || MODE == 2) {
    const int ncol = p.tiles_n;
    const int rt = (q / ncol) * 8 + xcd, ct = q % ncol;
    int e, row0, row_end;
    if (!(rtab.use ? locate_row_tile(rtab, rt, e, row0, row_end) : locate_row_tile(p.offsets, p.n_experts, rt, lane, e, row0, row_end))) return false;
    const int c0 = ct * kTile;
    t.m_valid = row_end - row0 < kTile ? row_end - row0 : kTile;
    t.ldo = p.ldo;
    t.out = reinterpret_cast<unsigned char*>(p.out) + ((int64_t)row0 * p.ldo + c0) * 2;
    if (MODE == 1) {  // x [M, K], w [E, n_first, K], w2 [E, N - n_first, K]
      const bool second = c0 >= p.n_first;
      const int wn = second ? p.n - p.n_first : p.n_first, nw0 = second ? c0 - p.n_first : c0;
      t.xb = p.x + (int64_t)row0 * p.ldx;
      t.x_rows = row_end - row0;
      t.wb = (second ? p.w2 : p.w) + ((int64_t)e * wn + nw0) * p.ldw;
      t.w_rows = wn - nw0;
      t.k = p.k;
      t.n_valid = wn - nw0 < kTile ? wn - nw0 : kTile;
      t.nk = k_tiles(p.k);
    } else {          // dy [M, n_dim = K], w [E, n_first, k_out = N] read K-major, then w2 [E, K - n_first, N]
      const int na = p.n_first, nb = p.k - p.n_first;
      const bool second = part == 1;
      const int kk = second ? nb : na;
      t.xb = p.x + (int64_t)row0 * p.ldx + (second ? na : 0);
      t.x_rows = row_end - row0;
      t.wb = (second ? p.w2 + (int64_t)e * nb * p.ldw : p.w + (int64_t)e * na * p.ldw) + c0;
      t.w_rows = p.n - c0;
      t.k = kk;
      t.n_valid = p.n - c0 < kTile ? p.n - c0 : kTile;
      t.nk = k_tiles(kk);
      t.first = !second;
      t.last = second || nb == 0;
      if (second && nb == 0) return false;  // a single weight set: no second part
    }
    return true;
  }
  // MODE 3: per expert (n_dim / 256) x (k_dim / 256) tiles; X = dy [M, n_dim] K-major (M-side = n), W = x [M, k_dim] K-major (N-side = k)
  const int per = p.tiles_m * p.tiles_n;
  int e, tile;
  if (rtab.use) {  // this XCD's piece of the cost-balanced tile line (see build_wgrad_range)
    const int g = rtab.g0 + q;
    if (g >= rtab.g1) return false;
    e = __builtin_amdgcn_readfirstlane(g / per);
    tile = g % per;
  } else {  // more experts than the table holds: expert e on XCD e % 8
    e = __builtin_amdgcn_readfirstlane((q / per) * 8 + xcd);
    tile = q % per;
    if (e >= p.n_experts) return false;
  }
  // a SCALAR load (the address is uniform; hipcc emits a vector load because the kernel also stores): a vector load here makes the unit switch wait on
  // vmcnt(0), i.e. for every LDS-DMA piece in flight (see RowTab)
  int lo, hi;
  {
    typedef int int2_t __attribute__((ext_vector_type(2)));
    int2_t lh;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lh) : "s"(p.offsets + e) : "memory");
    lo = lh[0];
    hi = lh[1];
  }
  const int tmn = tile / p.tiles_n, tk = tile % p.tiles_n;
  const int n0 = tmn * kTile, k0 = tk * kTile;
  t.slot = e * per + tile;
  t.xb = p.x + (int64_t)lo * p.ldx + n0;
  t.x_rows = p.m - n0;
  t.wb = p.w + (int64_t)lo * p.ldw + k0;
  t.w_rows = p.n - k0;
  t.k = hi - lo;
  const bool second = n0 >= p.n_first;
  const int on = second ? p.m - p.n_first : p.n_first, on0 = second ? n0 - p.n_first : n0;
  t.out = reinterpret_cast<unsigned char*>(second ? p.out2 : p.out) + (((int64_t)e * on + on0) * p.ldo + k0) * (p.out_f32 ? 4 : 2);
  t.ldo = p.ldo;
  t.m_valid = on - on0 < kTile ? on - on0 : kTile;
  t.n_valid = p.n - k0 < kTile ? p.n - k0 : kTile;
  t.nk = k_tiles(hi - lo);
  return true;
}

// the next unit of this workgroup's stream at or after (v, part); false when the stream is exhausted
template <int MODE, bool XK, bool WK>
__device__ __forceinline__ bool next_unit(const Params& p, int& v, int& part, int lane, const RowTab& rtab, Tile& t) {
  for (; v < p.n_blocks;) {
    if (tile_at<MODE, XK, WK>(p, v, part, lane, rtab, t)) return true;
    if (MODE == 3 && rtab.use) break;  // the XCD's piece of the tile line is contiguous: the first miss is its end
    part = 0;
    v += gridDim.x;
  }
  v = p.n_blocks;
  return false;
}
__device__ __forceinline__ void advance_unit(int& v, int& part, int parts) {
  if (++part >= parts) {
    part = 0;
    v += gridDim.x;
  }
}

// per-lane DMA source offsets + fragment read addresses of one operand (tile independent).  `wsel`: this wave's first row inside a half (64 wv for
// W, 32 wt for X); `lds0`: LDS byte address of the operand's window of the ring.
template <bool KM>
__device__ __forceinline__ void setup_operand(int64_t ld, int wave, int lane, int wsel, unsigned lds0, int (&voff)[2], int& half_step, int& k_step, unsigned (&rd)[4]) {
  const int lr = lane & 15, lq = lane >> 4;
  if (!KM) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int rho = (wave * 2 + q) * 8 + (lane >> 3);  // slot row of this lane's 16 bytes (piece = 8 rows x 128 bytes)
      voff[q] = (int)(rho * ld * 2) + (((lane & 7) ^ ((rho >> 1) & 7)) << 4) - q * 1024;
    }
    half_step = (int)(128 * ld * 2);
    k_step = kBK * 2;
    // fragment f of the wave starts 16 rows further: (row >> 1) & 7 does not change, so f is an immediate; the k-step flips chunk bit 2: a second register
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) rd[kk] = lds0 + (wsel + lr) * 128 + (((kk * 4 + lq) ^ ((lr >> 1) & 7)) << 4);
    rd[2] = rd[3] = 0;
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int krow = (wave * 2 + q) * 4 + (lane >> 4);  // piece = 4 k-rows x 256 bytes
      voff[q] = (int)(krow * ld * 2) + (((lane & 15) ^ (((krow & 3) << 2) | (((krow >> 3) & 1) << 1))) << 4) - q * 1024;
    }
    half_step = 128 * 2;
    k_step = (int)(kBK * ld * 2);
    const int krow = lq * 8 + (lr >> 2);  // kk = 0; kk adds 32 k-rows = 8 KiB, the second transpose read 4 k-rows = 1 KiB (immediates; neither changes the swizzle)
    const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
#pragma unroll
    for (int f = 0; f < 4; ++f) {  // 16 operand rows per fragment = 2 chunks
      const int chunk = (wsel >> 3) + f * 2 + ((lr & 3) >> 1);
      rd[f] = lds0 + krow * 256 + ((chunk ^ swz) << 4) + (lr & 1) * 8;
    }
  }
}

// MODE 8 epilogue: the tile's logits leave as partial softmax states (lm_head.py:76-86, 140-148).  acc[i][f][j][g][r] is the logit of token row
// j * 128 + wt * 32 + g * 16 + lr and vocab column i * 128 + wv * 64 + f * 16 + lq * 4 + r of the tile, so a lane holds 32 logits of each of its 4 token
// rows: (max, sum exp, sum exp * l, target) per lane, merged over the 4 lanes (lq) that share a token row with two shuffles; one float4 per (token, vocab
// tile, wave group) goes to p.lm_partial[(2 tn + wv) * m + token], 16 tokens = 256 contiguous bytes per store.  Logits are rounded to bf16 first (what the
// stored-logits path writes and K1 reads), scaled by 1 / T in f32, columns past the vocabulary masked (last vocab tile only).
__device__ __forceinline__ void lm_fwd_epilogue(const Params& p, int tm, int tn, f32x4 (&acc)[2][4][2][2], int wv, int wt, int lr, int lq) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  constexpr float kLog2e = 1.4426950408889634f, kNegBig = -3.0e38f;
  const int col0 = tn * kTile + wv * 64 + lq * 4;  // + i * 128 + f * 16 + r
  const bool ragged = (tn + 1) * kTile > p.n;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int tok = tm * kTile + j * 128 + wt * 32 + g * 16 + lr;
      const bool live = tok < p.m;
      const float inv_t = live ? p.lm_inv_temp[tok] : 1.f;
      const int rel = live ? (int)p.lm_labels[tok] - col0 : -1;
      f32x2 x[2][4][2];
      float mx = kNegBig;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const f32x4 c = acc[i][f][j][g];
          const uint32_t u0 = pack_bf16_hw(c[0], c[1]), u1 = pack_bf16_hw(c[2], c[3]);
          f32x2 a = f32x2{bf16lo(u0), bf16hi(u0)} * inv_t, b = f32x2{bf16lo(u1), bf16hi(u1)} * inv_t;
          if (ragged) {
            const int cc = col0 + i * 128 + f * 16;
            a[0] = cc < p.n ? a[0] : kNegBig; a[1] = cc + 1 < p.n ? a[1] : kNegBig;
            b[0] = cc + 2 < p.n ? b[0] : kNegBig; b[1] = cc + 3 < p.n ? b[1] : kNegBig;
          }
          x[i][f][0] = a; x[i][f][1] = b;
          mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(b[0], b[1]));
        }
      float tg = kNegBig;
      if (rel >= 0 && rel < 256 && (rel & 0x4c) == 0) {  // the label is one of this lane's 32 columns (one lane in ~1000 per tile)
        const int sel = ((rel >> 7) << 4) | (((rel >> 4) & 3) << 2) | (rel & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) tg = sel == ((i << 4) | (f << 2) | r) ? x[i][f][r >> 1][r & 1] : tg;
      }
      f32x2 s2 = f32x2{0.f, 0.f}, t2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 xx = x[i][f][h], d = (xx - mx) * kLog2e;
            const f32x2 e = f32x2{__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};  // exp(-huge) = 0 for masked columns
            s2 += e;
            t2 += e * (ragged ? f32x2{xx[0] > kNegBig ? xx[0] : 0.f, xx[1] > kNegBig ? xx[1] : 0.f} : xx);
          }
      float m = mx, sm = s2[0] + s2[1], tt = t2[0] + t2[1];
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), sb = __shfl_xor(sm, o, 64), tb = __shfl_xor(tt, o, 64), g2 = __shfl_xor(tg, o, 64);
        const float mn = fmaxf(m, m2);
        const float e1 = __builtin_amdgcn_exp2f((m - mn) * kLog2e), e2 = __builtin_amdgcn_exp2f((m2 - mn) * kLog2e);
        sm = sm * e1 + sb * e2;
        tt = tt * e1 + tb * e2;
        m = mn;
        tg = fmaxf(tg, g2);
      }
      if (lq == 0 && live) *reinterpret_cast<f32x4*>(p.lm_partial + ((int64_t)(tn * 2 + wv) * p.m + tok) * 4) = f32x4{m, sm, tt, tg};
    }
}

template <int V, bool XK, bool WK, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const Params p) {
  constexpr int TM = tile_mode<MODE>();          // which tile stream
  constexpr bool SWB = MODE == 5 || MODE == 6;  // SwiGLU-backward epilogue
  unsigned char* smem = lds_base();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = wave >> 2, wt = wave & 3;  // wv: which 64 W rows of a half (also the ping-pong group), wt: which 32 X rows of a half
  int v = blockIdx.x, part = 0;
  Tile cur;
  RowTab rtab;
  rtab.use = false;
  if constexpr (TM == 1 || TM == 2 || TM == 7) build_row_tab(p.offsets, p.n_experts, lane, rtab);  // the expert offsets, once, into registers
  if constexpr (TM == 3) build_wgrad_range(p.offsets, p.n_experts, p.tiles_m * p.tiles_n, lane, blockIdx.x & 7, rtab);
  if (!next_unit<TM, XK, WK>(p, v, part, lane, rtab, cur)) return;

  Wave s;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  setup_operand<WK>(p.ldw, wave, lane, wv * 64, lds0, s.vw, s.wstep, s.wk, s.aw);
  if (MODE == 4) s.wstep = (int)(p.n_first * p.ldw * 2);  // W^1 = the up rows of the same features
  if (MODE == 7) s.wstep = (int)((p.w2 - p.w) * 2);  // W^1 = the same rows of the second weight tensor
  setup_operand<XK>(p.ldx, wave, lane, wt * 32, lds0 + 4 * kSlot, s.vx, s.xstep, s.xk, s.ax);  // the X window is the second 64 KiB of the ring
  s.wr = panel_rsrc<WK, MODE != 4 && MODE != 7>(cur.wb, p.ldw, cur.w_rows, cur.k);
  s.xr = panel_rsrc<XK>(cur.xb, p.ldx, cur.x_rows, cur.k);
  s.piece_lds = wave * 2048;
  if (V & 16) {  // timing ablation: fragments are never read; give them defined (non-zero) contents
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          s.wf[i][f][kk] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          s.xf[i][f & 1][kk] = u32x4{0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u};
          s.wlo[i][f][kk] = s.whi[i][f][kk] = s.xlo[i][f & 1][kk] = s.xhi[i][f & 1][kk] = u32x2{0x3f803f80u, 0x3f803f80u + lane};
        }
  }
  // ---- prologue of the FIRST unit: the issue order continues into the loop's (X1, W1, W0, X0) rotation
  reset_offsets(s);
  stage<SLOT_W0, 0>(s, smem);
  stage<SLOT_X0, 0>(s, smem);
  stage<SLOT_X1, 0>(s, smem);
  stage<SLOT_W1, 0>(s, smem);
  stage<SLOT_W0, 1>(s, smem);
  stage<SLOT_X0, 1>(s, smem);
  if constexpr ((V & 256) != 0) {  // two-phase K-tile: X1(1) as well (what B(-1) would have staged); W0(0), X0(0), X1(0) have landed behind vmcnt(8)
    stage<SLOT_X1, 1>(s, smem);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // W0(0) and X0(0) of this wave have landed
  __builtin_amdgcn_s_barrier();
  if ((V & 2) != 0 && !(V & 16)) read_w<0, 0, WK>(s);  // balanced reads: W0 of K-tile 0 is read here (tied in phase 1), later ones in phase 4 of the previous K-tile
  if (wv == 1) __builtin_amdgcn_s_barrier();           // the second wave group runs one barrier behind the first
  bool staggered = true;

  for (;;) {  // ---- the workgroup's stream of units (output tiles, or parts of an accumulation chain)
    {
      // Re-derive the per-lane DMA offsets / read addresses from an OPAQUE copy of the lane id at the top of every unit: otherwise they stay live across
      // the epilogue (the register-hungriest part), get spilled, and every DMA in the K loop waits for a scratch reload with `vmcnt(1)`.
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      setup_operand<WK>(p.ldw, wave, lane_o, wv * 64, lds0, s.vw, s.wstep, s.wk, s.aw);
      if (MODE == 4) s.wstep = (int)(p.n_first * p.ldw * 2);  // W^1 = the up rows of the same features
      if (MODE == 7) s.wstep = (int)((p.w2 - p.w) * 2);
      setup_operand<XK>(p.ldx, wave, lane_o, wt * 32, lds0 + 4 * kSlot, s.vx, s.xstep, s.xk, s.ax);
    }
    // Only what the epilogue needs of `cur` stays live across the K loop; the NEXT unit is looked up at the point where the staging switches to it,
    // its two descriptors are built and the rest is dropped again (it is looked up once more as `cur` at the top of its own turn): scalar registers
    // are the scarce resource of this kernel (spilled SGPRs land in VGPR lanes and push the vector allocation over 256).
    int v_n = v, part_n = part;
    bool has_next = false;
    if (cur.first) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) s.acc[i][f][j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nk = cur.nk;
    if constexpr ((V & 256) != 0) {
      for (int t = 0; t < nk; t += 2) {
        if (!(V & 8)) stage<SLOT_W1, 1>(s, smem);  // A(t), first stage: W1(t + 1)
        if (t + 2 >= nk) {
          advance_unit(v_n, part_n, TM == 0 ? p.k_parts : parts_of<TM>());
          Tile nxt;
          has_next = next_unit<TM, XK, WK>(p, v_n, part_n, lane, rtab, nxt);
          if (has_next) {
            s.wr = panel_rsrc<WK, MODE != 4 && MODE != 7>(nxt.wb, p.ldw, nxt.w_rows, nxt.k);
            s.xr = panel_rsrc<XK>(nxt.xb, p.ldx, nxt.x_rows, nxt.k);
          }
          reset_offsets(s);
        }
        superphase<0, 0, V, XK, WK>(s, smem);
        superphase<1, 0, V, XK, WK>(s, smem);
        if (!(V & 8)) stage<SLOT_W1, 0>(s, smem);  // A(t + 1), first stage: W1(t + 2)
        superphase<0, 1, V, XK, WK>(s, smem);
        superphase<1, 1, V, XK, WK>(s, smem);
      }
    } else
    for (int t = 0; t < nk; t += 2) {  // two K-tiles per iteration: t in buffer 0, t + 1 in buffer 1
      phase<0, 0, V, XK, WK>(s, smem);
      phase<1, 0, V, XK, WK>(s, smem);
      if (t + 2 >= nk) {  // the last pair of this unit: every half-tile staged from here on belongs to the next one
        advance_unit(v_n, part_n, TM == 0 ? p.k_parts : parts_of<TM>());
        Tile nxt;
        has_next = next_unit<TM, XK, WK>(p, v_n, part_n, lane, rtab, nxt);
        if (has_next) {  // (after the last unit the stream re-stages this unit's first K-tiles: never read, drained before the kernel ends)
          s.wr = panel_rsrc<WK, MODE != 4 && MODE != 7>(nxt.wb, p.ldw, nxt.w_rows, nxt.k);
          s.xr = panel_rsrc<XK>(nxt.xb, p.ldx, nxt.x_rows, nxt.k);
        }
        reset_offsets(s);
      }
      phase<2, 0, V, XK, WK>(s, smem);
      phase<3, 0, V, XK, WK>(s, smem);
      phase<0, 1, V, XK, WK>(s, smem);
      phase<1, 1, V, XK, WK>(s, smem);
      phase<2, 1, V, XK, WK>(s, smem);
      phase<3, 1, V, XK, WK>(s, smem);
    }
    // Seam between two tiles (round 5).  Inside the K loop the second wave group runs one barrier behind the first; left like that across the epilogue, the
    // groups' epilogues SERIALIZE: group 0 runs its epilogue while group 1 finishes its last compute segment and then sits at the barrier group 0 reaches
    // only after its epilogue + first load segment, and vice versa one barrier later — two epilogue times of a mostly idle matrix pipe per tile (measured
    // with s_memtime: the epilogue is 4.6k cycles, the K loop of a 64-K-tile product was 12k cycles slower per tile with it than without, profiles/r05_gemm.md
    // section 6; aligned: -5 % cycles per tile on zero-filled operands, +1 % on random ones).  So group 0 takes one extra barrier here (= waits for the partner's last compute segment), both groups run their epilogues at the same
    // time (their LDS staging blocks are per wave), and group 1 takes the extra barrier again in front of the next unit's K loop.
    if (cur.last) {
      if (wv == 0) __builtin_amdgcn_s_barrier();
      staggered = false;
    }
    if (cur.last) {
    // ---- epilogue.  A lane of a C fragment owns ONE output row (lr) and 4 consecutive columns of every 16-column block: stored directly, one wave
    // instruction would touch 16 different rows with 8 / 16 bytes each, and the CU drains such stores at ~8 B/clk.  So each wave transposes through a
    // private 4 KiB LDS block ([32 rows][128 bytes], 16-byte chunk c of row r at c ^ (r & 7)): fragment registers in by ds_write_b64 / b128, full rows
    // out by ds_read_b128, and every global store instruction writes 8 complete 128-byte row segments.  LDS operations of one wave execute in order:
    // no barrier.  Coordinates are relative to the output block of the tile (cur.out, cur.m_valid x cur.n_valid).
    {
      int lane_e = lane;  // opaque per tile: keeps the epilogue's dozens of per-lane addresses from being hoisted out of the tile loop and spilled across the K loop
      asm volatile("" : "+v"(lane_e));
      const int lr = lane_e & 15, lq = lane_e >> 4;
      float ss = 0.f;  // squared norm of what this lane stores (p.sumsq: the gradient-norm pass fused into the weight-gradient epilogue)
      unsigned char* stg = smem + kRing + wave * 4096;
      const int srow = lane_e >> 3, schunk = lane_e & 7;
      const bool f32o = (MODE == 0 || MODE == 3) && p.out_f32;  // f32 results: the dense product and the grouped weight gradient (accumulated in f32)
      const bool wide = f32o ? (p.n % 4 == 0) : (p.n % 8 == 0 && cur.ldo % 8 == 0);
      // the output block as a buffer (bf16 results): rows >= m_valid lie past its end
      const int ldo_b = (int)cur.ldo * 2;
      const __amdgpu_buffer_rsrc_t orsrc =
          out_rsrc(cur.out, cur.m_valid > 0 ? ((cur.m_valid - 1) * ldo_b + (MODE == 4 || MODE == 7 ? p.n_first + 128 : cur.n_valid) * 2) * (f32o ? 2 : 1) : 0);
      const bf16_t* bias = MODE == 0 && p.bias ? p.bias + (reinterpret_cast<unsigned char*>(cur.out) - reinterpret_cast<unsigned char*>(p.out)) / (f32o ? 4 : 2) % p.ldo : nullptr;
      // SwiGLU-backward epilogue: the tile's block of the gate half of p.gu (same rows and columns as its block of OUT), the up halves p.n columns to the right
      int ldgu_b = 0;
      __amdgpu_buffer_rsrc_t grsrc = orsrc, ursrc = orsrc, o2rsrc = orsrc;
      if constexpr (SWB) {
        const int64_t eo = (reinterpret_cast<unsigned char*>(cur.out) - reinterpret_cast<unsigned char*>(p.out)) / 2;
        const bf16_t* gt = p.gu + eo / p.ldo * p.ldgu + eo % p.ldo;
        ldgu_b = (int)p.ldgu * 2;
        const int gbytes = cur.m_valid > 0 ? (cur.m_valid - 1) * ldgu_b + cur.n_valid * 2 : 0;
        grsrc = out_rsrc(gt, gbytes);
        ursrc = out_rsrc(gt + p.n, gbytes);
        o2rsrc = out_rsrc(reinterpret_cast<bf16_t*>(cur.out) + p.n, cur.m_valid > 0 ? (cur.m_valid - 1) * ldo_b + cur.n_valid * 2 : 0);
      }
      int lm_tm = 0, lm_tn = 0;  // MODE 8 / 9: the tile's place in the (token, vocab) grid
      if constexpr (MODE == 8 || MODE == 9) (void)tile_of_block(cur.slot, p.tiles_m, p.tiles_n, p.sweep_n, lm_tm, lm_tn);
      if constexpr (MODE == 8) {
        lm_fwd_epilogue(p, lm_tm, lm_tn, s.acc, wv, wt, lr, lq);
      } else
      if (wide) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ncol0 = (MODE == 4 || MODE == 7 ? i * p.n_first : i * 128) + wv * 64;  // MODE 4: the up half sits n_first columns to the right of the gate half
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int mrow0 = j * 128 + wt * 32;
            if (bias) {
#pragma unroll
              for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int nb = ncol0 + f * 16 + lq * 4 + r;
                  const float bv = nb < cur.n_valid ? bf16_to_f32(bias[nb]) : 0.f;
                  s.acc[i][f][j][0][r] += bv;
                  s.acc[i][f][j][1][r] += bv;
                }
            }
            if (!f32o) {  // block = [32 rows: g, lr][64 columns: f, lq, r] bf16
#pragma unroll
              for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  const f32x4& c = s.acc[i][f][j][g];
                  const int ml = g * 16 + lr;
                  *reinterpret_cast<u32x2*>(stg + ml * 128 + (((f * 2 + (lq >> 1)) ^ (ml & 7)) << 4) + (lq & 1) * 8) = u32x2{pack_bf16_hw(c[0], c[1]), pack_bf16_hw(c[2], c[3])};
                }
              // four row-chunk reads, then four BUFFER stores: rows past the block are out of the descriptor's range, a column past it turns the lane's
              // offset into -1 (one v_cndmask) — no per-store branch, so the four LDS reads are in flight together and the stores issue back to back
              u32x4 val[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) val[q] = *reinterpret_cast<const u32x4*>(stg + (q * 8 + srow) * 128 + ((schunk ^ srow) << 4));
              if constexpr (SWB) {
                // val = the gradient of act = silu(gate) * up, rounded to bf16 (what the plain product would have stored); gate / up of the same 8
                // (token, feature) pairs come from the saved projection, and [d_gate | d_up] leave instead (pm_swiglu_bwd's arithmetic: swiglu_bwd_pair)
                int offo[4];
                u32x4 gv[4], uv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int m = mrow0 + q * 8 + srow, nn = ncol0 + schunk * 8, past = (cur.n_valid - 1 - nn) >> 31;
                  const int offg = (m * ldgu_b + nn * 2) | past;
                  offo[q] = (m * ldo_b + nn * 2) | past;
                  gv[q] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, offg, 0, 0);
                  uv[q] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, offg, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  u32x4 dgv, duv;
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    float dg0, du0, dg1, du1;
                    swiglu_bwd_pair(bf16lo(val[q][c]), bf16lo(gv[q][c]), bf16lo(uv[q][c]), dg0, du0);
                    swiglu_bwd_pair(bf16hi(val[q][c]), bf16hi(gv[q][c]), bf16hi(uv[q][c]), dg1, du1);
                    dgv[c] = pack_bf16_hw(dg0, dg1);
                    duv[c] = pack_bf16_hw(du0, du1);
                  }
                  __builtin_amdgcn_raw_buffer_store_b128(dgv, orsrc, offo[q], 0, 0);
                  __builtin_amdgcn_raw_buffer_store_b128(duv, o2rsrc, offo[q], 0, 0);
                }
              } else if constexpr (MODE == 9) {
                // val = 8 recomputed logits of one token, rounded to bf16; out goes dl = g inv_T (onehot(label) - exp(l inv_T - logz)) (lm_head.py:184-191,
                // the arithmetic of lmhead_kernel<BWD>); the label's column is patched in by the one lane in ~19 000 that holds it
                constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int m = mrow0 + q * 8 + srow, nn = ncol0 + schunk * 8;
                  const int tok = lm_tm * kTile + m;
                  const bool live = m < cur.m_valid;
                  const float inv_t = live ? p.lm_inv_temp[tok] : 1.f, lz = live ? p.lm_logz[tok] : 0.f;
                  const float gsc = live ? p.lm_grad[tok] * inv_t : 0.f;
                  const int rel = live ? (int)p.lm_labels[tok] - (lm_tn * kTile + nn) : -1;
                  float d[8];
#pragma unroll
                  for (int c = 0; c < 4; ++c) {
                    const float p0 = __builtin_amdgcn_exp2f((bf16lo(val[q][c]) * inv_t - lz) * kLog2e), p1 = __builtin_amdgcn_exp2f((bf16hi(val[q][c]) * inv_t - lz) * kLog2e);
                    d[2 * c] = gsc * (0.f - p0);
                    d[2 * c + 1] = gsc * (0.f - p1);
                  }
                  if ((unsigned)rel < 8u) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                      const float l = (c & 1) ? bf16hi(val[q][c >> 1]) : bf16lo(val[q][c >> 1]);
                      d[c] = rel == c ? gsc * (1.f - __builtin_amdgcn_exp2f((l * inv_t - lz) * kLog2e)) : d[c];
                    }
                  }
                  const u32x4 dv = u32x4{pack_bf16_hw(d[0], d[1]), pack_bf16_hw(d[2], d[3]), pack_bf16_hw(d[4], d[5]), pack_bf16_hw(d[6], d[7])};
                  __builtin_amdgcn_raw_buffer_store_b128(dv, orsrc, (m * ldo_b + nn * 2) | ((cur.n_valid - 1 - nn) >> 31), 0, 0);
                }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int m = mrow0 + q * 8 + srow, nn = ncol0 + schunk * 8;
                  __builtin_amdgcn_raw_buffer_store_b128(val[q], orsrc, (m * ldo_b + nn * 2) | ((cur.n_valid - 1 - nn) >> 31), 0, 0);  // nn >= n_valid: offset -1
                }
              }
            } else {  // two blocks of [32 rows][32 columns: f & 1, lq, r] f32
#pragma unroll
              for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int fl = 0; fl < 2; ++fl)
#pragma unroll
                  for (int g = 0; g < 2; ++g) {
                    const int ml = g * 16 + lr;
                    *reinterpret_cast<f32x4*>(stg + ml * 128 + (((fl * 4 + lq) ^ (ml & 7)) << 4)) = s.acc[i][a * 2 + fl][j][g];
                  }
                // as in the bf16 path: four LDS reads in flight, then branch-free buffer accesses (rows past the block: out of the descriptor's range — stores
                // dropped, loads return 0; a column past it: offset -1)
                f32x4 val[4];
                int off[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  val[q] = *reinterpret_cast<const f32x4*>(stg + (q * 8 + srow) * 128 + ((schunk ^ srow) << 4));
                  const int m = mrow0 + q * 8 + srow, nn = ncol0 + a * 32 + schunk * 4;
                  off[q] = (m * ldo_b * 2 + nn * 4) | ((cur.n_valid - 1 - nn) >> 31);
                }
                if (p.accumulate) {
#pragma unroll
                  for (int q = 0; q < 4; ++q) val[q] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, off[q], 0, 0));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val[q]), orsrc, off[q], 0, 0);
                if ((MODE == 0 || MODE == 3) && p.sumsq) {
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    const bool ok = off[q] != -1 && mrow0 + q * 8 + srow < cur.m_valid;
                    const float sq = fmaf(val[q][0], val[q][0], fmaf(val[q][1], val[q][1], fmaf(val[q][2], val[q][2], val[q][3] * val[q][3])));
                    ss += ok ? sq : 0.f;
                  }
                }
              }
            }
          }
        }
        if constexpr (MODE == 4 || MODE == 7) {  // act[m][f0 + 64 wv + ..] = bf16(bf16(silu(gate)) * up) on the ROUNDED gate / up values (what pm_swiglu_fwd computes from OUT)
          bf16_t* act = reinterpret_cast<bf16_t*>(p.out2) + ((reinterpret_cast<unsigned char*>(cur.out) - reinterpret_cast<unsigned char*>(p.out)) / 2 / p.ldo) * p.ldo2 +
                        ((reinterpret_cast<unsigned char*>(cur.out) - reinterpret_cast<unsigned char*>(p.out)) / 2 % p.ldo);
          const int ldo2_b = (int)p.ldo2 * 2;
          const __amdgpu_buffer_rsrc_t arsrc = out_rsrc(act, cur.m_valid > 0 ? (cur.m_valid - 1) * ldo2_b + 128 * 2 : 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int mrow0 = j * 128 + wt * 32;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const f32x4& cg = s.acc[0][f][j][g];
                const f32x4& cu = s.acc[1][f][j][g];
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = swiglu_act(round_bf16(cg[r]), round_bf16(cu[r]));
                const int ml = g * 16 + lr;
                *reinterpret_cast<u32x2*>(stg + ml * 128 + (((f * 2 + (lq >> 1)) ^ (ml & 7)) << 4) + (lq & 1) * 8) = u32x2{pack_bf16_hw(o[0], o[1]), pack_bf16_hw(o[2], o[3])};
              }
            u32x4 val[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) val[q] = *reinterpret_cast<const u32x4*>(stg + (q * 8 + srow) * 128 + ((schunk ^ srow) << 4));
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(val[q], arsrc, (mrow0 + q * 8 + srow) * ldo2_b + (wv * 64 + schunk * 8) * 2, 0, 0);
          }
        }
      } else {  // ragged column counts: the direct form
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int m = j * 128 + wt * 32 + g * 16 + lr;
            if (m >= cur.m_valid) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int nb = i * 128 + wv * 64 + f * 16 + lq * 4 + r;
                  if (nb >= cur.n_valid) continue;
                  float val = s.acc[i][f][j][g][r];
                  if (bias) val += bf16_to_f32(bias[nb]);
                  if (f32o) {
                    float* dst = reinterpret_cast<float*>(cur.out) + (int64_t)m * cur.ldo + nb;
                    if (p.accumulate) val += *dst;
                    *dst = val;
                    if ((MODE == 0 || MODE == 3) && p.sumsq) ss = fmaf(val, val, ss);
                  } else {
                    reinterpret_cast<bf16_t*>(cur.out)[(int64_t)m * cur.ldo + nb] = f32_to_bf16(val);
                  }
                }
          }
      }
      if ((MODE == 0 || MODE == 3) && p.sumsq) {  // wave-uniform
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (lane_e == 0) p.sumsq[(int64_t)cur.slot * 8 + wave] = ss;
      }
    }
    }  // cur.last
    if (!has_next) break;
    if (!staggered) {  // re-establish the ping-pong offset for the next unit
      if (wv == 1) __builtin_amdgcn_s_barrier();
      staggered = true;
    }
    v = v_n;
    part = part_n;
    tile_at<TM, XK, WK>(p, v, part, lane, rtab, cur);
  }  // units
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (unused) half-tiles must have landed before this workgroup's LDS is handed on
  if (staggered && wv == 0) __builtin_amdgcn_s_barrier();
}


#if PM_VARIANTS  // `make VARIANTS=1`: the measured-and-not-adopted builds (profiles/r04_gemm.md, r05_gemm.md) stay out of the product library
// =============================================================================================================================================
// Round 4: the ONE-WAVE-PER-SIMD form of the same core (`gemm4_kernel`).  Same 256 x 256 x 64 tile, same LDS images (so the DMA swizzles and the fragment
// read addresses above carry over), same tile stream / units / epilogue arithmetic; what changes is who holds what:
//   * FOUR waves (2 x 2), one per SIMD, 512 registers each: wave (wv, wt) owns W rows [128 wv, +128) x X rows [128 wt, +128) = 8 x 8 accumulator
//     fragments of v_mfma_f32_16x16x32_bf16 = 256 registers (the accumulator half of the file).  A wave reads (128 + 128) x 64 operand elements per
//     K-tile for 128 x 128 x 64 MACs — 2/3 of the LDS fragment bytes per flop of the 128 x 64 register tile of the 8-wave form (profiles/r03_gemm.md
//     section 2: what the library's kernel has over the ping-pong core under the chip's power limit).
//   * no partner wave: ONE software-pipelined stream per SIMD.  A K-tile is two k-steps of 32; the fragments of the NEXT k-step (16 ds_read_b128, or 32
//     transpose reads) are fetched into a second register set while the 64 MFMAs of the current one issue, one read per four MFMAs.
//   * the ring holds two K-tiles; ONE barrier per K-tile.  Steps of iteration (t, t + 1), buffers 0 / 1:
//        A: MFMA (t, k0)      | read (t, k1) <- buf 0                          | end: lgkmcnt(0), vmcnt(0) [K-tile t + 1 landed], s_barrier
//        B: MFMA (t, k1)      | read (t + 1, k0) <- buf 1 | DMA K-tile t + 2 -> buf 0 (16 pieces per wave, one per four MFMAs)
//        C: MFMA (t + 1, k0)  | read (t + 1, k1) <- buf 1                      | end: lgkmcnt(0), vmcnt(0) [K-tile t + 2 landed], s_barrier
//        D: MFMA (t + 1, k1)  | read (t + 2, k0) <- buf 0 | DMA K-tile t + 3 -> buf 1
//     buf 0 is re-staged after the barrier that follows every wave's last read of it (end of A); a K-tile's DMA has one whole step (64 MFMAs ~ 1 k
//     cycles) plus its own issue step to land before the wait that retires it.
// V bit 3 / 4: the timing ablations of the 8-wave kernel (no DMA inside the loop / no fragment reads).
namespace g4 {

constexpr int kSmem4 = kRing + 4 * 4096;

struct Wave4 {
  f32x4 acc[8][8];                    // [f: 16-row W fragment][g: 16-row X fragment]
  u32x4 wf[2][8], xf[2][8];           // K-contiguous operands: [register set = k-step of the K-tile][fragment]
  u32x2 wlo[2][8], whi[2][8], xlo[2][8], xhi[2][8];  // K-major operands: the two transpose-read halves of a fragment
  __amdgpu_buffer_rsrc_t wr, xr;
  int vw[4], vx[4];                   // per-lane byte offsets of this wave's four DMA pieces of a slot (piece q minus its q KiB instruction offset)
  int off[4];                         // wave-uniform running source offsets per slot kind
  int wk, xk, wstep, xstep, piece_lds;
  unsigned aw[8], ax[8];              // fragment read addresses: K-contiguous [k-step] (fragment = immediate); K-major [fragment] (k-step = immediate)
  // register staging (RS): the K-tile travels HBM / L2 -> VGPRs (buffer_load_dwordx4, cheap to issue) -> LDS (ds_write_b128 one step later) instead of
  // by LDS-DMA, whose ~60-cycle issue cost has no partner wave to hide behind at one wave per SIMD (profiles/r04_gemm.md)
  u32x4 stg[16];
  unsigned wrw, wrx;                  // per-lane LDS byte address of this wave's first piece in the W / X window (slot + piece = immediate)
};

template <bool KM>
__device__ __forceinline__ void setup_operand4(int64_t ld, int wave, int lane, unsigned lds_slot0, int (&voff)[4], int& half_step, int& k_step, unsigned (&rd)[8]) {
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) rd[i] = 0;
  if (!KM) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rho = (wave * 4 + q) * 8 + (lane >> 3);  // slot row of this lane's 16 bytes (piece = 8 rows x 128 bytes)
      voff[q] = (int)(rho * ld * 2) + (((lane & 7) ^ ((rho >> 1) & 7)) << 4) - q * 1024;
    }
    half_step = (int)(128 * ld * 2);
    k_step = kBK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) rd[kk] = lds_slot0 + lr * 128 + (((kk * 4 + lq) ^ ((lr >> 1) & 7)) << 4);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int krow = (wave * 4 + q) * 4 + (lane >> 4);  // piece = 4 k-rows x 256 bytes
      voff[q] = (int)(krow * ld * 2) + (((lane & 15) ^ (((krow & 3) << 2) | (((krow >> 3) & 1) << 1))) << 4) - q * 1024;
    }
    half_step = 128 * 2;
    k_step = (int)(kBK * ld * 2);
    const int krow = lq * 8 + (lr >> 2);
    const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int chunk = f * 2 + ((lr & 3) >> 1);
      rd[f] = lds_slot0 + krow * 256 + ((chunk ^ swz) << 4) + (lr & 1) * 8;
    }
  }
}

// one DMA piece (1 KiB of this wave's 4 KiB of slot SLOT of buffer BUF); the fourth piece advances the slot's source offset by one K-tile
template <int SLOT, int BUF, int Q>
__device__ __forceinline__ void stage_piece(Wave4& s, unsigned char* smem) {
  constexpr int slot_off = slot_offset(BUF, SLOT);
  constexpr bool is_w = SLOT == SLOT_W0 || SLOT == SLOT_W1;
  const __amdgpu_buffer_rsrc_t rsrc = is_w ? s.wr : s.xr;
  unsigned char* dst = smem + slot_off + s.piece_lds;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)dst, 16, is_w ? s.vw[Q] : s.vx[Q], s.off[SLOT], Q * 1024, 0);
  if (Q == 3) s.off[SLOT] += is_w ? s.wk : s.xk;
}
template <int BUF>
__device__ __forceinline__ void stage_tile(Wave4& s, unsigned char* smem) {  // the whole K-tile at once (prologue of the first unit)
  stage_piece<SLOT_W0, BUF, 0>(s, smem); stage_piece<SLOT_W0, BUF, 1>(s, smem); stage_piece<SLOT_W0, BUF, 2>(s, smem); stage_piece<SLOT_W0, BUF, 3>(s, smem);
  stage_piece<SLOT_X0, BUF, 0>(s, smem); stage_piece<SLOT_X0, BUF, 1>(s, smem); stage_piece<SLOT_X0, BUF, 2>(s, smem); stage_piece<SLOT_X0, BUF, 3>(s, smem);
  stage_piece<SLOT_W1, BUF, 0>(s, smem); stage_piece<SLOT_W1, BUF, 1>(s, smem); stage_piece<SLOT_W1, BUF, 2>(s, smem); stage_piece<SLOT_W1, BUF, 3>(s, smem);
  stage_piece<SLOT_X1, BUF, 0>(s, smem); stage_piece<SLOT_X1, BUF, 1>(s, smem); stage_piece<SLOT_X1, BUF, 2>(s, smem); stage_piece<SLOT_X1, BUF, 3>(s, smem);
}
// piece I (0..15) of the K-tile staged into buffer BUF during a B / D step
template <int BUF, int I>
__device__ __forceinline__ void stage_nth(Wave4& s, unsigned char* smem) {
  constexpr int slot = I < 4 ? SLOT_W0 : I < 8 ? SLOT_X0 : I < 12 ? SLOT_W1 : SLOT_X1;
  stage_piece<slot, BUF, I & 3>(s, smem);
}

// ---- register staging: piece I of the NEXT-BUT-ONE K-tile is loaded into stg[I] in a B / D step and written to its ring slot in the following C / A step
template <int I>
__device__ __forceinline__ void load_nth(Wave4& s) {
  constexpr int slot = I < 4 ? SLOT_W0 : I < 8 ? SLOT_X0 : I < 12 ? SLOT_W1 : SLOT_X1;
  constexpr bool is_w = slot == SLOT_W0 || slot == SLOT_W1;
  constexpr int q = I & 3;
  // (vw / vx hold piece q's offset minus q KiB — the LDS-DMA form moves source and destination together through the instruction offset; undo it here)
  s.stg[I] = __builtin_amdgcn_raw_buffer_load_b128(is_w ? s.wr : s.xr, (is_w ? s.vw[q] : s.vx[q]) + q * 1024, s.off[slot], 0);
  if (q == 3) s.off[slot] += is_w ? s.wk : s.xk;
}
template <int BUF, int I>
__device__ __forceinline__ void write_nth(Wave4& s) {
  constexpr int slot = I < 4 ? SLOT_W0 : I < 8 ? SLOT_X0 : I < 12 ? SLOT_W1 : SLOT_X1;
  constexpr bool is_w = slot == SLOT_W0 || slot == SLOT_W1;
  constexpr int imm = slot_in_window(BUF, slot) + (I & 3) * 1024;
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(is_w ? s.wrw : s.wrx), "v"(s.stg[I]), "n"(imm) : "memory");
}

// fragment read I (0..15) of k-step SET of the K-tile in buffer BUF: 0-7 = W fragments, 8-15 = X fragments.  The wave's own half (W^wv, X^wt) is part of
// the per-lane base address; the buffer is an immediate.
template <int SET, int BUF, int I, bool XK, bool WK>
__device__ __forceinline__ void read_nth(Wave4& s) {
  constexpr int slot = BUF * 2 * kSlot;
  if constexpr (I < 8) {
    if constexpr (!WK) s.wf[SET][I] = b128_read<slot + I * 2048>(s.aw[SET]);
    else {
      s.wlo[SET][I] = tr_read<slot + SET * 8192>(s.aw[I]);
      s.whi[SET][I] = tr_read<slot + SET * 8192 + 1024>(s.aw[I]);
    }
  } else {
    constexpr int G = I - 8;
    if constexpr (!XK) s.xf[SET][G] = b128_read<slot + G * 2048>(s.ax[SET]);
    else {
      s.xlo[SET][G] = tr_read<slot + SET * 8192>(s.ax[G]);
      s.xhi[SET][G] = tr_read<slot + SET * 8192 + 1024>(s.ax[G]);
    }
  }
}
template <int SET, bool XK, bool WK>
__device__ __forceinline__ void tie_set(Wave4& s) {
  if constexpr (WK) {
    tie(s.wlo[SET][0], s.whi[SET][0], s.wlo[SET][1], s.whi[SET][1], s.wlo[SET][2], s.whi[SET][2], s.wlo[SET][3], s.whi[SET][3]);
    tie(s.wlo[SET][4], s.whi[SET][4], s.wlo[SET][5], s.whi[SET][5], s.wlo[SET][6], s.whi[SET][6], s.wlo[SET][7], s.whi[SET][7]);
  } else {
    tie(s.wf[SET][0], s.wf[SET][1], s.wf[SET][2], s.wf[SET][3]);
    tie(s.wf[SET][4], s.wf[SET][5], s.wf[SET][6], s.wf[SET][7]);
  }
  if constexpr (XK) {
    tie(s.xlo[SET][0], s.xhi[SET][0], s.xlo[SET][1], s.xhi[SET][1], s.xlo[SET][2], s.xhi[SET][2], s.xlo[SET][3], s.xhi[SET][3]);
    tie(s.xlo[SET][4], s.xhi[SET][4], s.xlo[SET][5], s.xhi[SET][5], s.xlo[SET][6], s.xhi[SET][6], s.xlo[SET][7], s.xhi[SET][7]);
  } else {
    tie(s.xf[SET][0], s.xf[SET][1], s.xf[SET][2], s.xf[SET][3]);
    tie(s.xf[SET][4], s.xf[SET][5], s.xf[SET][6], s.xf[SET][7]);
  }
}

// group I (0..15) of a step: four MFMAs (W fragment I / 2 against X fragments 4 (I % 2) .. + 3), then one fragment read of the next register set and, in a
// staging step, one DMA piece.  `sched_barrier(0)` after every group pins the interleave: asm reads are invisible to the scheduler's instruction classes.
template <int SET, int NBUF, int DBUF, bool DMA, int I, int V, bool XK, bool WK>
__device__ __forceinline__ void group(Wave4& s, unsigned char* smem) {
  constexpr bool RS = (V & 1) != 0;
  constexpr int f = I >> 1, g0 = (I & 1) * 4;
  const u32x4 wfrag = WK ? join(s.wlo[SET][f], s.whi[SET][f]) : s.wf[SET][f];
#pragma unroll
  for (int g = g0; g < g0 + 4; ++g) {
    const u32x4 xfrag = XK ? join(s.xlo[SET][g], s.xhi[SET][g]) : s.xf[SET][g];
    s.acc[f][g] = mfma16(wfrag, xfrag, s.acc[f][g]);
  }
  if (!(V & 16)) read_nth<SET ^ 1, NBUF, I, XK, WK>(s);
  if (!(V & 8)) {
    if (!RS) {
      if (DMA) stage_nth<DBUF, I>(s, smem);
    } else if (DMA) {
      load_nth<I>(s);
    } else {
      write_nth<DBUF, I>(s);  // a C / A step: the K-tile loaded during the previous step goes to the buffer the NEXT step reads
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// One step: MFMAs on register set SET; reads into SET ^ 1 from buffer NBUF; DMA (staging steps) of the next-but-one K-tile into buffer DBUF — or, with
// register staging, its loads (staging steps) / its ds_writes into DBUF (the step after).
// SYNC: the step ends a K-tile's reads of its buffer -> wait for the other buffer's DMA (or the ds_writes), barrier.
template <int SET, int NBUF, int DBUF, bool DMA, bool SYNC, int V, bool XK, bool WK>
__device__ __forceinline__ void step(Wave4& s, unsigned char* smem) {
  group<SET, NBUF, DBUF, DMA, 0, V, XK, WK>(s, smem);  group<SET, NBUF, DBUF, DMA, 1, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 2, V, XK, WK>(s, smem);  group<SET, NBUF, DBUF, DMA, 3, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 4, V, XK, WK>(s, smem);  group<SET, NBUF, DBUF, DMA, 5, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 6, V, XK, WK>(s, smem);  group<SET, NBUF, DBUF, DMA, 7, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 8, V, XK, WK>(s, smem);  group<SET, NBUF, DBUF, DMA, 9, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 10, V, XK, WK>(s, smem); group<SET, NBUF, DBUF, DMA, 11, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 12, V, XK, WK>(s, smem); group<SET, NBUF, DBUF, DMA, 13, V, XK, WK>(s, smem);
  group<SET, NBUF, DBUF, DMA, 14, V, XK, WK>(s, smem); group<SET, NBUF, DBUF, DMA, 15, V, XK, WK>(s, smem);
  if (SYNC) {
    if (V & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // register staging: nothing LDS-bound is on the vector-memory counter
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (!(V & 16)) tie_set<SET ^ 1, XK, WK>(s);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- the FINE schedule (V bit 1; variant bit 8): the main loop the library's 4-wave kernel runs (profiles/r04_gemm.md section 3: disassembled from the shipped
// code object), on this kernel's LDS images.  One K-tile = 32 groups of four MFMAs (16 on the k-step-0 registers, 16 on the k-step-1 registers):
//     groups 0-2    read the W fragments of (t, k1)                                  | end of group 4: lgkmcnt(0), BARRIER 1 -> the W half of buffer b is free
//     groups 5-7    read the X fragments of (t, k1)                                  | end of group 9: lgkmcnt(0), BARRIER 2 -> the X half of buffer b is free
//     groups 5-15   the 8 W pieces of K-tile t + 2 -> buffer b, two every three groups
//     groups 17-27  the 8 X pieces of K-tile t + 2 -> buffer b                       | end of group 21: vmcnt(12), BARRIER 3 -> K-tile t + 1 has landed in buffer b ^ 1
//     groups 22-27  read the fragments of (t + 1, k0) from buffer b ^ 1              | end of group 31: lgkmcnt(0)
// Against the two-step schedule above: the DMA pieces are spread over 23 of the 32 groups instead of bunched into 16 (one texture-addresser queue, four
// waves), a piece has MORE than a whole K-tile to land (issued in iteration t, waited for at barrier 3 of iteration t + 1; `vmcnt` never 0), and every
// fragment read is issued >= 2 groups before the wait that retires it.  Three barriers per K-tile.
__host__ __device__ constexpr int fine_dma_piece(int j) {  // the piece (0-3 W0, 4-7 X0, 8-11 W1, 12-15 X1) issued after group j, or -1
  if (j < 5 || j > 27) return -1;
  const int r = j - 5;
  if (r % 3 == 2) return -1;
  const int n = (r / 3) * 2 + r % 3;  // 0-7: the W pieces, 8-15: the X pieces
  return n < 4 ? n : n < 8 ? n + 4 : n < 12 ? n - 4 : n;
}
template <int SET, int BUF, int I0, int N, bool XK, bool WK>
__device__ __forceinline__ void read_run(Wave4& s) {
  if constexpr (N > 0) {
    read_nth<SET, BUF, I0, XK, WK>(s);
    read_run<SET, BUF, I0 + 1, N - 1, XK, WK>(s);
  }
}
template <int BUF, int J, int V, bool XK, bool WK>
__device__ __forceinline__ void fine_group(Wave4& s, unsigned char* smem) {
  constexpr int SET = J >> 4, I = J & 15, f = I >> 1, g0 = (I & 1) * 4;
  const u32x4 wfrag = WK ? join(s.wlo[SET][f], s.whi[SET][f]) : s.wf[SET][f];
#pragma unroll
  for (int g = g0; g < g0 + 4; ++g) {
    const u32x4 xfrag = XK ? join(s.xlo[SET][g], s.xhi[SET][g]) : s.xf[SET][g];
    s.acc[f][g] = mfma16(wfrag, xfrag, s.acc[f][g]);
  }
  if (!(V & 16)) {
    if constexpr (J == 0) read_run<1, BUF, 0, 3, XK, WK>(s);
    if constexpr (J == 1) read_run<1, BUF, 3, 3, XK, WK>(s);
    if constexpr (J == 2) read_run<1, BUF, 6, 2, XK, WK>(s);
    if constexpr (J == 5) read_run<1, BUF, 8, 3, XK, WK>(s);
    if constexpr (J == 6) read_run<1, BUF, 11, 3, XK, WK>(s);
    if constexpr (J == 7) read_run<1, BUF, 14, 2, XK, WK>(s);
    if constexpr (J == 22) read_run<0, BUF ^ 1, 0, 3, XK, WK>(s);
    if constexpr (J == 23) read_run<0, BUF ^ 1, 3, 3, XK, WK>(s);
    if constexpr (J == 24) read_run<0, BUF ^ 1, 6, 3, XK, WK>(s);
    if constexpr (J == 25) read_run<0, BUF ^ 1, 9, 3, XK, WK>(s);
    if constexpr (J == 26) read_run<0, BUF ^ 1, 12, 2, XK, WK>(s);
    if constexpr (J == 27) read_run<0, BUF ^ 1, 14, 2, XK, WK>(s);
  }
  if constexpr (fine_dma_piece(J) >= 0) {
    if (!(V & 8)) stage_nth<BUF, fine_dma_piece(J)>(s, smem);
  }
  if constexpr (J == 4 || J == 9) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (J == 9 && !(V & 16)) tie_set<1, XK, WK>(s);
  }
  if constexpr (J == 21) {
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (J == 31) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(V & 16)) tie_set<0, XK, WK>(s);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int BUF, int J, int V, bool XK, bool WK>
__device__ __forceinline__ void fine_groups(Wave4& s, unsigned char* smem) {
  if constexpr (J < 32) {
    fine_group<BUF, J, V, XK, WK>(s, smem);
    fine_groups<BUF, J + 1, V, XK, WK>(s, smem);
  }
}

template <int V, bool XK, bool WK, int MODE>
__global__ __launch_bounds__(256, 1) void gemm4_kernel(const Params p) {
  unsigned char* smem = lds_base();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = wave >> 1, wt = wave & 1;
  int v = blockIdx.x, part = 0;
  Tile cur;
  RowTab rtab;
  rtab.use = false;
  if (!next_unit<MODE, XK, WK>(p, v, part, lane, rtab, cur)) return;

  Wave4 s;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  setup_operand4<WK>(p.ldw, wave, lane, lds0 + wv * kSlot, s.vw, s.wstep, s.wk, s.aw);
  setup_operand4<XK>(p.ldx, wave, lane, lds0 + 4 * kSlot + wt * kSlot, s.vx, s.xstep, s.xk, s.ax);
  s.wr = panel_rsrc<WK>(cur.wb, p.ldw, cur.w_rows, cur.k);
  s.xr = panel_rsrc<XK>(cur.xb, p.ldx, cur.x_rows, cur.k);
  s.piece_lds = wave * 4096;
  if (V & 16) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        s.wf[st][f] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        s.xf[st][f] = u32x4{0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u};
        s.wlo[st][f] = s.whi[st][f] = s.xlo[st][f] = s.xhi[st][f] = u32x2{0x3f803f80u, 0x3f803f80u + lane};
      }
  }
  // ---- prologue of the first unit: K-tiles 0 and 1 on their way, k-step 0 of K-tile 0 in register set 0
  s.off[SLOT_W0] = 0; s.off[SLOT_W1] = s.wstep; s.off[SLOT_X0] = 0; s.off[SLOT_X1] = s.xstep;
  s.wrw = lds0 + s.piece_lds + lane * 16;
  s.wrx = s.wrw + 4 * kSlot;
  if (V & 1) {  // register staging: K-tile 0 through the registers into buffer 0, K-tile 1 stays in the staging registers until the first A step writes it
    load_nth<0>(s);  load_nth<1>(s);  load_nth<2>(s);  load_nth<3>(s);  load_nth<4>(s);  load_nth<5>(s);  load_nth<6>(s);  load_nth<7>(s);
    load_nth<8>(s);  load_nth<9>(s);  load_nth<10>(s); load_nth<11>(s); load_nth<12>(s); load_nth<13>(s); load_nth<14>(s); load_nth<15>(s);
    write_nth<0, 0>(s);  write_nth<0, 1>(s);  write_nth<0, 2>(s);  write_nth<0, 3>(s);  write_nth<0, 4>(s);  write_nth<0, 5>(s);  write_nth<0, 6>(s);  write_nth<0, 7>(s);
    write_nth<0, 8>(s);  write_nth<0, 9>(s);  write_nth<0, 10>(s); write_nth<0, 11>(s); write_nth<0, 12>(s); write_nth<0, 13>(s); write_nth<0, 14>(s); write_nth<0, 15>(s);
    __builtin_amdgcn_sched_barrier(0);
    load_nth<0>(s);  load_nth<1>(s);  load_nth<2>(s);  load_nth<3>(s);  load_nth<4>(s);  load_nth<5>(s);  load_nth<6>(s);  load_nth<7>(s);
    load_nth<8>(s);  load_nth<9>(s);  load_nth<10>(s); load_nth<11>(s); load_nth<12>(s); load_nth<13>(s); load_nth<14>(s); load_nth<15>(s);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    stage_tile<0>(s, smem);
    stage_tile<1>(s, smem);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (!(V & 16)) {
    read_nth<0, 0, 0, XK, WK>(s);  read_nth<0, 0, 1, XK, WK>(s);  read_nth<0, 0, 2, XK, WK>(s);  read_nth<0, 0, 3, XK, WK>(s);
    read_nth<0, 0, 4, XK, WK>(s);  read_nth<0, 0, 5, XK, WK>(s);  read_nth<0, 0, 6, XK, WK>(s);  read_nth<0, 0, 7, XK, WK>(s);
    read_nth<0, 0, 8, XK, WK>(s);  read_nth<0, 0, 9, XK, WK>(s);  read_nth<0, 0, 10, XK, WK>(s); read_nth<0, 0, 11, XK, WK>(s);
    read_nth<0, 0, 12, XK, WK>(s); read_nth<0, 0, 13, XK, WK>(s); read_nth<0, 0, 14, XK, WK>(s); read_nth<0, 0, 15, XK, WK>(s);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_set<0, XK, WK>(s);
    __builtin_amdgcn_sched_barrier(0);
  }

  for (;;) {  // ---- the workgroup's stream of units
    {
      int lane_o = lane;  // per-lane DMA offsets / read addresses re-derived per unit: they must not stay live (and spill) across the epilogue
      asm volatile("" : "+v"(lane_o));
      int dummy_a, dummy_b;
      setup_operand4<WK>(p.ldw, wave, lane_o, lds0 + wv * kSlot, s.vw, dummy_a, dummy_b, s.aw);
      setup_operand4<XK>(p.ldx, wave, lane_o, lds0 + 4 * kSlot + wt * kSlot, s.vx, dummy_a, dummy_b, s.ax);
      s.wrw = lds0 + s.piece_lds + lane_o * 16;
      s.wrx = s.wrw + 4 * kSlot;
    }
    int v_n = v, part_n = part;
    bool has_next = false;
    if (cur.first) {
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int g = 0; g < 8; ++g) s.acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nk = cur.nk;
    for (int t = 0; t < nk; t += 2) {
      if (t + 2 >= nk) {  // the last pair of this unit: both K-tiles staged in this iteration belong to the next one
        advance_unit(v_n, part_n, MODE == 0 ? p.k_parts : parts_of<MODE>());
        Tile nxt;
        has_next = next_unit<MODE, XK, WK>(p, v_n, part_n, lane, rtab, nxt);
        if (has_next) {
          s.wr = panel_rsrc<WK>(nxt.wb, p.ldw, nxt.w_rows, nxt.k);
          s.xr = panel_rsrc<XK>(nxt.xb, p.ldx, nxt.x_rows, nxt.k);
        }
        s.off[SLOT_W0] = 0; s.off[SLOT_W1] = s.wstep; s.off[SLOT_X0] = 0; s.off[SLOT_X1] = s.xstep;
      }
      if constexpr ((V & 2) != 0) {  // the fine schedule: K-tile t out of buffer 0, K-tile t + 1 out of buffer 1
        fine_groups<0, 0, V, XK, WK>(s, smem);
        fine_groups<1, 0, V, XK, WK>(s, smem);
      } else {
        step<0, 0, 1, false, true, V, XK, WK>(s, smem);   // A   (register staging: writes the K-tile loaded in the previous D into buffer 1)
        step<1, 1, 0, true, false, V, XK, WK>(s, smem);   // B
        step<0, 1, 0, false, true, V, XK, WK>(s, smem);   // C   (register staging: writes the K-tile loaded in B into buffer 0)
        step<1, 0, 1, true, false, V, XK, WK>(s, smem);   // D
      }
    }
    if (cur.last) {
      // ---- epilogue: the 8-wave kernel's LDS-staged form (every global store instruction writes 8 full 128-byte row segments), walked over this wave's
      // 4 (row blocks of 32) x 2 (column blocks of 64) blocks.  acc[f][g][r] = OUT[128 wt + 16 g + lr][128 wv + 16 f + 4 lq + r].
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int lr = lane_e & 15, lq = lane_e >> 4;
      float ss = 0.f;
      unsigned char* stg = smem + kRing + wave * 4096;
      const int srow = lane_e >> 3, schunk = lane_e & 7;
      const bool f32o = (MODE == 0 || MODE == 3) && p.out_f32;
      const bool wide = f32o ? (p.n % 4 == 0) : (p.n % 8 == 0 && cur.ldo % 8 == 0);
      const bf16_t* bias = MODE == 0 && p.bias ? p.bias + (reinterpret_cast<unsigned char*>(cur.out) - reinterpret_cast<unsigned char*>(p.out)) / (f32o ? 4 : 2) % p.ldo : nullptr;
      if (wide) {
#pragma unroll
        for (int fq = 0; fq < 2; ++fq) {
          const int ncol0 = wv * 128 + fq * 64;
#pragma unroll
          for (int gp = 0; gp < 4; ++gp) {
            const int mrow0 = wt * 128 + gp * 32;
            if (bias) {
#pragma unroll
              for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int nb = ncol0 + f * 16 + lq * 4 + r;
                  const float bv = nb < cur.n_valid ? bf16_to_f32(bias[nb]) : 0.f;
                  s.acc[fq * 4 + f][gp * 2][r] += bv;
                  s.acc[fq * 4 + f][gp * 2 + 1][r] += bv;
                }
            }
            if (!f32o) {
#pragma unroll
              for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  const f32x4& c = s.acc[fq * 4 + f][gp * 2 + g];
                  const int ml = g * 16 + lr;
                  *reinterpret_cast<u32x2*>(stg + ml * 128 + (((f * 2 + (lq >> 1)) ^ (ml & 7)) << 4) + (lq & 1) * 8) = u32x2{pack_bf16_hw(c[0], c[1]), pack_bf16_hw(c[2], c[3])};
                }
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int ml = q * 8 + srow;
                const u32x4 val = *reinterpret_cast<const u32x4*>(stg + ml * 128 + ((schunk ^ srow) << 4));
                const int m = mrow0 + ml, nn = ncol0 + schunk * 8;
                if (m < cur.m_valid && nn < cur.n_valid) {
                  u32x4* dstv = reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(cur.out) + (int64_t)m * cur.ldo + nn);
                  if (p.nt_out) __builtin_nontemporal_store(val, dstv);
                  else *dstv = val;
                }
              }
            } else {
#pragma unroll
              for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int fl = 0; fl < 2; ++fl)
#pragma unroll
                  for (int g = 0; g < 2; ++g) {
                    const int ml = g * 16 + lr;
                    *reinterpret_cast<f32x4*>(stg + ml * 128 + (((fl * 4 + lq) ^ (ml & 7)) << 4)) = s.acc[fq * 4 + a * 2 + fl][gp * 2 + g];
                  }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int ml = q * 8 + srow;
                  f32x4 val = *reinterpret_cast<const f32x4*>(stg + ml * 128 + ((schunk ^ srow) << 4));
                  const int m = mrow0 + ml, nn = ncol0 + a * 32 + schunk * 4;
                  if (m < cur.m_valid && nn < cur.n_valid) {
                    float* dst = reinterpret_cast<float*>(cur.out) + (int64_t)m * cur.ldo + nn;
                    if (p.accumulate) val += *reinterpret_cast<const f32x4*>(dst);
                    if (p.nt_out) __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(dst));
                    else *reinterpret_cast<f32x4*>(dst) = val;
                    if (MODE == 0 && p.sumsq) ss = fmaf(val[0], val[0], fmaf(val[1], val[1], fmaf(val[2], val[2], fmaf(val[3], val[3], ss))));
                  }
                }
              }
            }
          }
        }
      } else {  // ragged column counts: the direct form
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int m = wt * 128 + g * 16 + lr;
          if (m >= cur.m_valid) continue;
#pragma unroll
          for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int nb = wv * 128 + f * 16 + lq * 4 + r;
              if (nb >= cur.n_valid) continue;
              float val = s.acc[f][g][r];
              if (bias) val += bf16_to_f32(bias[nb]);
              if (f32o) {
                float* dst = reinterpret_cast<float*>(cur.out) + (int64_t)m * cur.ldo + nb;
                if (p.accumulate) val += *dst;
                *dst = val;
                if (MODE == 0 && p.sumsq) ss = fmaf(val, val, ss);
              } else {
                reinterpret_cast<bf16_t*>(cur.out)[(int64_t)m * cur.ldo + nb] = f32_to_bf16(val);
              }
            }
        }
      }
      if (MODE == 0 && p.sumsq) {  // 8 slots per tile (the 8-wave kernel's layout): this kernel fills the first four and clears the rest
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (lane_e == 0) {
          p.sumsq[(int64_t)v * 8 + wave] = ss;
          p.sumsq[(int64_t)v * 8 + 4 + wave] = 0.f;
        }
      }
    }  // cur.last
    if (!has_next) break;
    v = v_n;
    part = part_n;
    tile_at<MODE, XK, WK>(p, v, part, lane, rtab, cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (unused) K-tiles must have landed before this workgroup's LDS is handed on
}

template <int V, bool XK, bool WK, int MODE = 0>
static void launch_v4(const Params& p, hipStream_t st, dim3 grid) {
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)gemm4_kernel<V, XK, WK, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem4);
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL((gemm4_kernel<V, XK, WK, MODE>), grid, dim3(256), kSmem4, st, p);
}

}  // namespace g4
#endif  // PM_VARIANTS

template <int V, bool XK, bool WK, int MODE = 0>
static void launch_v(const Params& p, hipStream_t st, dim3 grid) {
  static bool attr_done[64] = {};  // per device: the attribute belongs to the function object of the device the launch goes to
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<V, XK, WK, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    attr_done[dev & 63] = true;
  }
  hipLaunchKernelGGL((gemm_kernel<V, XK, WK, MODE>), grid, dim3(512), kSmem, st, p);
}

// The shipped K loop (round 5): setprio around the MFMA clusters, two-phase K-tile, split B segment.  profiles/r05_gemm.md has the A/B against the
// four-phase loop of rounds 3-4 (weight gradients +2-4 %, dgrad on the stored [out, in] weight +5-7 %, forward +0-2 %, bit-identical results).
constexpr int kShippedLoop = 1 | 256 | 1024;

template <bool XK, bool WK>
static void launch(const Params& p, hipStream_t st, int g_variant) {
  // bit 2 of the variant: one persistent workgroup per CU walking its tiles (the next tile's first K-tiles are staged under the current tile's last
  // ones); otherwise one workgroup per tile.  n_blocks is a multiple of 256, so both grids keep "block b runs on XCD b % 8".
  const dim3 grid((g_variant & 4) ? std::min(p.n_blocks, 256) : p.n_blocks);
#if PM_VARIANTS  // A/B partners (`make VARIANTS=1`); the numbers are the round-3 / round-4 variant words so that the profiles' command lines still run
  if (g_variant & (64 | 128 | 256)) {  // bit 6: the one-wave-per-SIMD kernel (bits 3 / 4: its timing ablations)
    const int v4 = g_variant & (8 | 16);
    if (g_variant & 256) {  // bit 8: the fine schedule (bits 3 / 4: its timing ablations, forward layout)
      if (!XK && !WK && v4 == 8) g4::launch_v4<2 | 8, false, false>(p, st, grid);
      else if (!XK && !WK && v4 == 16) g4::launch_v4<2 | 16, false, false>(p, st, grid);
      else g4::launch_v4<2, XK, WK>(p, st, grid);
    } else if (g_variant & 128) g4::launch_v4<1, XK, WK>(p, st, grid);  // bit 7: register staging instead of LDS-DMA
    else if (!XK && !WK && v4 == 8) g4::launch_v4<8, false, false>(p, st, grid);
    else if (!XK && !WK && v4 == 16) g4::launch_v4<16, false, false>(p, st, grid);
    else if (!XK && !WK && v4 == 24) g4::launch_v4<24, false, false>(p, st, grid);
    else g4::launch_v4<0, XK, WK>(p, st, grid);
    return;
  }
  if (g_variant & 65536) {  // bit 16: the FOUR-PHASE loop of rounds 3-4 (bits 0 / 1: setprio / balanced reads; 3 / 4: ablations; 11 / 12: DMA piece placement)
    const int v = g_variant & (1 | 2 | 8 | 16 | 2048 | 4096);
    if (v == (3 | 2048)) launch_v<3 | 32, XK, WK>(p, st, grid);
    else if (v == (3 | 4096)) launch_v<3 | 64, XK, WK>(p, st, grid);
    else if (v == 0) launch_v<0, XK, WK>(p, st, grid);
    else if (v == 1) launch_v<1, XK, WK>(p, st, grid);
    else if (!XK && !WK && v == (1 | 8)) launch_v<1 | 8, false, false>(p, st, grid);  // timing ablations, forward layout only
    else if (!XK && !WK && v == (1 | 16)) launch_v<1 | 16, false, false>(p, st, grid);
    else if (!XK && !WK && v == (1 | 8 | 16)) launch_v<1 | 8 | 16, false, false>(p, st, grid);
    else launch_v<3, XK, WK>(p, st, grid);
    return;
  }
  if (g_variant & 32768) {  // bit 15: other forms of the two-phase loop
    const int v = g_variant & (1 | 2 | 8 | 16 | 131072);
    if (v == 3) launch_v<3 | 256, XK, WK>(p, st, grid);                                        // W0 pre-read (spills on the K-major layouts)
    else if (v == 1) launch_v<1 | 256, XK, WK>(p, st, grid);                                   // B segment not split
    else if (v == 0) launch_v<0 | 256, XK, WK>(p, st, grid);                                   // no setprio
    else if (!XK && !WK && v == (1 | 8)) launch_v<1 | 8 | 256, false, false>(p, st, grid);     // timing ablations (forward layout)
    else if (!XK && !WK && v == (1 | 16)) launch_v<1 | 16 | 256, false, false>(p, st, grid);
    else launch_v<kShippedLoop, XK, WK>(p, st, grid);
    return;
  }
#endif
  launch_v<kShippedLoop, XK, WK>(p, st, grid);
}
// grouped products: the shipped loop; persistent unless the debug variant clears bit 2
template <bool XK, bool WK, int MODE>
static void launch_grouped(const Params& p, hipStream_t st, int g_variant) {
  const dim3 grid((g_variant & 4) ? std::min(p.n_blocks, 256) : p.n_blocks);
#if PM_VARIANTS
  if (g_variant & 65536) { launch_v<3, XK, WK, MODE>(p, st, grid); return; }
#endif
  launch_v<kShippedLoop, XK, WK, MODE>(p, st, grid);
}

// out[tile] += sum over parts >= 1 of the workspace tiles of a split tail tile, in part order; with `sumsq` also the sum of squares of the FINAL tile.
// Eight workgroups per tail tile slot (32 rows each, 16-byte accesses); workgroup j's partial of the squared norm goes to wave slot j of part 0's unit,
// the slots the GEMM epilogues wrote for the other parts of this tile are reset by workgroup 0.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Params p) {
  __shared__ float red[4];
  const int slot = blockIdx.x >> 3, sub = blockIdx.x & 7;  // (patch slot, position inside the patch), 32-row band
  const int x0 = slot >> 5, within = slot & 31;
  const int vv = p.split_v0 + within * 8 + x0;
  int tm, tn;
  if (!tile_of_block(vv, p.tiles_m, p.tiles_n, p.sweep_n, tm, tn)) return;
  const int64_t m0 = (int64_t)tm * kTile, n0 = (int64_t)tn * kTile;
  const int mv = (int)(p.m - m0 < kTile ? p.m - m0 : kTile), nv = (int)(p.n - n0 < kTile ? p.n - n0 : kTile);
  float* out = reinterpret_cast<float*>(p.out) + m0 * p.ldo + n0;
  const float* ws = p.split_ws + (int64_t)slot * (p.split_s - 1) * (kTile * kTile);
  const int parts = (p.k + p.split_k - 1) / p.split_k;
  const int c = (threadIdx.x & 63) * 4;  // 64 lanes x 16 bytes = one 256-float row; the 4 waves take 4 rows at a time
  float ss = 0.f;
  const bool vec = nv == kTile && (p.ldo & 3) == 0;
#pragma unroll 2
  for (int r = sub * 32 + (threadIdx.x >> 6); r < sub * 32 + 32 && r < mv; r += 4) {
    if (vec) {
      f32x4 v = *reinterpret_cast<const f32x4*>(out + (int64_t)r * p.ldo + c);
      for (int q = 1; q < parts; ++q) v += *reinterpret_cast<const f32x4*>(ws + (int64_t)(q - 1) * (kTile * kTile) + r * kTile + c);
      *reinterpret_cast<f32x4*>(out + (int64_t)r * p.ldo + c) = v;
      ss = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], ss))));
    } else {
      for (int e = 0; e < 4 && c + e < nv; ++e) {
        float v = out[(int64_t)r * p.ldo + c + e];
        for (int q = 1; q < parts; ++q) v += ws[(int64_t)(q - 1) * (kTile * kTile) + r * kTile + c + e];
        out[(int64_t)r * p.ldo + c + e] = v;
        ss = fmaf(v, v, ss);
      }
    }
  }
  if (p.sumsq) {
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) p.sumsq[(int64_t)((vv & ~7) | x0) * 8 + sub] = (red[0] + red[1]) + (red[2] + red[3]);
    if (sub == 0 && threadIdx.x >= 8 && threadIdx.x < 8 * p.split_s) {  // wave slots of the units of parts >= 1
      const int part = threadIdx.x >> 3, w = threadIdx.x & 7;
      p.sumsq[(int64_t)((vv & ~7) | (part * p.split_rp + x0)) * 8 + w] = 0.f;
    }
  }
}

// ---- grouped products on the same core (called by the pm_grouped_gemm_* entry points in grouped_gemm.hip) ----------------------------------------
static void grouped_common(Params& p, const int32_t* offsets, int n_experts, int64_t n_first) {
  p.bias = nullptr;
  p.out_f32 = 0;
  p.nt_out = 0;
  p.sweep_n = 0;
  p.accumulate = 0;
  p.sumsq = nullptr;
  p.split_s = 1; p.split_rp = 0; p.split_v0 = 0; p.split_k = 0; p.split_ws = nullptr;
  p.k_parts = 1; p.k_chunk = 0;
  p.offsets = offsets;
  p.n_experts = n_experts;
  p.n_first = (int)n_first;
  p.w2 = nullptr;
  p.out2 = nullptr;
  p.gu = nullptr; p.ldgu = 0;
}
bool grouped_shapes_ok(int64_t n_first, int64_t n, int64_t k_or_second) { return n_first % 256 == 0 && n % 256 == 0 && k_or_second % 128 == 0; }

// y[m] = x[m] . [w | w2][e(m)]^T : x [M, K], w [E, n_first, K], w2 [E, N - n_first, K] -> y [M, N]
int grouped_nt(const bf16_t* x, const bf16_t* w, const bf16_t* w2, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t n_first, int64_t n_out,
               int64_t k_dim, bf16_t* y, hipStream_t st, int sched) {
  Params p;
  grouped_common(p, offsets, n_experts, n_first);
  p.x = x; p.w = w; p.w2 = w2; p.out = y;
  p.ldx = k_dim; p.ldw = k_dim; p.ldo = n_out;
  p.m = (int)n_rows; p.n = (int)n_out; p.k = (int)k_dim;
  p.tiles_m = 0; p.tiles_n = (int)(n_out / kTile);
  p.n_blocks = (int)(cdiv(cdiv(n_rows, kTile) + n_experts, 8) * 8 * p.tiles_n);
  launch_grouped<false, false, 1>(p, st, eff_sched(sched));
  return PM_OK;
}
// dx[m] = dy[m, :n_first] . w[e] + dy[m, n_first:] . w2[e] : dy [M, n_dim], w [E, n_first, k_out], w2 [E, n_dim - n_first, k_out] -> dx [M, k_out]
int grouped_nn(const bf16_t* dy, const bf16_t* w, const bf16_t* w2, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t n_first, int64_t n_dim,
               int64_t k_out, bf16_t* dx, hipStream_t st, int sched) {
  Params p;
  grouped_common(p, offsets, n_experts, n_first);
  p.x = dy; p.w = w; p.w2 = w2; p.out = dx;
  p.ldx = n_dim; p.ldw = k_out; p.ldo = k_out;
  p.m = (int)n_rows; p.n = (int)k_out; p.k = (int)n_dim;
  p.tiles_m = 0; p.tiles_n = (int)(k_out / kTile);
  p.n_blocks = (int)(cdiv(cdiv(n_rows, kTile) + n_experts, 8) * 8 * p.tiles_n);
  launch_grouped<false, true, 2>(p, st, eff_sched(sched));
  return PM_OK;
}
// gate_up[m] = x[m] . [w | w2][e(m)]^T and act[m] = silu(gate) * up in ONE launch (MODE 7): x [M, K], w / w2 [E, inter, K], gate_up [M, 2 inter], act [M, inter]
int grouped_nt_swiglu(const bf16_t* x, const bf16_t* w, const bf16_t* w2, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t inter, int64_t k_dim,
                      bf16_t* gate_up, bf16_t* act, hipStream_t st, int sched) {
  Params p;
  grouped_common(p, offsets, n_experts, inter);
  p.x = x; p.w = w; p.w2 = w2; p.out = gate_up; p.out2 = act;
  p.ldx = k_dim; p.ldw = k_dim; p.ldo = 2 * inter; p.ldo2 = inter;
  p.m = (int)n_rows; p.n = (int)(2 * inter); p.k = (int)k_dim;
  p.tiles_m = 0; p.tiles_n = (int)(inter / 128);
  p.n_blocks = (int)(cdiv(cdiv(n_rows, kTile) + n_experts, 8) * 8 * p.tiles_n);
  launch_grouped<false, false, 7>(p, st, eff_sched(sched));
  return PM_OK;
}
// [d_gate | d_up][m] = SwiGLU'(gate_up[m]) applied to (dy[m] . w[e]) : dy [M, n_dim], w [E, n_dim, inter] (the experts' down projection), gate_up and
// d_gate_up [M, 2 inter] — grouped_nn through one weight set with the SwiGLU-backward epilogue (MODE 6); d_gate_up may be gate_up itself
int grouped_nn_swiglu_bwd(const bf16_t* dy, const bf16_t* w, const int32_t* offsets, int n_experts, int64_t n_rows, int64_t n_dim, int64_t inter,
                          const bf16_t* gate_up, bf16_t* d_gate_up, hipStream_t st, int sched) {
  Params p;
  grouped_common(p, offsets, n_experts, n_dim);
  p.x = dy; p.w = w; p.out = d_gate_up;
  p.gu = gate_up; p.ldgu = 2 * inter;
  p.ldx = n_dim; p.ldw = inter; p.ldo = 2 * inter;
  p.m = (int)n_rows; p.n = (int)inter; p.k = (int)n_dim;
  p.tiles_m = 0; p.tiles_n = (int)(inter / kTile);
  p.n_blocks = (int)(cdiv(cdiv(n_rows, kTile) + n_experts, 8) * 8 * p.tiles_n);
  launch_grouped<false, true, 6>(p, st, eff_sched(sched));
  return PM_OK;
}
// dw[e] = dy[rows of e]^T . x[rows of e] : dy [M, n_dim], x [M, k_dim] -> dw [E, n_first, k_dim], dw2 [E, n_dim - n_first, k_dim]
int grouped_tn(const bf16_t* dy, const bf16_t* x, const int32_t* offsets, int n_experts, int64_t n_first, int64_t n_dim, int64_t k_dim, void* dw, void* dw2,
               int out_f32, int accumulate, float* tile_sumsq, hipStream_t st, int sched) {
  Params p;
  grouped_common(p, offsets, n_experts, n_first);
  p.sumsq = out_f32 ? tile_sumsq : nullptr;  // [n_experts][tiles per expert][8]: sum of squares of the FINAL f32 values each wave stored (the gradient norm, fused)
  p.out_f32 = out_f32;
  p.accumulate = out_f32 ? accumulate : 0;
  p.x = dy; p.w = x; p.out = dw; p.out2 = dw2;
  p.ldx = n_dim; p.ldw = k_dim; p.ldo = k_dim;
  p.m = (int)n_dim; p.n = (int)k_dim; p.k = 0;
  p.tiles_m = (int)(n_dim / kTile); p.tiles_n = (int)(k_dim / kTile);
  // virtual ids: 8 XCD slots x the longest piece an XCD can own.  Up to 256 experts the tile line is cut by cost (build_wgrad_range) and a piece may hold
  // (nearly) all tiles; a workgroup stops at the first miss, so the surplus ids cost nothing.  More experts: expert e on XCD e % 8.
  p.n_blocks = (int)((n_experts <= 256 ? (int64_t)n_experts : cdiv(n_experts, 8)) * 8 * p.tiles_m * p.tiles_n);
  launch_grouped<true, true, 3>(p, st, eff_sched(sched));
  return PM_OK;
}

}  // namespace gm
}  // namespace pm

using namespace pm;

// Schedule word of the dense core — the `sched` argument of every launch (-1 = the library default, 7).  The product library has ONE K loop (kShippedLoop); the word selects around it: bit 2 = persistent grid (one workgroup
// per CU walking its tiles) instead of one workgroup per tile, bit 5 = no split-K of a half-empty last tile round, bit 10 = the round-3 patch sweep order (A/B); bits 0 / 1 are accepted and ignored (rounds 3-4 wrote 3 / 7 for "per tile" / "persistent").  Every other bit
// selects a build that was measured and not adopted and exists only in a `make VARIANTS=1` library — PM_ERR_UNSUPPORTED otherwise (the GPU tests skip).
extern "C" int pm_gemm_debug_set_variant(int32_t v) {
#if PM_VARIANTS
  gm::g_variant_override = v;  // >= 0: every launch of this process runs under this word, whatever the caller passes; -1: back to the per-call words
  return PM_OK;
#else
  PM_REQUIRE(v < 0, PM_ERR_UNSUPPORTED, "pm_gemm_debug_set_variant(%d): the product library holds no schedule state — pass the word as the `sched` argument of the "
             "launch (a process-wide override exists only in a `make VARIANTS=1` build)", (int)v);
  return PM_OK;
#endif
}

// K-contiguous operands need K % 128 == 0 (a K-tile past the end of a row would read the next row); two K-major operands take any K
extern "C" int pm_gemm_supported(int64_t m, int64_t n, int64_t k, int32_t x_kmajor, int32_t w_kmajor) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  if (x_kmajor && !w_kmajor) return 0;  // no caller has that pair
  if (!(x_kmajor && w_kmajor) && k % 128 != 0) return 0;
  if (x_kmajor && (m % 8 != 0 || m < 128)) return 0;  // 16-byte chunks along the rows of a K-major operand; its row stride (>= rows) covers the 1 KiB piece offset
  if (w_kmajor && (n % 8 != 0 || n < 128)) return 0;
  return 1;
}

extern "C" int64_t pm_gemm_sumsq_slots(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  return (int64_t)gm::grid_blocks((int)cdiv(m, gm::kTile), (int)cdiv(n, gm::kTile)) * 8;
}

static int gemm_impl(const void* x, int64_t ldx, int32_t x_kmajor, const void* w, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo, int32_t out_dtype,
                     const void* bias, int32_t accumulate, int64_t m, int64_t n, int64_t k, float* tile_sumsq, void* workspace, int64_t workspace_bytes,
                     int32_t sched, void* stream);

extern "C" int pm_gemm_bf16(const void* x, int64_t ldx, int32_t x_kmajor, const void* w, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo,
                            int32_t out_dtype, const void* bias, int32_t accumulate, int64_t m, int64_t n, int64_t k, int32_t sched, void* stream) {
  return gemm_impl(x, ldx, x_kmajor, w, ldw, w_kmajor, out, ldo, out_dtype, bias, accumulate, m, n, k, nullptr, nullptr, 0, sched, stream);
}

extern "C" int pm_gemm_bf16_sumsq(const void* x, int64_t ldx, int32_t x_kmajor, const void* w, int64_t ldw, int32_t w_kmajor, float* out, int64_t ldo,
                                  int32_t accumulate, int64_t m, int64_t n, int64_t k, float* tile_sumsq, int64_t tile_sumsq_len, int32_t sched, void* stream) {
  PM_REQUIRE(tile_sumsq && tile_sumsq_len >= pm_gemm_sumsq_slots(m, n), PM_ERR_WORKSPACE, "pm_gemm_bf16_sumsq: tile_sumsq needs pm_gemm_sumsq_slots(m, n) floats");
  return gemm_impl(x, ldx, x_kmajor, w, ldw, w_kmajor, out, ldo, PM_DTYPE_F32, nullptr, accumulate, m, n, k, tile_sumsq, nullptr, 0, sched, stream);
}

static int gemm_impl(const void* x, int64_t ldx, int32_t x_kmajor, const void* w, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo, int32_t out_dtype,
                     const void* bias, int32_t accumulate, int64_t m, int64_t n, int64_t k, float* tile_sumsq, void* workspace, int64_t workspace_bytes,
                     int32_t sched, void* stream) {
  if (m == 0 || n == 0) return PM_OK;
#if !PM_VARIANTS
  PM_REQUIRE(sched < 0 || (sched & ~(1 | 2 | 4 | 32 | 512 | 1024)) == 0, PM_ERR_UNSUPPORTED, "pm_gemm_bf16: schedule word %d selects a build that exists only in a `make VARIANTS=1` library",
             (int)sched);
#endif
  const int g_variant = gm::eff_sched(sched);
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
  PM_REQUIRE(k_chunk >= 128 || k_chunk == k, PM_ERR_UNSUPPORTED, "pm_gemm_bf16: a K-major operand with rows beyond 8M elements");
  PM_REQUIRE(k_chunk == k || (out_dtype == PM_DTYPE_F32 && x_kmajor && w_kmajor && !bias), PM_ERR_UNSUPPORTED,
             "pm_gemm_bf16: a K-major operand beyond 2 GB needs the f32-accumulating wgrad form");
  gm::Params p;
  p.offsets = nullptr; p.w2 = nullptr; p.out2 = nullptr; p.n_experts = 0; p.n_first = 0; p.gu = nullptr; p.ldgu = 0;
  p.out = out; p.bias = (const bf16_t*)bias;
  p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
  p.m = (int)m; p.n = (int)n;
  p.tiles_m = (int)cdiv(m, gm::kTile); p.tiles_n = (int)cdiv(n, gm::kTile);
  p.n_blocks = gm::grid_blocks(p.tiles_m, p.tiles_n);
  p.out_f32 = out_dtype == PM_DTYPE_F32;
  p.nt_out = (g_variant & 512) ? 1 : 0;
  // cycle the smaller operand past the held block of the bigger one (X has m x k elements, W n x k); bit 10 of the debug variant: always the round-3 order (A/B)
  p.sweep_n = (m > n && !(g_variant & 1024)) ? 1 : 0;
  hipStream_t st = as_stream(stream);
  // split-K of the last round (see Params): f32 results without bias, the whole contraction in one launch, a workspace from the caller
  p.split_s = 1; p.split_rp = 0; p.split_v0 = 0; p.split_k = 0; p.split_ws = nullptr;
  {
    const gm::SplitPlan sp = gm::plan_split(p.tiles_m, p.tiles_n, k);
    if (sp.s > 1 && !(g_variant & 32) && p.out_f32 && !bias && k_chunk == k && workspace && workspace_bytes >= sp.ws_bytes) {  // (bit 5 of the debug variant: no split)
      p.split_s = sp.s; p.split_rp = sp.rp; p.split_v0 = p.n_blocks - 256; p.split_k = sp.k_part; p.split_ws = (float*)workspace;
      if (accumulate) (void)hipMemsetAsync(workspace, 0, (size_t)sp.ws_bytes, st);  // parts >= 1 then add their product onto zeros
    }
  }
  // a contraction whose K-major span exceeds the 32-bit offsets runs as chained parts of every tile's accumulation in ONE launch (the accumulators stay
  // in registers across the parts; only the operand descriptors switch): no read-modify-write of the f32 result between K-chunks
  p.x = (const bf16_t*)x;
  p.w = (const bf16_t*)w;
  p.k = (int)k;
  p.k_chunk = (int)k_chunk;
  p.k_parts = (int)cdiv(k, k_chunk);
  p.accumulate = accumulate ? 1 : 0;
  p.sumsq = tile_sumsq;
  if (!x_kmajor && !w_kmajor) gm::launch<false, false>(p, st, g_variant);
  else if (!x_kmajor && w_kmajor) gm::launch<false, true>(p, st, g_variant);
  else gm::launch<true, true>(p, st, g_variant);
  PM_CHECK_LAUNCH("gemm_kernel");
  if (p.split_s > 1) {
    hipLaunchKernelGGL(gm::splitk_reduce_kernel, dim3((unsigned)(p.split_rp * 32 * 8)), dim3(256), 0, st, p);
    PM_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return PM_OK;
}

extern "C" int64_t pm_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  return gm::plan_split((int)cdiv(m, gm::kTile), (int)cdiv(n, gm::kTile), k).ws_bytes;
}

extern "C" int pm_gemm_bf16_ws(const void* x, int64_t ldx, int32_t x_kmajor, const void* w, int64_t ldw, int32_t w_kmajor, void* out, int64_t ldo, int32_t out_dtype,
                               const void* bias, int32_t accumulate, int64_t m, int64_t n, int64_t k, float* tile_sumsq, int64_t tile_sumsq_len, void* workspace,
                               int64_t workspace_bytes, int32_t sched, void* stream) {
  PM_REQUIRE(!tile_sumsq || (out_dtype == PM_DTYPE_F32 && !bias && tile_sumsq_len >= pm_gemm_sumsq_slots(m, n)), PM_ERR_WORKSPACE,
             "pm_gemm_bf16_ws: tile_sumsq needs an f32 output, no bias and pm_gemm_sumsq_slots(m, n) floats");
  return gemm_impl(x, ldx, x_kmajor, w, ldw, w_kmajor, out, ldo, out_dtype, bias, accumulate, m, n, k, tile_sumsq, workspace, workspace_bytes, sched, stream);
}

// ---- gate|up forward with the SwiGLU epilogue (MODE 4 of the tile stream) ------------------------------------------------------------------------------
extern "C" int pm_gemm_swiglu_supported(int64_t m, int64_t inter, int64_t k) {
  return m > 0 && inter > 0 && inter % 128 == 0 && k > 0 && k % 128 == 0 && 2 * inter < (1ll << 31) && m < (1ll << 31);
}

extern "C" int pm_gemm_swiglu_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, void* gate_up, int64_t ldo, void* act, int64_t ldo2, int64_t m, int64_t inter,
                                   int64_t k, int32_t sched, void* stream) {
  if (m == 0 || inter == 0) return PM_OK;
  const int g_variant = gm::eff_sched(sched);
  PM_REQUIRE(x && w && gate_up && act, PM_ERR_INVALID_ARG, "pm_gemm_swiglu_bf16: null pointer");
  PM_REQUIRE(pm_gemm_swiglu_supported(m, inter, k), PM_ERR_UNSUPPORTED, "pm_gemm_swiglu_bf16: needs inter %% 128 == 0 and k %% 128 == 0 (m %lld, inter %lld, k %lld)",
             (long long)m, (long long)inter, (long long)k);
  PM_REQUIRE(ldx >= k && ldw >= k && ldo >= 2 * inter && ldo2 >= inter, PM_ERR_INVALID_ARG, "pm_gemm_swiglu_bf16: leading dimension too small");
  PM_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0 && ldo2 % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)gate_up & 15) == 0 &&
                 ((uintptr_t)act & 15) == 0,
             PM_ERR_INVALID_ARG, "pm_gemm_swiglu_bf16: rows must start 16-byte aligned (leading dimensions multiples of 8, 16-byte aligned bases)");
  const int64_t kLimit = (1ll << 31) - 4096;
  PM_REQUIRE(256 * ldx * 2 < kLimit && (inter + 128) * ldw * 2 < kLimit, PM_ERR_UNSUPPORTED, "pm_gemm_swiglu_bf16: a tile's weight panel (inter + 128 rows) beyond int32 offsets");
  gm::Params p;
  p.offsets = nullptr; p.w2 = nullptr; p.n_experts = 0; p.bias = nullptr; p.sumsq = nullptr; p.gu = nullptr; p.ldgu = 0;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w;
  p.out = gate_up; p.out2 = act;
  p.ldx = ldx; p.ldw = ldw; p.ldo = ldo; p.ldo2 = ldo2;
  p.m = (int)m; p.n = (int)(2 * inter); p.k = (int)k; p.n_first = (int)inter;
  p.tiles_m = (int)cdiv(m, gm::kTile); p.tiles_n = (int)(inter / 128);
  p.n_blocks = gm::grid_blocks(p.tiles_m, p.tiles_n);
  p.out_f32 = 0; p.accumulate = 0;
  p.sweep_n = (m > 2 * inter && !(g_variant & 1024)) ? 1 : 0;
  p.split_s = 1; p.split_rp = 0; p.split_v0 = 0; p.split_k = 0; p.split_ws = nullptr;
  p.k_chunk = (int)k; p.k_parts = 1;
  gm::launch_grouped<false, false, 4>(p, as_stream(stream), g_variant);
  PM_CHECK_LAUNCH("gemm_kernel<swiglu>");
  return PM_OK;
}

// ---- down-projection row gradient with the SwiGLU-backward epilogue (MODE 5 of the tile stream) -------------------------------------------------------
// d_act = dy . W_down (W_down [k = hidden, inter] as stored, read K-major) never reaches HBM: each value, rounded to bf16 as the plain product would store
// it, meets gate / up of its (token, feature) in the epilogue and leaves as d_gate and d_up (pm_swiglu_bwd's arithmetic, bit for bit).
extern "C" int pm_gemm_swiglu_bwd_supported(int64_t m, int64_t inter, int64_t k) {
  return pm_gemm_supported(m, inter, k, 0, 1) && inter % 8 == 0 && 2 * inter < (1ll << 31);
}

extern "C" int pm_gemm_swiglu_bwd_bf16(const void* dy, int64_t lddy, const void* w_down, int64_t ldw, const void* gate_up, int64_t ldgu, void* grad_gate_up, int64_t ldo,
                                       int64_t m, int64_t inter, int64_t k, int32_t sched, void* stream) {
  if (m == 0 || inter == 0) return PM_OK;
  const int g_variant = gm::eff_sched(sched);
  PM_REQUIRE(dy && w_down && gate_up && grad_gate_up, PM_ERR_INVALID_ARG, "pm_gemm_swiglu_bwd_bf16: null pointer");
  PM_REQUIRE(pm_gemm_swiglu_bwd_supported(m, inter, k), PM_ERR_UNSUPPORTED, "pm_gemm_swiglu_bwd_bf16: unsupported shape (m %lld, inter %lld, k %lld)", (long long)m,
             (long long)inter, (long long)k);
  PM_REQUIRE(lddy >= k && ldw >= inter && ldgu >= 2 * inter && ldo >= 2 * inter, PM_ERR_INVALID_ARG, "pm_gemm_swiglu_bwd_bf16: leading dimension too small");
  PM_REQUIRE(lddy % 8 == 0 && ldw % 8 == 0 && ldgu % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)w_down & 15) == 0 &&
                 ((uintptr_t)gate_up & 15) == 0 && ((uintptr_t)grad_gate_up & 15) == 0,
             PM_ERR_INVALID_ARG, "pm_gemm_swiglu_bwd_bf16: rows must start 16-byte aligned (leading dimensions multiples of 8, 16-byte aligned bases)");
  const int64_t kLimit = (1ll << 31) - 4096;
  PM_REQUIRE(256 * lddy * 2 < kLimit && k * ldw * 2 < kLimit && 256 * ldgu * 2 < kLimit && 256 * ldo * 2 < kLimit && m < (1ll << 31), PM_ERR_UNSUPPORTED,
             "pm_gemm_swiglu_bwd_bf16: sizes beyond int32 offsets");
  gm::Params p;
  p.offsets = nullptr; p.w2 = nullptr; p.out2 = nullptr; p.n_experts = 0; p.n_first = 0; p.bias = nullptr; p.sumsq = nullptr;
  p.x = (const bf16_t*)dy; p.w = (const bf16_t*)w_down;
  p.out = grad_gate_up; p.gu = (const bf16_t*)gate_up;
  p.ldx = lddy; p.ldw = ldw; p.ldo = ldo; p.ldo2 = 0; p.ldgu = ldgu;
  p.m = (int)m; p.n = (int)inter; p.k = (int)k;
  p.tiles_m = (int)cdiv(m, gm::kTile); p.tiles_n = (int)cdiv(inter, gm::kTile);
  p.n_blocks = gm::grid_blocks(p.tiles_m, p.tiles_n);
  p.out_f32 = 0; p.accumulate = 0; p.nt_out = 0;
  p.sweep_n = (m > inter && !(g_variant & 1024)) ? 1 : 0;
  p.split_s = 1; p.split_rp = 0; p.split_v0 = 0; p.split_k = 0; p.split_ws = nullptr;
  p.k_chunk = (int)k; p.k_parts = 1;
  gm::launch_grouped<false, true, 5>(p, as_stream(stream), g_variant);
  PM_CHECK_LAUNCH("gemm_kernel<swiglu_bwd>");
  return PM_OK;
}

// ---- the fused LM head on the tile stream (MODE 8 / 9; entry points pm_lmhead_logprob_fwd / _bwd in lmhead.hip) ---------------------------------------
namespace pm {
namespace gm {
bool lmhead_tiles_supported(const void* hidden, const void* weight, int64_t n, int64_t h, int64_t v) {
  const int64_t kLimit = (1ll << 31) - 4096;
  return pm_gemm_supported(n, v, h, 0, 0) && h % 8 == 0 && ((uintptr_t)hidden & 15) == 0 && ((uintptr_t)weight & 15) == 0 && 256 * h * 2 < kLimit && n < (1ll << 31) &&
         v < (1ll << 31) - 256;
}
int64_t lmhead_tiles_partial_floats(int64_t n, int64_t v) { return n * 2 * cdiv(v, (int64_t)kTile) * 4; }

static void lmhead_params(Params& p, const bf16_t* hidden, const bf16_t* weight, int64_t n, int64_t h, int64_t v, const int64_t* labels, const float* inv_temp) {
  p.offsets = nullptr; p.w2 = nullptr; p.out2 = nullptr; p.n_experts = 0; p.n_first = 0; p.gu = nullptr; p.ldgu = 0; p.bias = nullptr; p.sumsq = nullptr;
  p.x = hidden; p.w = weight; p.out = nullptr;
  p.ldx = h; p.ldw = h; p.ldo = v; p.ldo2 = 0;
  p.m = (int)n; p.n = (int)v; p.k = (int)h;
  p.tiles_m = (int)cdiv(n, (int64_t)kTile); p.tiles_n = (int)cdiv(v, (int64_t)kTile);
  p.n_blocks = grid_blocks(p.tiles_m, p.tiles_n);
  p.out_f32 = 0; p.accumulate = 0; p.nt_out = 0;
  p.sweep_n = n > v ? 1 : 0;  // cycle the smaller operand past the held block of the bigger one (gemm_impl)
  p.split_s = 1; p.split_rp = 0; p.split_v0 = 0; p.split_k = 0; p.split_ws = nullptr;
  p.k_chunk = (int)h; p.k_parts = 1;
  p.lm_labels = labels; p.lm_inv_temp = inv_temp; p.lm_logz = nullptr; p.lm_grad = nullptr; p.lm_partial = nullptr;
}

void lmhead_fwd_tiles(const bf16_t* hidden, const bf16_t* weight, int64_t n, int64_t h, int64_t v, const int64_t* labels, const float* inv_temp, float* partial,
                      hipStream_t st) {
  Params p;
  lmhead_params(p, hidden, weight, n, h, v, labels, inv_temp);
  p.lm_partial = partial;
  launch_grouped<false, false, 8>(p, st, kDefaultSched);
}

void lmhead_bwd_tiles(const bf16_t* hidden, const bf16_t* weight, int64_t n, int64_t h, int64_t v, const int64_t* labels, const float* inv_temp, const float* logz,
                      const float* grad, bf16_t* dlogits, int64_t ld, hipStream_t st) {
  Params p;
  lmhead_params(p, hidden, weight, n, h, v, labels, inv_temp);
  p.out = dlogits; p.ldo = ld;
  p.lm_logz = logz; p.lm_grad = grad;
  launch_grouped<false, false, 9>(p, st, kDefaultSched);
}
}  // namespace gm
}  // namespace pm
