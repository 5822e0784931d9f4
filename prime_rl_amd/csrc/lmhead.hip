// lmhead.hip — J1: the fused LM head.  logits = hidden . W^T is computed tile by tile on the MFMA units and consumed in registers by the
// log-prob / entropy epilogue (forward) or turned into dlogits (backward); the [N, V] logits never exist in HBM.
//
// Reference: FusedOutputLinear.forward -> _SequenceChunkedLogProbEntropyFn (src/prime_rl/trainer/models/layers/lm_head.py:42-202):
//   forward  (:89-156)  per token chunk x vocab chunk: logits = hidden_chunk @ w_chunk.t() (bf16 GEMM, bf16 result), .float() * inv_t,
//                       online (m, s, t) update (:76-86), branch-free label pick (:140-143); logz = m + log s,
//                       logprob = target - logz, entropy = logz - t / s;
//   backward (:158-202) recompute the chunk's logits, p = exp(l - logz), dl = (g * inv_t) * (onehot(label) - p)   [then two GEMMs].
// The reference round-trips every logits chunk through HBM ~10 times; here one kernel does GEMM + epilogue.
//
// MI355X design
//   * TRANSPOSED product: the MFMA A operand is the W tile (rows = vocab), the B operand the hidden tile (rows = tokens), so the C
//     fragment of v_mfma_f32_16x16x32_bf16 gives a lane 4 CONSECUTIVE VOCAB entries of ONE token (row = (lane>>4)*4 + r <-> vocab,
//     col = lane&15 <-> token).  The softmax reduction over the vocab is then almost entirely per lane: a lane owns 4 token slots and
//     keeps its own online (m, s, t, target) state across ALL vocab tiles of its workgroup; lanes are merged (two shuffles) exactly
//     once, at the very end.  In backward the same layout turns into 8-byte dlogits stores (4 bf16 along the vocab).
//   * tile 128 (vocab) x 128 (tokens) x 64 (k), 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 accumulator fragments; LDS 2 x 32 KB
//     (double buffer) -> two workgroups per CU (two independent barrier domains).  Both operands are K-contiguous in global memory
//     ([V, H] and [N, H]): no transposed copy of the head weight is needed (the library path keeps an [H, V] copy for its dgrad).
//   * global -> LDS with 16-byte LDS-DMA (`global_load_lds_dwordx4`): no staging VGPRs, no ds_write pass.  The DMA writes lane-linear,
//     so the bank-conflict swizzle (16-byte chunk c of row r lives at chunk c ^ (r & 7)) is applied to the per-lane SOURCE address and
//     undone in the ds_read_b128 fragment reads (guide 5.4 rule 21).  With 128-byte tile rows that XOR makes every ds_read_b128 lane
//     group hit 16 distinct 16-byte slots of the 256-byte bank row (checked lane group by lane group: conflict-free).
//   * a workgroup owns one 128-token block and one vocab SUPER-CHUNK (1/S of the vocab) and walks its vocab tiles with the k-loop
//     pipelined across tile boundaries; grid = token blocks x S.  blockIdx is remapped so that the 64 workgroups resident on one XCD
//     form an 8 x 8 patch (8 token blocks x 8 super-chunks): they walk k in step and share every hidden / weight slice through that
//     XCD's L2 (8x less traffic beyond L2 than a row-major block order).
//   * per-token partial states (one per super-chunk and vocab half of the workgroup) go to a small workspace
//     (N x 2S x 4 floats) and a finalize kernel merges them: logz, logprob, entropy.
//
// Round 6: from 1024 tokens up (bf16 logits, hidden % 128 == 0, no raw-logits copy) both passes run on the GEMM core's tile stream instead — gemm.hip
// MODE 8 (forward: the tile's logits leave as partial softmax states per (token, vocab tile, wave group)) and MODE 9 (backward: the recomputed tile
// leaves as dlogits through the LDS-transposed row stores) — 8-wave ping-pong, LDS-DMA ring, persistent grid: 16 384 x 4096 x 151 936 forward 16.99 ->
// 14.31 ms (0.478 -> 0.57 of 2.5 PFLOP/s: the plain product WITH its 5 GB result store takes 14.34), backward 17.4 -> 14.78 ms.  lmhead_kernel below stays
// for small token counts, f32-logits semantics and the `logits_out` copy.
#include "common.hip.h"

namespace pm {

typedef __attribute__((ext_vector_type(8))) __bf16 lm_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float lm_f32x4;

namespace {

constexpr int BK = 64;           // k per stage
// Two tile configurations of the same kernel:
//   small  128 (vocab) x 128 (tokens), 4 waves 2 x 2, 64 x 64 accumulators per wave, LDS 64 KB, two workgroups per CU;
//   big    256 x 256, 8 waves 2 (vocab) x 4 (tokens), 128 x 64 accumulators per wave (128 VGPRs), LDS 128 KB, one workgroup per CU —
//          half the L2 -> LDS bytes per flop (a 128 x 128 tile streams 64 flop/byte: at the full MFMA rate that is a CU's whole 64 B/clk L2
//          port), 12 instead of 16 fragment reads per 32 MFMAs, 64 MFMAs per wave between barriers; used from 1024 tokens up.
template <bool BIG>
struct Cfg {
  static constexpr int BV = BIG ? 256 : 128;            // vocab rows per tile (MFMA A operand)
  static constexpr int BT = BIG ? 256 : 128;            // tokens per tile (MFMA B operand)
  static constexpr int NVI = BV / 32;                   // 16-row vocab fragments per wave (2 waves along the vocab)
  static constexpr int NWT = BT / 64;                   // waves along the tokens (64 tokens = 4 fragments each)
  static constexpr int kThreads = 2 * NWT * 64;
  static constexpr int kTileBytes = BV * BK * 2;        // bytes per operand per stage (BV == BT)
  static constexpr int kStageBytes = 2 * kTileBytes;    // A then B
  static constexpr int kSmemBytes = 2 * kStageBytes;    // double buffer
};
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -3.0e38f;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ lm_f32x4 mfma(const u32x4& a, const u32x4& b, lm_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lm_bf16x8_t, a), __builtin_bit_cast(lm_bf16x8_t, b), c, 0, 0, 0);
}

struct Args {
  const bf16_t* w;        // [V, H]
  const bf16_t* x;        // [N, H]
  const int64_t* labels;  // [N]
  const float* inv_temp;  // [N]
  const float* logz;      // [N]      (backward)
  const float* grad;      // [N]      (backward: dL/dlogprob)
  float* partial;         // [N][P][4] (forward)
  bf16_t* tile_out;       // [N, ld]  (forward: optional logits; backward: dlogits)
  int64_t ld;
  int n, h, v;
  int n_super;            // S
  int tiles_per_super;    // vocab tiles a workgroup walks
  int tb_count;           // token blocks
  int round_bf16;         // round the GEMM result to bf16 before the epilogue (the reference's bf16 logits)
};

// One stage = the [BV x 64] k-slice of the W tile and of the hidden tile as BV / 8 one-KiB DMA pieces per operand (4 per wave):
// piece p covers tile rows 8p..8p+7; lane L of the issuing wave lands at LDS byte p*1024 + L*16 = (row 8p + L/8, chunk L%8) and must
// therefore FETCH chunk (L%8) ^ (row & 7) of that row.  Wave w issues pieces 4w..4w+3 of each operand.
template <bool BIG>
__device__ __forceinline__ void stage_tiles(const Args& a, unsigned char* smem, int buf, int v0, int t0, int k0, int wave, int lane) {
  constexpr int kStageBytes = Cfg<BIG>::kStageBytes, kTileBytes = Cfg<BIG>::kTileBytes;
  const int sub = lane >> 3, ch = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave * 4 + i;
    const int row = p * 8 + sub;
    const int src_chunk = ch ^ (row & 7);
    int vr = v0 + row;
    vr = vr < a.v ? vr : a.v - 1;  // rows past the vocab are clamped (masked in the epilogue)
    int tr = t0 + row;
    tr = tr < a.n ? tr : a.n - 1;
    const bf16_t* ga = a.w + (int64_t)vr * a.h + k0 + src_chunk * 8;
    const bf16_t* gb = a.x + (int64_t)tr * a.h + k0 + src_chunk * 8;
    unsigned char* la = smem + buf * kStageBytes + p * 1024;  // wave-uniform base; the DMA adds lane * 16
    unsigned char* lb = la + kTileBytes;
    __builtin_amdgcn_global_load_lds((gbl_void*)ga, (lds_void*)la, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)gb, (lds_void*)lb, 16, 0, 0);
  }
}

// fragment (16 rows x 32 k) of a swizzled [128][64] bf16 tile: lane (lr, lq) reads row r0 + lr, 16-byte chunk kk*4 + lq
__device__ __forceinline__ u32x4 frag(const unsigned char* tile, int r0, int kk, int lr, int lq) {
  const int row = r0 + lr;
  return *reinterpret_cast<const u32x4*>(tile + row * 128 + (((kk * 4 + lq) ^ (row & 7)) << 4));
}

template <bool BWD, bool STORE, bool BIG>
__global__ __launch_bounds__(Cfg<BIG>::kThreads, BIG ? 1 : 2) void lmhead_kernel(const Args a) {
  constexpr int BV = Cfg<BIG>::BV, BT = Cfg<BIG>::BT, NVI = Cfg<BIG>::NVI, NWT = Cfg<BIG>::NWT;
  constexpr int kStageBytes = Cfg<BIG>::kStageBytes, kTileBytes = Cfg<BIG>::kTileBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- blockIdx -> (token block, super-chunk): 8 x 8 patches per XCD (block b runs on XCD b % 8; speed only, any mapping is correct)
  const int patches_t = (a.tb_count + 7) >> 3, patches_s = (a.n_super + 7) >> 3;
  const int b = blockIdx.x;
  const int xcd = b & 7, seq = b >> 3;
  const int patch = (seq >> 6) * 8 + xcd, within = seq & 63;
  if (patch >= patches_t * patches_s) return;
  const int tb = (patch % patches_t) * 8 + (within & 7);
  const int sc = (patch / patches_t) * 8 + (within >> 3);
  if (tb >= a.tb_count || sc >= a.n_super) return;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wv = wave / NWT, wt = wave % NWT;
  const int t0 = tb * BT;
  const int tile_first = sc * a.tiles_per_super;
  const int n_vtiles = (a.v + BV - 1) / BV;
  int tile_last = tile_first + a.tiles_per_super;
  tile_last = tile_last < n_vtiles ? tile_last : n_vtiles;
  if (tile_first >= tile_last) {  // an empty super-chunk still owes its (neutral) partials
    if (!BWD && lq == 0) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const int tok = t0 + wt * 64 + ti * 16 + lr;
        if (tok < a.n) {
          float* p = a.partial + ((int64_t)tok * (2 * a.n_super) + sc * 2 + wv) * 4;
          p[0] = kNegBig; p[1] = 0.f; p[2] = 0.f; p[3] = kNegBig;
        }
      }
    }
    return;
  }
  const int nk = a.h / BK;

  // per-lane token slots: token = t0 + wt*64 + ti*16 + lr
  float inv_t[4], st_m[4], st_s[4], st_t[4], st_tgt[4], lz[4], gsc[4];
  int label[4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    int tok = t0 + wt * 64 + ti * 16 + lr;
    tok = tok < a.n ? tok : a.n - 1;
    inv_t[ti] = a.inv_temp[tok];
    label[ti] = (int)a.labels[tok];
    st_m[ti] = kNegBig; st_s[ti] = 0.f; st_t[ti] = 0.f; st_tgt[ti] = kNegBig;
    if (BWD) {
      lz[ti] = a.logz[tok];
      gsc[ti] = a.grad[tok] * inv_t[ti];
    }
  }

  lm_f32x4 acc[NVI][4];
#pragma unroll
  for (int vi = 0; vi < NVI; ++vi)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) acc[vi][ti] = lm_f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- software pipeline over the flattened (tile, k) space: stage s+1 is in flight while stage s is multiplied
  const int total = (tile_last - tile_first) * nk;
  stage_tiles<BIG>(a, smem, 0, tile_first * BV, t0, 0, wave, lane);
  int tile = tile_first, k = 0;
  for (int s = 0; s < total; ++s) {
    const int buf = s & 1;
    // An LDS-DMA is invisible to hipcc's waitcnt bookkeeping (it emitted a bare `s_waitcnt lgkmcnt(0); s_barrier` here): wait for this
    // wave's own pieces of stage s by hand, THEN rendezvous — after the barrier every wave's pieces have landed and every wave has
    // finished reading the buffer stage s+1 is about to overwrite.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < total) {
      int nt = tile, nkk = k + 1;
      if (nkk == nk) { nkk = 0; ++nt; }
      stage_tiles<BIG>(a, smem, buf ^ 1, nt * BV, t0, nkk * BK, wave, lane);
    }
    const unsigned char* ta = smem + buf * kStageBytes;
    const unsigned char* tbp = ta + kTileBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[NVI], bfr[4];
#pragma unroll
      for (int i = 0; i < NVI; ++i) af[i] = frag(ta, wv * (NVI * 16) + i * 16, kk, lr, lq);
#pragma unroll
      for (int i = 0; i < 4; ++i) bfr[i] = frag(tbp, wt * 64 + i * 16, kk, lr, lq);
#pragma unroll
      for (int vi = 0; vi < NVI; ++vi)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[vi][ti] = mfma(af[vi], bfr[ti], acc[vi][ti]);
    }
    if (++k == nk) {
      // ---- epilogue of vocab tile `tile`: acc[vi][ti][r] = logits[token(ti)][vocab = v0 + wv*(NVI*16) + vi*16 + lq*4 + r]
      const int vbase = tile * BV + wv * (NVI * 16) + lq * 4;
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        float l[NVI * 4];
#pragma unroll
        for (int vi = 0; vi < NVI; ++vi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[vi][ti][r];
            if (a.round_bf16) x = round_bf16(x);
            l[vi * 4 + r] = x;
          }
        const int tok = t0 + wt * 64 + ti * 16 + lr;
        if (STORE && !BWD && tok < a.n) {  // optional raw-logits tile (bf16), 8 bytes per (token, 4 vocab entries)
#pragma unroll
          for (int vi = 0; vi < NVI; ++vi) {
            const int vc = vbase + vi * 16;
            bf16_t* dst = a.tile_out + (int64_t)tok * a.ld + vc;
            if (vc + 3 < a.v) {
              *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16(l[vi * 4], l[vi * 4 + 1]), pack_bf16(l[vi * 4 + 2], l[vi * 4 + 3])};
            } else {
              for (int r = 0; r < 4; ++r)
                if (vc + r < a.v) dst[r] = f32_to_bf16(l[vi * 4 + r]);
            }
          }
        }
        if (!BWD) {
          float mx = kNegBig;
#pragma unroll
          for (int vi = 0; vi < NVI; ++vi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int vc = vbase + vi * 16 + r;
              float x = l[vi * 4 + r] * inv_t[ti];
              x = vc < a.v ? x : kNegBig;
              l[vi * 4 + r] = x;
              mx = fmaxf(mx, x);
              st_tgt[ti] = (vc == label[ti]) ? x : st_tgt[ti];
            }
          const float m_new = fmaxf(st_m[ti], mx);
          const float sc_old = __builtin_amdgcn_exp2f((st_m[ti] - m_new) * kLog2e);
          float ssum = 0.f, tsum = 0.f;
#pragma unroll
          for (int j = 0; j < NVI * 4; ++j) {
            const float e = __builtin_amdgcn_exp2f((l[j] - m_new) * kLog2e);  // exp(-huge) = 0 for masked entries
            ssum += e;
            tsum = fmaf(e, l[j] > kNegBig ? l[j] : 0.f, tsum);
          }
          st_s[ti] = fmaf(st_s[ti], sc_old, ssum);
          st_t[ti] = fmaf(st_t[ti], sc_old, tsum);
          st_m[ti] = m_new;
        } else if (tok < a.n) {
          // dl = g * inv_t * (onehot(label) - exp(l * inv_t - logz))   (lm_head.py:184-191)
#pragma unroll
          for (int vi = 0; vi < NVI; ++vi) {
            const int vc = vbase + vi * 16;
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = __builtin_amdgcn_exp2f((l[vi * 4 + r] * inv_t[ti] - lz[ti]) * kLog2e);
              d[r] = gsc[ti] * (((vc + r) == label[ti] ? 1.f : 0.f) - p);
            }
            bf16_t* dst = a.tile_out + (int64_t)tok * a.ld + vc;
            if (vc + 3 < a.v) {
              *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3])};
            } else {
              for (int r = 0; r < 4; ++r)
                if (vc + r < a.v) dst[r] = f32_to_bf16(d[r]);
            }
          }
        }
      }
#pragma unroll
      for (int vi = 0; vi < NVI; ++vi)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[vi][ti] = lm_f32x4{0.f, 0.f, 0.f, 0.f};
      k = 0;
      ++tile;
    }
  }

  if (!BWD) {
    // merge the four lanes (lq = 0..3) that hold partial states of the same token, then one partial row per (token, super-chunk, wv)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      float m = st_m[ti], s = st_s[ti], tt = st_t[ti], tg = st_tgt[ti];
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64), t2 = __shfl_xor(tt, o, 64), g2 = __shfl_xor(tg, o, 64);
        const float mn = fmaxf(m, m2);
        const float e1 = __builtin_amdgcn_exp2f((m - mn) * kLog2e), e2 = __builtin_amdgcn_exp2f((m2 - mn) * kLog2e);
        s = s * e1 + s2 * e2;
        tt = tt * e1 + t2 * e2;
        m = mn;
        tg = fmaxf(tg, g2);
      }
      const int tok = t0 + wt * 64 + ti * 16 + lr;
      if (lq == 0 && tok < a.n) {
        float* p = a.partial + ((int64_t)tok * (2 * a.n_super) + sc * 2 + wv) * 4;
        *reinterpret_cast<lm_f32x4*>(p) = lm_f32x4{m, s, tt, tg};
      }
    }
  }
}

// one thread per token: merge the 2S partial states (lm_head.py:145-148 finalisation)
__global__ void lmhead_finalize_kernel(const float* __restrict__ partial, int n, int parts, float* __restrict__ logprobs, float* __restrict__ entropy,
                                       float* __restrict__ logz) {
  const int tok = blockIdx.x * blockDim.x + threadIdx.x;
  if (tok >= n) return;
  const lm_f32x4* p = reinterpret_cast<const lm_f32x4*>(partial) + (int64_t)tok * parts;
  float m = kNegBig, tg = kNegBig;
  for (int i = 0; i < parts; ++i) {
    m = fmaxf(m, p[i][0]);
    tg = fmaxf(tg, p[i][3]);
  }
  float s = 0.f, tt = 0.f;
  for (int i = 0; i < parts; ++i) {
    const float e = __builtin_amdgcn_exp2f((p[i][0] - m) * kLog2e);
    s = fmaf(p[i][1], e, s);
    tt = fmaf(p[i][2], e, tt);
  }
  const float lzv = m + __logf(s);
  logz[tok] = lzv;
  logprobs[tok] = tg - lzv;
  entropy[tok] = lzv - tt / s;
}

// merge of the tile stream's partial states (MODE 8 of gemm.hip): [parts][n] float4.  A workgroup = 64 tokens x 4 part streams (wave w merges parts w, w + 4,
// ... online, 1 KiB coalesced per row), then wave 0 merges the four streams in a fixed order through LDS.
__global__ __launch_bounds__(256) void lmhead_finalize_parts_kernel(const float* __restrict__ partial, int n, int parts, float* __restrict__ logprobs,
                                                                    float* __restrict__ entropy, float* __restrict__ logz) {
  __shared__ lm_f32x4 sm[4][64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int tok = blockIdx.x * 64 + lane;
  float m = kNegBig, tg = kNegBig, s = 0.f, tt = 0.f;
  if (tok < n) {
    const lm_f32x4* p = reinterpret_cast<const lm_f32x4*>(partial) + tok;
#pragma unroll 4
    for (int i = sub; i < parts; i += 4) {
      const lm_f32x4 v = p[(int64_t)i * n];
      const float mn = fmaxf(m, v[0]);
      const float e1 = __builtin_amdgcn_exp2f((m - mn) * kLog2e), e2 = __builtin_amdgcn_exp2f((v[0] - mn) * kLog2e);
      s = fmaf(s, e1, v[1] * e2);
      tt = fmaf(tt, e1, v[2] * e2);
      m = mn;
      tg = fmaxf(tg, v[3]);
    }
  }
  sm[sub][lane] = lm_f32x4{m, s, tt, tg};
  __syncthreads();
  if (sub != 0 || tok >= n) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const lm_f32x4 v = sm[w][lane];
    const float mn = fmaxf(m, v[0]);
    const float e1 = __builtin_amdgcn_exp2f((m - mn) * kLog2e), e2 = __builtin_amdgcn_exp2f((v[0] - mn) * kLog2e);
    s = fmaf(s, e1, v[1] * e2);
    tt = fmaf(tt, e1, v[2] * e2);
    m = mn;
    tg = fmaxf(tg, v[3]);
  }
  const float lzv = m + __logf(s);
  logz[tok] = lzv;
  logprobs[tok] = tg - lzv;
  entropy[tok] = lzv - tt / s;
}

bool use_big(int64_t n_tokens) { return n_tokens >= 1024; }  // 256 x 256 tiles once there are at least four 256-token blocks

int choose_super(int64_t n_tokens, int64_t vocab) {
  const int BT = use_big(n_tokens) ? 256 : 128, BV = BT;
  // ~2 workgroups per CU per round: token blocks x S close to a multiple of 512, S a multiple of 8 (8 x 8 XCD patches), capped so that
  // a super-chunk keeps >= 8 vocab tiles (the k-pipeline restarts only at workgroup start)
  const int64_t tb = cdiv(n_tokens, BT), vt = cdiv(vocab, BV);
  const int64_t want = use_big(n_tokens) ? 512 : 1024;  // workgroups: ~2 rounds of (one | two) per CU
  int s = 8;
  while (s < 64 && tb * s < want && vt / (s * 2) >= 8) s *= 2;
  return s;
}

int fill_args(Args& a, const void* hidden, const void* weight, int64_t n, int64_t h, int64_t v, const int64_t* labels, const float* inv_temp,
              int round_logits_bf16, const char* who) {
  PM_REQUIRE(hidden && weight && labels && inv_temp, PM_ERR_INVALID_ARG, "%s: null pointer", who);
  PM_REQUIRE(n > 0 && v > 0 && h > 0 && h % BK == 0, PM_ERR_UNSUPPORTED, "%s: hidden size must be a positive multiple of %d (got %lld)", who, BK, (long long)h);
  PM_REQUIRE(n < (1ll << 31) && v < (1ll << 31), PM_ERR_UNSUPPORTED, "%s: sizes beyond int32", who);
  a.w = (const bf16_t*)weight;
  a.x = (const bf16_t*)hidden;
  a.labels = labels;
  a.inv_temp = inv_temp;
  a.n = (int)n; a.h = (int)h; a.v = (int)v;
  const int tile = use_big(n) ? 256 : 128;
  a.n_super = choose_super(n, v);
  a.tiles_per_super = (int)cdiv(cdiv(v, tile), a.n_super);
  a.tb_count = (int)cdiv(n, tile);
  a.round_bf16 = round_logits_bf16 ? 1 : 0;
  a.logz = nullptr; a.grad = nullptr; a.partial = nullptr; a.tile_out = nullptr; a.ld = 0;
  return PM_OK;
}

int grid_blocks(const Args& a) {
  const int patches = ((a.tb_count + 7) / 8) * ((a.n_super + 7) / 8);
  return ((patches + 7) / 8) * 8 * 64;  // every XCD walks ceil(patches / 8) patches of 64 workgroups
}

}  // namespace

// The same two passes on the GEMM core's tile stream (gemm.hip MODE 8 / 9: 8-wave ping-pong, LDS-DMA ring, persistent grid): from 1024 tokens up, with the
// reference's bf16 logits (`round_logits_bf16`), no raw-logits copy requested.  lmhead_kernel stays for small token counts, f32-logits semantics and `logits_out`.
namespace gm {
bool lmhead_tiles_supported(const void* hidden, const void* weight, int64_t n, int64_t h, int64_t v);
int64_t lmhead_tiles_partial_floats(int64_t n, int64_t v);
void lmhead_fwd_tiles(const bf16_t* hidden, const bf16_t* weight, int64_t n, int64_t h, int64_t v, const int64_t* labels, const float* inv_temp, float* partial,
                      hipStream_t st);
void lmhead_bwd_tiles(const bf16_t* hidden, const bf16_t* weight, int64_t n, int64_t h, int64_t v, const int64_t* labels, const float* inv_temp, const float* logz,
                      const float* grad, bf16_t* dlogits, int64_t ld, hipStream_t st);
}  // namespace gm
namespace {
bool on_tile_stream(const void* hidden, const void* weight, int64_t n, int64_t h, int64_t v, int round_logits_bf16) {
  return use_big(n) && round_logits_bf16 && gm::lmhead_tiles_supported(hidden, weight, n, h, v);
}
}  // namespace
}  // namespace pm

using namespace pm;

extern "C" int64_t pm_lmhead_workspace_bytes(int64_t n_tokens, int64_t vocab) {
  if (n_tokens <= 0 || vocab <= 0) return 256;
  const int64_t own = n_tokens * 2 * choose_super(n_tokens, vocab) * 4 * (int64_t)sizeof(float);
  const int64_t tiles = use_big(n_tokens) ? gm::lmhead_tiles_partial_floats(n_tokens, vocab) * (int64_t)sizeof(float) : 0;  // [2 vocab tiles][tokens] float4
  return own > tiles ? own : tiles;
}

extern "C" int pm_lmhead_logprob_fwd(const void* hidden_bf16, const void* weight_bf16, int64_t n_tokens, int64_t hidden, int64_t vocab,
                                     const int64_t* labels, const float* inv_temp, int32_t round_logits_bf16, float* logprobs, float* entropy,
                                     float* logz, void* logits_out_bf16, int64_t logits_ld, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n_tokens == 0) return PM_OK;
  Args a;
  int rc = fill_args(a, hidden_bf16, weight_bf16, n_tokens, hidden, vocab, labels, inv_temp, round_logits_bf16, "pm_lmhead_logprob_fwd");
  if (rc != PM_OK) return rc;
  PM_REQUIRE(logprobs && entropy && logz && workspace, PM_ERR_INVALID_ARG, "pm_lmhead_logprob_fwd: null output");
  PM_REQUIRE(workspace_bytes >= pm_lmhead_workspace_bytes(n_tokens, vocab), PM_ERR_WORKSPACE, "pm_lmhead_logprob_fwd: workspace too small");
  PM_REQUIRE(!logits_out_bf16 || logits_ld >= vocab, PM_ERR_INVALID_ARG, "pm_lmhead_logprob_fwd: logits_ld < vocab");
  a.partial = (float*)workspace;
  a.tile_out = (bf16_t*)logits_out_bf16;
  a.ld = logits_ld;
  hipStream_t st = as_stream(stream);
  if (!logits_out_bf16 && on_tile_stream(hidden_bf16, weight_bf16, n_tokens, hidden, vocab, round_logits_bf16)) {
    gm::lmhead_fwd_tiles(a.x, a.w, n_tokens, hidden, vocab, labels, inv_temp, a.partial, st);
    PM_CHECK_LAUNCH("gemm_kernel<lmhead fwd>");
    hipLaunchKernelGGL(lmhead_finalize_parts_kernel, dim3((unsigned)cdiv(n_tokens, 64)), dim3(256), 0, st, a.partial, a.n, (int)(2 * cdiv(vocab, 256)), logprobs, entropy,
                       logz);
    PM_CHECK_LAUNCH("lmhead_finalize_parts_kernel");
    return PM_OK;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)lmhead_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<false>::kSmemBytes);
    (void)hipFuncSetAttribute((const void*)lmhead_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<false>::kSmemBytes);
    (void)hipFuncSetAttribute((const void*)lmhead_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<true>::kSmemBytes);
    (void)hipFuncSetAttribute((const void*)lmhead_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<true>::kSmemBytes);
    attr_done = true;
  }
  const int grid = grid_blocks(a);
  if (use_big(n_tokens)) {
    if (logits_out_bf16) hipLaunchKernelGGL((lmhead_kernel<false, true, true>), dim3(grid), dim3(Cfg<true>::kThreads), Cfg<true>::kSmemBytes, st, a);
    else hipLaunchKernelGGL((lmhead_kernel<false, false, true>), dim3(grid), dim3(Cfg<true>::kThreads), Cfg<true>::kSmemBytes, st, a);
  } else {
    if (logits_out_bf16) hipLaunchKernelGGL((lmhead_kernel<false, true, false>), dim3(grid), dim3(Cfg<false>::kThreads), Cfg<false>::kSmemBytes, st, a);
    else hipLaunchKernelGGL((lmhead_kernel<false, false, false>), dim3(grid), dim3(Cfg<false>::kThreads), Cfg<false>::kSmemBytes, st, a);
  }
  PM_CHECK_LAUNCH("lmhead_kernel<fwd>");
  hipLaunchKernelGGL(lmhead_finalize_kernel, dim3((unsigned)cdiv(n_tokens, 256)), dim3(256), 0, st, a.partial, a.n, 2 * a.n_super, logprobs, entropy, logz);
  PM_CHECK_LAUNCH("lmhead_finalize_kernel");
  return PM_OK;
}

extern "C" int pm_lmhead_logprob_bwd(const void* hidden_bf16, const void* weight_bf16, int64_t n_tokens, int64_t hidden, int64_t vocab,
                                     const int64_t* labels, const float* inv_temp, const float* logz, const float* grad_logprobs,
                                     int32_t round_logits_bf16, void* dlogits_bf16, int64_t dlogits_ld, void* stream) {
  if (n_tokens == 0) return PM_OK;
  Args a;
  int rc = fill_args(a, hidden_bf16, weight_bf16, n_tokens, hidden, vocab, labels, inv_temp, round_logits_bf16, "pm_lmhead_logprob_bwd");
  if (rc != PM_OK) return rc;
  PM_REQUIRE(logz && grad_logprobs && dlogits_bf16 && dlogits_ld >= vocab, PM_ERR_INVALID_ARG, "pm_lmhead_logprob_bwd: bad arguments");
  PM_REQUIRE(dlogits_ld % 4 == 0, PM_ERR_INVALID_ARG, "pm_lmhead_logprob_bwd: dlogits_ld must be a multiple of 4 (8-byte stores)");
  a.logz = logz;
  a.grad = grad_logprobs;
  a.tile_out = (bf16_t*)dlogits_bf16;
  a.ld = dlogits_ld;
  if (vocab % 8 == 0 && dlogits_ld % 8 == 0 && ((uintptr_t)dlogits_bf16 & 15) == 0 && 256 * dlogits_ld * 2 < (1ll << 31) - 4096 &&
      on_tile_stream(hidden_bf16, weight_bf16, n_tokens, hidden, vocab, round_logits_bf16)) {
    gm::lmhead_bwd_tiles(a.x, a.w, n_tokens, hidden, vocab, labels, inv_temp, logz, grad_logprobs, a.tile_out, dlogits_ld, as_stream(stream));
    PM_CHECK_LAUNCH("gemm_kernel<lmhead bwd>");
    return PM_OK;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)lmhead_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<false>::kSmemBytes);
    (void)hipFuncSetAttribute((const void*)lmhead_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<true>::kSmemBytes);
    attr_done = true;
  }
  if (use_big(n_tokens))
    hipLaunchKernelGGL((lmhead_kernel<true, true, true>), dim3(grid_blocks(a)), dim3(Cfg<true>::kThreads), Cfg<true>::kSmemBytes, as_stream(stream), a);
  else
    hipLaunchKernelGGL((lmhead_kernel<true, true, false>), dim3(grid_blocks(a)), dim3(Cfg<false>::kThreads), Cfg<false>::kSmemBytes, as_stream(stream), a);
  PM_CHECK_LAUNCH("lmhead_kernel<bwd>");
  return PM_OK;
}
