// attention.hip — varlen causal attention with GQA over packed documents (forward + deterministic backward), MFMA bf16.
//
// Reference: FlashAttention._compute_attention (src/prime_rl/trainer/models/layers/attn.py:94-110), i.e.
// flash_attn_varlen_func(q, k, v, cu_seqlens, cu_seqlens, max_seqlen, max_seqlen, causal=True) with GQA and
// scale head_dim^-0.5 (attn.py:60).  flash-attn itself is a third-party dependency (flash-attn 2.8.3, uv.lock:1737);
// the arithmetic restated here is its published algorithm: S = scale*Q K^T masked to (same document, j <= i),
// P = softmax_row(S), O = P V; backward dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dO o O)), dQ = scale dS K,
// dK = scale dS^T Q.
//
// MI355X design (this file: shared helpers, layout probes, delta kernel, host dispatch; kernels in attention_kernels.hip.h; earlier
// generations are in the git history only):
//   * tiles live on the GLOBAL token grid of the packed row: row i may attend to j in [seq_start[i], i].  No per-document
//     tile scheduling on the host, no max_seqlen, no host sync: grid = ceil(N/BM) * heads.
//   * v_mfma_f32_16x16x32_bf16; every A/B fragment is 16 bytes along the contraction dim (lane l: row l&15,
//     k = (l>>4)*8..+8), C layout row = (l>>4)*4 + r, col = l&15 (guide §3, verified by pm_mfma_probe_16x16x32).
//   * TRANSPOSED products (S^T = K.Q^T, ...): probabilities stay in registers, the C layout is the B-fragment layout up to a
//     permutation of the contraction index (attention_kernels.hip.h).
//   * operands whose contraction dim is the token axis (V in PV; Q, dO in dK/dV; K in dQ) are read out of the row-major LDS
//     tile with ds_read_b64_tr_b16 (pinned by pm_ds_read_tr16_probe): no transposed copies in global memory, no pre-pass.
//   * blockIdx -> (kv head fastest): the workgroups that share K/V land on the same XCD (block b runs on XCD b%8),
//     q tiles are issued heaviest-first (causal imbalance).
//   * backward is two kernels (dK/dV per kv tile, dQ per q tile): no atomics, bitwise reproducible.
#include "common.hip.h"
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
// 16-byte LDS accessors
__device__ __forceinline__ u32x4 lds16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void sts16(bf16_t* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

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
// Row statistics of the backward pass, [hq][n] each (16 lanes per (token, head)):
//   ndelta[h][i] = -sum_d dO[i,h,d] * O[i,h,d]      nlse[h][i] = -lse[h][i] / scale
// negated (and the lse pre-divided by the softmax scale) because the dK/dV and dQ kernels START their dP and S accumulators from these
// rows instead of zero: P = exp2((S - lse / scale) * scale * log2 e), dS = P * (dP - delta) * scale.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, const float* __restrict__ lse,
                                                        int64_t n, int hq, float inv_scale, float* __restrict__ ndelta, float* __restrict__ nlse) {
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
    const int64_t o = (int64_t)h * n + tok;
    ndelta[o] = -s;
    nlse[o] = -lse[o] * inv_scale;
  }
}

#include "attention_kernels.hip.h"
#include "attention_fwd_p3.hip.h"
#include "attention_dkdv_p2.hip.h"
#if PM_VARIANTS  // `make VARIANTS=1`: the measured-and-not-adopted builds (profiles/r03_attention.md, r04_attention.md)
#include "attention_fwd_pp.hip.h"
#endif

static inline int64_t align256(int64_t b) { return cdiv(b, 256) * 256; }
// Debug-only variant codes (pm_attn_debug_set_variant; tools/microbench.py attnbwd / attndq time them interleaved in ONE process because
// boxes differ by ~5 %).  0 = the default of each kernel.  dK/dV: 0 = attn_bwd_dkdv_p2_kernel (round 6); 7 = the round-2..5 kernel attn_bwd_dkdv_kernel
// <2, 4, 64, 1>, 1 = its fragment batch 2, 2 = fragment batch 8, 3 = 32-row q tiles at three workgroups per CU, 4-6 = its 32-kv-rows-per-wave builds with
// compiler scheduling, 8-11 = its timing ablations; 12 + 32 V = timing ablations of the default.  dQ: 1 = 32 kv rows per barrier.  Measured alternatives and
// the generations they replaced: profiles/r02_attention.md, r04_attention.md, r06_attention.md.
#if PM_VARIANTS  // the A/B selectors are process-wide state of `make VARIANTS=1` libraries only; the product library holds constants (no mutable globals)
static int g_fwd_var = -1;  // forward: A/B bits of fwd_tile (attention_kernels.hip.h); -1 = the default build
static int g_dkdv_var = 0;
static int g_dq_var = 0;
#else
constexpr int g_fwd_var = -1, g_dkdv_var = 0, g_dq_var = 0;
#endif
}  // namespace pm

using namespace pm;

// Product library: only the shipped builds (value <= 0).  Every other value selects an A/B build of a `make VARIANTS=1` library.
extern "C" int pm_attn_debug_set_variant(int32_t kernel, int32_t value) {
  PM_REQUIRE(kernel >= 0 && kernel <= 2, PM_ERR_INVALID_ARG, "pm_attn_debug_set_variant: kernel %d", (int)kernel);
#if !PM_VARIANTS
  // no state to set: the default of each kernel (value <= 0) is accepted as a no-op
  PM_REQUIRE(value <= 0, PM_ERR_UNSUPPORTED, "pm_attn_debug_set_variant(%d, %d): this build exists only in a `make VARIANTS=1` library",
             (int)kernel, (int)value);
  return PM_OK;
#else
  if (kernel == 0) {
    if (value > 15 && value < 64 && value != 47) return PM_ERR_INVALID_ARG;
    // -1: the default (= 129).  0-15: option bits of the round-1..5 kernel attn_fwd_kernel (15 = its shipped form); 47 = 15 + 32: eight waves x 16 q rows;
    // 64 + OPT: the 8-wave ping-pong kernel (attention_fwd_pp.hip.h); 128 / 129: the three-stage pipelined forward (attention_fwd_p3.hip.h) with
    // compiler-scheduled stages / with the hand-placed tile loop; 130-140: A/B builds and timing ablations of 129 (tools/gen_attn_p3_asm.py VARIANTS)
    g_fwd_var = value;
    return PM_OK;
  }
  if (kernel == 1) g_dkdv_var = value < 0 ? 0 : value;
  else g_dq_var = value < 0 ? 0 : value;
  return PM_OK;
#endif
}

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

// The kernels address K / V / Q / dO through buffer descriptors with 32-bit byte offsets.
static bool fits_buffer_range(int64_t n, int64_t row_elems) { return n * row_elems * 2 < (int64_t)1 << 31; }

// Workspace: none in forward (the entry point keeps the argument for ABI stability); backward: the two row-statistics arrays of
// attn_delta_kernel, [hq][n] f32 each.
extern "C" int64_t pm_attn_fwd_workspace_bytes(int64_t n_tokens, int32_t n_kv_heads, int32_t head_dim) {
  (void)n_kv_heads; (void)head_dim;
  return n_tokens <= 0 ? 0 : 256;
}

// One implementation for the ordinary call (q_row0 = 0, n_q = n_tokens) and the row-window call: k, v, seq_start hold rows [0, q_row0 + n_q),
// q / out / lse the n_q rows from q_row0 on.
static int attn_fwd_impl(const char* who, const void* q, const void* k, const void* v, int64_t v_row_stride, const int32_t* seq_start, int64_t q_row0,
                         int64_t n_q, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale, void* out, float* lse, void* stream) {
  PM_REQUIRE(q && k && v && seq_start && out && lse, PM_ERR_INVALID_ARG, "%s: null pointer", who);
  PM_REQUIRE(head_dim == kD, PM_ERR_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", who, head_dim);
  PM_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, PM_ERR_INVALID_ARG, "%s: bad head counts", who);
  const int64_t n = q_row0 + n_q;
  PM_REQUIRE(n_q >= 0 && q_row0 >= 0 && n < (1ll << 31) - 256 && v_row_stride >= (int64_t)n_kv_heads * kD && v_row_stride % 8 == 0, PM_ERR_INVALID_ARG,
             "%s: bad sizes", who);
  PM_REQUIRE(q_row0 % 128 == 0, PM_ERR_UNSUPPORTED, "%s: q_row0=%lld must be a multiple of 128 (q tiles must not straddle the window start)", who,
             (long long)q_row0);
  if (n_q == 0) return PM_OK;
  PM_REQUIRE(fits_buffer_range(n, v_row_stride) && fits_buffer_range(n, (int64_t)n_kv_heads * kD), PM_ERR_UNSUPPORTED,
             "%s: k / v exceed the 2 GiB buffer-offset range", who);
#define PM_LAUNCH_FWD(OPT_)                                                                                                                  \
  hipLaunchKernelGGL((attn_fwd_kernel<2, OPT_>), dim3((unsigned)(cdiv(n_q, 128) * n_q_heads)), dim3(256), 0, as_stream(stream), (const bf16_t*)q, \
                     (const bf16_t*)k, (const bf16_t*)v, v_row_stride, seq_start, n, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse,  \
                     (int)q_row0)
#if PM_VARIANTS
  if (g_fwd_var >= 128 && g_fwd_var <= 140) {
#define PM_LAUNCH_P3(KERNEL_)                                                                                                                      \
  hipLaunchKernelGGL((KERNEL_), dim3((unsigned)(cdiv(n_q, 128) * n_q_heads)), dim3(256), 0, as_stream(stream), (const bf16_t*)q, (const bf16_t*)k, \
                     (const bf16_t*)v, v_row_stride, seq_start, n, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse, (int)q_row0)
    switch (g_fwd_var) {  // 129 + V: p3_loop_asm<V> (tools/gen_attn_p3_asm.py VARIANTS: 0 product, 1-4 placement A/B, 5-8 timing ablations with wrong results)
      case 128: PM_LAUNCH_P3(attn_fwd_p3_kernel<2>); break;  // compiler-scheduled stages
      case 129: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 0>)); break;
      case 130: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 1>)); break;
      case 131: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 2>)); break;
      case 132: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 3>)); break;
      case 133: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 4>)); break;
      case 134: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 5>)); break;
      case 135: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 6>)); break;
      case 136: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 7>)); break;
      case 137: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 8>)); break;
      case 138: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 9>)); break;
      case 139: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 10>)); break;
      default: PM_LAUNCH_P3((attn_fwd_p3a_kernel<2, 11>)); break;
    }
#undef PM_LAUNCH_P3
    PM_CHECK_LAUNCH(who);
    return PM_OK;
  }
  if (g_fwd_var >= 64) {
    const int group = n_q_heads / n_kv_heads;
    const int by_rows = (group & 1) ? 1 : 0;
    const unsigned grid = by_rows ? (unsigned)(cdiv(n_q, 256) * n_q_heads) : (unsigned)(cdiv(n_q, 128) * n_kv_heads * (group / 2));
#define PM_LAUNCH_PP(OPT_)                                                                                                                             \
  case OPT_: {                                                                                                                                         \
    static bool attr_done = false;                                                                                                                     \
    if (!attr_done) {                                                                                                                                  \
      (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<OPT_>, hipFuncAttributeMaxDynamicSharedMemorySize, kPPSmem);                           \
      attr_done = true;                                                                                                                                \
    }                                                                                                                                                  \
    hipLaunchKernelGGL((attn_fwd_pp_kernel<OPT_>), dim3(grid), dim3(512), kPPSmem, as_stream(stream), (const bf16_t*)q, (const bf16_t*)k,             \
                       (const bf16_t*)v, v_row_stride, seq_start, n, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse, (int)q_row0, by_rows);   \
  } break
    switch (g_fwd_var - 64) {
      PM_LAUNCH_PP(0);
      PM_LAUNCH_PP(1);
      PM_LAUNCH_PP(32);
      PM_LAUNCH_PP(33);
      PM_LAUNCH_PP(36);
      PM_LAUNCH_PP(40);
      PM_LAUNCH_PP(44);
      PM_LAUNCH_PP(48);
      PM_LAUNCH_PP(96);
      PM_LAUNCH_PP(112);
      PM_LAUNCH_PP(124);
      default: PM_REQUIRE(false, PM_ERR_INVALID_ARG, "%s: ping-pong forward variant %d is not built", who, g_fwd_var - 64);
    }
#undef PM_LAUNCH_PP
    PM_CHECK_LAUNCH(who);
    return PM_OK;
  }
#endif
#define PM_LAUNCH_FWD_P3A()                                                                                                                           \
  hipLaunchKernelGGL((attn_fwd_p3a_kernel<2, 0>), dim3((unsigned)(cdiv(n_q, 128) * n_q_heads)), dim3(256), 0, as_stream(stream), (const bf16_t*)q,    \
                     (const bf16_t*)k, (const bf16_t*)v, v_row_stride, seq_start, n, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse, (int)q_row0)
#if PM_VARIANTS
  if (g_fwd_var < 0) {  // the default of every build: the three-stage pipelined forward with the hand-placed tile loop (attention_fwd_p3.hip.h)
    PM_LAUNCH_FWD_P3A();
    PM_CHECK_LAUNCH(who);
    return PM_OK;
  }
  switch (g_fwd_var) {
    case 0: PM_LAUNCH_FWD(0); break;
    case 1: PM_LAUNCH_FWD(1); break;
    case 2: PM_LAUNCH_FWD(2); break;
    case 3: PM_LAUNCH_FWD(3); break;
    case 7: PM_LAUNCH_FWD(7); break;
    case 47:  // eight waves x 16 q rows per workgroup, four waves per SIMD (A/B: twice the K / V fragment reads per MFMA, twice the waves to hide latency)
      hipLaunchKernelGGL((attn_fwd_kernel<2, 15, 1, 8>), dim3((unsigned)(cdiv(n_q, 128) * n_q_heads)), dim3(512), 0, as_stream(stream), (const bf16_t*)q,
                         (const bf16_t*)k, (const bf16_t*)v, v_row_stride, seq_start, n, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse, (int)q_row0);
      break;
    default: PM_LAUNCH_FWD(15); break;
  }
#else
  PM_LAUNCH_FWD_P3A();
#endif
#undef PM_LAUNCH_FWD_P3A
#undef PM_LAUNCH_FWD
  PM_CHECK_LAUNCH(who);
  return PM_OK;
}

extern "C" int pm_attn_varlen_fwd(const void* q, const void* k, const void* v, int64_t v_row_stride, const int32_t* seq_start, int64_t n_tokens,
                                  int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale, void* out, float* lse,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  return attn_fwd_impl("pm_attn_varlen_fwd", q, k, v, v_row_stride, seq_start, 0, n_tokens, n_q_heads, n_kv_heads, head_dim, softmax_scale, out, lse, stream);
}

extern "C" int pm_attn_varlen_fwd_rows(const void* q, const void* k, const void* v, int64_t v_row_stride, const int32_t* seq_start, int64_t q_row0,
                                       int64_t n_q, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale, void* out, float* lse,
                                       void* stream) {
  return attn_fwd_impl("pm_attn_varlen_fwd_rows", q, k, v, v_row_stride, seq_start, q_row0, n_q, n_q_heads, n_kv_heads, head_dim, softmax_scale, out, lse,
                       stream);
}

extern "C" int64_t pm_attn_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim) {
  if (n_tokens <= 0) return 0;
  (void)n_kv_heads; (void)head_dim;
  return 2 * align256((int64_t)n_q_heads * n_tokens * 4);
}

static int attn_bwd_impl(const char* who, const void* grad_out, const void* q, const void* k, const void* v, int64_t v_row_stride, const void* out,
                         const float* lse, const int32_t* seq_start, int64_t q_row0, int64_t n_q, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                         float softmax_scale, void* grad_q, void* grad_k, void* grad_v, int64_t grad_v_row_stride, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(grad_out && q && k && v && out && lse && seq_start && grad_q && grad_k && grad_v, PM_ERR_INVALID_ARG, "%s: null pointer", who);
  PM_REQUIRE(head_dim == kD, PM_ERR_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", who, head_dim);
  PM_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, PM_ERR_INVALID_ARG, "%s: bad head counts", who);
  PM_REQUIRE(n_q >= 0 && q_row0 >= 0 && v_row_stride % 8 == 0 && grad_v_row_stride >= (int64_t)n_kv_heads * kD && softmax_scale > 0.f, PM_ERR_INVALID_ARG,
             "%s: bad sizes", who);
  PM_REQUIRE(q_row0 % 128 == 0, PM_ERR_UNSUPPORTED, "%s: q_row0=%lld must be a multiple of 128", who, (long long)q_row0);
  if (n_q == 0) return PM_OK;
  const int64_t need = pm_attn_bwd_workspace_bytes(n_q, n_q_heads, n_kv_heads, head_dim);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "%s: workspace %lld < %lld", who, (long long)workspace_bytes, (long long)need);
  const int64_t n = q_row0 + n_q;
  PM_REQUIRE(n < (1ll << 31) - 256 && fits_buffer_range(n, (int64_t)n_q_heads * kD) && fits_buffer_range(n, v_row_stride), PM_ERR_UNSUPPORTED,
             "%s: q / k / v exceed the 2 GiB buffer-offset range", who);
  hipStream_t st = as_stream(stream);
  float* ndelta = (float*)workspace;
  float* nlse = ndelta + align256(n_q * n_q_heads * 4) / 4;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv(n_q * n_q_heads, 16)), dim3(256), 0, st, (const bf16_t*)grad_out, (const bf16_t*)out, lse, n_q,
                     n_q_heads, 1.0f / softmax_scale, ndelta, nlse);
  PM_CHECK_LAUNCH(who);

#define PM_LAUNCH_DKDV(MINB_, FB_, QT_, KVB_, ...)                                                                                                \
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<MINB_, FB_, QT_, KVB_, ##__VA_ARGS__>), dim3((unsigned)(cdiv(n, 64 * KVB_) * n_kv_heads)), dim3(256), 0, st, (const bf16_t*)q, \
                     (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, nlse, ndelta, seq_start, n, n_q_heads, n_kv_heads, \
                     softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride, (int)q_row0, (g_dkdv_var & 16) ? 1 : 0)
#define PM_LAUNCH_DKDV_P2(V_)                                                                                                                        \
  do {                                                                                                                                               \
    static bool attr_done = false;                                                                                                                   \
    if (!attr_done) {                                                                                                                                \
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_p2_kernel<V_>, hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvP2Smem);                       \
      attr_done = true;                                                                                                                              \
    }                                                                                                                                                \
    hipLaunchKernelGGL(attn_bwd_dkdv_p2_kernel<V_>, dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(256), kDkdvP2Smem, st, (const bf16_t*)q,            \
                       (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, nlse, ndelta, seq_start, n, n_q_heads, n_kv_heads, \
                       softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride, (int)q_row0, (g_dkdv_var & 16) ? 1 : 0);                  \
  } while (0)
#if PM_VARIANTS
  if ((g_dkdv_var & 15) == 0 || (g_dkdv_var & 15) == 12) {  // the default: 32 kv rows per wave, one wave per SIMD, hand-placed two-stage tile loop
    switch ((g_dkdv_var & 15) == 12 ? g_dkdv_var >> 5 : 0) {  // (attention_dkdv_p2.hip.h); 12 + 32 V: V = 1 its rotating-slot form (A/B), V >= 2 timing ablations
      case 1: PM_LAUNCH_DKDV_P2(1); break;
      case 2: PM_LAUNCH_DKDV_P2(2); break;
      case 3: PM_LAUNCH_DKDV_P2(3); break;
      case 4: PM_LAUNCH_DKDV_P2(4); break;
      case 5: PM_LAUNCH_DKDV_P2(5); break;
      case 6: PM_LAUNCH_DKDV_P2(6); break;
      case 7: PM_LAUNCH_DKDV_P2(7); break;  // v_pk_mul_f32 instead of the scalar multiplies (same bits, 1-2 % slower: profiles/r06_attention.md); 8-10: its other distances
      case 8: PM_LAUNCH_DKDV_P2(8); break;
      case 9: PM_LAUNCH_DKDV_P2(9); break;
      case 10: PM_LAUNCH_DKDV_P2(10); break;
      default: PM_LAUNCH_DKDV_P2(0); break;
    }
  }
  else if ((g_dkdv_var & 15) == 1) PM_LAUNCH_DKDV(2, 2, 64, 1);
  else if ((g_dkdv_var & 15) == 2) PM_LAUNCH_DKDV(2, 8, 64, 1);
  else if ((g_dkdv_var & 15) == 3) PM_LAUNCH_DKDV(3, 4, 32, 1);
  else if ((g_dkdv_var & 15) == 4) PM_LAUNCH_DKDV(1, 4, 64, 2);  // 32 kv rows per wave, one wave per SIMD (accumulators in AGPRs)
  else if ((g_dkdv_var & 15) == 5) PM_LAUNCH_DKDV(1, 8, 64, 2);
  else if ((g_dkdv_var & 15) == 6) PM_LAUNCH_DKDV(1, 4, 32, 2);
  else if ((g_dkdv_var & 15) == 8) PM_LAUNCH_DKDV(2, 4, 64, 1, 1);   // timing ablations of the shipped build (wrong results): no exp / mask VALU
  else if ((g_dkdv_var & 15) == 9) PM_LAUNCH_DKDV(2, 4, 64, 1, 2);   // no fragment reads
  else if ((g_dkdv_var & 15) == 10) PM_LAUNCH_DKDV(2, 4, 64, 1, 4);  // no LDS-DMA
  else if ((g_dkdv_var & 15) == 11) PM_LAUNCH_DKDV(2, 4, 64, 1, 8);  // no MFMAs
  else PM_LAUNCH_DKDV(2, 4, 64, 1);  // 7: the round-2..5 kernel (16 kv rows per wave, two workgroups per CU)
#else
  PM_LAUNCH_DKDV_P2(0);
#endif
#undef PM_LAUNCH_DKDV_P2
#undef PM_LAUNCH_DKDV
  PM_CHECK_LAUNCH(who);

#define PM_LAUNCH_DQ(KT_)                                                                                                                        \
  hipLaunchKernelGGL((attn_bwd_dq_kernel<2, KT_>), dim3((unsigned)(cdiv(n_q, 128) * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, \
                     (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, nlse, ndelta, seq_start, n, n_q_heads, n_kv_heads, softmax_scale,    \
                     (bf16_t*)grad_q, (int)q_row0, (g_dq_var & 16) ? 1 : 0)
#if PM_VARIANTS
  if ((g_dq_var & 15) == 1) PM_LAUNCH_DQ(32);
  else
#endif
  PM_LAUNCH_DQ(64);
#undef PM_LAUNCH_DQ
  PM_CHECK_LAUNCH(who);
  return PM_OK;
}

extern "C" int pm_attn_varlen_bwd(const void* grad_out, const void* q, const void* k, const void* v, int64_t v_row_stride, const void* out,
                                  const float* lse, const int32_t* seq_start, int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads,
                                  int32_t head_dim, float softmax_scale, void* grad_q, void* grad_k, void* grad_v, int64_t grad_v_row_stride,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  return attn_bwd_impl("pm_attn_varlen_bwd", grad_out, q, k, v, v_row_stride, out, lse, seq_start, 0, n_tokens, n_q_heads, n_kv_heads, head_dim,
                       softmax_scale, grad_q, grad_k, grad_v, grad_v_row_stride, workspace, workspace_bytes, stream);
}

extern "C" int pm_attn_varlen_bwd_rows(const void* grad_out, const void* q, const void* k, const void* v, int64_t v_row_stride, const void* out,
                                       const float* lse, const int32_t* seq_start, int64_t q_row0, int64_t n_q, int32_t n_q_heads, int32_t n_kv_heads,
                                       int32_t head_dim, float softmax_scale, void* grad_q, void* grad_k, void* grad_v, int64_t grad_v_row_stride,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  return attn_bwd_impl("pm_attn_varlen_bwd_rows", grad_out, q, k, v, v_row_stride, out, lse, seq_start, q_row0, n_q, n_q_heads, n_kv_heads, head_dim,
                       softmax_scale, grad_q, grad_k, grad_v, grad_v_row_stride, workspace, workspace_bytes, stream);
}
