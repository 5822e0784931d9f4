// attention.hip — varlen causal attention with GQA over packed documents (forward + deterministic backward), MFMA bf16.
//
// Reference: FlashAttention._compute_attention (src/prime_rl/trainer/models/layers/attn.py:94-110), i.e.
// flash_attn_varlen_func(q, k, v, cu_seqlens, cu_seqlens, max_seqlen, max_seqlen, causal=True) with GQA and
// scale head_dim^-0.5 (attn.py:60).  flash-attn itself is a third-party dependency (flash-attn 2.8.3, uv.lock:1737);
// the arithmetic restated here is its published algorithm: S = scale*Q K^T masked to (same document, j <= i),
// P = softmax_row(S), O = P V; backward dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dO o O)), dQ = scale dS K,
// dK = scale dS^T Q.
//
// MI355X design (this file: shared helpers, layout probes, delta kernel, host dispatch; kernels in attention_v4.cuh; the earlier
// generations — correct-first v1, LDS-round-trip v2, padded-LDS / transposed-global-copy v3 — are in the git history only):
//   * tiles live on the GLOBAL token grid of the packed row: row i may attend to j in [seq_start[i], i].  No per-document
//     tile scheduling on the host, no max_seqlen, no host sync: grid = ceil(N/BM) * heads.
//   * v_mfma_f32_16x16x32_bf16; every A/B fragment is 16 bytes along the contraction dim (lane l: row l&15,
//     k = (l>>4)*8..+8), C layout row = (l>>4)*4 + r, col = l&15 (guide §3, verified by pm_mfma_probe_16x16x32).
//   * TRANSPOSED products (S^T = K.Q^T, ...): probabilities stay in registers, the C layout is the B-fragment layout up to a
//     permutation of the contraction index (attention_v4.cuh header).
//   * operands whose contraction dim is the token axis (V in PV; Q, dO in dK/dV; K in dQ) are read out of the row-major LDS
//     tile with ds_read_b64_tr_b16 (pinned by pm_ds_read_tr16_probe): no transposed copies in global memory, no pre-pass.
//   * blockIdx -> (kv head fastest): the workgroups that share K/V land on the same XCD (block b runs on XCD b%8),
//     q tiles are issued heaviest-first (causal imbalance).
//   * backward is two kernels (dK/dV per kv tile, dQ per q tile): no atomics, bitwise reproducible.
#include "common.cuh"
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
// delta[h][i] = sum_d dO[i,h,d] * O[i,h,d]   (16 lanes per (token, head))
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, int64_t n, int hq,
                                                        float* __restrict__ delta) {
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
    delta[(int64_t)h * n + tok] = s;
  }
}

#include "attention_v4.cuh"

static inline int64_t align256(int64_t b) { return cdiv(b, 256) * 256; }
constexpr int kFwdDefaultVar = 412;
static int g_fwd_var_override = -1;  // set through pm_attn_debug_set_variant (microbenchmark A/B only)
static int g_bwd_var_override = -1;
}  // namespace pm

using namespace pm;

// A/B hook for tools/microbench.py: kernel == 0 selects the forward variant (100 * waves + OPT bits), value < 0 restores the default.
extern "C" int pm_attn_debug_set_variant(int32_t kernel, int32_t value) {
  if (kernel == 0) g_fwd_var_override = value;
  else if (kernel == 1) g_bwd_var_override = value;
  else return PM_ERR_INVALID_ARG;
  return PM_OK;
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

// Workspace: none in forward (the entry point keeps the argument for ABI stability), delta[h][n] f32 in backward.
extern "C" int64_t pm_attn_fwd_workspace_bytes(int64_t n_tokens, int32_t n_kv_heads, int32_t head_dim) {
  (void)n_kv_heads; (void)head_dim;
  return n_tokens <= 0 ? 0 : 256;
}

extern "C" int pm_attn_varlen_fwd(const void* q, const void* k, const void* v, int64_t v_row_stride, const int32_t* seq_start, int64_t n_tokens,
                                  int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float softmax_scale, void* out, float* lse,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(q && k && v && seq_start && out && lse, PM_ERR_INVALID_ARG, "pm_attn_varlen_fwd: null pointer");
  PM_REQUIRE(head_dim == kD, PM_ERR_UNSUPPORTED, "pm_attn_varlen_fwd: head_dim=%d (only 128 is built)", head_dim);
  PM_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, PM_ERR_INVALID_ARG, "pm_attn_varlen_fwd: bad head counts");
  PM_REQUIRE(n_tokens >= 0 && n_tokens < (1ll << 31) - 256 && v_row_stride >= (int64_t)n_kv_heads * kD && v_row_stride % 8 == 0, PM_ERR_INVALID_ARG,
             "pm_attn_varlen_fwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  PM_REQUIRE(n_tokens * v_row_stride * 2 < (1ll << 31) && n_tokens * (int64_t)n_kv_heads * kD * 2 < (1ll << 31), PM_ERR_UNSUPPORTED,
             "pm_attn_varlen_fwd: K / V beyond the 2 GiB a 32-bit buffer offset addresses (%lld tokens)", (long long)n_tokens);
  (void)workspace; (void)workspace_bytes;
  hipStream_t st = as_stream(stream);
  {  // V^T fragments are LDS transpose-reads of the row-major V tile
    // PM_ATTN_FWD_VAR = 100 * waves per workgroup + OPT bits (attention_v4.cuh): A/B with tools/microbench.py attnfwd, which times every
    // variant in one process.
    static const int var_env = getenv("PM_ATTN_FWD_VAR") ? atoi(getenv("PM_ATTN_FWD_VAR")) : -1;
    const int var = g_fwd_var_override >= 0 ? g_fwd_var_override : (var_env >= 0 ? var_env : kFwdDefaultVar);
#define PM_LAUNCH_FWD_V4(MINW_, NW_, OPT_)                                                                                                        \
  do {                                                                                                                                            \
    static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_v4_kernel<MINW_, NW_, OPT_>),                             \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kV4FwdSmemBytes) == hipSuccess;                       \
    PM_REQUIRE(ok_, PM_ERR_LAUNCH, "pm_attn_varlen_fwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");                                 \
    hipLaunchKernelGGL((attn_fwd_v4_kernel<MINW_, NW_, OPT_>), dim3((unsigned)(cdiv(n_tokens, NW_ * 32) * n_q_heads)), dim3(NW_ * 64),            \
                       kV4FwdSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, v_row_stride, seq_start, n_tokens, n_q_heads,   \
                       n_kv_heads, softmax_scale, (bf16_t*)out, lse);                                                                             \
  } while (0)
    switch (var) {
      case 400: PM_LAUNCH_FWD_V4(2, 4, 0); break;
      case 404: PM_LAUNCH_FWD_V4(2, 4, 4); break;
      case 408: PM_LAUNCH_FWD_V4(2, 4, 8); break;
      case 412: PM_LAUNCH_FWD_V4(2, 4, 12); break;
      case 414: PM_LAUNCH_FWD_V4(2, 4, 14); break;
      case 415: PM_LAUNCH_FWD_V4(2, 4, 15); break;
      case 428: PM_LAUNCH_FWD_V4(2, 4, 28); break;
      case 444: PM_LAUNCH_FWD_V4(2, 4, 44); break;
      case 460: PM_LAUNCH_FWD_V4(2, 4, 60); break;
      case 462: PM_LAUNCH_FWD_V4(2, 4, 62); break;
      default: PM_LAUNCH_FWD_V4(2, 4, 12); break;
    }
#undef PM_LAUNCH_FWD_V4
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_fwd");
  return PM_OK;
}

extern "C" int64_t pm_attn_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim) {
  if (n_tokens <= 0) return 0;
  (void)n_kv_heads; (void)head_dim;
  return align256((int64_t)n_q_heads * n_tokens * 4);
}

extern "C" int pm_attn_varlen_bwd(const void* grad_out, const void* q, const void* k, const void* v, int64_t v_row_stride, const void* out,
                                  const float* lse, const int32_t* seq_start, int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads,
                                  int32_t head_dim, float softmax_scale, void* grad_q, void* grad_k, void* grad_v, int64_t grad_v_row_stride,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(grad_out && q && k && v && out && lse && seq_start && grad_q && grad_k && grad_v, PM_ERR_INVALID_ARG, "pm_attn_varlen_bwd: null pointer");
  PM_REQUIRE(head_dim == kD, PM_ERR_UNSUPPORTED, "pm_attn_varlen_bwd: head_dim=%d (only 128 is built)", head_dim);
  PM_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, PM_ERR_INVALID_ARG, "pm_attn_varlen_bwd: bad head counts");
  PM_REQUIRE(n_tokens >= 0 && v_row_stride % 8 == 0 && grad_v_row_stride >= (int64_t)n_kv_heads * kD, PM_ERR_INVALID_ARG, "pm_attn_varlen_bwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  const int64_t need = pm_attn_bwd_workspace_bytes(n_tokens, n_q_heads, n_kv_heads, head_dim);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_attn_varlen_bwd: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  const int64_t n = n_tokens;
  float* delta = (float*)workspace;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv(n * n_q_heads, 16)), dim3(256), 0, st, (const bf16_t*)grad_out, (const bf16_t*)out, n,
                     n_q_heads, delta);
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(delta)");

  // ---- dK / dV
  {
    // PM_ATTN_DKDV_V4_VAR (A/B; whole backward per 16k-token layer): 4 (default) = 64-row q tiles, 4 waves, 2 workgroups/CU (1.37 ms);
    // 0 = 32-row q tiles (1.47 ms); 1 / 2 = 3 workgroups/CU with 2- / 4-fragment batches (register spills: +9 % / +34 % over 0);
    // 3 = one 8-wave workgroup with 128 kv rows (+5 % over 0)
    static const int var = getenv("PM_ATTN_DKDV_V4_VAR") ? atoi(getenv("PM_ATTN_DKDV_V4_VAR")) : 4;
#define PM_LAUNCH_DKDV_V4(NW_, MINB_, FB_)                                                                                                       \
  hipLaunchKernelGGL((attn_bwd_dkdv_v4_kernel<NW_, MINB_, FB_>), dim3((unsigned)(cdiv(n, NW_ * 16) * n_kv_heads)), dim3(NW_ * 64), kV4SmemBytes, st, \
                     (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n,        \
                     n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride)
    if (var == 1) PM_LAUNCH_DKDV_V4(4, 3, 2);
    else if (var == 2) PM_LAUNCH_DKDV_V4(4, 3, 4);
    else if (var == 3) PM_LAUNCH_DKDV_V4(8, 1, 4);
    else if (var == 0) PM_LAUNCH_DKDV_V4(4, 2, 4);
    else if (var == 5) {  // 64-row q tiles, one 8-wave workgroup per CU with 128 kv rows (half the staging traffic of the default)
      constexpr int kSmemQ64 = 4 * 64 * kD * 2 + 6 * 64 * 4 + 16;
      static const bool ok8 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v4q_kernel<8, 1, 4, 64>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, kSmemQ64) == hipSuccess;
      PM_REQUIRE(ok8, PM_ERR_LAUNCH, "pm_attn_varlen_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      hipLaunchKernelGGL((attn_bwd_dkdv_v4q_kernel<8, 1, 4, 64>), dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(512), kSmemQ64, st, (const bf16_t*)q,
                         (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    } else {
      constexpr int kSmemQ64 = 4 * 64 * kD * 2 + 6 * 64 * 4 + 16;
      static const bool q64_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v4q_kernel<4, 2, 4, 64>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kSmemQ64) == hipSuccess;
      PM_REQUIRE(q64_ok, PM_ERR_LAUNCH, "pm_attn_varlen_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      hipLaunchKernelGGL((attn_bwd_dkdv_v4q_kernel<4, 2, 4, 64>), dim3((unsigned)(cdiv(n, 64) * n_kv_heads)), dim3(256), kSmemQ64, st, (const bf16_t*)q,
                         (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n, n_q_heads, n_kv_heads,
                         softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
    }
#undef PM_LAUNCH_DKDV_V4
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dkdv)");

  // ---- dQ
  {
    static const bool kt64 = getenv("PM_ATTN_DQ_KT32") == nullptr;  // default: 64 kv rows per barrier (backward 1.41 -> 1.37 ms per layer); PM_ATTN_DQ_KT32 = A/B
    if (kt64) {
      static const bool ok64 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_v4_kernel<2, 64>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kV4DqSmemBytes<64>) == hipSuccess;
      PM_REQUIRE(ok64, PM_ERR_LAUNCH, "pm_attn_varlen_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      hipLaunchKernelGGL((attn_bwd_dq_v4_kernel<2, 64>), dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), kV4DqSmemBytes<64>, st,
                         (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n,
                         n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)grad_q);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_v4_kernel<2, 32>), dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), kV4DqSmemBytes<32>, st,
                         (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, lse, delta, seq_start, n,
                         n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)grad_q);
    }
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dq)");
  return PM_OK;
}
