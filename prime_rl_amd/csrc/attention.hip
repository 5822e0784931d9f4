// attention.hip — varlen causal attention with GQA over packed documents (forward + deterministic backward), MFMA bf16.
//
// Reference: FlashAttention._compute_attention (src/prime_rl/trainer/models/layers/attn.py:94-110), i.e.
// flash_attn_varlen_func(q, k, v, cu_seqlens, cu_seqlens, max_seqlen, max_seqlen, causal=True) with GQA and
// scale head_dim^-0.5 (attn.py:60).  flash-attn itself is a third-party dependency (flash-attn 2.8.3, uv.lock:1737);
// the arithmetic restated here is its published algorithm: S = scale*Q K^T masked to (same document, j <= i),
// P = softmax_row(S), O = P V; backward dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dO o O)), dQ = scale dS K,
// dK = scale dS^T Q.
//
// MI355X design (this file: shared helpers, layout probes, delta kernel, host dispatch; kernels in attention_v4.cuh = default and
// attention_v3.cuh = the previous generation kept behind PM_ATTN_{FWD,DKDV,DQ}_V3 for A/B; the correct-first v1 and the
// LDS-round-trip v2 kernels are in the git history only):
//   * tiles live on the GLOBAL token grid of the packed row: row i may attend to j in [seq_start[i], i].  No per-document
//     tile scheduling on the host, no max_seqlen, no host sync: grid = ceil(N/BM) * heads.
//   * v_mfma_f32_16x16x32_bf16; every A/B fragment is 16 bytes along the contraction dim (lane l: row l&15,
//     k = (l>>4)*8..+8), C layout row = (l>>4)*4 + r, col = l&15 (guide §3, verified by pm_mfma_probe_16x16x32).
//   * TRANSPOSED products (S^T = K.Q^T, ...): probabilities stay in registers, the C layout is the B-fragment layout up to a
//     permutation of the contraction index (attention_v3.cuh header).
//   * operands whose contraction dim is the token axis (V in PV; Q, dO in dK/dV; K in dQ): v4 reads them out of the row-major LDS
//     tile with ds_read_b64_tr_b16 (pinned by pm_ds_read_tr16_probe); v3 reads [head][128][N_pad] transposed copies written by a
//     pre-pass (transpose_heads_kernel) into the caller's workspace.
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
// padded LDS row lengths of the v3 kernels (elements) and the 16-byte LDS accessors shared by v3 / v4
constexpr int kLd128 = 136;  // 128 + 8
constexpr int kLd64 = 72;    // 64 + 8
constexpr int kLd32 = 40;    // 32 + 8
__device__ __forceinline__ u32x4 lds16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void sts16(bf16_t* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
constexpr int kPadLd = 72;    // 64 + 8 bf16: 144-byte LDS rows

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
// [n][row_stride] (head h at column h*128) -> [heads][128][n_pad], zero padding for tokens >= n.
__global__ __launch_bounds__(256) void transpose_heads_kernel(const bf16_t* __restrict__ src, int64_t row_stride, int64_t n, int64_t n_pad,
                                                             bf16_t* __restrict__ dst) {
  __shared__ bf16_t tile[64][66];
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  const int d0 = blockIdx.y * 64;
  const int h = blockIdx.z;
  const int t = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int tr = (t >> 3) + pass * 32;
    const int c8 = (t & 7) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t0 + tr < n) v = ldg16(src + (t0 + tr) * row_stride + (int64_t)h * kD + d0 + c8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tile[tr][c8 + 2 * k] = (bf16_t)(w[k] & 0xffffu);
      tile[tr][c8 + 2 * k + 1] = (bf16_t)(w[k] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int d = (t >> 3) + pass * 32;
    const int tk = (t & 7) * 8;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[tk + 2 * k][d] | ((uint32_t)tile[tk + 2 * k + 1][d] << 16);
    u32x4 v = {w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4*>(dst + ((int64_t)h * kD + d0 + d) * n_pad + t0 + tk) = v;
  }
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

#include "attention_v3.cuh"
#include "attention_v4.cuh"

static inline int64_t pad64(int64_t n) { return cdiv(n, 64) * 64; }
static inline int64_t align256(int64_t b) { return cdiv(b, 256) * 256; }
// A/B switches: the v3 kernels (padded LDS tiles + transposed global copies written by a pre-pass) stay selectable per kernel.
static bool fwd_v3() {
  static const bool f = getenv("PM_ATTN_FWD_V3") != nullptr;
  return f;
}
static bool dkdv_v3() {
  static const bool f = getenv("PM_ATTN_DKDV_V3") != nullptr;
  return f;
}
static bool dq_v3() {
  static const bool f = getenv("PM_ATTN_DQ_V3") != nullptr || getenv("PM_ATTN_DKDV_V3") != nullptr;
  return f;
}

static int launch_transpose(const bf16_t* src, int64_t row_stride, int64_t n, int64_t n_pad, int heads, bf16_t* dst, hipStream_t st) {
  hipLaunchKernelGGL(transpose_heads_kernel, dim3((unsigned)(n_pad / 64), 2, (unsigned)heads), dim3(256), 0, st, src, row_stride, n, n_pad, dst);
  PM_CHECK_LAUNCH("attention transpose");
  return PM_OK;
}

}  // namespace pm

using namespace pm;

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

// Workspace: the default (v4) kernels need none in forward and only delta[h][n] in backward; the sizes below also cover the
// transposed copies of the v3 kernels so that the A/B switches work with the same caller-owned buffer.
extern "C" int64_t pm_attn_fwd_workspace_bytes(int64_t n_tokens, int32_t n_kv_heads, int32_t head_dim) {
  if (n_tokens <= 0) return 0;
  return align256((int64_t)n_kv_heads * head_dim * pad64(n_tokens) * 2);
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
  const int64_t need = pm_attn_fwd_workspace_bytes(n_tokens, n_kv_heads, head_dim);
  PM_REQUIRE(workspace && workspace_bytes >= need, PM_ERR_WORKSPACE, "pm_attn_varlen_fwd: workspace %lld < %lld", (long long)workspace_bytes,
             (long long)need);
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)(cdiv(n_tokens, 128) * n_q_heads);
  if (!fwd_v3()) {  // v4: V^T fragments are LDS transpose-reads of the row-major V tile
    static const bool smem_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_v4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    kV4FwdSmemBytes) == hipSuccess;
    PM_REQUIRE(smem_ok, PM_ERR_LAUNCH, "pm_attn_varlen_fwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipLaunchKernelGGL((attn_fwd_v4_kernel<2>), dim3(grid), dim3(256), kV4FwdSmemBytes, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       v_row_stride, seq_start, n_tokens, n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  } else {
    const int64_t n_pad = pad64(n_tokens);
    bf16_t* vt = (bf16_t*)workspace;
    const int rc = launch_transpose((const bf16_t*)v, v_row_stride, n_tokens, n_pad, n_kv_heads, vt, st);
    if (rc != PM_OK) return rc;
    hipLaunchKernelGGL((attn_fwd_v3_kernel<2>), dim3(grid), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, vt, seq_start, n_tokens, n_pad,
                       n_q_heads, n_kv_heads, softmax_scale, (bf16_t*)out, lse);
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_fwd");
  return PM_OK;
}

extern "C" int64_t pm_attn_bwd_workspace_bytes(int64_t n_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim) {
  if (n_tokens <= 0) return 0;
  const int64_t n_pad = pad64(n_tokens);
  const int64_t qt = align256((int64_t)n_q_heads * head_dim * n_pad * 2);
  const int64_t kt = align256((int64_t)n_kv_heads * head_dim * n_pad * 2);
  const int64_t dl = align256((int64_t)n_q_heads * n_tokens * 4);
  return qt * 2 + kt + dl;
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
  const int64_t n = n_tokens, n_pad = pad64(n);
  const int64_t qt_b = align256((int64_t)n_q_heads * kD * n_pad * 2);
  const int64_t kt_b = align256((int64_t)n_kv_heads * kD * n_pad * 2);
  char* ws = (char*)workspace;
  bf16_t* qt = (bf16_t*)ws;
  bf16_t* dot_ = (bf16_t*)(ws + qt_b);
  bf16_t* kt = (bf16_t*)(ws + 2 * qt_b);
  float* delta = (float*)(ws + 2 * qt_b + kt_b);
  int rc;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv(n * n_q_heads, 16)), dim3(256), 0, st, (const bf16_t*)grad_out, (const bf16_t*)out, n,
                     n_q_heads, delta);
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(delta)");

  // ---- dK / dV
  if (!dkdv_v3()) {
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
  } else {
    static const bool smem_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_v3_kernel<8, 1, 0>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvSmemBytes) == hipSuccess;
    PM_REQUIRE(smem_ok, PM_ERR_LAUNCH, "pm_attn_varlen_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    if ((rc = launch_transpose((const bf16_t*)q, (int64_t)n_q_heads * kD, n, n_pad, n_q_heads, qt, st)) != PM_OK) return rc;
    if ((rc = launch_transpose((const bf16_t*)grad_out, (int64_t)n_q_heads * kD, n, n_pad, n_q_heads, dot_, st)) != PM_OK) return rc;
    hipLaunchKernelGGL((attn_bwd_dkdv_v3_kernel<8, 1>), dim3((unsigned)(cdiv(n, 128) * n_kv_heads)), dim3(512), kDkdvSmemBytes, st, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, qt, dot_, lse, delta, seq_start, n, n_pad, n_q_heads,
                       n_kv_heads, softmax_scale, (bf16_t*)grad_k, (bf16_t*)grad_v, grad_v_row_stride);
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dkdv)");

  // ---- dQ
  if (!dq_v3()) {
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
  } else {
    if ((rc = launch_transpose((const bf16_t*)k, (int64_t)n_kv_heads * kD, n, n_pad, n_kv_heads, kt, st)) != PM_OK) return rc;
    hipLaunchKernelGGL((attn_bwd_dq_v3_kernel<2>), dim3((unsigned)(cdiv(n, 128) * n_q_heads)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, v_row_stride, (const bf16_t*)grad_out, kt, lse, delta, seq_start, n, n_pad, n_q_heads, n_kv_heads,
                       softmax_scale, (bf16_t*)grad_q);
  }
  PM_CHECK_LAUNCH("pm_attn_varlen_bwd(dq)");
  return PM_OK;
}
