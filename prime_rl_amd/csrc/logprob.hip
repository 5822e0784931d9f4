// logprob.hip — K1: per-token log-prob gather over the vocab + entropy (forward, backward, vocab-chunked form).
//
// Reference arithmetic (src/prime_rl/): selective_log_softmax / compute_entropy (trainer/rl/loss.py:48-63) with the
// temperature divide of trainer/rl/train.py:441, and the online (m, s, t) update of
// trainer/models/layers/lm_head.py:76-86 / 118-148; backward lm_head.py:184-191.
//
// Design (HBM-bound streaming reduction; guide Appendix B "Reduction", G13):
//   * one 256-thread workgroup (4 wave64) per logits row; each lane streams 16-byte vectors (float4 / 8 x bf16),
//     UNROLL independent loads in flight per lane -> >=32 KiB outstanding per workgroup;
//   * ONE pass: per-lane online softmax state (m, s = sum e^(l-m), t = sum e^(l-m)*l) in registers, rescaled once
//     per UNROLL*VEC elements (branch-free), 4 partial accumulators to break the add chain;
//   * wave64 shuffle merge + 4-entry LDS merge across waves; logz, logprob, entropy written by lane 0;
//   * the label logit is an exact indexed load of logits[row, label] (bit-exact gather);
//   * rows that are not 16-byte aligned (odd vocab / stride) peel a scalar head and tail around the vector body;
//   * small n_rows: split the vocab across several workgroups per row (partials in workspace + finalize kernel)
//     so a launch still has >=2048 workgroups for 256 CUs.
// Algorithmic bytes: (sizeof(T)*V + 20) per row forward; (sizeof(Tin)+sizeof(Tout))*V backward.
#include "common.hip.h"
#include <float.h>

namespace pm {

struct Lse {
  float m, s, t;
};

__device__ __forceinline__ Lse lse_empty() { return Lse{-FLT_MAX, 0.f, 0.f}; }

__device__ __forceinline__ Lse lse_merge(const Lse& a, const Lse& b) {
  const float m = fmaxf(a.m, b.m);
  const float ea = __expf(a.m - m), eb = __expf(b.m - m);
  return Lse{m, a.s * ea + b.s * eb, a.t * ea + b.t * eb};
}

__device__ __forceinline__ Lse lse_wave_reduce(Lse v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Lse other{__shfl_xor(v.m, o, 64), __shfl_xor(v.s, o, 64), __shfl_xor(v.t, o, 64)};
    v = lse_merge(v, other);
  }
  return v;
}

template <int NWAVES>
__device__ __forceinline__ Lse lse_block_reduce(Lse v, float* smem /* 3*NWAVES */) {
  v = lse_wave_reduce(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    smem[w * 3 + 0] = v.m;
    smem[w * 3 + 1] = v.s;
    smem[w * 3 + 2] = v.t;
  }
  __syncthreads();
  Lse r{smem[0], smem[1], smem[2]};
#pragma unroll
  for (int i = 1; i < NWAVES; ++i) r = lse_merge(r, Lse{smem[i * 3], smem[i * 3 + 1], smem[i * 3 + 2]});
  return r;
}

template <typename T>
struct VecTraits;
template <>
struct VecTraits<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* out) {
    out[0] = __uint_as_float(v.x);
    out[1] = __uint_as_float(v.y);
    out[2] = __uint_as_float(v.z);
    out[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ float scalar(const float* p) { return *p; }
};
template <>
struct VecTraits<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* out) {
    out[0] = bf16lo(v.x); out[1] = bf16hi(v.x);
    out[2] = bf16lo(v.y); out[3] = bf16hi(v.y);
    out[4] = bf16lo(v.z); out[5] = bf16hi(v.z);
    out[6] = bf16lo(v.w); out[7] = bf16hi(v.w);
  }
  __device__ static __forceinline__ float scalar(const bf16_t* p) { return bf16_to_f32(*p); }
};

// Per-lane online-softmax state with 4 partial (s, t) accumulators sharing one running max.
struct LaneAcc {
  float m;
  float s[4];
  float t[4];
  __device__ __forceinline__ void init() {
    m = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = t[i] = 0.f;
  }
  template <int K>
  __device__ __forceinline__ void update(const float (&l)[K]) {
    float bm = l[0];
#pragma unroll
    for (int i = 1; i < K; ++i) bm = fmaxf(bm, l[i]);
    const float mn = fmaxf(m, bm);
    const float sc = __expf(m - mn);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] *= sc;
      t[i] *= sc;
    }
    m = mn;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float e = __expf(l[i] - mn);
      s[i & 3] += e;
      t[i & 3] = fmaf(e, l[i], t[i & 3]);
    }
  }
  __device__ __forceinline__ void update1(float l) {
    const float mn = fmaxf(m, l);
    const float sc = __expf(m - mn);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] *= sc;
      t[i] *= sc;
    }
    m = mn;
    const float e = __expf(l - mn);
    s[0] += e;
    t[0] = fmaf(e, l, t[0]);
  }
  __device__ __forceinline__ Lse fold() const { return Lse{m, (s[0] + s[1]) + (s[2] + s[3]), (t[0] + t[1]) + (t[2] + t[3])}; }
};

// Stream elements [e_begin, e_end) of one row into `acc`.  Handles arbitrary alignment by peeling scalars.
template <typename T, int BLOCK, int UNROLL>
__device__ __forceinline__ void row_stream(const T* __restrict__ row, int64_t e_begin, int64_t e_end, float inv_t, LaneAcc& acc) {
  constexpr int VEC = VecTraits<T>::N;
  const int tid = threadIdx.x;
  const T* p = row + e_begin;
  int64_t n = e_end - e_begin;
  if (n <= 0) return;
  // scalar head until 16-byte aligned
  const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
  int64_t head = (int64_t)(((16 - (addr & 15)) & 15) / sizeof(T));
  if (head > n) head = n;
  if (tid < head) acc.update1(VecTraits<T>::scalar(p + tid) * inv_t);
  p += head;
  n -= head;
  const int64_t nvec = n / VEC;
  const u32x4* __restrict__ pv = reinterpret_cast<const u32x4*>(p);
  int64_t i = tid;
  for (; i + (int64_t)(UNROLL - 1) * BLOCK < nvec; i += (int64_t)UNROLL * BLOCK) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(pv + i + (int64_t)u * BLOCK);
    float l[UNROLL * VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) VecTraits<T>::unpack(v[u], l + u * VEC);
#pragma unroll
    for (int k = 0; k < UNROLL * VEC; ++k) l[k] *= inv_t;
    acc.update(l);
  }
  for (; i < nvec; i += BLOCK) {
    const u32x4 v = __builtin_nontemporal_load(pv + i);
    float l[VEC];
    VecTraits<T>::unpack(v, l);
#pragma unroll
    for (int k = 0; k < VEC; ++k) l[k] *= inv_t;
    acc.update(l);
  }
  // scalar tail
  const int64_t tail = n - nvec * VEC;
  if (tid < tail) acc.update1(VecTraits<T>::scalar(p + nvec * VEC + tid) * inv_t);
}

template <typename T>
__device__ __forceinline__ float target_logit(const T* row, int64_t label, int64_t v_lo, int64_t v_hi, float inv_t, float fallback) {
  // mirrors the branchless in_range / where of lm_head.py:140-143 (out-of-range label keeps the old value)
  if (label >= v_lo && label < v_hi) return VecTraits<T>::scalar(row + (label - v_lo)) * inv_t;
  return fallback;
}

// ---- single-pass kernel: one workgroup per row ----------------------------------------------------
template <typename T, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void logprob_fwd_row_kernel(const T* __restrict__ logits, int64_t row_stride, int64_t vocab,
                                                               const int64_t* __restrict__ labels, const float* __restrict__ inv_temp,
                                                               float* __restrict__ logprobs, float* __restrict__ entropy,
                                                               float* __restrict__ logz) {
  __shared__ float smem[3 * (BLOCK / 64)];
  const int64_t row = blockIdx.x;
  const T* rp = logits + row * row_stride;
  const float inv_t = inv_temp[row];
  LaneAcc acc;
  acc.init();
  row_stream<T, BLOCK, UNROLL>(rp, 0, vocab, inv_t, acc);
  const Lse r = lse_block_reduce<BLOCK / 64>(acc.fold(), smem);
  if (threadIdx.x == 0) {
    const float tl = target_logit<T>(rp, labels[row], 0, vocab, inv_t, 0.f);
    const float lz = r.m + logf(r.s);
    logz[row] = lz;
    logprobs[row] = tl - lz;
    entropy[row] = lz - r.t / r.s;
  }
}

// ---- split-vocab kernels (small n_rows) ------------------------------------------------------------
template <typename T, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void logprob_fwd_split_kernel(const T* __restrict__ logits, int64_t row_stride, int64_t vocab,
                                                                 const float* __restrict__ inv_temp, int splits, int64_t per_split,
                                                                 float* __restrict__ partials /* [n_rows][splits][3] */) {
  __shared__ float smem[3 * (BLOCK / 64)];
  const int64_t row = blockIdx.x;
  const int sp = blockIdx.y;
  const T* rp = logits + row * row_stride;
  const float inv_t = inv_temp[row];
  int64_t b = (int64_t)sp * per_split, e = b + per_split;
  if (e > vocab) e = vocab;
  LaneAcc acc;
  acc.init();
  row_stream<T, BLOCK, UNROLL>(rp, b, e, inv_t, acc);
  const Lse r = lse_block_reduce<BLOCK / 64>(acc.fold(), smem);
  if (threadIdx.x == 0) {
    float* o = partials + (row * splits + sp) * 3;
    o[0] = r.m;
    o[1] = r.s;
    o[2] = r.t;
  }
}

template <typename T>
__global__ void logprob_fwd_split_finalize_kernel(const T* __restrict__ logits, int64_t row_stride, int64_t vocab,
                                                  const int64_t* __restrict__ labels, const float* __restrict__ inv_temp,
                                                  const float* __restrict__ partials, int splits, int64_t n_rows,
                                                  float* __restrict__ logprobs, float* __restrict__ entropy, float* __restrict__ logz) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  Lse r = lse_empty();
  for (int sp = 0; sp < splits; ++sp) {
    const float* o = partials + (row * splits + sp) * 3;
    r = lse_merge(r, Lse{o[0], o[1], o[2]});
  }
  const float inv_t = inv_temp[row];
  const float tl = target_logit<T>(logits + row * row_stride, labels[row], 0, vocab, inv_t, 0.f);
  const float lz = r.m + logf(r.s);
  logz[row] = lz;
  logprobs[row] = tl - lz;
  entropy[row] = lz - r.t / r.s;
}

// ---- vocab-chunked online update (fused LM-head path) ----------------------------------------------
__global__ void chunk_init_kernel(float* m, float* s, float* t, float* tgt, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    m[i] = -FLT_MAX;
    s[i] = 0.f;
    t[i] = 0.f;
    tgt[i] = 0.f;
  }
}

template <typename T, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void logprob_chunk_update_kernel(const T* __restrict__ chunk, int64_t row_stride, int64_t chunk_vocab,
                                                                    int64_t vocab_start, const int64_t* __restrict__ labels,
                                                                    const float* __restrict__ inv_temp, float* __restrict__ sm,
                                                                    float* __restrict__ ss, float* __restrict__ st, float* __restrict__ stgt) {
  __shared__ float smem[3 * (BLOCK / 64)];
  const int64_t row = blockIdx.x;
  const T* rp = chunk + row * row_stride;
  const float inv_t = inv_temp[row];
  LaneAcc acc;
  acc.init();
  row_stream<T, BLOCK, UNROLL>(rp, 0, chunk_vocab, inv_t, acc);
  const Lse r = lse_block_reduce<BLOCK / 64>(acc.fold(), smem);
  if (threadIdx.x == 0) {
    const Lse merged = lse_merge(Lse{sm[row], ss[row], st[row]}, r);
    sm[row] = merged.m;
    ss[row] = merged.s;
    st[row] = merged.t;
    stgt[row] = target_logit<T>(rp, labels[row], vocab_start, vocab_start + chunk_vocab, inv_t, stgt[row]);
  }
}

__global__ void chunk_finalize_kernel(const float* m, const float* s, const float* t, const float* tgt, int64_t n,
                                      float* logprobs, float* entropy, float* logz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float lz = m[i] + logf(s[i]);
    logz[i] = lz;
    logprobs[i] = tgt[i] - lz;
    entropy[i] = lz - t[i] / s[i];
  }
}

// ---- backward: dlogits = g*inv_t*(onehot - softmax) ------------------------------------------------
template <typename TO, int VEC>
struct alignas(VEC * sizeof(TO)) OutVec {
  TO v[VEC];
};
__device__ __forceinline__ void store_out(float x, float* p) { *p = x; }
__device__ __forceinline__ void store_out(float x, bf16_t* p) { *p = f32_to_bf16(x); }

// VEC == VecTraits<TI>::N : aligned fast path (rows 16-byte aligned for TI, VEC*sizeof(TO) aligned for TO);  VEC == 1: generic.
template <typename TI, typename TO, int VEC, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void logprob_bwd_kernel(const TI* __restrict__ logits, int64_t row_stride, int64_t vocab,
                                                           int64_t vocab_start, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ inv_temp, const float* __restrict__ logz,
                                                           const float* __restrict__ grad, TO* __restrict__ dlogits, int64_t out_stride) {
  const int64_t row = blockIdx.x;
  const float inv_t = inv_temp[row];
  const float a = grad[row] * inv_t;
  const float lz = logz[row];
  const int64_t lab = labels[row] - vocab_start;  // local label (may be out of this chunk)
  const TI* rp = logits + row * row_stride;
  TO* op = dlogits + row * out_stride;
  const int64_t nvec = vocab / VEC;  // fast path requires vocab % VEC == 0
  const int64_t base = (int64_t)blockIdx.y * BLOCK * UNROLL + threadIdx.x;
  if (a == 0.f) {  // masked token: gradient row is exactly zero, skip the logits read
    OutVec<TO, VEC> z;
#pragma unroll
    for (int k = 0; k < VEC; ++k) store_out(0.f, &z.v[k]);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * BLOCK;
      if (i < nvec) reinterpret_cast<OutVec<TO, VEC>*>(op)[i] = z;
    }
    return;
  }
  if constexpr (VEC == 1) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * BLOCK;
      if (i < nvec) {
        const float l = VecTraits<TI>::scalar(rp + i) * inv_t;
        const float d = a * ((i == lab ? 1.f : 0.f) - __expf(l - lz));
        store_out(d, op + i);
      }
    }
  } else {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * BLOCK;
      if (i < nvec) v[u] = reinterpret_cast<const u32x4*>(rp)[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * BLOCK;
      if (i < nvec) {
        float l[VEC];
        VecTraits<TI>::unpack(v[u], l);
        OutVec<TO, VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float lk = l[k] * inv_t;
          const float d = a * (((i * VEC + k) == lab ? 1.f : 0.f) - __expf(lk - lz));
          store_out(d, &o.v[k]);
        }
        reinterpret_cast<OutVec<TO, VEC>*>(op)[i] = o;
      }
    }
  }
}

// ---- host-side dispatch -----------------------------------------------------------------------------
static constexpr int kBlock = 256;
static constexpr int kSplitTargetWGs = 2048;

static int choose_splits(int64_t n_rows, int64_t vocab) {
  if (n_rows >= 1024) return 1;
  int64_t s = cdiv(kSplitTargetWGs, n_rows > 0 ? n_rows : 1);
  const int64_t max_by_work = vocab / 8192 > 0 ? vocab / 8192 : 1;  // keep >= 8K elements per split
  if (s > max_by_work) s = max_by_work;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return (int)s;
}

template <typename T>
static int fwd_dispatch(const T* logits, int64_t n_rows, int64_t vocab, int64_t row_stride, const int64_t* labels, const float* inv_temp,
                        float* logprobs, float* entropy, float* logz, void* ws, int64_t ws_bytes, hipStream_t st) {
  constexpr int UNROLL = sizeof(T) == 4 ? 8 : 4;  // 128 B (f32) / 64 B (bf16) per lane in flight -> 32 / 16 elements per rescale
  const int splits = choose_splits(n_rows, vocab);
  if (splits == 1) {
    hipLaunchKernelGGL((logprob_fwd_row_kernel<T, kBlock, UNROLL>), dim3((unsigned)n_rows), dim3(kBlock), 0, st, logits, row_stride, vocab,
                       labels, inv_temp, logprobs, entropy, logz);
    PM_CHECK_LAUNCH("pm_logprob_entropy_fwd");
    return PM_OK;
  }
  const int64_t need = n_rows * splits * 3 * (int64_t)sizeof(float);
  PM_REQUIRE(ws != nullptr && ws_bytes >= need, PM_ERR_WORKSPACE, "pm_logprob_entropy_fwd: workspace %lld < %lld", (long long)ws_bytes,
             (long long)need);
  constexpr int VEC = VecTraits<T>::N;
  int64_t per = cdiv(vocab, splits);
  per = cdiv(per, VEC * 64) * VEC * 64;  // keep split boundaries vector-friendly
  hipLaunchKernelGGL((logprob_fwd_split_kernel<T, kBlock, UNROLL>), dim3((unsigned)n_rows, (unsigned)splits), dim3(kBlock), 0, st, logits,
                     row_stride, vocab, inv_temp, splits, per, (float*)ws);
  PM_CHECK_LAUNCH("pm_logprob_entropy_fwd(split)");
  hipLaunchKernelGGL((logprob_fwd_split_finalize_kernel<T>), dim3((unsigned)cdiv(n_rows, 256)), dim3(256), 0, st, logits, row_stride, vocab,
                     labels, inv_temp, (const float*)ws, splits, n_rows, logprobs, entropy, logz);
  PM_CHECK_LAUNCH("pm_logprob_entropy_fwd(finalize)");
  return PM_OK;
}

template <typename TI, typename TO>
static int bwd_dispatch(const TI* logits, int64_t n_rows, int64_t vocab, int64_t row_stride, int64_t vocab_start, const int64_t* labels,
                        const float* inv_temp, const float* logz, const float* grad, TO* dlogits, int64_t out_stride, hipStream_t st,
                        const char* name) {
  constexpr int VEC = VecTraits<TI>::N;
  constexpr int UNROLL = 4;
  const bool aligned = (vocab % VEC == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) && ((row_stride * sizeof(TI)) % 16 == 0) &&
                       ((reinterpret_cast<uintptr_t>(dlogits) % (VEC * sizeof(TO))) == 0) && ((out_stride * sizeof(TO)) % (VEC * sizeof(TO)) == 0);
  if (aligned) {
    const int64_t nvec = vocab / VEC;
    dim3 grid((unsigned)n_rows, (unsigned)cdiv(nvec, (int64_t)kBlock * UNROLL));
    hipLaunchKernelGGL((logprob_bwd_kernel<TI, TO, VEC, kBlock, UNROLL>), grid, dim3(kBlock), 0, st, logits, row_stride, vocab, vocab_start,
                       labels, inv_temp, logz, grad, dlogits, out_stride);
  } else {
    dim3 grid((unsigned)n_rows, (unsigned)cdiv(vocab, (int64_t)kBlock * UNROLL));
    hipLaunchKernelGGL((logprob_bwd_kernel<TI, TO, 1, kBlock, UNROLL>), grid, dim3(kBlock), 0, st, logits, row_stride, vocab, vocab_start,
                       labels, inv_temp, logz, grad, dlogits, out_stride);
  }
  PM_CHECK_LAUNCH(name);
  return PM_OK;
}

static int bwd_common(const void* logits, int dtype, int64_t n_rows, int64_t vocab, int64_t row_stride, int64_t vocab_start,
                      const int64_t* labels, const float* inv_temp, const float* logz, const float* grad, void* dlogits, int out_dtype,
                      int64_t out_stride, void* stream, const char* name) {
  PM_REQUIRE(logits && labels && inv_temp && logz && grad && dlogits, PM_ERR_INVALID_ARG, "%s: null pointer", name);
  PM_REQUIRE(n_rows >= 0 && vocab > 0 && row_stride >= vocab && out_stride >= vocab, PM_ERR_INVALID_ARG, "%s: bad sizes", name);
  if (n_rows == 0) return PM_OK;
  hipStream_t st = as_stream(stream);
  if (dtype == PM_DTYPE_F32 && out_dtype == PM_DTYPE_F32)
    return bwd_dispatch<float, float>((const float*)logits, n_rows, vocab, row_stride, vocab_start, labels, inv_temp, logz, grad, (float*)dlogits, out_stride, st, name);
  if (dtype == PM_DTYPE_F32 && out_dtype == PM_DTYPE_BF16)
    return bwd_dispatch<float, bf16_t>((const float*)logits, n_rows, vocab, row_stride, vocab_start, labels, inv_temp, logz, grad, (bf16_t*)dlogits, out_stride, st, name);
  if (dtype == PM_DTYPE_BF16 && out_dtype == PM_DTYPE_BF16)
    return bwd_dispatch<bf16_t, bf16_t>((const bf16_t*)logits, n_rows, vocab, row_stride, vocab_start, labels, inv_temp, logz, grad, (bf16_t*)dlogits, out_stride, st, name);
  if (dtype == PM_DTYPE_BF16 && out_dtype == PM_DTYPE_F32)
    return bwd_dispatch<bf16_t, float>((const bf16_t*)logits, n_rows, vocab, row_stride, vocab_start, labels, inv_temp, logz, grad, (float*)dlogits, out_stride, st, name);
  set_error("%s: unsupported dtype %d/%d", name, dtype, out_dtype);
  return PM_ERR_INVALID_ARG;
}

}  // namespace pm

using namespace pm;

extern "C" int64_t pm_logprob_workspace_bytes(int64_t n_rows, int64_t vocab) {
  if (n_rows <= 0 || vocab <= 0) return 0;
  return n_rows * choose_splits(n_rows, vocab) * 3 * (int64_t)sizeof(float);
}

extern "C" int pm_logprob_entropy_fwd(const void* logits, int dtype, int64_t n_rows, int64_t vocab, int64_t row_stride, const int64_t* labels,
                                      const float* inv_temp, float* logprobs, float* entropy, float* logz, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(logits && labels && inv_temp && logprobs && entropy && logz, PM_ERR_INVALID_ARG, "pm_logprob_entropy_fwd: null pointer");
  PM_REQUIRE(n_rows >= 0 && vocab > 0 && row_stride >= vocab, PM_ERR_INVALID_ARG, "pm_logprob_entropy_fwd: bad sizes n=%lld v=%lld stride=%lld",
             (long long)n_rows, (long long)vocab, (long long)row_stride);
  if (n_rows == 0) return PM_OK;
  hipStream_t st = as_stream(stream);
  if (dtype == PM_DTYPE_F32)
    return fwd_dispatch<float>((const float*)logits, n_rows, vocab, row_stride, labels, inv_temp, logprobs, entropy, logz, workspace, workspace_bytes, st);
  if (dtype == PM_DTYPE_BF16)
    return fwd_dispatch<bf16_t>((const bf16_t*)logits, n_rows, vocab, row_stride, labels, inv_temp, logprobs, entropy, logz, workspace, workspace_bytes, st);
  set_error("pm_logprob_entropy_fwd: unsupported dtype %d", dtype);
  return PM_ERR_INVALID_ARG;
}

extern "C" int pm_logprob_bwd(const void* logits, int dtype, int64_t n_rows, int64_t vocab, int64_t row_stride, const int64_t* labels,
                              const float* inv_temp, const float* logz, const float* grad_logprobs, void* dlogits, int dlogits_dtype,
                              int64_t dlogits_stride, void* stream) {
  return bwd_common(logits, dtype, n_rows, vocab, row_stride, 0, labels, inv_temp, logz, grad_logprobs, dlogits, dlogits_dtype, dlogits_stride,
                    stream, "pm_logprob_bwd");
}

extern "C" int pm_logprob_chunk_init(float* m, float* s, float* t, float* tgt, int64_t n_rows, void* stream) {
  PM_REQUIRE(m && s && t && tgt && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_logprob_chunk_init: bad args");
  if (n_rows == 0) return PM_OK;
  hipLaunchKernelGGL(chunk_init_kernel, dim3((unsigned)cdiv(n_rows, 256)), dim3(256), 0, as_stream(stream), m, s, t, tgt, n_rows);
  PM_CHECK_LAUNCH("pm_logprob_chunk_init");
  return PM_OK;
}

extern "C" int pm_logprob_chunk_update(const void* chunk, int dtype, int64_t n_rows, int64_t chunk_vocab, int64_t row_stride,
                                       int64_t vocab_start, const int64_t* labels, const float* inv_temp, float* m, float* s, float* t,
                                       float* tgt, void* stream) {
  PM_REQUIRE(chunk && labels && inv_temp && m && s && t && tgt, PM_ERR_INVALID_ARG, "pm_logprob_chunk_update: null pointer");
  PM_REQUIRE(n_rows >= 0 && chunk_vocab > 0 && row_stride >= chunk_vocab && vocab_start >= 0, PM_ERR_INVALID_ARG, "pm_logprob_chunk_update: bad sizes");
  if (n_rows == 0) return PM_OK;
  hipStream_t st = as_stream(stream);
  if (dtype == PM_DTYPE_F32) {
    hipLaunchKernelGGL((logprob_chunk_update_kernel<float, kBlock, 4>), dim3((unsigned)n_rows), dim3(kBlock), 0, st, (const float*)chunk, row_stride,
                       chunk_vocab, vocab_start, labels, inv_temp, m, s, t, tgt);
  } else if (dtype == PM_DTYPE_BF16) {
    hipLaunchKernelGGL((logprob_chunk_update_kernel<bf16_t, kBlock, 2>), dim3((unsigned)n_rows), dim3(kBlock), 0, st, (const bf16_t*)chunk, row_stride,
                       chunk_vocab, vocab_start, labels, inv_temp, m, s, t, tgt);
  } else {
    set_error("pm_logprob_chunk_update: unsupported dtype %d", dtype);
    return PM_ERR_INVALID_ARG;
  }
  PM_CHECK_LAUNCH("pm_logprob_chunk_update");
  return PM_OK;
}

extern "C" int pm_logprob_chunk_finalize(const float* m, const float* s, const float* t, const float* tgt, int64_t n_rows, float* logprobs,
                                         float* entropy, float* logz, void* stream) {
  PM_REQUIRE(m && s && t && tgt && logprobs && entropy && logz && n_rows >= 0, PM_ERR_INVALID_ARG, "pm_logprob_chunk_finalize: bad args");
  if (n_rows == 0) return PM_OK;
  hipLaunchKernelGGL(chunk_finalize_kernel, dim3((unsigned)cdiv(n_rows, 256)), dim3(256), 0, as_stream(stream), m, s, t, tgt, n_rows, logprobs,
                     entropy, logz);
  PM_CHECK_LAUNCH("pm_logprob_chunk_finalize");
  return PM_OK;
}

extern "C" int pm_logprob_chunk_bwd(const void* chunk, int dtype, int64_t n_rows, int64_t chunk_vocab, int64_t row_stride, int64_t vocab_start,
                                    const int64_t* labels, const float* inv_temp, const float* logz, const float* grad_logprobs, void* dlogits,
                                    int dlogits_dtype, int64_t dlogits_stride, void* stream) {
  return bwd_common(chunk, dtype, n_rows, chunk_vocab, row_stride, vocab_start, labels, inv_temp, logz, grad_logprobs, dlogits, dlogits_dtype,
                    dlogits_stride, stream, "pm_logprob_chunk_bwd");
}
