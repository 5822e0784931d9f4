// elementwise.hip — SwiGLU fwd/bwd, embedding gather / deterministic scatter-add, gradient accumulation,
// sum-of-squares, clip coefficient and fused AdamW on flat f32 shards.  All HBM-bound streaming kernels with
// 16-byte accesses per lane and capped grids (guide G11/G13).
//
// Reference arithmetic (src/prime_rl/): MLP.forward (trainer/models/layers/mlp.py:27-29), nn.Embedding
// (trainer/models/qwen3/modeling_qwen3.py:139,174), scale_gradients_ / clip_grad_norm_ (trainer/utils.py:78-98),
// AdamW (trainer/optim/__init__.py:101-108; torch.optim.AdamW semantics: decoupled decay, bias correction).
#include "common.hip.h"

namespace pm {

__device__ __forceinline__ void unpack8e(const u32x4& v, float* o) {
  o[0] = bf16lo(v.x); o[1] = bf16hi(v.x); o[2] = bf16lo(v.y); o[3] = bf16hi(v.y);
  o[4] = bf16lo(v.z); o[5] = bf16hi(v.z); o[6] = bf16lo(v.w); o[7] = bf16hi(v.w);
}
__device__ __forceinline__ u32x4 pack8e(const float* o) {
  u32x4 v;
  v.x = pack_bf16(o[0], o[1]); v.y = pack_bf16(o[2], o[3]); v.z = pack_bf16(o[4], o[5]); v.w = pack_bf16(o[6], o[7]);
  return v;
}

static inline unsigned capped_grid(int64_t work_items, int block) {
  int64_t g = cdiv(work_items, block);
  const int64_t cap = 256 * 16;  // 256 CUs x 16 workgroups
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// Optimizer kernels (AdamW, SGD, SignSGD): contiguous blocks per workgroup (round 6; rounds 1-5 walked a grid stride, which profiles/r04_adamw_grid_probe.log
// had at 96 % of the achievable streaming rate of THAT walk at every grid size).
// adamw_kernel: contiguous blocks per workgroup.  Block = 16 Ki vectors (256 KiB per f32 stream), halved until there are >= 1024 blocks (small spans keep
// every CU busy: 512 vectors = one double iteration of a workgroup at the least); grid = one workgroup per CU walking them.
static inline void update_blocks(int64_t n_vec, unsigned& grid, int64_t& blk) {
  blk = 16384;
  while (blk > 512 && n_vec / blk < 1024) blk >>= 1;
  const int64_t nb = cdiv(n_vec, blk);
  grid = (unsigned)(nb < 1 ? 1 : nb < 512 ? nb : 512);  // (pm_adamw_step on 2 Gi parameters: 256 workgroups 5.6, 512 6.28, 1024 6.04, 2048 6.18, 4096 6.06 TB/s)
}
static inline unsigned update_grid(int64_t n_vec) {
  int64_t g = cdiv(n_vec, (int64_t)256);
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)(g < 1 ? 1 : g);
}

// ---------------------------------- SwiGLU -------------------------------------------------------
// act = bf16( bf16(silu(gate)) * up ): torch computes silu on bf16 in f32 and rounds, then a bf16 multiply.
// Thread layout (both kernels): blockIdx.x * 256 + threadIdx.x = 16-byte column vector, blockIdx.y = block of kSwigluRows rows walked
// kSwigluUnroll at a time with every load of the batch issued before the first use (no 64-bit divide per vector, 2-3 x kSwigluUnroll
// independent 16-byte loads in flight per lane; each wave touches 1 KB contiguous per array per row).
constexpr int kSwigluRows = 16, kSwigluUnroll = 4;

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, int64_t n_rows, int64_t inter, bf16_t* __restrict__ act) {
  const int64_t vec_per_row = inter >> 3;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= vec_per_row) return;
  const int64_t r0 = (int64_t)blockIdx.y * kSwigluRows;
#pragma unroll 1
  for (int64_t rb = r0; rb < r0 + kSwigluRows && rb < n_rows; rb += kSwigluUnroll) {
    u32x4 gv[kSwigluUnroll], uv[kSwigluUnroll];
#pragma unroll
    for (int j = 0; j < kSwigluUnroll; ++j) {
      const int64_t r = rb + j < n_rows ? rb + j : n_rows - 1;
      const u32x4* gp = reinterpret_cast<const u32x4*>(gu + r * 2 * inter);
      gv[j] = gp[c];
      uv[j] = gp[vec_per_row + c];
    }
#pragma unroll
    for (int j = 0; j < kSwigluUnroll; ++j) {
      if (rb + j >= n_rows) break;
      float g[8], u[8], o[8];
      unpack8e(gv[j], g);
      unpack8e(uv[j], u);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o[k] = swiglu_act(g[k], u[k]);
      }
      reinterpret_cast<u32x4*>(act + (rb + j) * inter)[c] = pack8e(o);
    }
  }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ ga, const bf16_t* __restrict__ gu, int64_t n_rows,
                                                        int64_t inter, bf16_t* __restrict__ ggu) {
  const int64_t vec_per_row = inter >> 3;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= vec_per_row) return;
  const int64_t r0 = (int64_t)blockIdx.y * kSwigluRows;
#pragma unroll 1
  for (int64_t rb = r0; rb < r0 + kSwigluRows && rb < n_rows; rb += kSwigluUnroll) {
    u32x4 gv[kSwigluUnroll], uv[kSwigluUnroll], dv[kSwigluUnroll];
#pragma unroll
    for (int j = 0; j < kSwigluUnroll; ++j) {
      const int64_t r = rb + j < n_rows ? rb + j : n_rows - 1;
      const u32x4* gp = reinterpret_cast<const u32x4*>(gu + r * 2 * inter);
      gv[j] = gp[c];
      uv[j] = gp[vec_per_row + c];
      dv[j] = reinterpret_cast<const u32x4*>(ga + r * inter)[c];
    }
#pragma unroll
    for (int j = 0; j < kSwigluUnroll; ++j) {
      if (rb + j >= n_rows) break;
      float g[8], u[8], d[8], dg[8], du[8];
      unpack8e(gv[j], g);
      unpack8e(uv[j], u);
      unpack8e(dv[j], d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        swiglu_bwd_pair(d[k], g[k], u[k], dg[k], du[k]);
      }
      u32x4* op = reinterpret_cast<u32x4*>(ggu + (rb + j) * 2 * inter);
      op[c] = pack8e(dg);
      op[vec_per_row + c] = pack8e(du);
    }
  }
}

// ---------------------------------- embedding ----------------------------------------------------
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids, int64_t n_tokens,
                                                           int64_t hidden, int64_t vocab, bf16_t* __restrict__ out) {
  const int64_t vec_per_row = hidden >> 3;
  const int64_t total = n_tokens * vec_per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / vec_per_row, c = i - t * vec_per_row;
    int64_t id = ids[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    reinterpret_cast<u32x4*>(out + t * hidden)[c] = reinterpret_cast<const u32x4*>(table + id * hidden)[c];
  }
}

// one workgroup per sorted position; only the first position of each run of equal ids does work, summing the run in order.
// The destination is a WINDOW [elem_lo, elem_hi) of the flattened [vocab, hidden] table (both multiples of 8): `gwin` points at element
// elem_lo.  The whole table is the window [0, vocab * hidden); a ZeRO shard of the embedding bucket is a window that may start and end
// inside a row (trainer/engine.py: the last pass's embedding gradient goes to the rank that owns the rows instead of through the dense
// reduce-scatter).  Rows outside the window, negative ids (padding of the gathered id lists) and ids >= vocab are skipped.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const bf16_t* __restrict__ gout, const int64_t* __restrict__ ids_sorted,
                                                           const int64_t* __restrict__ order, int64_t n_tokens, int64_t hidden, int64_t vocab,
                                                           int64_t elem_lo, int64_t elem_hi, float* __restrict__ gwin) {
  const int64_t p = blockIdx.x;
  const int64_t id = ids_sorted[p];
  if (id < 0 || id >= vocab) return;  // never write outside the table
  if (p > 0 && ids_sorted[p - 1] == id) return;
  const int64_t row0 = id * hidden;
  if (row0 + hidden <= elem_lo || row0 >= elem_hi) return;
  int64_t e = p + 1;
  while (e < n_tokens && ids_sorted[e] == id) ++e;
  const int64_t vec_per_row = hidden >> 3;
  for (int64_t c = threadIdx.x; c < vec_per_row; c += 256) {
    const int64_t at = row0 + c * 8;
    if (at < elem_lo || at >= elem_hi) continue;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int64_t q = p; q < e; ++q) {
      float g[8];
      unpack8e(reinterpret_cast<const u32x4*>(gout + order[q] * hidden)[c], g);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += g[k];
    }
    float4* dst = reinterpret_cast<float4*>(gwin + (at - elem_lo));
    float4 a = dst[0], b = dst[1];
    a.x += acc[0]; a.y += acc[1]; a.z += acc[2]; a.w += acc[3];
    b.x += acc[4]; b.y += acc[5]; b.z += acc[6]; b.w += acc[7];
    dst[0] = a;
    dst[1] = b;
  }
}

// ---------------------------------- gradients / optimizer ---------------------------------------
__global__ __launch_bounds__(256) void grad_accumulate_kernel(float* __restrict__ mg, const bf16_t* __restrict__ g, int64_t n) {
  const int64_t nvec = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float gf[8];
    unpack8e(reinterpret_cast<const u32x4*>(g)[i], gf);
    float4* dst = reinterpret_cast<float4*>(mg + i * 8);
    float4 a = dst[0], b = dst[1];
    a.x += gf[0]; a.y += gf[1]; a.z += gf[2]; a.w += gf[3];
    b.x += gf[4]; b.y += gf[5]; b.z += gf[6]; b.w += gf[7];
    dst[0] = a;
    dst[1] = b;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = nvec * 8 + threadIdx.x; i < n; i += 256) mg[i] += bf16_to_f32(g[i]);
  }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));  // clang ext vector: usable with the non-temporal builtins

// The gradient shard is read exactly once here and once by AdamW (tens of GB: nothing survives in L2 / Infinity Cache), so both
// kernels stream with non-temporal accesses and keep several independent 16-byte loads in flight per lane.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partials) {
  __shared__ float smem[4];
  const int64_t nvec = n >> 2;
  const f32x4v* xv = reinterpret_cast<const f32x4v*>(x);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    const f32x4v a = __builtin_nontemporal_load(xv + i), b = __builtin_nontemporal_load(xv + i + stride),
                 c = __builtin_nontemporal_load(xv + i + 2 * stride), d = __builtin_nontemporal_load(xv + i + 3 * stride);
    s0 = fmaf(a.x, a.x, s0); s1 = fmaf(a.y, a.y, s1); s2 = fmaf(a.z, a.z, s2); s3 = fmaf(a.w, a.w, s3);
    s0 = fmaf(b.x, b.x, s0); s1 = fmaf(b.y, b.y, s1); s2 = fmaf(b.z, b.z, s2); s3 = fmaf(b.w, b.w, s3);
    s0 = fmaf(c.x, c.x, s0); s1 = fmaf(c.y, c.y, s1); s2 = fmaf(c.z, c.z, s2); s3 = fmaf(c.w, c.w, s3);
    s0 = fmaf(d.x, d.x, s0); s1 = fmaf(d.y, d.y, s1); s2 = fmaf(d.z, d.z, s2); s3 = fmaf(d.w, d.w, s3);
  }
  for (; i < nvec; i += stride) {
    const f32x4v v = __builtin_nontemporal_load(xv + i);
    s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
  }
  if (blockIdx.x == 0) {
    for (int64_t t = nvec * 4 + threadIdx.x; t < n; t += 256) s0 = fmaf(x[t], x[t], s0);
  }
  const float tot = block_sum<4>((s0 + s1) + (s2 + s3), smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// out[0] += sum(partials), one workgroup, fixed order.  The fused gradient norm hands it ~1 M tile partials (4 MB): 1024 threads with eight 16-byte
// loads in flight each (a single load per thread and trip made this 1.5 ms of pure latency); the unaligned head / tail go through scalar loads.
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ partials, int n_partials, float* __restrict__ out) {
  __shared__ float smem[16];
  const int t = threadIdx.x;
  int head = (int)((16 - (reinterpret_cast<uintptr_t>(partials) & 15)) & 15) >> 2;
  if (head > n_partials) head = n_partials;
  const int nvec = (n_partials - head) >> 2;
  const f32x4v* pv = reinterpret_cast<const f32x4v*>(partials + head);
  f32x4v a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
  int i = t;
  for (; i + 7 * 1024 < nvec; i += 8 * 1024) {
    f32x4v v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pv[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += v[u];
  }
  for (; i < nvec; i += 1024) a[0] += pv[i];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
  if (t < head) s += partials[t];
  const int tail0 = head + (nvec << 2);
  if (tail0 + t < n_partials) s += partials[tail0 + t];
  const float tot = block_sum<16>(s, smem);
  if (t == 0) out[0] += tot;
}

// One workgroup per segment (offset, length) of a flat f32 buffer: the small tensors between the GEMM weights of a bucket (norm weights, biases), whose
// squared norm the weight-gradient epilogues do not cover (pm_gemm_bf16_sumsq).  Callers split anything large into segments of <= 1 Mi elements.
__global__ __launch_bounds__(256) void sumsq_segments_kernel(const float* __restrict__ base, const int64_t* __restrict__ segs, float* __restrict__ partials) {
  __shared__ float smem[4];
  const int64_t off = segs[2 * blockIdx.x], len = segs[2 * blockIdx.x + 1];
  const float* x = base + off;
  float s0 = 0.f, s1 = 0.f;
  int64_t i = threadIdx.x;
  for (; i + 256 < len; i += 512) {
    const float a = x[i], b = x[i + 256];
    s0 = fmaf(a, a, s0);
    s1 = fmaf(b, b, s1);
  }
  if (i < len) s0 = fmaf(x[i], x[i], s0);
  const float tot = block_sum<4>(s0 + s1, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void clip_coef_kernel(const float* __restrict__ total_sumsq, float grad_scale, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float norm = sqrtf(total_sumsq[0]) * grad_scale;
  if (norm_out) norm_out[0] = norm;
  // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1.0
  float c = (max_norm > 0.f) ? max_norm / (norm + 1e-6f) : 1.0f;
  coef[0] = c > 1.0f ? 1.0f : c;
}

template <bool BLOCKED>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float* __restrict__ g,
                                                   bf16_t* __restrict__ pb, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, float grad_scale, const float* __restrict__ clip_coef,
                                                   int zero_grad, int64_t blk) {
  const float gs = grad_scale * (clip_coef ? clip_coef[0] : 1.0f);
  const float step_size = lr / bc1;
  const int64_t nvec = n >> 2;
  f32x4v* pv = reinterpret_cast<f32x4v*>(p);
  f32x4v* mv = reinterpret_cast<f32x4v*>(m);
  f32x4v* vv4 = reinterpret_cast<f32x4v*>(v);
  f32x4v* gv = reinterpret_cast<f32x4v*>(g);
  const float decay = 1.0f - lr * wd, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  auto update = [&](int64_t i, f32x4v P, f32x4v M, f32x4v V, const f32x4v G) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gs;
      P[k] = P[k] * decay;                              // param.mul_(1 - lr*wd)
      M[k] = M[k] + (gr - M[k]) * omb1;                 // exp_avg.lerp_(grad, 1-beta1)
      V[k] = V[k] * b2 + omb2 * gr * gr;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - step_size * (M[k] / denom);         // param.addcdiv_(exp_avg, denom, -step_size)
    }
    __builtin_nontemporal_store(P, pv + i);
    __builtin_nontemporal_store(M, mv + i);
    __builtin_nontemporal_store(V, vv4 + i);
    if (zero_grad) __builtin_nontemporal_store(f32x4v{0.f, 0.f, 0.f, 0.f}, gv + i);
    if (pb) {
      u32x2 o;
      o.x = pack_bf16(P[0], P[1]);
      o.y = pack_bf16(P[2], P[3]);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(pb) + i);
    }
  };
  // A workgroup walks CONTIGUOUS blocks of `blk` vectors of every stream (256 KiB of each f32 array from 4 Mi vectors up), block b, b + gridDim.x, ...:
  // consecutive iterations touch adjacent 4 KiB pieces of each of the 8 streams instead of pieces a whole grid stride (16 MiB) apart.  Same bytes, same
  // per-element arithmetic; tools/probes/adamw_stream_probe.hip: 5.7-5.8 -> 6.2-6.3 TB/s on this access pattern (DRAM page / TLB locality); this kernel on
  // 2 Gi parameters 5.82-5.89 -> 6.19 TB/s in one process, the step -1.8 ms.
  if constexpr (!BLOCKED) {  // the grid-stride walk of rounds 1-5 (PM_ADAMW_BLOCKED=0: A/B)
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < nvec; i += 2 * stride) {
      const f32x4v P0 = __builtin_nontemporal_load(pv + i), M0 = __builtin_nontemporal_load(mv + i), V0 = __builtin_nontemporal_load(vv4 + i),
                   G0 = __builtin_nontemporal_load(gv + i);
      const f32x4v P1 = __builtin_nontemporal_load(pv + i + stride), M1 = __builtin_nontemporal_load(mv + i + stride),
                   V1 = __builtin_nontemporal_load(vv4 + i + stride), G1 = __builtin_nontemporal_load(gv + i + stride);
      update(i, P0, M0, V0, G0);
      update(i + stride, P1, M1, V1, G1);
    }
    for (; i < nvec; i += stride)
      update(i, __builtin_nontemporal_load(pv + i), __builtin_nontemporal_load(mv + i), __builtin_nontemporal_load(vv4 + i), __builtin_nontemporal_load(gv + i));
  } else
  for (int64_t b = blockIdx.x; b * blk < nvec; b += gridDim.x) {
    const int64_t e = (b + 1) * blk < nvec ? (b + 1) * blk : nvec;
    int64_t i = b * blk + threadIdx.x;
    for (; i + 256 < e; i += 512) {  // two independent vectors of every stream in flight
      const f32x4v P0 = __builtin_nontemporal_load(pv + i), M0 = __builtin_nontemporal_load(mv + i), V0 = __builtin_nontemporal_load(vv4 + i),
                   G0 = __builtin_nontemporal_load(gv + i);
      const f32x4v P1 = __builtin_nontemporal_load(pv + i + 256), M1 = __builtin_nontemporal_load(mv + i + 256),
                   V1 = __builtin_nontemporal_load(vv4 + i + 256), G1 = __builtin_nontemporal_load(gv + i + 256);
      update(i, P0, M0, V0, G0);
      update(i + 256, P1, M1, V1, G1);
    }
    if (i < e)
      update(i, __builtin_nontemporal_load(pv + i), __builtin_nontemporal_load(mv + i), __builtin_nontemporal_load(vv4 + i), __builtin_nontemporal_load(gv + i));
  }
  if (blockIdx.x == 0) {
    for (int64_t i = nvec * 4 + threadIdx.x; i < n; i += 256) {
      const float gr = g[i] * gs;
      float pp = p[i] * (1.0f - lr * wd);
      const float mm = m[i] + (gr - m[i]) * (1.0f - b1);
      const float vv = v[i] * b2 + (1.0f - b2) * gr * gr;
      pp = pp - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (zero_grad) g[i] = 0.f;
      if (pb) pb[i] = f32_to_bf16(pp);
    }
  }
}

constexpr int kTrTile = 64, kTrLd = 66;  // 2-D transpose tiles (transpose16_kernel): 64 x 64 elements, rows 33 dwords apart

template <bool SIGN>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ buf, float* __restrict__ g, bf16_t* __restrict__ pb,
                                                 int64_t n, float lr, float momentum, float wd, int nesterov, int first_step, float grad_scale,
                                                 const float* __restrict__ clip_coef, int zero_grad, int64_t blk) {
  const float gs = grad_scale * (clip_coef ? clip_coef[0] : 1.0f);
  auto one = [&](float P, float& B, float G) -> float {
    float gr = G * gs;
    if constexpr (SIGN) {
      if (wd > 0.f) P = P + (-lr * wd) * P;
      const float sg = gr > 0.f ? 1.0f : (gr < 0.f ? -1.0f : 0.0f);
      return P + (-lr) * sg;
    } else {
      if (wd != 0.f) gr = gr + wd * P;
      if (momentum != 0.f) {
        B = first_step ? gr : momentum * B + gr;
        gr = nesterov ? gr + momentum * B : B;
      }
      return P + (-lr) * gr;
    }
  };
  const bool has_buf = !SIGN && momentum != 0.f;
  const int64_t nvec = n >> 2;
  f32x4v* pv = reinterpret_cast<f32x4v*>(p);
  f32x4v* bv = reinterpret_cast<f32x4v*>(buf);
  f32x4v* gv = reinterpret_cast<f32x4v*>(g);
  for (int64_t b = blockIdx.x; b * blk < nvec; b += gridDim.x)  // contiguous blocks per workgroup, as adamw_kernel
  for (int64_t i = b * blk + threadIdx.x, e = (b + 1) * blk < nvec ? (b + 1) * blk : nvec; i < e; i += 256) {
    f32x4v P = __builtin_nontemporal_load(pv + i);
    const f32x4v G = __builtin_nontemporal_load(gv + i);
    f32x4v B = {0.f, 0.f, 0.f, 0.f};
    if (has_buf && !first_step) B = __builtin_nontemporal_load(bv + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float b = B[k];
      P[k] = one(P[k], b, G[k]);
      B[k] = b;
    }
    __builtin_nontemporal_store(P, pv + i);
    if (has_buf) __builtin_nontemporal_store(B, bv + i);
    if (zero_grad) __builtin_nontemporal_store(f32x4v{0.f, 0.f, 0.f, 0.f}, gv + i);
    if (pb) {
      u32x2 o;
      o.x = pack_bf16(P[0], P[1]);
      o.y = pack_bf16(P[2], P[3]);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(pb) + i);
    }
  }
  if (blockIdx.x == 0) {
    for (int64_t i = nvec * 4 + threadIdx.x; i < n; i += 256) {
      float b = (has_buf && !first_step) ? buf[i] : 0.f;
      const float pp = one(p[i], b, g[i]);
      p[i] = pp;
      if (has_buf) buf[i] = b;
      if (zero_grad) g[i] = 0.f;
      if (pb) pb[i] = f32_to_bf16(pp);
    }
  }
}

// ================================= 2-D transpose (16-bit elements) ============================
// dst[c][r] = src[r][c].  One 64x64 tile per workgroup through LDS: 16-byte global loads along src rows, tile rows padded to 33
// dwords so that the column gathers (8 x ds_read_u16 per 16-byte output vector) are bank-conflict free, 16-byte global stores
// along dst rows.  HBM-bound: 2 bytes read + 2 written per element.  rows, cols and both leading dimensions are multiples of 8.

__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ src, int64_t rows, int64_t cols, int64_t src_ld,
                                                         uint16_t* __restrict__ dst, int64_t dst_ld) {
  __shared__ uint32_t tile32[kTrTile * kTrLd / 2];
  const uint16_t* tile = reinterpret_cast<const uint16_t*>(tile32);
  const int64_t r0 = (int64_t)blockIdx.y * kTrTile, c0 = (int64_t)blockIdx.x * kTrTile;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = threadIdx.x + 256 * k;
    const int r = vi >> 3, cv = vi & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < rows && c0 + cv * 8 < cols) v = *reinterpret_cast<const u32x4*>(src + (r0 + r) * src_ld + c0 + cv * 8);
    uint32_t* t = tile32 + r * (kTrLd / 2) + cv * 4;
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ch = threadIdx.x + 256 * k;
    const int j = ch & 7, c = ch >> 3;
    const int64_t rr = r0 + j * 8;
    if (c0 + c < cols && rr < rows) {
      uint32_t e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = tile[(j * 8 + i) * kTrLd + c];
      uint16_t* d = dst + (c0 + c) * dst_ld + rr;
      if (rr + 8 <= rows) {
        u32x4 o = {e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        *reinterpret_cast<u32x4*>(d) = o;
      } else {  // ragged last vector of a dst row (rows is not a multiple of 8: e.g. an unpadded token count)
        for (int i = 0; i < 8 && rr + i < rows; ++i) d[i] = (uint16_t)e[i];
      }
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" int pm_transpose_2d(const void* src, int64_t rows, int64_t cols, int64_t src_ld, void* dst, int64_t dst_ld, void* stream) {
  PM_REQUIRE(src && dst, PM_ERR_INVALID_ARG, "pm_transpose_2d: null pointer");
  PM_REQUIRE(rows >= 0 && cols >= 0 && cols % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0 && src_ld >= cols && dst_ld >= rows, PM_ERR_UNSUPPORTED,
             "pm_transpose_2d: rows=%lld cols=%lld src_ld=%lld dst_ld=%lld (cols and both leading dimensions multiples of 8, ld >= extent)",
             (long long)rows, (long long)cols, (long long)src_ld, (long long)dst_ld);
  if (rows == 0 || cols == 0) return PM_OK;
  const int64_t gx = cdiv(cols, (int64_t)kTrTile), gy = cdiv(rows, (int64_t)kTrTile);
  PM_REQUIRE(gy <= 65535, PM_ERR_UNSUPPORTED, "pm_transpose_2d: rows=%lld too large", (long long)rows);
  hipLaunchKernelGGL(transpose16_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, as_stream(stream), (const uint16_t*)src, rows, cols, src_ld,
                     (uint16_t*)dst, dst_ld);
  PM_CHECK_LAUNCH("pm_transpose_2d");
  return PM_OK;
}

extern "C" int pm_swiglu_fwd(const void* gate_up, int64_t n_rows, int64_t inter, void* act, void* stream) {
  PM_REQUIRE(gate_up && act, PM_ERR_INVALID_ARG, "pm_swiglu_fwd: null pointer");
  PM_REQUIRE(n_rows >= 0 && inter > 0 && inter % 8 == 0, PM_ERR_UNSUPPORTED, "pm_swiglu_fwd: inter=%lld must be a multiple of 8", (long long)inter);
  if (n_rows == 0) return PM_OK;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)cdiv(inter >> 3, (int64_t)256), (unsigned)cdiv(n_rows, (int64_t)kSwigluRows)), dim3(256), 0,
                     as_stream(stream), (const bf16_t*)gate_up, n_rows, inter, (bf16_t*)act);
  PM_CHECK_LAUNCH("pm_swiglu_fwd");
  return PM_OK;
}

extern "C" int pm_swiglu_bwd(const void* grad_act, const void* gate_up, int64_t n_rows, int64_t inter, void* grad_gate_up, void* stream) {
  PM_REQUIRE(grad_act && gate_up && grad_gate_up, PM_ERR_INVALID_ARG, "pm_swiglu_bwd: null pointer");
  PM_REQUIRE(n_rows >= 0 && inter > 0 && inter % 8 == 0, PM_ERR_UNSUPPORTED, "pm_swiglu_bwd: inter=%lld must be a multiple of 8", (long long)inter);
  if (n_rows == 0) return PM_OK;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)cdiv(inter >> 3, (int64_t)256), (unsigned)cdiv(n_rows, (int64_t)kSwigluRows)), dim3(256), 0,
                     as_stream(stream), (const bf16_t*)grad_act, (const bf16_t*)gate_up, n_rows, inter, (bf16_t*)grad_gate_up);
  PM_CHECK_LAUNCH("pm_swiglu_bwd");
  return PM_OK;
}

extern "C" int pm_embedding_fwd(const void* table, const int64_t* ids, int64_t n_tokens, int64_t hidden, int64_t vocab, void* out, void* stream) {
  PM_REQUIRE(table && ids && out, PM_ERR_INVALID_ARG, "pm_embedding_fwd: null pointer");
  PM_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden % 8 == 0 && vocab > 0, PM_ERR_UNSUPPORTED, "pm_embedding_fwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(capped_grid(n_tokens * (hidden >> 3), 256)), dim3(256), 0, as_stream(stream), (const bf16_t*)table,
                     ids, n_tokens, hidden, vocab, (bf16_t*)out);
  PM_CHECK_LAUNCH("pm_embedding_fwd");
  return PM_OK;
}

extern "C" int pm_embedding_bwd(const void* grad_out, const int64_t* ids_sorted, const int64_t* order, int64_t n_tokens, int64_t hidden,
                                int64_t vocab, float* grad_table, void* stream) {
  PM_REQUIRE(grad_out && ids_sorted && order && grad_table, PM_ERR_INVALID_ARG, "pm_embedding_bwd: null pointer");
  PM_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden % 8 == 0, PM_ERR_UNSUPPORTED, "pm_embedding_bwd: bad sizes");
  if (n_tokens == 0) return PM_OK;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)n_tokens), dim3(256), 0, as_stream(stream), (const bf16_t*)grad_out, ids_sorted, order,
                     n_tokens, hidden, vocab, (int64_t)0, vocab * hidden, grad_table);
  PM_CHECK_LAUNCH("pm_embedding_bwd");
  return PM_OK;
}

extern "C" int pm_embedding_bwd_window(const void* grad_out, const int64_t* ids_sorted, const int64_t* order, int64_t n_tokens, int64_t hidden,
                                       int64_t vocab, int64_t elem_lo, int64_t elem_count, float* grad_window, void* stream) {
  PM_REQUIRE(n_tokens >= 0 && hidden > 0 && hidden % 8 == 0 && vocab > 0, PM_ERR_UNSUPPORTED, "pm_embedding_bwd_window: bad sizes");
  PM_REQUIRE(elem_lo >= 0 && elem_count >= 0 && elem_lo % 8 == 0 && elem_count % 8 == 0, PM_ERR_INVALID_ARG,
             "pm_embedding_bwd_window: the window must start and end on a multiple of 8 elements");
  if (n_tokens == 0 || elem_count == 0) return PM_OK;
  PM_REQUIRE(grad_out && ids_sorted && order && grad_window, PM_ERR_INVALID_ARG, "pm_embedding_bwd_window: null pointer");
  PM_REQUIRE((reinterpret_cast<uintptr_t>(grad_window) & 15) == 0, PM_ERR_INVALID_ARG, "pm_embedding_bwd_window: grad_window must be 16-byte aligned");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)n_tokens), dim3(256), 0, as_stream(stream), (const bf16_t*)grad_out, ids_sorted, order,
                     n_tokens, hidden, vocab, elem_lo, elem_lo + elem_count, grad_window);
  PM_CHECK_LAUNCH("pm_embedding_bwd_window");
  return PM_OK;
}

extern "C" int pm_grad_accumulate(float* main_grad, const void* grad_bf16, int64_t n, void* stream) {
  PM_REQUIRE(main_grad && grad_bf16 && n >= 0, PM_ERR_INVALID_ARG, "pm_grad_accumulate: bad args");
  PM_REQUIRE((reinterpret_cast<uintptr_t>(main_grad) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_bf16) & 15) == 0, PM_ERR_INVALID_ARG,
             "pm_grad_accumulate: buffers must be 16-byte aligned");
  if (n == 0) return PM_OK;
  hipLaunchKernelGGL(grad_accumulate_kernel, dim3(capped_grid(n >> 3, 256)), dim3(256), 0, as_stream(stream), main_grad, (const bf16_t*)grad_bf16, n);
  PM_CHECK_LAUNCH("pm_grad_accumulate");
  return PM_OK;
}

static int sumsq_grid(int64_t n) { return (int)capped_grid(n >> 2, 256); }

extern "C" int64_t pm_sumsq_workspace_bytes(int64_t n) { return n <= 0 ? 0 : (int64_t)sumsq_grid(n) * (int64_t)sizeof(float); }

extern "C" int pm_sumsq(const float* x, int64_t n, float* out_sumsq, void* workspace, int64_t workspace_bytes, void* stream) {
  PM_REQUIRE(x && out_sumsq && n >= 0, PM_ERR_INVALID_ARG, "pm_sumsq: bad args");
  PM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, PM_ERR_INVALID_ARG, "pm_sumsq: x must be 16-byte aligned");
  if (n == 0) return PM_OK;
  const int grid = sumsq_grid(n);
  PM_REQUIRE(workspace && workspace_bytes >= (int64_t)grid * 4, PM_ERR_WORKSPACE, "pm_sumsq: workspace too small");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, st, x, n, (float*)workspace);
  PM_CHECK_LAUNCH("pm_sumsq");
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)workspace, grid, out_sumsq);
  PM_CHECK_LAUNCH("pm_sumsq(final)");
  return PM_OK;
}

extern "C" int pm_sumsq_segments(const float* base, const int64_t* segments, int32_t n_segments, float* partials, void* stream) {
  if (n_segments == 0) return PM_OK;
  PM_REQUIRE(base && segments && partials && n_segments > 0, PM_ERR_INVALID_ARG, "pm_sumsq_segments: bad args");
  hipLaunchKernelGGL(sumsq_segments_kernel, dim3((unsigned)n_segments), dim3(256), 0, as_stream(stream), base, segments, partials);
  PM_CHECK_LAUNCH("pm_sumsq_segments");
  return PM_OK;
}

// out[0] += sum(partials) in a fixed order (one workgroup): the last stage of the fused gradient norm
extern "C" int pm_sum_partials(const float* partials, int64_t n, float* out, void* stream) {
  PM_REQUIRE(partials && out && n >= 0 && n < (1ll << 31), PM_ERR_INVALID_ARG, "pm_sum_partials: bad args");
  if (n == 0) return PM_OK;
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partials, (int)n, out);
  PM_CHECK_LAUNCH("pm_sum_partials");
  return PM_OK;
}

extern "C" int pm_clip_coef(const float* total_sumsq, float grad_scale, float max_norm, float* clip_coef, float* grad_norm_out, void* stream) {
  PM_REQUIRE(total_sumsq && clip_coef, PM_ERR_INVALID_ARG, "pm_clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, as_stream(stream), total_sumsq, grad_scale, max_norm, clip_coef, grad_norm_out);
  PM_CHECK_LAUNCH("pm_clip_coef");
  return PM_OK;
}

extern "C" int pm_sgd_step(float* master, float* momentum_buf, float* grad, void* param_bf16, int64_t n, float lr, float momentum, float weight_decay,
                           int32_t nesterov, int64_t step, float grad_scale, const float* clip_coef, int32_t zero_grad, void* stream) {
  PM_REQUIRE(master && grad && n >= 0 && step >= 1 && momentum >= 0.f, PM_ERR_INVALID_ARG, "pm_sgd_step: bad args");
  PM_REQUIRE(momentum == 0.f || momentum_buf, PM_ERR_INVALID_ARG, "pm_sgd_step: momentum %g needs a momentum buffer", (double)momentum);
  PM_REQUIRE(!nesterov || momentum > 0.f, PM_ERR_INVALID_ARG, "pm_sgd_step: Nesterov momentum requires a momentum (torch.optim.SGD)");
  PM_REQUIRE(((reinterpret_cast<uintptr_t>(master) | reinterpret_cast<uintptr_t>(momentum_buf) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(param_bf16) & 7) == 0,
             PM_ERR_INVALID_ARG, "pm_sgd_step: buffers must be 16-byte (f32) / 8-byte (bf16) aligned");
  if (n == 0) return PM_OK;
  unsigned grid;
  int64_t blk;
  update_blocks(n >> 2, grid, blk);
  hipLaunchKernelGGL(sgd_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), master, momentum_buf, grad, (bf16_t*)param_bf16, n,
                     lr, momentum, weight_decay, (int)nesterov, (int)(step == 1), grad_scale, clip_coef, (int)zero_grad, blk);
  PM_CHECK_LAUNCH("pm_sgd_step");
  return PM_OK;
}

extern "C" int pm_sign_sgd_step(float* master, float* grad, void* param_bf16, int64_t n, float lr, float weight_decay, float grad_scale,
                                const float* clip_coef, int32_t zero_grad, void* stream) {
  PM_REQUIRE(master && grad && n >= 0, PM_ERR_INVALID_ARG, "pm_sign_sgd_step: bad args");
  PM_REQUIRE(((reinterpret_cast<uintptr_t>(master) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0 && (reinterpret_cast<uintptr_t>(param_bf16) & 7) == 0,
             PM_ERR_INVALID_ARG, "pm_sign_sgd_step: buffers must be 16-byte (f32) / 8-byte (bf16) aligned");
  if (n == 0) return PM_OK;
  unsigned grid;
  int64_t blk;
  update_blocks(n >> 2, grid, blk);
  hipLaunchKernelGGL(sgd_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), master, (float*)nullptr, grad, (bf16_t*)param_bf16,
                     n, lr, 0.f, weight_decay, 0, 0, grad_scale, clip_coef, (int)zero_grad, blk);
  PM_CHECK_LAUNCH("pm_sign_sgd_step");
  return PM_OK;
}

extern "C" int pm_adamw_step(float* master, float* exp_avg, float* exp_avg_sq, float* grad, void* param_bf16, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, float grad_scale, const float* clip_coef, int32_t zero_grad,
                             void* stream) {
  PM_REQUIRE(master && exp_avg && exp_avg_sq && grad && n >= 0 && step >= 1, PM_ERR_INVALID_ARG, "pm_adamw_step: bad args");
  PM_REQUIRE(((reinterpret_cast<uintptr_t>(master) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq) |
               reinterpret_cast<uintptr_t>(grad)) & 15) == 0 && (reinterpret_cast<uintptr_t>(param_bf16) & 7) == 0,
             PM_ERR_INVALID_ARG, "pm_adamw_step: buffers must be 16-byte (f32) / 8-byte (bf16) aligned");
  if (n == 0) return PM_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  unsigned grid;
  int64_t blk;
  update_blocks(n >> 2, grid, blk);
#if PM_VARIANTS
  static const bool blocked = !(getenv("PM_ADAMW_BLOCKED") && atoi(getenv("PM_ADAMW_BLOCKED")) == 0);  // A/B: the grid-stride walk (5.82-5.89 against 6.19 TB/s in one process)
#else
  constexpr bool blocked = true;
#endif
  if (blocked)
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), master, exp_avg, exp_avg_sq, grad,
                       (bf16_t*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, clip_coef, zero_grad, blk);
  else
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(update_grid(n >> 2)), dim3(256), 0, as_stream(stream), master, exp_avg, exp_avg_sq, grad,
                       (bf16_t*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, clip_coef, zero_grad, blk);
  PM_CHECK_LAUNCH("pm_adamw_step");
  return PM_OK;
}

