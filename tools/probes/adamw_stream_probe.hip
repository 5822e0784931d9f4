// How fast can the AdamW access pattern go on MI355X?  4 f32 read streams (p, m, v, g) + 3 f32 write streams (p, m, v) + one bf16 write stream = 30 B per
// parameter (csrc/elementwise.hip adamw_kernel, 6.0 TB/s in the step).  Variants of the SAME arithmetic-free skeleton (the update is replaced by three
// fmas per element: the kernel is nowhere near VALU-bound), 4 Gi... 2 G parameters, grid-stride unless noted:
//   0  the product's form: non-temporal loads and stores, 2 vectors of every stream in flight per lane
//   1  plain loads, non-temporal stores
//   2  plain loads and stores
//   3  non-temporal, 4 vectors in flight
//   4  non-temporal, each workgroup walks ONE contiguous 1 MiB block of every stream at a time (256 lanes x 16 B x 256 iterations) instead of the grid stride
//   5  non-temporal, 1 vector in flight
//   6  like 0 with 512-thread workgroups
// hipcc --offload-arch=gfx950 -O3 tools/probes/adamw_stream_probe.hip -o tools/probes/adamw_stream_probe && tools/probes/adamw_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) unsigned u2;

template <bool NT>
__device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(f4 v, f4* p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT>
__device__ __forceinline__ void stb(u2 v, u2* p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NTL, bool NTS>
__device__ __forceinline__ void upd(int64_t i, f4 P, f4 M, f4 V, f4 G, f4* p, f4* m, f4* v, u2* pb) {
  M = M + (G - M) * 0.1f;
  V = V * 0.999f + G * G * 0.001f;
  P = P * 0.9999f - M * 1e-3f / (V + 1e-8f);
  st<NTS>(P, p + i); st<NTS>(M, m + i); st<NTS>(V, v + i);
  u2 o;
  o.x = (__float_as_uint(P[0]) >> 16) | (__float_as_uint(P[1]) & 0xffff0000u);
  o.y = (__float_as_uint(P[2]) >> 16) | (__float_as_uint(P[3]) & 0xffff0000u);
  stb<NTS>(o, pb + i);
}

template <bool NTL, bool NTS, int INFLIGHT>
__global__ void k_stride(f4* p, f4* m, f4* v, const f4* g, u2* pb, int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (INFLIGHT - 1) * stride < nvec; i += INFLIGHT * stride) {
    f4 P[INFLIGHT], M[INFLIGHT], V[INFLIGHT], G[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      P[j] = ld<NTL>(p + i + j * stride); M[j] = ld<NTL>(m + i + j * stride); V[j] = ld<NTL>(v + i + j * stride); G[j] = ld<NTL>(g + i + j * stride);
    }
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) upd<NTL, NTS>(i + j * stride, P[j], M[j], V[j], G[j], p, m, v, pb);
  }
  for (; i < nvec; i += stride) upd<NTL, NTS>(i, ld<NTL>(p + i), ld<NTL>(m + i), ld<NTL>(v + i), ld<NTL>(g + i), p, m, v, pb);
}

__global__ void k_block(f4* p, f4* m, f4* v, const f4* g, u2* pb, int64_t nvec, int64_t blk) {  // contiguous blocks of `blk` vectors per workgroup turn
  for (int64_t b = blockIdx.x; b * blk < nvec; b += gridDim.x) {
    const int64_t e = (b + 1) * blk < nvec ? (b + 1) * blk : nvec;
    for (int64_t i = b * blk + threadIdx.x; i < e; i += 2 * 256) {
      const bool two = i + 256 < e;
      f4 P0 = ld<true>(p + i), M0 = ld<true>(m + i), V0 = ld<true>(v + i), G0 = ld<true>(g + i);
      f4 P1, M1, V1, G1;
      if (two) { P1 = ld<true>(p + i + 256); M1 = ld<true>(m + i + 256); V1 = ld<true>(v + i + 256); G1 = ld<true>(g + i + 256); }
      upd<true, true>(i, P0, M0, V0, G0, p, m, v, pb);
      if (two) upd<true, true>(i + 256, P1, M1, V1, G1, p, m, v, pb);
    }
  }
}

int main() {
  const int64_t n = 2ll << 30, nvec = n / 4;
  f4 *p, *m, *v, *g; u2* pb;
  hipMalloc(&p, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&g, n * 4); hipMalloc(&pb, n * 2);
  hipMemset(p, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(g, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"nt loads + nt stores, 2 in flight (product)", "plain loads, nt stores", "plain loads + stores", "nt, 4 in flight",
                         "nt, contiguous 1 MiB blocks per workgroup", "nt, 1 in flight", "nt, 2 in flight, 512-thread workgroups"};
  for (int rep = 0; rep < 2; ++rep)
    for (int var = 0; var < 7; ++var) {
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        const dim3 grid(4096), blk(var == 6 ? 512 : 256);
        switch (var) {
          case 0: case 6: hipLaunchKernelGGL((k_stride<true, true, 2>), grid, blk, 0, 0, p, m, v, g, pb, nvec); break;
          case 1: hipLaunchKernelGGL((k_stride<false, true, 2>), grid, blk, 0, 0, p, m, v, g, pb, nvec); break;
          case 2: hipLaunchKernelGGL((k_stride<false, false, 2>), grid, blk, 0, 0, p, m, v, g, pb, nvec); break;
          case 3: hipLaunchKernelGGL((k_stride<true, true, 4>), grid, blk, 0, 0, p, m, v, g, pb, nvec); break;
          case 4: hipLaunchKernelGGL(k_block, dim3(2048), blk, 0, 0, p, m, v, g, pb, nvec, (int64_t)65536); break;
          default: hipLaunchKernelGGL((k_stride<true, true, 1>), grid, blk, 0, 0, p, m, v, g, pb, nvec); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      if (rep == 1) printf("variant %d  %-48s %7.3f ms  %5.2f TB/s\n", var, names[var], best, 30.0 * n / (best * 1e-3) / 1e12);
    }
  // block size x grid sweep of the contiguous-block form
  for (int64_t bv : {4096ll, 16384ll, 65536ll, 262144ll, 1048576ll})
    for (int gr : {256, 512, 1024, 2048, 4096}) {
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_block, dim3(gr), dim3(256), 0, 0, p, m, v, g, pb, nvec, bv);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("block %8lld vectors (%6lld KiB per stream)  grid %4d  %7.3f ms  %5.2f TB/s\n", (long long)bv, (long long)(bv * 16 / 1024), gr, best, 30.0 * n / (best * 1e-3) / 1e12);
    }
  return 0;
}
