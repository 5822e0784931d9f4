// What one s_barrier hand-off costs on gfx950, by workgroup shape and by what the waves do between barriers.  Question behind it
// (profiles/r03_attention.md): the 8-wave ping-pong attention kernel measured ~350 cycles per EMPTY barrier-delimited phase, while the GEMM core
// of csrc/gemm.hip runs 256-cycle MFMA phases at ~90 % matrix-pipe occupancy through the same barrier pattern.
// hipcc --offload-arch=gfx950 -O3 tools/probes/barrier_probe.hip -o tools/probes/barrier_probe && tools/probes/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: bare barriers.  1: group 0 / group 1 staggered by one barrier (the ping-pong pattern), nothing between.  2: staggered, the "compute" group
// issues NM MFMAs per phase while the other group idles.  3: like 2 plus s_setprio 1 around the MFMAs.  4: staggered, 24 dependent-free VALU per phase
// in the compute group.
template <int MODE, int NM>
__global__ __launch_bounds__(512, 1) void probe(long long* out, int iters, float seed) {
  extern __shared__ unsigned char smem[];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = (blockDim.x == 512) ? (w >> 2) : 0;
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{seed, seed, seed, seed};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * i;
  if (MODE >= 1 && g == 1) __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    // phase A: group 0 computes, group 1 idles; phase B: the other way round (in this wave's own time: compute then idle)
    if (MODE == 2 || MODE == 3) {
      if (MODE == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NM; ++i) c[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i & 7], 0, 0, 0);
      if (MODE == 3) __builtin_amdgcn_s_setprio(0);
    }
    if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 24; ++i) v[i & 7] = v[i & 7] * seed + 1.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (MODE >= 1 && g == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + v[i];
  if (s == 12345.678f) out[1000] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, int NM>
static void run(const char* what, int threads, size_t lds, long long* d) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)probe<MODE, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, NM>), dim3(256), dim3(threads), lds, 0, d, iters, 1.0f);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-78s threads %3d  lds %6zu : %7.1f cycles per barrier (2 per iteration)\n", what, threads, lds, (double)h / iters / 2);
}

int main() {
  long long* d;
  hipMalloc(&d, 8192 * 8);
  run<0, 0>("bare barriers", 512, 0, d);
  run<0, 0>("bare barriers", 256, 0, d);
  run<0, 0>("bare barriers", 128, 0, d);
  run<0, 0>("bare barriers", 64, 0, d);
  run<0, 0>("bare barriers, 96 KiB LDS", 512, 96 * 1024, d);
  run<1, 0>("two groups staggered by one barrier, nothing between", 512, 96 * 1024, d);
  run<2, 16>("staggered, 16 MFMA (16x16x32) in every second phase of each group", 512, 96 * 1024, d);
  run<3, 16>("same with s_setprio 1 around the MFMAs", 512, 96 * 1024, d);
  run<2, 32>("staggered, 32 MFMA in every second phase", 512, 96 * 1024, d);
  run<3, 32>("same with s_setprio 1", 512, 96 * 1024, d);
  run<4, 0>("staggered, 24 VALU in every second phase", 512, 96 * 1024, d);
  return 0;
}
