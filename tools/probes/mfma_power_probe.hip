// Sustained MFMA throughput under the chip's power limit, by instruction shape and operand data: v_mfma_f32_32x32x16_bf16 vs
// v_mfma_f32_16x16x32_bf16, register operands only (no memory traffic), zero vs random operands, ~10 ms launches, all 256 CUs x 4 SIMDs x 1|2 waves.
// Question behind it (profiles/r03_gemm.md): the own GEMM core holds the matrix pipe at 88 % but is clocked at 1.59 GHz where hipBLASLt's 16x16x32
// kernel runs 1.81 GHz at 86 % — is the 32x32x16 shape itself the more power-hungry one (it moves 0.625 B of register data per MAC against 0.5)?
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const u32x4* __restrict__ ops, float* __restrict__ sink, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = ops[(t * 8 + i) & 0xffff];
    b[i] = ops[(t * 8 + 4 + i) & 0xffff];
  }
  if (SHAPE == 32) {
    f32x16 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 1) & 3]), c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) s += c[i][r];
    sink[t] = s;
  } else {
    f32x4 c[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 2) & 3]), c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    sink[t] = s;
  }
}

int main() {
  const int n_ops = 1 << 16;
  std::vector<unsigned> h(n_ops * 4);
  u32x4* d_ops;
  float* d_sink;
  hipMalloc(&d_ops, n_ops * 16);
  hipMalloc(&d_sink, 256 * 2 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (auto& x : h) {  // bf16 pairs: zero, or N(0,1)-like random values (random sign / exponent near 1 / mantissa)
      auto one = []() { unsigned m = rand() & 0x7f, e = 124 + (rand() % 6), s = rand() & 1; return (s << 15) | (e << 7) | m; };
      x = data ? (one() | (one() << 16)) : 0u;
    }
    hipMemcpy(d_ops, h.data(), n_ops * 16, hipMemcpyHostToDevice);
    for (int wpb = 256; wpb <= 512; wpb += 256) {  // 4 waves (one per SIMD) or 8 waves (two per SIMD) per CU
      for (int shape = 16; shape <= 32; shape += 16) {
        // equal MAC count per launch: a 32x32x16 instruction = 2 x a 16x16x32; 8 vs 32 instructions per iteration -> iters ratio 2
        const int iters = shape == 32 ? 40000 : 20000;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(e0);
          if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(256), dim3(wpb), 0, 0, d_ops, d_sink, iters);
          else hipLaunchKernelGGL(mfma_loop<16>, dim3(256), dim3(wpb), 0, 0, d_ops, d_sink, iters);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;  // first launch = warm-up
        }
        const double flops = 256.0 * (wpb / 64) * (shape == 32 ? 8.0 * 32768 : 32.0 * 16384) * iters;
        printf("{\"probe\": \"mfma_power\", \"data\": \"%s\", \"waves_per_cu\": %d, \"shape\": \"%s\", \"ms\": %.3f, \"TFLOPs\": %.1f}\n", data ? "random" : "zero", wpb / 64,
               shape == 32 ? "32x32x16" : "16x16x32", best, flops / best / 1e9);
      }
    }
  }
  return 0;
}
