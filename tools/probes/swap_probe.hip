// Probe of v_permlane32_swap / v_permlane16_swap as emitted by hipcc: the raw rows below call the builtin with the SAME value twice
// (both results come back equal: the pitfall), xrow_max uses two distinct VGPRs.
// hipcc --offload-arch=gfx950 -O3 tools/probes/swap_probe.hip -o /tmp/swap_probe && /tmp/swap_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__device__ float xrow_max_swap(float v) {  // the version used by attention_kernels.hip.h: inline asm, both registers read-write
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a = fmaxf(a, b);
  b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__global__ void k(const float* in, float* out, float* raw) {
  const float v = in[threadIdx.x];
  out[threadIdx.x] = xrow_max_swap(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const u2 a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  raw[threadIdx.x] = __builtin_bit_cast(float, a.x);
  raw[64 + threadIdx.x] = __builtin_bit_cast(float, a.y);
  const u2 b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  raw[128 + threadIdx.x] = __builtin_bit_cast(float, b.x);
  raw[192 + threadIdx.x] = __builtin_bit_cast(float, b.y);
}
int main() {
  float h[64], o[64], r[256], *di, *dout, *dr;
  for (int i = 0; i < 64; ++i) h[i] = (float)i;
  hipMalloc(&di, 256); hipMalloc(&dout, 256); hipMalloc(&dr, 1024);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, dr);
  hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost); hipMemcpy(r, dr, 1024, hipMemcpyDeviceToHost);
  printf("in = lane id.  xrow_max (want 48+lr):"); for (int i = 0; i < 64; ++i) printf(" %g", o[i]); printf("\n");
  const char* names[4] = {"permlane32.x", "permlane32.y", "permlane16.x", "permlane16.y"};
  for (int j = 0; j < 4; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; ++i) printf(" %g", r[j * 64 + i]); printf("\n"); }
  return 0;
}
