// common.hip.h — shared device/host helpers for libprime_mi355x (gfx950 only; wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/prime_mi355x.h"

namespace pm {

constexpr int kWave = 64;  // CDNA wavefront; hard-coded on purpose (guide §1)

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define PM_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      ::pm::set_error(__VA_ARGS__);            \
      return (code);                           \
    }                                          \
  } while (0)

#define PM_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      ::pm::set_error("%s: launch failed: %s", name, hipGetErrorString(_e));    \
      return PM_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 (raw bits; RNE, NaN preserved) --------------------------------------------
typedef uint16_t bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-byte vector (clang ext vector: usable with nontemporal builtins)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// f32 -> bf16 is the hardware conversion v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN) everywhere: the integer emulation
// (NaN test + biased add + shift) costs ~6 VALU ops per value, which made several "HBM-bound" kernels VALU-bound on their stores
// (q/k-norm + RoPE forward: 124 -> 67 us per 16k-token layer after the switch).
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16_hw(float lo, float hi) {
  hw_f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hw_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16_hw(f, 0.f) & 0xffffu); }
// round an f32 to the nearest bf16 and return it as f32 (emulates a bf16 torch op's output rounding)
__device__ __forceinline__ float round_bf16(float f) { return __uint_as_float(pack_bf16_hw(f, 0.f) << 16); }

// SwiGLU activation on bf16-valued inputs: bf16(silu(gate)) * up — torch computes silu on a bf16 tensor in f32 and rounds, then a bf16 multiply whose
// result the caller rounds (mlp.py:27-29).  ONE definition for pm_swiglu_fwd and the gate|up GEMM's fused epilogue (bit-identical by construction).
__device__ __forceinline__ float swiglu_act(float gate, float up) {
  const float s = gate / (1.0f + __expf(-gate));
  return round_bf16(s) * up;
}

// Its gradient on bf16-valued inputs (d = the gradient of act): d_up = d * silu(gate), d_gate = d * up * silu'(gate); the caller rounds both to bf16.
// ONE definition for pm_swiglu_bwd and the down-projection row-gradient GEMM's fused epilogue (bit-identical by construction).
__device__ __forceinline__ void swiglu_bwd_pair(float d, float gate, float up, float& d_gate, float& d_up) {
  const float sig = 1.0f / (1.0f + __expf(-gate));
  const float s = gate * sig;
  d_up = d * s;
  d_gate = d * up * (sig * (1.0f + gate * (1.0f - sig)));
}

// low / high halves of a packed pair
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { return pack_bf16_hw(lo, hi); }

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 16-lane group a lane belongs to
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-lane row reductions on the DPP path (VALU speed; __shfl_xor is an LDS-pipe ds_bpermute with ~100-cycle latency).
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_ror:4 = 0x124, row_ror:8 = 0x128: every lane of a row gets the result.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x124>(v));
  v = fmaxf(v, dpp_f32<0x128>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x124>(v);
  v += dpp_f32<0x128>(v);
  return v;
}
// raw v_exp_f32
__device__ __forceinline__ bf16_t f32_to_bf16_hw(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// block-wide sum for blockDim.x = NWAVES*64; `smem` needs NWAVES floats. Result valid in every thread.
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVES; ++i) r += smem[i];
  return r;
}

}  // namespace pm
