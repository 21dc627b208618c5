// Shared device helpers for the gfx950 kernels (wave64, CDNA4).  gfx950 only: no other targets.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/mixdq_hip.h"

namespace mixdq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));

constexpr int kNumCU = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr int kNumXCD = 8;

// Launch-status helper: kernels are launched asynchronously; only launch-time errors are seen.
inline int launch_status() {
  return hipGetLastError() == hipSuccess ? MIXDQ_OK : MIXDQ_ERR_LAUNCH;
}

// ---- arithmetic specification ------------------------------------------
// Quantize one value: q = clamp(rint(x * s_inv + zp)).  rint = round-half-to-even (v_rndne_f32),
// the float->int conversion saturates and maps NaN to 0 (v_cvt_i32_f32), as cvt.rni does on the
// reference's hardware.  FUSED = one FMA (variant A); otherwise mul then add (variant B).
template <bool UNFUSED>
__device__ __forceinline__ int quantize_one(float x, float s_inv, float zp) {
  float t = UNFUSED ? __fadd_rn(__fmul_rn(x, s_inv), zp) : __builtin_fmaf(x, s_inv, zp);
  int i = (int)__builtin_rintf(t);
  return min(max(i, -128), 127);
}

// FP32 -> FP16 with the FP32 value made opaque first.  Without the barrier hipcc folds
// `__float2half_rn(a * b + c)` into v_fma_mixlo_f16, which rounds the exact product-sum ONCE to
// FP16 (and adds +0, turning -0 into +0): not the specified "round to FP32, then to FP16".
__device__ __forceinline__ __half f32_to_f16_rn(float v) {
  asm("" : "+v"(v));
  return __float2half_rn(v);
}

// Epilogue: (f32(acc) - bias0) * scale [+ bias] -> f16 (RNE, overflow -> inf).
__device__ __forceinline__ __half epilogue_one(int acc, float bias0, float scale, float bias,
                                               bool has_bias, bool unfused) {
  float v = __fsub_rn((float)acc, bias0);   // v_cvt_f32_i32 (RNE) then subtract
  float r;
  if (!has_bias) {
    r = __fmul_rn(v, scale);
  } else if (unfused) {
    r = __fadd_rn(__fmul_rn(v, scale), bias);
  } else {
    r = __builtin_fmaf(v, scale, bias);
  }
  return f32_to_f16_rn(r);
}

}  // namespace mixdq
