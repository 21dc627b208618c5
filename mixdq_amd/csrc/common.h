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

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute is a property of the kernel ON A
// DEVICE: it is applied once per (instantiation, device) -- `seen` is the instantiation's own static
// table -- so that a process that drives several devices through the C-ABI gets it on each of them.
inline int lds_opt_in(const void* kernel, int bytes, bool (&seen)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MIXDQ_ERR_LAUNCH;
  if (!seen[dev]) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      return MIXDQ_ERR_LAUNCH;
    seen[dev] = true;
  }
  return MIXDQ_OK;
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

// Eight values at once, packed: the same arithmetic (v_rndne_f32, v_cvt_i32_f32: saturating, NaN -> 0), the clamp
// and the packing by v_ashr_pk_i8_i32 (gfx950: two INT32 -> two saturated INT8 in one instruction) + v_perm_b32 --
// 3.75 vector instructions per element where the scalar form with min / max / mask / shift / or spends ~7.
template <bool UNFUSED>
__device__ __forceinline__ uint2 quantize_pack8(const float (&x)[8], float s_inv, float zp) {
  uint32_t w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t0 = UNFUSED ? __fadd_rn(__fmul_rn(x[2 * e], s_inv), zp) : __builtin_fmaf(x[2 * e], s_inv, zp);
    const float t1 = UNFUSED ? __fadd_rn(__fmul_rn(x[2 * e + 1], s_inv), zp) : __builtin_fmaf(x[2 * e + 1], s_inv, zp);
    w[e] = __builtin_amdgcn_ashr_pk_i8_i32((int)__builtin_rintf(t0), (int)__builtin_rintf(t1), 0);
  }
  return make_uint2(__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), __builtin_amdgcn_perm(w[3], w[2], 0x05040100u));
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

// ---- kernel arguments, all at once ---------------------------------------------------------------
// hipcc loads a kernel's arguments lazily, next to their first uses: several DEPENDENT round trips to
// a scalar cache that is cold at kernel start (~0.5 us each, tools/stamp_report.py), in front of the
// first useful memory request of every launch of the batch-1 chain.  MIXDQ_ARGS_NOW(a, b, ...) as
// the first statement of a kernel makes every listed argument live in scalar registers at that
// point, so that all of them are requested together behind one wait (at most 16 per statement).
#define MIXDQ_ARG1_(x) "s"(x)
#define MIXDQ_ARGS_NOW(...) asm volatile("" ::MIXDQ_FOR_EACH_(MIXDQ_ARG1_, __VA_ARGS__))
#define MIXDQ_FE_1(m, a) m(a)
#define MIXDQ_FE_2(m, a, ...) m(a), MIXDQ_FE_1(m, __VA_ARGS__)
#define MIXDQ_FE_3(m, a, ...) m(a), MIXDQ_FE_2(m, __VA_ARGS__)
#define MIXDQ_FE_4(m, a, ...) m(a), MIXDQ_FE_3(m, __VA_ARGS__)
#define MIXDQ_FE_5(m, a, ...) m(a), MIXDQ_FE_4(m, __VA_ARGS__)
#define MIXDQ_FE_6(m, a, ...) m(a), MIXDQ_FE_5(m, __VA_ARGS__)
#define MIXDQ_FE_7(m, a, ...) m(a), MIXDQ_FE_6(m, __VA_ARGS__)
#define MIXDQ_FE_8(m, a, ...) m(a), MIXDQ_FE_7(m, __VA_ARGS__)
#define MIXDQ_FE_9(m, a, ...) m(a), MIXDQ_FE_8(m, __VA_ARGS__)
#define MIXDQ_FE_10(m, a, ...) m(a), MIXDQ_FE_9(m, __VA_ARGS__)
#define MIXDQ_FE_11(m, a, ...) m(a), MIXDQ_FE_10(m, __VA_ARGS__)
#define MIXDQ_FE_12(m, a, ...) m(a), MIXDQ_FE_11(m, __VA_ARGS__)
#define MIXDQ_FE_13(m, a, ...) m(a), MIXDQ_FE_12(m, __VA_ARGS__)
#define MIXDQ_FE_14(m, a, ...) m(a), MIXDQ_FE_13(m, __VA_ARGS__)
#define MIXDQ_FE_15(m, a, ...) m(a), MIXDQ_FE_14(m, __VA_ARGS__)
#define MIXDQ_FE_16(m, a, ...) m(a), MIXDQ_FE_15(m, __VA_ARGS__)
#define MIXDQ_FE_N_(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, _11, _12, _13, _14, _15, _16, N, ...) N
#define MIXDQ_FOR_EACH_(m, ...)                                                                     \
  MIXDQ_FE_N_(__VA_ARGS__, MIXDQ_FE_16, MIXDQ_FE_15, MIXDQ_FE_14, MIXDQ_FE_13, MIXDQ_FE_12,        \
              MIXDQ_FE_11, MIXDQ_FE_10, MIXDQ_FE_9, MIXDQ_FE_8, MIXDQ_FE_7, MIXDQ_FE_6, MIXDQ_FE_5, \
              MIXDQ_FE_4, MIXDQ_FE_3, MIXDQ_FE_2, MIXDQ_FE_1)(m, __VA_ARGS__)

// ---- GELU on two values at a time (packed-FP32 VALU: v_pk_mul / v_pk_fma / v_pk_add_f32) ----------
// Each half performs EXACTLY the IEEE operations of the scalar specification include/mixdq_math.h
// (mixdq_geluf -> mixdq_erff -> mixdq_expf), in the same order; the two erf branches are both
// evaluated and selected per element (a wave of gate values takes both anyway), so the result is
// the specification's bit for bit -- at about half the vector instructions per element.  It is the
// epilogue arithmetic of the largest launch of the step (ff.net.0.proj + GEGLU: ~5 of its ~25 us).
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, float c) { return __builtin_elementwise_fma(a, b, v2f{c, c}); }
__device__ __forceinline__ v2f pk_fma(float a, v2f b, float c) { return __builtin_elementwise_fma(v2f{a, a}, b, v2f{c, c}); }

// mixdq_expf for arguments <= 0 (or -inf): erf's large branch never passes anything else, so the
// NaN / overflow / e > 127 cases of the specification cannot occur and are not evaluated
__device__ __forceinline__ v2f expf2_nonpos(v2f x) {
  const v2f n = __builtin_elementwise_rint(x * 1.44269504088896341f);
  v2f r = pk_fma(n, v2f{-0.693359375f, -0.693359375f}, x);
  r = pk_fma(n, v2f{2.12194440e-4f, 2.12194440e-4f}, r);
  v2f p = v2f{1.9875691500e-4f, 1.9875691500e-4f};
  p = pk_fma(p, r, 1.3981999507e-3f);
  p = pk_fma(p, r, 8.3334519073e-3f);
  p = pk_fma(p, r, 4.1665795894e-2f);
  p = pk_fma(p, r, 1.6666665459e-1f);
  p = pk_fma(p, r, 5.0000001201e-1f);
  p = pk_fma(p * r, r, r) + 1.0f;
  const v2i e = __builtin_convertvector(n, v2i);                 // -126 .. 0 where the result is used
  const v2u sb = __builtin_bit_cast(v2u, e + 127) << 23;
  v2f y = p * __builtin_bit_cast(v2f, sb);
  y[0] = x[0] < -87.33654f ? 0.0f : y[0];
  y[1] = x[1] < -87.33654f ? 0.0f : y[1];
  return y;
}

__device__ __forceinline__ v2f erff2(v2f a) {
  const v2u ua = __builtin_bit_cast(v2u, a);
  const v2f t = __builtin_bit_cast(v2f, ua & 0x7fffffffu);
  const v2f s = a * a;
  v2f r = pk_fma(-1.72853470e-5f, t, 3.83197126e-4f);
  const v2f u = pk_fma(-3.88396438e-3f, t, 2.42546219e-2f);
  r = pk_fma(r, s, u);
  r = pk_fma(r, t, -1.06777877e-1f);
  r = pk_fma(r, t, -6.34846687e-1f);
  r = pk_fma(r, t, -1.28717512e-1f);
  r = pk_fma(r, t, -t);
  r = 1.0f - expf2_nonpos(r);
  const v2u ur = __builtin_bit_cast(v2u, r);
  const v2f big = __builtin_bit_cast(v2f, (ur & 0x7fffffffu) | (ua & 0x80000000u));
  v2f q = v2f{-5.96761703e-4f, -5.96761703e-4f};
  q = pk_fma(q, s, 4.99119423e-3f);
  q = pk_fma(q, s, -2.67681349e-2f);
  q = pk_fma(q, s, 1.12819925e-1f);
  q = pk_fma(q, s, -3.76125336e-1f);
  q = pk_fma(q, s, 1.28379166e-1f);
  q = pk_fma(q, a, a);
  v2f out;
  out[0] = t[0] >= 0.921875f ? big[0] : q[0];      // NaN compares false: the small branch propagates it
  out[1] = t[1] >= 0.921875f ? big[1] : q[1];
  return out;
}

__device__ __forceinline__ v2f geluf2(v2f x) {
  return (0.5f * x) * (1.0f + erff2(x * 0.70710678118654752440f));
}

}  // namespace mixdq
