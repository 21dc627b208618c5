/*
 * mixdq_math.h -- the arithmetic SPECIFICATION of the transcendental steps used by the fused
 * producer kernels (SiLU, GELU), written so that a host C compiler and hipcc produce bit-identical
 * results: only IEEE-754 binary32 +, -, *, /, fmaf, rintf and integer bit operations, every fused
 * multiply-add explicit (compile with -ffp-contract=off), no libm transcendental, no fast-math.
 *
 * The reference computes SiLU / GELU with stock PyTorch FP16 ops; those are
 * accurate to ~1 ulp of FP32 before rounding to FP16, as are these, so results agree with PyTorch to
 * within one FP16 ulp (tests state the tolerance).  Between the HIP kernels and the CPU oracle the
 * agreement is exact by construction, which is what lets the fused kernels be tested bit-for-bit.
 */
#ifndef MIXDQ_MATH_H_
#define MIXDQ_MATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MIXDQ_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define MIXDQ_HD static inline
#endif

MIXDQ_HD float mixdq_bits_to_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* exp(x), |rel err| < 2 ulp.  Cephes-style: n = rint(x*log2(e)), r = x - n*ln2 (two-term),
 * degree-5 polynomial, scale by 2^n through the exponent bits.  Results below 2^-126 flush to 0,
 * above FLT_MAX give +inf. */
MIXDQ_HD float mixdq_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283f) return mixdq_bits_to_float(0x7f800000u);
  if (x < -87.33654f) return 0.0f;
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  p = __builtin_fmaf(p * r, r, r) + 1.0f;
  int e = (int)n;            /* -126 .. 128 */
  float scale;
  if (e > 127) {             /* split the scale so 2^128 never has to be formed */
    p = p * 2.0f;
    e -= 1;
  }
  scale = mixdq_bits_to_float((uint32_t)(e + 127) << 23);
  return p * scale;
}

/* SiLU on an FP32 value: x / (1 + exp(-x)) with a correctly rounded division. */
MIXDQ_HD float mixdq_siluf(float x) { return x / (1.0f + mixdq_expf(-x)); }

/* erf(x), max error 1.5 ulp (|abs err| < 7e-8), two branches, one exp, no division:
 *   |x| <  0.921875 : x + x * P(x^2)                      (degree-5 polynomial in x^2)
 *   |x| >= 0.921875 : +-(1 - exp(-|x| + |x| * Q(|x|)))     (degree-6 polynomial, mixed x / x^2)
 * Minimax coefficients as published by N. Juffa for single-precision erff. */
MIXDQ_HD float mixdq_erff(float a) {
  uint32_t ua;
  memcpy(&ua, &a, 4);
  const float t = mixdq_bits_to_float(ua & 0x7fffffffu);   /* |a| */
  const float s = a * a;
  float r;
  if (t >= 0.921875f) {
    float u;
    r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    r = 1.0f - mixdq_expf(r);
    uint32_t ur;
    memcpy(&ur, &r, 4);
    r = mixdq_bits_to_float((ur & 0x7fffffffu) | (ua & 0x80000000u));   /* copysign(r, a) */
  } else {                     /* also taken by NaN: propagates */
    r = -5.96761703e-4f;
    r = __builtin_fmaf(r, s, 4.99119423e-3f);
    r = __builtin_fmaf(r, s, -2.67681349e-2f);
    r = __builtin_fmaf(r, s, 1.12819925e-1f);
    r = __builtin_fmaf(r, s, -3.76125336e-1f);
    r = __builtin_fmaf(r, s, 1.28379166e-1f);
    r = __builtin_fmaf(r, a, a);
  }
  return r;
}

/* GELU (erf form, torch's default): 0.5 * x * (1 + erf(x / sqrt(2))). */
MIXDQ_HD float mixdq_geluf(float x) {
  return 0.5f * x * (1.0f + mixdq_erff(x * 0.70710678118654752440f));
}

#endif /* MIXDQ_MATH_H_ */
