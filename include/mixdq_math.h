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

/* erf(x), |abs err| < 2e-7.  Abramowitz-Stegun 7.1.26 is too coarse (1.5e-7 needs double); this
 * is the FreeBSD msun / musl erff decomposition restated with explicit fmaf:
 *   |x| < 0.84375 : x + x*P(x^2)/Q(x^2)
 *   |x| < 1.25    : +-(erx + P(|x|-1)/Q(|x|-1))
 *   |x| < 6       : +-(1 - exp(-z*z - 0.5625) * exp((z-|x|)*(z+|x|) + R/S) / |x|),  z = |x| with
 *                   the low 13 mantissa bits cleared
 *   else          : +-1                                                                          */
MIXDQ_HD float mixdq_erff(float x) {
  const float erx = 8.4506291151e-01f;
  const float efx8 = 1.0270333290e+00f;
  const float pp0 = 1.2837916613e-01f, pp1 = -3.2504209876e-01f, pp2 = -2.8481749818e-02f,
              pp3 = -5.7702702470e-03f, pp4 = -2.3763017452e-05f;
  const float qq1 = 3.9791721106e-01f, qq2 = 6.5022252500e-02f, qq3 = 5.0813062117e-03f,
              qq4 = 1.3249473704e-04f, qq5 = -3.9602282413e-06f;
  const float pa0 = -2.3621185683e-03f, pa1 = 4.1485610604e-01f, pa2 = -3.7220788002e-01f,
              pa3 = 3.1834661961e-01f, pa4 = -1.1089469492e-01f, pa5 = 3.5478305072e-02f,
              pa6 = -2.1663755178e-03f;
  const float qa1 = 1.0642088205e-01f, qa2 = 5.4039794207e-01f, qa3 = 7.1828655899e-02f,
              qa4 = 1.2617121637e-01f, qa5 = 1.3637083583e-02f, qa6 = 1.1984500103e-02f;
  const float ra0 = -9.8649440333e-03f, ra1 = -6.9385856390e-01f, ra2 = -1.0558626175e+01f,
              ra3 = -6.2375331879e+01f, ra4 = -1.6239666748e+02f, ra5 = -1.8460508728e+02f,
              ra6 = -8.1287437439e+01f, ra7 = -9.8143291473e+00f;
  const float sa1 = 1.9651271820e+01f, sa2 = 1.3765776062e+02f, sa3 = 4.3456588745e+02f,
              sa4 = 6.4538726807e+02f, sa5 = 4.2900814819e+02f, sa6 = 1.0863500214e+02f,
              sa7 = 6.5702495575e+00f, sa8 = -6.0424413532e-02f;
  const float rb0 = -9.8649431020e-03f, rb1 = -7.9928326607e-01f, rb2 = -1.7757955551e+01f,
              rb3 = -1.6063638306e+02f, rb4 = -6.3756646729e+02f, rb5 = -1.0250950928e+03f,
              rb6 = -4.8351919556e+02f;
  const float sb1 = 3.0338060379e+01f, sb2 = 3.2579251099e+02f, sb3 = 1.5367296143e+03f,
              sb4 = 3.1998581543e+03f, sb5 = 2.5530502930e+03f, sb6 = 4.7452853394e+02f,
              sb7 = -2.2440952301e+01f;
  uint32_t ux;
  memcpy(&ux, &x, 4);
  const uint32_t ix = ux & 0x7fffffffu;
  const int neg = (int)(ux >> 31);
  if (ix >= 0x7f800000u) {   /* erf(nan) = nan, erf(+-inf) = +-1 */
    if (ix > 0x7f800000u) return x;
    return neg ? -1.0f : 1.0f;
  }
  const float ax = mixdq_bits_to_float(ix);
  if (ix < 0x3f580000u) {    /* |x| < 0.84375 */
    if (ix < 0x31800000u)    /* |x| < 2^-28: avoid underflow */
      return 0.125f * __builtin_fmaf(efx8, x, 8.0f * x);
    const float z = x * x;
    float r = __builtin_fmaf(z, pp4, pp3);
    r = __builtin_fmaf(z, r, pp2);
    r = __builtin_fmaf(z, r, pp1);
    r = __builtin_fmaf(z, r, pp0);
    float s = __builtin_fmaf(z, qq5, qq4);
    s = __builtin_fmaf(z, s, qq3);
    s = __builtin_fmaf(z, s, qq2);
    s = __builtin_fmaf(z, s, qq1);
    s = __builtin_fmaf(z, s, 1.0f);
    return __builtin_fmaf(x, r / s, x);
  }
  if (ix < 0x3fa00000u) {    /* 0.84375 <= |x| < 1.25 */
    const float s = ax - 1.0f;
    float P = __builtin_fmaf(s, pa6, pa5);
    P = __builtin_fmaf(s, P, pa4);
    P = __builtin_fmaf(s, P, pa3);
    P = __builtin_fmaf(s, P, pa2);
    P = __builtin_fmaf(s, P, pa1);
    P = __builtin_fmaf(s, P, pa0);
    float Q = __builtin_fmaf(s, qa6, qa5);
    Q = __builtin_fmaf(s, Q, qa4);
    Q = __builtin_fmaf(s, Q, qa3);
    Q = __builtin_fmaf(s, Q, qa2);
    Q = __builtin_fmaf(s, Q, qa1);
    Q = __builtin_fmaf(s, Q, 1.0f);
    return neg ? -erx - P / Q : erx + P / Q;
  }
  if (ix >= 0x40c00000u)     /* |x| >= 6 */
    return neg ? -1.0f : 1.0f;
  const float s = 1.0f / (ax * ax);
  float R, S;
  if (ix < 0x4036db6eu) {    /* |x| < 1/0.35 */
    R = __builtin_fmaf(s, ra7, ra6);
    R = __builtin_fmaf(s, R, ra5);
    R = __builtin_fmaf(s, R, ra4);
    R = __builtin_fmaf(s, R, ra3);
    R = __builtin_fmaf(s, R, ra2);
    R = __builtin_fmaf(s, R, ra1);
    R = __builtin_fmaf(s, R, ra0);
    S = __builtin_fmaf(s, sa8, sa7);
    S = __builtin_fmaf(s, S, sa6);
    S = __builtin_fmaf(s, S, sa5);
    S = __builtin_fmaf(s, S, sa4);
    S = __builtin_fmaf(s, S, sa3);
    S = __builtin_fmaf(s, S, sa2);
    S = __builtin_fmaf(s, S, sa1);
    S = __builtin_fmaf(s, S, 1.0f);
  } else {
    R = __builtin_fmaf(s, rb6, rb5);
    R = __builtin_fmaf(s, R, rb4);
    R = __builtin_fmaf(s, R, rb3);
    R = __builtin_fmaf(s, R, rb2);
    R = __builtin_fmaf(s, R, rb1);
    R = __builtin_fmaf(s, R, rb0);
    S = __builtin_fmaf(s, sb7, sb6);
    S = __builtin_fmaf(s, S, sb5);
    S = __builtin_fmaf(s, S, sb4);
    S = __builtin_fmaf(s, S, sb3);
    S = __builtin_fmaf(s, S, sb2);
    S = __builtin_fmaf(s, S, sb1);
    S = __builtin_fmaf(s, S, 1.0f);
  }
  const float z = mixdq_bits_to_float(ix & 0xffffe000u);
  const float r = mixdq_expf(__builtin_fmaf(-z, z, -0.5625f)) *
                  mixdq_expf(__builtin_fmaf(z - ax, z + ax, R / S));
  return neg ? r / ax - 1.0f : 1.0f - r / ax;
}

/* GELU (erf form, torch's default): 0.5 * x * (1 + erf(x / sqrt(2))). */
MIXDQ_HD float mixdq_geluf(float x) {
  return 0.5f * x * (1.0f + mixdq_erff(x * 0.70710678118654752440f));
}

#endif /* MIXDQ_MATH_H_ */
