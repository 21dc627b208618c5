/*
 * mixdq_oracle.c -- CPU restatement of the MixDQ W8A8 operator arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mixdq_amd/ may import, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Each function follows one reference file (paths relative to the reference checkout,
 * kernels/mixdq_extension/...).  The reference's native path cannot be built here (CUDA +
 * an un-vendored, unpinned NVIDIA/cutlass submodule: .gitmodules:1-3, kernels/setup.py:19-23), so
 * this file restates the specified arithmetic and is pinned by
 *   - torch.quantize_per_tensor on CPU, the reference's own exact check (op/quant.py:24-27);
 *   - the reference's in-file integer/FP reference formulas (op/qlinear.py:66-83,
 *     op/qconv2d.py:65-95) at the reference's tolerances;
 *   - the reference's imported Python (nn/Linear.py, nn/Conv2d.py, qdiff QuantLayer) run over
 *     this oracle, committed as tests/golden/ fixtures (tests/golden/gen_golden.py).
 * Bit-level agreement with the CUDA binary itself is UNPINNED in this container: whether nvcc
 * contracted the epilogue's mul+add into an FMA cannot be observed without a CUDA toolchain.
 * `variant` selects: 0 = A, fused (fmaf), the likelier one under nvcc -fmad=true; 1 = B, unfused.
 *
 * Plain C99, no dependencies.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- IEEE half <-> float, exact / round-to-nearest-even ---------------------------------- */
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t f2h(float f) { /* round-to-nearest-even; overflow -> inf (no saturation) */
  uint32_t x;
  memcpy(&x, &f, 4);
  uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u | ((ax >> 13) & 0x3ffu) : 0));
  }
  if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25: ties to even -> 0) */
    return sign;
  }
  int e = (int)(ax >> 23) - 127;
  uint32_t man = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
  int shift;
  uint32_t hexp;
  if (e < -14) { /* subnormal half */
    shift = 13 + (-14 - e);
    hexp = 0;
  } else {
    shift = 13;
    hexp = (uint32_t)(e + 15);
  }
  uint32_t q = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t out;
  if (hexp == 0) {
    out = q; /* may carry into exponent 1: correct */
  } else {
    out = ((hexp - 1) << 10) + q; /* q has the implicit bit at 0x400; carry propagates */
  }
  return (uint16_t)(sign | out);
}

/* Exposed for the Python-side tests of the converters themselves. */
float mixdq_oracle_h2f(uint16_t h) { return h2f(h); }
uint16_t mixdq_oracle_f2h(float f) { return f2h(f); }

static int8_t quant1(float xf, float s_inv, float zp, int variant) {
  /* quantize_kernel.cu:21-25: lrintf(x * scale_inv + zp), clamp, cast.  lrintf = RNE. */
  float t;
  if (variant == 0) {
    t = fmaf(xf, s_inv, zp);
  } else {
    volatile float p = xf * s_inv; /* volatile: forbid contraction */
    t = p + zp;
  }
  /* lrintf on NaN/huge is UB in C; the GPU's v_cvt_i32_f32 saturates and maps NaN to 0. */
  long r;
  if (t != t) r = 0;
  else if (t >= 2147483520.0f) r = 2147483647L;
  else if (t <= -2147483648.0f) r = -2147483647L - 1;
  else r = lrintf(t);
  if (r < -128) r = -128;
  if (r > 127) r = 127;
  return (int8_t)r;
}

/* a1: quantize_kernel.cu:10-27 / quantize_kernel_vectorized.cu:29-72, with the intended strided
 * read: logical index -> x offset via x_strides, out offset via
 * out_strides.  ndim <= 8.  Row-major logical order. */
void mixdq_oracle_quantize(const uint16_t* x, int8_t* out, const int64_t* sizes,
                           const int64_t* x_strides, const int64_t* out_strides, int ndim,
                           float s_inv, float zp, int variant) {
  int64_t idx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t numel = 1;
  for (int d = 0; d < ndim; d++) numel *= sizes[d];
  for (int64_t i = 0; i < numel; i++) {
    int64_t xo = 0, oo = 0;
    for (int d = 0; d < ndim; d++) {
      xo += idx[d] * x_strides[d];
      oo += idx[d] * out_strides[d];
    }
    out[oo] = quant1(h2f(x[xo]), s_inv, zp, variant);
    for (int d = ndim - 1; d >= 0; d--) {
      if (++idx[d] < sizes[d]) break;
      idx[d] = 0;
    }
  }
}

/* Epilogue: cutlassGemm_withBias_optimalAlignment.cu:29-95 (Accum -> minus Bias0 ->
 * multiplies Scale -> plus Bias1 -> store half, all FloatRoundStyle::round_to_nearest). */
static uint16_t epilogue(int32_t acc, float bias0, float scale, int has_bias, uint16_t bias_h,
                         int variant) {
  float v = (float)acc; /* cvt.rn.f32.s32 */
  volatile float d = v - bias0;
  float r;
  if (!has_bias) {
    volatile float m = d * scale;
    r = m;
  } else if (variant == 0) {
    r = fmaf(d, scale, h2f(bias_h));
  } else {
    volatile float m = d * scale;
    r = m + h2f(bias_h);
  }
  return f2h(r);
}

/* a2: qlinear.cc:13-137.  A [M,K], W [N,K], D [M,N] f16 bits.  acc_out (optional) receives the
 * exact int32 accumulators. */
void mixdq_oracle_qlinear(const int8_t* A, const int8_t* W, const float* bias0,
                          const float* scale, const uint16_t* bias_or_null, uint16_t* D,
                          int32_t* acc_out_or_null, int64_t M, int N, int K, int variant) {
  for (int64_t m = 0; m < M; m++) {
    const int8_t* a = A + m * (int64_t)K;
    for (int n = 0; n < N; n++) {
      const int8_t* w = W + (int64_t)n * K;
      int32_t acc = 0;
      for (int k = 0; k < K; k++) acc += (int32_t)a[k] * (int32_t)w[k];
      if (acc_out_or_null) acc_out_or_null[m * N + n] = acc;
      D[m * N + n] = epilogue(acc, bias0[n], scale[n], bias_or_null != NULL,
                              bias_or_null ? bias_or_null[n] : 0, variant);
    }
  }
}

/* a4: conv_act_zero_point_propagate.cu:23-51.  out [N,P,Q,K] f32.  `acc` is a float sum in
 * (r, s) order, exactly as the reference's template<float> instantiation. */
void mixdq_oracle_zp_propagate(const float* wsum_krs, float zp, float* out, int N, int H, int W,
                               int K, int R, int S, int P, int Q, int stride, int pad) {
  for (int n = 0; n < N; n++)
    for (int p = 0; p < P; p++)
      for (int q = 0; q < Q; q++)
        for (int k = 0; k < K; k++) {
          int h0 = -pad + p * stride, w0 = -pad + q * stride;
          float acc = 0.f;
          for (int r = 0; r < R; r++)
            for (int s = 0; s < S; s++) {
              int h = h0 + r, w = w0 + s;
              if (h >= 0 && h < H && w >= 0 && w < W) acc += wsum_krs[((int64_t)k * R + r) * S + s];
            }
          out[(((int64_t)n * P + p) * Q + q) * K + k] = acc * zp;
        }
}

/* a3: qconv2d.cc:27-206.  X [N,H,W,C], Wt [K,R,S,C], D [N,P,Q,K] f16 bits.
 * pad == 0: bias0 [K] (qconv2d.cc:117-120,144-146);  pad > 0: wsum [K,R,S] + zp -> per-pixel
 * bias0 (qconv2d.cc:131-136).  Zero padding is 0 in the INT8 domain. */
void mixdq_oracle_qconv2d(const int8_t* X, const int8_t* Wt, const float* scale,
                          const float* wsum_or_null, float zp, const float* bias0_or_null,
                          const uint16_t* bias_or_null, uint16_t* D, int32_t* acc_out_or_null,
                          int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                          int variant) {
  int P = (H + 2 * pad - (R - 1) - 1) / stride + 1;
  int Q = (W + 2 * pad - (S - 1) - 1) / stride + 1;
  for (int n = 0; n < N; n++)
    for (int p = 0; p < P; p++)
      for (int q = 0; q < Q; q++) {
        int h0 = -pad + p * stride, w0 = -pad + q * stride;
        for (int k = 0; k < K; k++) {
          int32_t acc = 0;
          float wacc = 0.f;
          for (int r = 0; r < R; r++)
            for (int s = 0; s < S; s++) {
              int h = h0 + r, w = w0 + s;
              if (h < 0 || h >= H || w < 0 || w >= W) continue;
              const int8_t* xp = X + (((int64_t)n * H + h) * W + w) * C;
              const int8_t* wp = Wt + (((int64_t)k * R + r) * S + s) * C;
              for (int c = 0; c < C; c++) acc += (int32_t)xp[c] * (int32_t)wp[c];
              if (pad > 0) wacc += wsum_or_null[((int64_t)k * R + r) * S + s];
            }
          float b0 = (pad > 0) ? wacc * zp : bias0_or_null[k];
          int64_t o = (((int64_t)n * P + p) * Q + q) * K + k;
          if (acc_out_or_null) acc_out_or_null[o] = acc;
          D[o] = epilogue(acc, b0, scale[k], bias_or_null != NULL,
                          bias_or_null ? bias_or_null[k] : 0, variant);
        }
      }
}

/* f16 + f16 -> f16 elementwise (torch's half add: f32 add, one rounding), used by the split
 * conv_shortcut: nn/Conv2d.py:345 `output = output + output_0`. */
void mixdq_oracle_add_f16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = f2h(h2f(a[i]) + h2f(b[i]));
}

/* FP16 debug GEMM: qlinear.cc:140-204 (B is [K,N] row-major, qlinear.cc:161).  FP32 accumulate
 * here; the reference accumulates in half (cutlassGemm_reference.cu:133) and is only checked to
 * rtol 1e-4 / atol 1e-2 (op/qlinear.py:95). */
void mixdq_oracle_gemm_f16(const uint16_t* A, const uint16_t* B, uint16_t* D, int64_t M, int N,
                           int K) {
  for (int64_t m = 0; m < M; m++)
    for (int n = 0; n < N; n++) {
      float acc = 0.f;
      for (int k = 0; k < K; k++) acc = fmaf(h2f(A[m * K + k]), h2f(B[(int64_t)k * N + n]), acc);
      D[m * N + n] = f2h(acc);
    }
}

/* =============================================================================================
 * Producer fusions (mixdq_amd/csrc/fused_norm.hip).  No reference counterpart: the reference runs
 * stock PyTorch FP16 ops and then its quantizer.  These restate the fused kernels' arithmetic
 * INCLUDING their fixed reduction orders (the order is part of the specification), with the
 * transcendental steps from include/mixdq_math.h; tests/ additionally hold them to PyTorch's
 * fp32-reference GroupNorm / LayerNorm / SiLU / GELU within one FP16 ulp.
 * ============================================================================================= */
#include "../include/mixdq_math.h"

static float rh(float v) { return h2f(f2h(v)); } /* round to fp16 and back */

/* 64-lane xor butterfly: all lanes end with the same bits; returns lane 0. */
static float wave_sum64(float* v) {
  for (int off = 32; off >= 1; off >>= 1) {
    float t[64];
    for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ off];
    for (int l = 0; l < 64; l++) v[l] = t[l];
  }
  return v[0];
}

/* Geometry rule of make_gn_geom() (fused_norm.hip): returns 0 if unsupported. */
static int gn_geom(int N, int64_t HW, int C, int G, int* cg, int* OC, int* PP, int* ppb,
                   int* nchunk) {
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0 || C % 8 != 0) return 0;
  *cg = C / G;
  *OC = C / 8;
  if (*OC > 1024) return 0;
  for (int o = 0; o < *OC; o++)
    if ((8 * o + 7) / *cg - (8 * o) / *cg > 1) return 0;
  *PP = *OC >= 256 ? 1 : 256 / *OC;
  if (G > *OC * *PP) return 0;
  int64_t target = 512; /* per image: independent of the batch (N) */
  int64_t p = (HW + target - 1) / target;
  p = ((p + *PP - 1) / *PP) * *PP;
  *ppb = (int)p;
  *nchunk = (int)((HW + p - 1) / p);
  return 1;
}

/* gn_stats_kernel + gn_finalize_kernel: mean / rstd per (n, group), in the kernels' summation
 * order.  Returns 0 when the shape is unsupported. */
int mixdq_oracle_groupnorm_stats(const uint16_t* x, float eps, float* mean, float* rstd, int N,
                                 int64_t HW, int C, int G) {
  int cg, OC, PP, ppb, nchunk;
  if (!gn_geom(N, HW, C, G, &cg, &OC, &PP, &ppb, &nchunk)) return 0;
  const int T = OC * PP;
  float* acc = (float*)malloc(sizeof(float) * 4 * T);
  float* ps = (float*)malloc(sizeof(float) * (size_t)nchunk * G);
  float* pq = (float*)malloc(sizeof(float) * (size_t)nchunk * G);
  for (int n = 0; n < N; n++) {
    for (int chunk = 0; chunk < nchunk; chunk++) {
      int64_t pb = (int64_t)chunk * ppb, pe = pb + ppb < HW ? pb + ppb : HW;
      for (int t = 0; t < T; t++) { /* gn_stats_kernel, one "thread" at a time */
        int o = t % OC, pp = t / OC, g0 = (8 * o) / cg;
        int jb = (g0 + 1) * cg - 8 * o;
        if (jb > 8) jb = 8;
        float s0 = 0, q0 = 0, s1 = 0, q1 = 0;
        for (int64_t p = pb + pp; p < pe; p += PP)
          for (int j = 0; j < 8; j++) {
            float v = h2f(x[((int64_t)n * HW + p) * C + 8 * o + j]);
            if (j < jb) { s0 = s0 + v; q0 = fmaf(v, v, q0); }
            else        { s1 = s1 + v; q1 = fmaf(v, v, q1); }
          }
        acc[4 * t] = s0; acc[4 * t + 1] = q0; acc[4 * t + 2] = s1; acc[4 * t + 3] = q1;
      }
      for (int g = 0; g < G; g++) {
        int olo = (g * cg) / 8, ohi = ((g + 1) * cg - 1) / 8;
        float s = 0, q = 0;
        for (int l = 0; l < PP; l++)
          for (int oo = olo; oo <= ohi; oo++) {
            int tt = l * OC + oo, first = (8 * oo) / cg, part = (first == g) ? 0 : 2;
            if (first == g || first + 1 == g) { s = s + acc[4 * tt + part]; q = q + acc[4 * tt + part + 1]; }
          }
        ps[(size_t)chunk * G + g] = s;
        pq[(size_t)chunk * G + g] = q;
      }
    }
    for (int g = 0; g < G; g++) { /* gn_finalize_kernel: lane partials + xor butterfly */
      float ls[64], lq[64];
      for (int l = 0; l < 64; l++) {
        float a = 0, b = 0;
        for (int c = l; c < nchunk; c += 64) { a = a + ps[(size_t)c * G + g]; b = b + pq[(size_t)c * G + g]; }
        ls[l] = a; lq[l] = b;
      }
      float s = wave_sum64(ls), q = wave_sum64(lq);
      float cnt = (float)((double)HW * cg);
      float m = s / cnt;
      float var = fmaf(-m, m, q / cnt);
      if (!(var > 0.f)) var = 0.f;
      mean[n * G + g] = m;
      rstd[n * G + g] = 1.0f / sqrtf(var + eps);
    }
  }
  free(acc); free(ps); free(pq);
  return 1;
}

/* x [N,HW,C] fp16 bits.  out_q / out_h may be NULL.  Returns 0 when the shape is unsupported. */
int mixdq_oracle_groupnorm_silu_quantize(const uint16_t* x, const uint16_t* gamma,
                                         const uint16_t* beta, float eps, int apply_silu,
                                         float s_inv, float zp, int8_t* out_q, uint16_t* out_h,
                                         int N, int64_t HW, int C, int G, int variant) {
  const int cg = (G > 0 && C % G == 0) ? C / G : 1;
  float* mean = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * (G > 0 ? G : 1));
  float* rstd = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * (G > 0 ? G : 1));
  if (!mixdq_oracle_groupnorm_stats(x, eps, mean, rstd, N, HW, C, G)) {
    free(mean); free(rstd);
    return 0;
  }
  for (int n = 0; n < N; n++) /* gn_apply_kernel (elementwise: order irrelevant) */
    for (int64_t p = 0; p < HW; p++)
      for (int c = 0; c < C; c++) {
        int g = c / cg;
        float a = rstd[n * G + g] * h2f(gamma[c]);
        float b = fmaf(-mean[n * G + g], a, h2f(beta[c]));
        int64_t i = ((int64_t)n * HW + p) * C + c;
        float y = rh(fmaf(h2f(x[i]), a, b));
        if (apply_silu) y = rh(mixdq_siluf(y));
        if (out_h) out_h[i] = f2h(y);
        if (out_q) out_q[i] = quant1(y, s_inv, zp, variant);
      }
  free(mean); free(rstd);
  return 1;
}

/* ln_quant_kernel: one 64-lane wave per row, lane l owns chunks l, l+64, ...; xor butterfly. */

/* LayerNorm + quantize.  The row statistics follow a reduction order that does not depend on how a kernel
 * tiles the row (round 5): csrc/fused_norm.hip's one-wave-per-row kernel and the GEMM epilogue of
 * csrc/igemm_ln.hip -- where the 16 column tiles of a row block each hold 80 of the 1280 columns and
 * exchange ONE record per row and tile -- produce the same bits.
 *   group g = columns 16g .. 16g+15 (C % 16 == 0):
 *     S1_g = (x0 + .. + x7) + (x8 + .. + x15)      each half summed left to right, FP32
 *     m_g  = S1_g / 16                              (exact)
 *     M2_g = (sum_{j<8} d_j^2) + (sum_{j>=8} d_j^2),  d_j = x_j - m_g, each half an fmaf chain from 0
 *   unit u = `per` consecutive groups, U = (largest power of two <= 16 dividing G = C / 16) units per row
 *   (C = 1280: 16 units of 5 groups = 80 columns, the GEMM's column tile; C = 640: 8 such units):
 *     S1_u = S1_g summed left to right;   m_u = S1_u * rn(1 / (16 per))          (rn: the FP32 reciprocal, one multiply)
 *     M2_u = sum, left to right, of fmaf(16 (m_g - m_u), m_g - m_u, M2_g)        (Chan's combination)
 *   row: the U unit values combined by a balanced binary tree (pairs at distance 1, 2, 4, 8):
 *     mean = tree(S1_u) * rn(1 / C)
 *     var  = tree( fmaf(16 per (m_u - mean), m_u - mean, M2_u) ) * rn(1 / C)
 *     rstd = 1 / sqrtf(var + eps)
 *   y = f16( fmaf((x - mean) * rstd, gamma, beta) ), each quantizer applied to y. */
static int ln_units(int G) {
  int u = 1;
  while (u < 16 && G % (2 * u) == 0) u *= 2;
  return u;
}

static float ln_tree(float* a, int U) {
  for (int off = 1; off < U; off *= 2) {
    float b[16];
    for (int i = 0; i < U; i++) b[i] = a[i] + a[i ^ off];
    for (int i = 0; i < U; i++) a[i] = b[i];
  }
  return a[0];
}

void mixdq_oracle_layernorm_quantize(const uint16_t* x, const uint16_t* gamma, const uint16_t* beta,
                                     float eps, int64_t M, int C, int n_out, const float* s_inv,
                                     const float* zp, int8_t** out_q, uint16_t* out_h,
                                     int variant) {
  const int G = C / 16, U = ln_units(G > 0 ? G : 1), per = (G > 0 ? G : 1) / U;
  const float n_u = (float)(16 * per);
  const float inv_nu = 1.0f / n_u, inv_c = 1.0f / (float)C;
  for (int64_t r = 0; r < M; r++) {
    const uint16_t* xr = x + r * C;
    float s1u[16], m2u[16], mu[16];
    for (int u = 0; u < U; u++) {
      float s1g[128], m2g[128];           /* per <= 128 (C <= 2048 in the kernels; any C here up to 32768) */
      for (int k = 0; k < per; k++) {
        const uint16_t* xg = xr + 16 * (u * per + k);
        float half[2];
        for (int h = 0; h < 2; h++) {
          float t = h2f(xg[8 * h]);
          for (int j = 1; j < 8; j++) t = t + h2f(xg[8 * h + j]);
          half[h] = t;
        }
        s1g[k] = half[0] + half[1];
        const float mg = s1g[k] * 0.0625f;
        for (int h = 0; h < 2; h++) {
          float t = 0;
          for (int j = 0; j < 8; j++) { float d = h2f(xg[8 * h + j]) - mg; t = fmaf(d, d, t); }
          half[h] = t;
        }
        m2g[k] = half[0] + half[1];
      }
      float t = s1g[0];
      for (int k = 1; k < per; k++) t = t + s1g[k];
      s1u[u] = t;
      mu[u] = t * inv_nu;
      float q = 0;
      for (int k = 0; k < per; k++) {
        const float e = s1g[k] * 0.0625f - mu[u];
        const float c = fmaf(e * 16.0f, e, m2g[k]);
        q = k == 0 ? c : q + c;
      }
      m2u[u] = q;
    }
    float a[16];
    for (int u = 0; u < U; u++) a[u] = s1u[u];
    const float mean = ln_tree(a, U) * inv_c;
    for (int u = 0; u < U; u++) {
      const float e = mu[u] - mean;
      a[u] = fmaf(e * n_u, e, m2u[u]);
    }
    const float rstd = 1.0f / sqrtf(ln_tree(a, U) * inv_c + eps);
    for (int c = 0; c < C; c++) {
      volatile float nrm = (h2f(xr[c]) - mean) * rstd;
      float y = rh(fmaf(nrm, h2f(gamma[c]), h2f(beta[c])));
      if (out_h) out_h[r * C + c] = f2h(y);
      for (int k = 0; k < n_out; k++) out_q[k][r * C + c] = quant1(y, s_inv[k], zp[k], variant);
    }
  }
}

void mixdq_oracle_geglu_quantize(const uint16_t* h, int64_t M, int D, float s_inv, float zp,
                                 int8_t* out_q, uint16_t* out_h, int variant) {
  for (int64_t m = 0; m < M; m++)
    for (int d = 0; d < D; d++) {
      float ge = rh(mixdq_geluf(h2f(h[m * 2 * D + D + d])));
      volatile float prod = h2f(h[m * 2 * D + d]) * ge;
      float y = rh(prod);
      if (out_h) out_h[m * D + d] = f2h(y);
      if (out_q) out_q[m * D + d] = quant1(y, s_inv, zp, variant);
    }
}

float mixdq_oracle_expf(float x) { return mixdq_expf(x); }
float mixdq_oracle_erff(float x) { return mixdq_erff(x); }
float mixdq_oracle_siluf(float x) { return mixdq_siluf(x); }
float mixdq_oracle_geluf(float x) { return mixdq_geluf(x); }
